#!/bin/bash
# three-way A/B on one box: round 3's library, this build with TS_EMIT=0, this build
cd "$(dirname "$0")/.."
run() { timeout -k 10 180 python3 bench.py --no-cpu-baseline --no-e2e --no-reads "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-14s %9.1f Gbases/s  step %.4f ms kernel %.4f ms' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" "$LABEL"; }
for i in 1 2 3; do
  LABEL=r03 TELOSCAN_LIB=$PWD/teloscope_amd/libteloscan_old.so run "$@"
  LABEL=emit0 run "$@"
  LABEL=emit1 TS_BENCH_EMIT=1 run "$@"
done
