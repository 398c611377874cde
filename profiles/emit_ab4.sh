#!/bin/bash
# A/B on one box: round 3's library; this build with the queue drained eagerly (-DTS_LAZY_DRAIN=0) and lazily; each plain and emitting
cd "$(dirname "$0")/.."
run() { timeout -k 10 180 python3 bench.py --no-cpu-baseline --no-e2e --no-reads "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-14s %9.1f Gbases/s  step %.4f ms kernel %.4f ms' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" "$LABEL"; }
for i in 1 2; do
  LABEL=r03 TELOSCAN_LIB=$PWD/teloscope_amd/libteloscan_old.so run "$@"
  LABEL=eager TELOSCAN_LIB=$PWD/teloscope_amd/libteloscan_eager.so run "$@"
  LABEL=eager-emit TELOSCAN_LIB=$PWD/teloscope_amd/libteloscan_eager.so TS_BENCH_EMIT=1 run "$@"
  LABEL=lazy run "$@"
  LABEL=lazy-emit TS_BENCH_EMIT=1 run "$@"
done
