#!/bin/bash
# A/B of the scan kernel with and without what it hands to block calling (TS_EMIT=0: round 3's behaviour), same build, one box.
cd "$(dirname "$0")/.."
run() { timeout -k 10 180 python3 bench.py --no-cpu-baseline --no-e2e --no-reads "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-14s %9.1f Gbases/s  step %.4f ms kernel %.4f ms' % (sys.argv[1], d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" "$LABEL"; }
for i in 1 2; do
  LABEL=emit0 TS_EMIT=0 run
  LABEL=emit1 run
done
LABEL=plant-emit0 TS_EMIT=0 run --flags "-c CCCTAAA -w 2000 -s 1000 -r -g -e -m -i"
LABEL=plant-emit1 run --flags "-c CCCTAAA -w 2000 -s 1000 -r -g -e -m -i"
LABEL=default-emit0 TS_EMIT=0 run --flags "-c TTAGGG -r -g -e -m -i"
LABEL=default-emit1 run --flags "-c TTAGGG -r -g -e -m -i"
