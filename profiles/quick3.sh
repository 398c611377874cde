#!/bin/bash
# Kernel time of the three reference configurations on the resident 3 Gb input (A/B runs while tuning).
cd "$(dirname "$0")/.."
run() { timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-e2e "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s %9.1f Gbases/s  kernel %.4f ms' % (sys.argv[1], d['value'], d['roofline']['kernel_ms']))" "$LABEL"; }
LABEL=headline run; LABEL=headline run
LABEL=plant run --flags "-c CCCTAAA -w 2000 -s 1000 -r -g -e -m -i"
LABEL=default run --flags "-c TTAGGG -r -g -e -m -i"
