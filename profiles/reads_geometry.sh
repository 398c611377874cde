#!/bin/bash
# Read batches (tips mode), whole bench (scan + predicate of the previous sub-batch on a second stream):
#   one workgroup of 16 waves against two of 10 (TS_GEOMETRY); LIBS="a b": libraries profiles/abx_<a>.so ... instead
cd "$(dirname "$0")/.."
rr() {
  if [ -n "$2" ]; then export TELOSCAN_LIB=$PWD/profiles/abx_$2.so; else unset TELOSCAN_LIB; fi
  TS_GEOMETRY=$1 timeout -k 10 200 python3 bench.py --reads --no-cpu-baseline 2> /tmp/rg.err \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-5s %-3s reads  value %.0f Gbases/s  ms_per_step %.3f  scan kernel %.3f ms' % ('$1', '$2', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" || tail -n 2 /tmp/rg.err
}
for r in 1 2 3; do
  if [ -n "$LIBS" ]; then for l in $LIBS; do rr 16,6 $l; done; else rr 16,6; rr 10,6; fi
done
