#!/bin/bash
# Read batches (tips mode): one workgroup of 16 waves against two of 10, whole bench (scan + predicate on a second stream).
cd "$(dirname "$0")/.."
rr() {
  TS_GEOMETRY=$1 timeout -k 10 200 python3 bench.py --reads --no-cpu-baseline 2> /tmp/rg.err \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-5s reads  value %.0f Gbases/s  ms_per_step %.3f  scan kernel %.3f ms' % ('$1', d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))" || tail -n 2 /tmp/rg.err
}
for r in 1 2 3 4; do rr 16,6; rr 10,6; done
