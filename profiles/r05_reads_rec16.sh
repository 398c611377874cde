#!/bin/bash
# the read predicate's blocks in flight per thread (TS_PRED_NB) and waves per SIMD with 16-bit records: bench.py's reads sub-record (5e5 reads)
cd "$(dirname "$0")/.."
line() { timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['reads']
print('$1 reads step', r['ms_per_step'], 'frac', r['roofline']['frac'], r['roofline']['over'][17:47])"; }
for i in 1 2; do
  unset TELOSCAN_LIB; line "NB=4 (default)"
  for v in pnb2 pnb8 pw6; do export TELOSCAN_LIB=$PWD/teloscope_amd/libteloscan_$v.so; line $v; done
done
