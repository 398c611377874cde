#!/bin/bash
# read filter with 16-bit records: parity (read-filter tests, the read fuzz), then the reads step with 16- and 32-bit records, interleaved
cd "$(dirname "$0")/.."
set -o pipefail
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "read" 2>&1 | tail -2 || exit 1
timeout -k 10 300 python3 -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2 || exit 1
for i in 1 2 3; do
  for v in 16 32; do
    if [ $v = 32 ]; then export TS_REC32=1; else unset TS_REC32; fi
    timeout -k 10 300 python3 bench.py --reads --n-reads 1e6 --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('records $v bits: step', d['ms_per_step'], 'ms  value', d['value'], ' scan kernel', r['kernel_ms'], r['frac'], ' whole step frac', round(d['config'].get('whole_step_frac', 0), 4) if 'whole_step_frac' in d['config'] else '')"
  done
done
unset TS_REC32
timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['reads']
print('default line reads:', r['ms_per_step'], r['roofline']['frac'], r['roofline']['over'][:90])"
