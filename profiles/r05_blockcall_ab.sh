#!/bin/bash
# terminal walks forced inline (no scratch, no flat addressing): block-calling parity, then the pack alone and the sharded step, new against old
cd "$(dirname "$0")/.."
set -o pipefail
timeout -k 10 600 python3 -m pytest tests/test_gpu_shard_results.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2 || exit 1
for i in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export TELOSCAN_LIB=$PWD/teloscope_amd/libteloscan_pretrim.so; else unset TELOSCAN_LIB; fi
    timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-e2e --no-reads 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['scan_plus_block_calling']
print('$v 3 Gb: plain', d['ms_per_step'], 'sharded', s['ms_per_step'], round(s['ms_per_step']/d['ms_per_step'],3), 'emit alone', s['emitting_scan_alone_ms'], 'pack alone', s['block_calling_and_pack_alone_ms'])"
    TS_BENCH_FORCE_STRONG=1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-e2e --no-reads --gbases 0.375 --contigs 25 --steps 50 --warmup 5 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
print('$v 0.375 Gb:', d['ms_per_step'], c.get('step_split'))"
  done
done
