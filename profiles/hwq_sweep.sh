#!/bin/bash
# one-rank sharded step: hardware queues (GPU_MAX_HW_QUEUES: HIP streams share them round-robin; kernels of two streams on one
# queue run one after the other) x buffer slots x pack streams
set -e
make -s -C teloscope_amd/csrc && make -s -C oracle
mkdir -p gpurun_out
OUT=gpurun_out/hwq_${1:-a}.txt
: > $OUT
for spec in "0.375 25" "3.0 200"; do
  for q in 4 8 16; do
  for cfg in "3 2" "4 2" "4 4"; do
    set -- $spec $cfg
    echo "gbases $1 hwq $q slots $3 pack_streams $4" >> $OUT
    GPU_MAX_HW_QUEUES=$q TS_BENCH_SLOTS=$3 TS_BENCH_PACK_STREAMS=$4 TS_BENCH_FORCE_STRONG=1 python3 bench.py --no-cpu-baseline --no-e2e --no-reads --gbases $1 --contigs $2 --steps 60 --warmup 6 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step %.4f  scan kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))
" >> $OUT
  done
  done
done
cat $OUT
