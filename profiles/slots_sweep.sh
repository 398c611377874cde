#!/bin/bash
# one-rank sharded step at the N = 8 and N = 1 sizes: buffer slots x pack streams
# usage: bash profiles/slots_sweep.sh <tag> -> gpurun_out/slots_<tag>.txt
set -e
make -s -C teloscope_amd/csrc && make -s -C oracle
mkdir -p gpurun_out
OUT=gpurun_out/slots_${1:-a}.txt
: > $OUT
for spec in "0.375 25" "3.0 200"; do
  for cfg in "2 1" "3 2" "3 3" "4 2" "4 4" "6 3" "6 6"; do
    set -- $spec $cfg
    echo "gbases $1 slots $3 pack_streams $4" >> $OUT
    TS_BENCH_SLOTS=$3 TS_BENCH_PACK_STREAMS=$4 TS_BENCH_FORCE_STRONG=1 python3 bench.py --no-cpu-baseline --no-e2e --no-reads --gbases $1 --contigs $2 --steps 60 --warmup 6 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step %.4f  scan kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))
" >> $OUT
  done
done
cat $OUT
