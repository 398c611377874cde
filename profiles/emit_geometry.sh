#!/bin/bash
# the sharded step (emitting scan + device block calling + pack, one rank on the RCCL group) against the scan's tiling
# (TS_GEOMETRY=waves,chunks): is the plain build's best tiling also the emitting build's?   usage: bash profiles/emit_geometry.sh <tag>
set -e
TAG=${1:-a}
make -s -C teloscope_amd/csrc && make -s -C oracle
mkdir -p gpurun_out
OUT=gpurun_out/emit_geometry_$TAG.txt
: > $OUT
for G in "" "10,6" "10,5" "16,6" "16,8" "12,6" "16,7"; do
  for spec in "0.375 25" "3.0 200"; do
    set -- $spec
    echo "TS_GEOMETRY=$G $1 Gb" >> $OUT
    TS_GEOMETRY=$G TS_BENCH_FORCE_STRONG=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-e2e --no-reads --gbases $1 --contigs $2 --steps 50 --warmup 5 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); c=d['config']
        print('   %.4f ms/step, scan kernel %.4f, split %s' % (d['ms_per_step'], d['roofline']['kernel_ms'], {k: v for k, v in c.get('step_split', {}).items() if k.endswith('_ms')}))" >> $OUT || echo "   failed" >> $OUT
  done
done
cat $OUT
