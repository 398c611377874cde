#!/bin/bash
# emit_geometry.sh at the N = 4 and N = 2 per-rank sizes; usage: bash profiles/emit_geometry2.sh <tag>
set -e
TAG=${1:-a}
make -s -C teloscope_amd/csrc && make -s -C oracle
mkdir -p gpurun_out
OUT=gpurun_out/emit_geometry2_$TAG.txt
: > $OUT
for G in "" "16,8" "" "16,8"; do
  for spec in "0.75 50" "1.5 100" "3.0 200"; do
    set -- $spec
    echo "TS_GEOMETRY=$G $1 Gb" >> $OUT
    TS_GEOMETRY=$G TS_BENCH_FORCE_STRONG=1 timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-e2e --no-reads --gbases $1 --contigs $2 --steps 50 --warmup 5 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); c=d['config']
        print('   %.4f ms/step, scan kernel %.4f' % (d['ms_per_step'], d['roofline']['kernel_ms']))" >> $OUT || echo "   failed" >> $OUT
  done
done
cat $OUT
