#!/bin/bash
# scan kernel at the N = 8 shard size (375 Mb, 25 contigs) under pinned tilings
set -e
make -s -C teloscope_amd/csrc && make -s -C oracle
mkdir -p gpurun_out
OUT=gpurun_out/small_geom_${1:-a}.txt
: > $OUT
for g in "" "10,6" "10,5" "10,4" "10,3" "16,8" "16,6" "16,4" "12,6" "8,6" "8,4"; do
  echo "TS_GEOMETRY=$g" >> $OUT
  TS_GEOMETRY=$g python3 bench.py --no-cpu-baseline --no-e2e --no-reads --gbases 0.375 --contigs 25 --steps 100 --warmup 10 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('   ms_per_step %.4f kernel %.4f tiles %d' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['config']['tiles']))
" >> $OUT || echo "   failed" >> $OUT
done
cat $OUT
