#!/bin/bash
# Kernel time per forced tiling (TS_GEOMETRY = waves,chunks) — the measurements behind plan_geometry's model.
cd "$(dirname "$0")/.."
FLAGS=${FLAGS:-"-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -x 1 -w 1000 -s 500 -r -g -e -m -i"}
for g in ${GEOMS:-16,4 16,5 16,6 16,7 16,8}; do
  if [ "$g" = auto ]; then unset TS_GEOMETRY; else export TS_GEOMETRY=$g; fi
  TS_TIMING=1 timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 30 --warmup 3 --flags "$FLAGS" 2> /tmp/geom.err | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$g', 'kernel %.4f ms' % d['roofline']['kernel_ms'])"
  grep -a ts_batch_create /tmp/geom.err | head -1 | sed 's/.*pair table), //'
done
