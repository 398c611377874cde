#!/bin/bash
# pcie_inclusive legs of bench.py with the packed upload on and off, staging thread counts
set -e
make -s -C teloscope_amd/csrc && make -s -C oracle
mkdir -p gpurun_out
OUT=gpurun_out/packed_${1:-a}.txt
: > $OUT
IFS=";" read -ra LIST <<< "${CFGS:-0 8;1 8;1 12;1 16}"
for cfg in "${LIST[@]}"; do
  IFS=" " read -r a b g <<< "$cfg"; set -- $a $b; export TS_GROUP_MB=${g:-256}; echo "TS_GROUP_MB=$TS_GROUP_MB" >> $OUT
  echo "TS_PACKED_UPLOAD=$1 TS_STAGE_THREADS=$2" >> $OUT
  TS_TIMING=1 TS_PACKED_UPLOAD=$1 TS_STAGE_THREADS=$2 python3 bench.py --no-cpu-baseline --no-reads --steps 5 --warmup 2 2> gpurun_out/packed_err.txt | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l)['pcie_inclusive']
        print('   blocks %.1f  matches %.1f  multi %.1f Gbases/s' % (d['blocks_windows_counts']['gbases_per_s'], d['with_match_vectors']['gbases_per_s'], d['writer_view_multi']['gbases_per_s']))
" >> $OUT
  grep "ts_scan_segments" gpurun_out/packed_err.txt | tail -4 | cut -c1-330 >> $OUT
done
cat $OUT
