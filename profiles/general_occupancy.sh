#!/bin/bash
# list kernel: grid size against what is resident (TS_GEN_WGS; 0 = what the occupancy query says); usage: bash profiles/general_occupancy.sh <tag>
set -e
TAG=${1:-a}
mkdir -p gpurun_out
O=gpurun_out/gen_occ_$TAG.txt
: > $O
make -s -C oracle && make -s -C teloscope_amd/csrc
for WGS in 0 1024 1280 1536 1792 2048 3072; do
  echo "TS_GEN_WGS=$WGS" >> $O
  TS_GEN_WGS=$WGS TS_TIMING=1 timeout -k 10 300 python3 profiles/general_path_rate.py 3.0 2>&1 | grep "kernels alone" | sed -e 's/.*kernels alone, HIP events: \([0-9.]*\) ms.*/\1/' | tr "\n" " " >> $O
  echo >> $O
done
cat $O
