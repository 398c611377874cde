#!/bin/bash
# round 4: the restructured list kernel — parity subset (list + strided forms), rates, stage ablation of the list form
# usage: bash profiles/general_r04.sh <tag>
set -e
TAG=${1:-a}
make -s -C teloscope_amd/csrc && make -s -C oracle
mkdir -p gpurun_out
O=gpurun_out/gen_r04_$TAG.txt
: > $O
timeout -k 10 600 python3 -m pytest tests/ -x -q -m gpu -k "generic or general or fuzz or mixed or outside or dense or packed_upload or text_pieces" > gpurun_out/gen_r04_tests_$TAG.txt 2>&1 || { tail -40 gpurun_out/gen_r04_tests_$TAG.txt; exit 1; }
tail -2 gpurun_out/gen_r04_tests_$TAG.txt >> $O
TS_TIMING=1 timeout -k 10 400 python3 profiles/general_path_rate.py ${GB:-3.0} > gpurun_out/gen_rate_$TAG.txt 2>&1 || { tail -30 gpurun_out/gen_rate_$TAG.txt; exit 1; }
grep -E "general path|gbases" gpurun_out/gen_rate_$TAG.txt >> $O
for abl in 16 32 64 96 112; do
  echo "TS_GEN_ABL=$abl" >> $O
  TS_GEN_ABL=$abl TS_TIMING=1 TS_GEN_ONLY=mixed_5_6 timeout -k 10 300 python3 profiles/general_path_rate.py 3.0 2>&1 | grep "general path" | tail -1 | sed -e 's/.*kernels alone, HIP events: \([0-9.]*\) ms.*/  kernels \1 ms/' >> $O
done
echo "TS_GEN_PREFETCH=0" >> $O
TS_GEN_PREFETCH=0 TS_TIMING=1 TS_GEN_ONLY=mixed_5_6 timeout -k 10 300 python3 profiles/general_path_rate.py 3.0 2>&1 | grep -E "general path|gbases" >> $O
cat $O
