#!/bin/bash
# Where the fused general kernel's time goes: stages switched off by TS_GEN_ABL (results are then wrong: timing only).
set -e
make -s -C teloscope_amd/csrc && make -s -C oracle
mkdir -p gpurun_out
OUT=gpurun_out/gen_ablate_${1:-a}.txt
: > $OUT
for abl in 0 1 2 4 3 6 7 8; do
  echo "TS_GEN_ABL=$abl" >> $OUT
  TS_GEN_ABL=$abl TS_TIMING=1 TS_GEN_ONLY=mixed_5_6 timeout -k 10 300 python3 profiles/general_path_rate.py 3.0 2>&1 | grep "general path" | tail -2 | sed -e 's/.*kernels alone, HIP events: \([0-9.]*\) ms.*/  kernels \1 ms/' >> $OUT
done
cat $OUT
