#!/bin/bash
# general path: list form vs strided form of the fused kernel (TS_GEN_LIST=0), parity tests under both, kernel times
set -e
make -s -C teloscope_amd/csrc && make -s -C oracle
mkdir -p gpurun_out
OUT=gpurun_out/gen_ab_${1:-a}.txt
: > $OUT
for L in 1 0; do
  echo "TS_GEN_LIST=$L" >> $OUT
  TS_GEN_LIST=$L timeout -k 10 500 python3 -m pytest tests/ -x -q -m gpu -k "generic or general or fuzz or mixed or outside or dense" 2>&1 | tail -2 >> $OUT
  TS_GEN_LIST=$L TS_TIMING=1 timeout -k 10 400 python3 profiles/general_path_rate.py 3.0 2>&1 | grep "kernels alone" | sed -e "s/.*HIP events: \([0-9.]*\) ms.*/\1/" | tr "\n" " " >> $OUT
  echo >> $OUT
done
cat $OUT
