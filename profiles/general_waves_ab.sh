#!/bin/bash
# list kernel: register budget (waves per SIMD of __launch_bounds__) A/B; usage: bash profiles/general_waves_ab.sh <tag>
set -e
TAG=${1:-a}
mkdir -p gpurun_out
O=gpurun_out/gen_waves_$TAG.txt
: > $O
make -s -C oracle
for WV in 4 5 6; do
  touch teloscope_amd/csrc/generic.hip
  make -s -C teloscope_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wextra -ffp-contract=off --offload-arch=gfx950 -DTS_GEN_WAVES=$WV"
  echo "TS_GEN_WAVES=$WV" >> $O
  TS_TIMING=1 timeout -k 10 300 python3 profiles/general_path_rate.py 3.0 2>&1 | grep "kernels alone" | sed -e 's/.*kernels alone, HIP events: \([0-9.]*\) ms.*/\1/' | tr "\n" " " >> $O
  echo >> $O
done
touch teloscope_amd/csrc/generic.hip
make -s -C teloscope_amd/csrc
timeout -k 10 600 python3 -m pytest tests/ -x -q -m gpu -k "generic or general or fuzz or mixed or outside or dense or packed_upload or text_pieces" 2>&1 | tail -2 >> $O
cat $O
