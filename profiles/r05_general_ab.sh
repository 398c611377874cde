#!/bin/bash
# the list kernel after its descriptors became scalar loads and its stores inline asm: general-path parity tests, the long parameter fuzz,
# then kernel times of three parameter sets (TS_TIMING: kernels alone, HIP events) against the library before (TELOSCAN_LIB)
cd "$(dirname "$0")/.."
set -o pipefail
timeout -k 10 800 python3 -m pytest tests/test_gpu_parity.py -x -q -k "general or generic or wide or mixed" 2>&1 | tail -2 || exit 1
timeout -k 10 300 python3 -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2 || exit 1
for i in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export TELOSCAN_LIB=$PWD/teloscope_amd/libteloscan_pretrim.so; else unset TELOSCAN_LIB; fi
    for set in mixed_5_6 k14 wrapped_start_index; do
      echo "$v $set: $(TS_GEN_ONLY=$set TS_TIMING=1 timeout -k 10 300 python3 profiles/general_path_rate.py 3.0 2>&1 | grep -o 'kernels alone, HIP events: [0-9.]* ms' | tr '\n' ' ')"
    done
  done
done
