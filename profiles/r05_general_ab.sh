#!/bin/bash
# the wide form after its list reads stopped being flat accesses: wide parity + the parameter fuzz, kernels alone new against old
cd "$(dirname "$0")/.."
set -o pipefail
timeout -k 10 800 python3 -m pytest tests/test_gpu_parity.py -x -q -k "general or generic or wide or mixed" 2>&1 | tail -2 || exit 1
timeout -k 10 300 python3 -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -2 || exit 1
for i in 1 2; do
  for v in new old; do
    if [ $v = old ]; then export TELOSCAN_LIB=$PWD/teloscope_amd/libteloscan_pretrim.so; else unset TELOSCAN_LIB; fi
    echo "$v wide_9_lengths: $(TS_GEN_ONLY=wide_9_lengths TS_TIMING=1 timeout -k 10 300 python3 profiles/general_path_rate.py 3.0 2>&1 | grep -o 'kernels alone, HIP events: [0-9.]* ms' | tr '\n' ' ')"
  done
done
