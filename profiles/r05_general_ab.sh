#!/bin/bash
# the list kernel at five / six / seven workgroups per CU (TS_GEN_WAVES; 74 VGPRs since its descriptors became scalar loads), kernels alone
cd "$(dirname "$0")/.."
for i in 1 2; do
  for v in 5 6 7; do
    if [ $v = 5 ]; then unset TELOSCAN_LIB; else export TELOSCAN_LIB=$PWD/teloscope_amd/libteloscan_gw$v.so; fi
    for set in mixed_5_6 k14 wrapped_start_index; do
      echo "waves $v $set: $(TS_GEN_ONLY=$set TS_TIMING=1 timeout -k 10 300 python3 profiles/general_path_rate.py 3.0 2>&1 | grep -o 'kernels alone, HIP events: [0-9.]* ms' | tr '\n' ' ')"
    done
  done
done
