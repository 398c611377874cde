#!/bin/bash
# the scan's packed window records with the ten-bit layout by constants: shard parity, then the sharded step new against old
cd "$(dirname "$0")/.."
set -o pipefail
timeout -k 10 400 python3 -m pytest tests/test_gpu_shard_results.py -x -q 2>&1 | tail -2 || exit 1
for i in 1 2 3; do
  LABEL=new STREAMS=probe EV=lib TIME_EVERY=0 python3 profiles/pack_abl_time.py 60 2>/dev/null
  LABEL=old TELOSCAN_LIB=$PWD/teloscope_amd/libteloscan_prewin.so STREAMS=probe EV=lib TIME_EVERY=0 python3 profiles/pack_abl_time.py 60 2>/dev/null
done
