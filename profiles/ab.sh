#!/bin/bash
# A/B of two builds in one gpurun call (box-to-box variance is larger than most kernel changes):
#   teloscope_amd/libteloscan_old.so (build of the previous commit) against teloscope_amd/libteloscan.so
cd "$(dirname "$0")/.."
for i in 1 2; do
  echo "--- old"; TELOSCAN_LIB=$PWD/teloscope_amd/libteloscan_old.so profiles/quick3.sh
  echo "--- new"; profiles/quick3.sh
done
