#!/bin/bash
# where the upload stage's time goes (TS_TIMING=1 TS_STAGE_TIMING=1); usage: bash profiles/stage_timing.sh <tag>
set -e
TAG=${1:-a}
make -s -C teloscope_amd/csrc && make -s -C oracle
mkdir -p gpurun_out
for M in 0 3; do
TS_PACK_MODE=$M TS_TIMING=1 TS_STAGE_TIMING=1 timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --no-reads --no-cpu-baseline > gpurun_out/stage_timing_${TAG}_$M.json 2> gpurun_out/stage_timing_${TAG}_$M.txt
grep -E "upload_pieces|ts_scan_segments" gpurun_out/stage_timing_${TAG}_$M.txt | head -40
done
