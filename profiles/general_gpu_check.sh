#!/bin/bash
# general path: parity subset + rate.  usage: bash profiles/general_gpu_check.sh <tag>
set -e
TAG=${1:-a}
make -s -C teloscope_amd/csrc && make -s -C oracle
mkdir -p gpurun_out
timeout -k 10 500 python3 -m pytest tests/ -x -q -m gpu -k "generic or general or fuzz or mixed or outside" > gpurun_out/gen_$TAG.txt 2>&1 || { tail -30 gpurun_out/gen_$TAG.txt; exit 1; }
tail -3 gpurun_out/gen_$TAG.txt
TS_TIMING=1 timeout -k 10 400 python3 profiles/general_path_rate.py ${GB:-3.0} > gpurun_out/gen_rate_$TAG.txt 2>&1 || { tail -30 gpurun_out/gen_rate_$TAG.txt; exit 1; }
grep -E "general path|gbases" gpurun_out/gen_rate_$TAG.txt | tail -20
