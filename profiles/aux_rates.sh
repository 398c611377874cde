#!/bin/bash
# Refreshes the auxiliary rates quoted in DESIGN.md (run on the GPU box through gpurun):
#   host API rate, read-filter rate, device block calling at 3 Gb, full-size verification.
set -e
cd "$(dirname "$0")/.."
python3 profiles/host_api_rate.py > gpurun_out/host_api_rate.txt 2>&1
python3 profiles/read_filter_rate.py > gpurun_out/read_filter_rate.txt 2>&1
python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --verify > gpurun_out/bench_verify_3gb.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/prof_blocks -- python3 $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --blocks > $OLDPWD/gpurun_out/bench_blocks.log 2>&1
