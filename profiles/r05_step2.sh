#!/bin/bash
# full GPU suite, then the sharded step at the per-rank sizes of 8 / 4 / 2 / 1 ranks for round 4's tree and this one (same box)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05_step2
( time timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q ) > gpurun_out/r05_step2/gputests.txt 2>&1; rc=$?
tail -6 gpurun_out/r05_step2/gputests.txt
[ $rc -ne 0 ] && exit $rc
(cd ab_old && bash profiles/strong_sizes.sh r04 > /dev/null 2>&1; tail -4 gpurun_out/strong_r04.txt) | tee gpurun_out/r05_step2/strong_r04.txt
bash profiles/strong_sizes.sh r05 > /dev/null 2>&1; tail -4 gpurun_out/strong_r05.txt | tee gpurun_out/r05_step2/strong_r05.txt
