#!/bin/bash
# the whole GPU suite, then a default bench.py line and the strong sizes, on the round's last tree
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05_final2
( time timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q ) > gpurun_out/r05_final2/gputests.txt 2>&1; rc=$?
tail -4 gpurun_out/r05_final2/gputests.txt
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 bench.py > gpurun_out/r05_final2/bench_default.json 2> gpurun_out/r05_final2/bench_default.err || exit 1
bash profiles/strong_sizes.sh r05 > gpurun_out/r05_final2/strong.log 2>&1; tail -4 gpurun_out/r05_final2/strong.log
