#!/bin/bash
# round 5's evidence in one call: the whole GPU suite, the rocprofv3 passes of bench.py (kernel trace + counters + HBM traffic), a default
# bench.py line, the sharded step at the per-rank sizes of 8 / 4 / 2 / 1 ranks, the read filter line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r05_final
( time timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q ) > gpurun_out/r05_final/gputests.txt 2>&1; rc=$?
tail -4 gpurun_out/r05_final/gputests.txt
[ $rc -ne 0 ] && exit $rc
bash profiles/run_profile.sh r05 > gpurun_out/r05_final/profile.log 2>&1 || exit 1
tail -30 gpurun_out/prof_r05/summary.txt
timeout -k 10 600 python3 bench.py > gpurun_out/r05_final/bench_default.json 2> gpurun_out/r05_final/bench_default.err || exit 1
bash profiles/strong_sizes.sh r05 > gpurun_out/r05_final/strong.log 2>&1; tail -4 gpurun_out/r05_final/strong.log
timeout -k 10 600 python3 bench.py --reads --n-reads 5e6 --no-cpu-baseline > gpurun_out/r05_final/bench_reads_5m.json 2> gpurun_out/r05_final/bench_reads_5m.err || exit 1
