set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "wide or push_ordered" > gpurun_out/wide_tests.log 2>&1 || { tail -30 gpurun_out/wide_tests.log; exit 1; }
tail -2 gpurun_out/wide_tests.log
TS_GEN_ONLY=wide_9_lengths TS_TIMING=1 timeout -k 10 280 python profiles/general_path_rate.py 3.0 > gpurun_out/wide_rate.log 2>&1 || { tail -20 gpurun_out/wide_rate.log; exit 1; }
grep -E "gbases_per_s" gpurun_out/wide_rate.log | head -4
grep "general path:" gpurun_out/wide_rate.log | tail -3 | cut -c1-330
