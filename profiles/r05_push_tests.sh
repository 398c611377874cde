set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -q -k "push_ordered or general_path_block_calling or wide_pattern or general_kernels" > gpurun_out/push_tests.log 2>&1 || { tail -40 gpurun_out/push_tests.log; exit 1; }
tail -5 gpurun_out/push_tests.log
