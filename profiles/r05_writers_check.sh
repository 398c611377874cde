set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_writers.py tests/test_cpp_mirror.py tests/test_io_selftest.py -q -x -m gpu > gpurun_out/writers_tests.log 2>&1 || { tail -30 gpurun_out/writers_tests.log; exit 1; }
tail -2 gpurun_out/writers_tests.log
bash profiles/r05_mirror_trace.sh
