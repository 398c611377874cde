# the whole GPU suite, then the extended parameter fuzz (1500 sets, a quarter of them wide, a second seed) on the tree with
# push-ordered streams called on the device
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests -q -m gpu -x > gpurun_out/push_suite.log 2>&1 || { tail -40 gpurun_out/push_suite.log; exit 1; }
tail -3 gpurun_out/push_suite.log
TS_FUZZ_WIDE=0.25 timeout -k 10 330 python profiles/fuzz_long.py 1500 11 > gpurun_out/push_fuzz_1500.log 2>&1 || { tail -30 gpurun_out/push_fuzz_1500.log; exit 1; }
tail -3 gpurun_out/push_fuzz_1500.log
