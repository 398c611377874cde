cd $GRAFT_REPO_ROOT && timeout -k 10 300 python profiles/debug/push_diff.py $1 > gpurun_out/push_diff_$1.log 2>&1; tail -30 gpurun_out/push_diff_$1.log
