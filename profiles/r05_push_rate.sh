# the two kinds of parameter set whose blocks used to be called on the host (a stream not in position order; the wide form),
# device route against the host route (TS_GEN_HOST_BLOCKS=1), same box
set -e
cd $GRAFT_REPO_ROOT
for set in wide_9_lengths mixed_6_14; do
  for host in 0 1; do
    echo "== $set TS_GEN_HOST_BLOCKS=$host"
    TS_GEN_ONLY=$set TS_GEN_HOST_BLOCKS=$host TS_TIMING=1 timeout -k 10 280 python profiles/general_path_rate.py 3.0 > gpurun_out/push_rate_${set}_$host.log 2>&1 || { tail -20 gpurun_out/push_rate_${set}_$host.log; exit 1; }
    grep -E "gbases_per_s|seconds" gpurun_out/push_rate_${set}_$host.log | head -4
    grep "general path:" gpurun_out/push_rate_${set}_$host.log | tail -3 | cut -c1-420
  done
done
