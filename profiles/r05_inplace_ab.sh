# blocks-only calls of the general path: block calling reads the records in the tiles' slots (default) against the dense stream made
# first (TS_GEN_COMPACT=1), same box; parity first
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -k "general or wide or push_ordered or generic or fuzz" > gpurun_out/inplace_tests.log 2>&1 || { tail -30 gpurun_out/inplace_tests.log; exit 1; }
tail -2 gpurun_out/inplace_tests.log
for rep in 1 2; do for cmp in 0 1; do
  echo "== TS_GEN_COMPACT=$cmp"
  TS_GEN_COMPACT=$cmp TS_GEN_ONLY=mixed_5_6 TS_TIMING=1 timeout -k 10 200 python profiles/general_path_rate.py 3.0 > gpurun_out/inplace_rate_$cmp.log 2>&1
  grep "gbases_per_s" gpurun_out/inplace_rate_$cmp.log | head -1
  grep "kernels alone" gpurun_out/inplace_rate_$cmp.log | sed -n 2p | cut -c1-160
done; done
