# extended fuzzes on the round's last tree (the suite runs shorter ones with other seeds): parameter sets (a quarter wide), read-filter
# sets, shard layouts
cd $GRAFT_REPO_ROOT
TS_FUZZ_WIDE=0.25 timeout -k 10 420 python profiles/fuzz_long.py 1500 31 > gpurun_out/fuzz_final_long.log 2>&1; tail -1 gpurun_out/fuzz_final_long.log
timeout -k 10 300 python profiles/fuzz_reads.py 300 31 > gpurun_out/fuzz_final_reads.log 2>&1; tail -1 gpurun_out/fuzz_final_reads.log
timeout -k 10 300 python profiles/fuzz_shards.py 300 31 > gpurun_out/fuzz_final_shards.log 2>&1; tail -1 gpurun_out/fuzz_final_shards.log
