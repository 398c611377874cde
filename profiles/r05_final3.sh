# validation of the round's last tree: the whole GPU suite, the default bench line (the sharded step's kernels are the MODE 0
# instantiations of blockcall.hip: their times must not have moved), the general-path rates
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 800 python -m pytest tests -q -m gpu -x > gpurun_out/final3_suite.log 2>&1 || { tail -40 gpurun_out/final3_suite.log; exit 1; }
tail -2 gpurun_out/final3_suite.log
timeout -k 10 400 python bench.py > gpurun_out/final3_bench.json 2> gpurun_out/final3_bench.err || { tail -20 gpurun_out/final3_bench.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/final3_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "roofline", d["roofline"]["frac"])
print("scan_plus_block_calling", json.dumps(d.get("scan_plus_block_calling"))[:600])
PY
timeout -k 10 280 python profiles/general_path_rate.py 3.0 > gpurun_out/final3_general.json 2>/dev/null
python - <<'PY'
import json
d = json.load(open("gpurun_out/final3_general.json"))
for k, v in d["results"].items():
    print(k, v["blocks_windows_counts"]["gbases_per_s"], v["with_match_vectors"]["gbases_per_s"], v["blocks_windows_counts"]["matches"])
PY
