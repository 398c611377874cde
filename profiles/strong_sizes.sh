#!/bin/bash
# The sharded code path (scan + device block calling + pack + exchange object) with ONE rank on the RCCL group, at the
# per-rank sizes of N = 8, 4, 2, 1 ranks of the 3 Gb assembly: per-step time, host enqueue time, step split.
# usage: bash profiles/strong_sizes.sh <tag>  -> gpurun_out/strong_<tag>.txt
set -e
TAG=${1:-r03}
make -s -C teloscope_amd/csrc && make -s -C oracle
mkdir -p gpurun_out
OUT=gpurun_out/strong_$TAG.txt
: > $OUT
for spec in "0.375 25" "0.75 50" "1.5 100" "3.0 200"; do
  set -- $spec
  TS_BENCH_FORCE_STRONG=1 python3 bench.py --no-cpu-baseline --no-e2e --no-reads --gbases $1 --contigs $2 --steps 50 --warmup 5 >> $OUT 2>&1
done
python3 - $OUT <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if line.startswith("{"):
        d = json.loads(line)
        c = d["config"]
        print("%.3f Gb: %.4f ms/step (events %.4f, host enqueue %.4f) scan kernel %.4f  split %s" % (c["bases"] / 1e9, d["ms_per_step"], c["device_ms_per_step_events"], c["host_enqueue_ms_per_step"], d["roofline"]["kernel_ms"], {k: v for k, v in c.get("step_split", {}).items() if k.endswith("_ms")}))
PY
