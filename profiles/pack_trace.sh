#!/bin/bash
# Kernel trace of the sharded path with ONE rank (TS_BENCH_FORCE_STRONG=1: scan + device block calling + pack, nothing
# on the wire): per-kernel times of the pack.  usage: bash profiles/pack_trace.sh <tag>   -> gpurun_out/pack_<tag>/
set -e
TAG=${1:-r03}
REPO=$(pwd)
make -s -C $REPO/teloscope_amd/csrc && make -s -C $REPO/oracle
OUT=$REPO/gpurun_out/pack_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export TS_BENCH_FORCE_STRONG=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --no-cpu-baseline --no-e2e --steps 20 --warmup 5 $BENCH_ARGS > $OUT/trace.log 2>&1
f=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $OUT/kernel_stats.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("%-70s %8s %12s %10s" % ("kernel", "calls", "avg_us", "pct"))
for r in rows[:20]:
    print("%-70s %8s %12.1f %10s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
cat $OUT/kernel_stats.txt
tail -c 600 $OUT/trace.log
