#!/bin/bash
# kernel timeline of the last read steps (bench.py --reads at 5e5 reads per step) under rocprofv3 --kernel-trace: start, duration, queue
# usage: bash profiles/reads_timeline.sh <tag> [ENV=val ...]  -> gpurun_out/readstl_<tag>.txt
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/readstl_$TAG
mkdir -p $OUT
for e in "$@"; do export "$e"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $REPO/bench.py --reads --n-reads 5e5 --no-cpu-baseline --steps 12 --warmup 3 > $OUT/trace.log 2>&1
python3 - $OUT > $REPO/gpurun_out/readstl_$TAG.txt <<'PY'
import csv, glob, os, sys, json
root = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
import re
scans = [i for i, r in enumerate(rows) if "ts_scan_tiles" in r[2]]
big = [i for i in scans if rows[i][1] - rows[i][0] > 1000000]      # the resident steps: scans of more than a millisecond
lo, hi = big[len(big) // 2], big[len(big) // 2 + 4]
t0 = rows[lo][0]
for s, e, name, q, st in rows[lo:hi + 8]:
    m = re.search(r"(ts_\w+|__amd\w+|\w+_kernel\w*)", name)
    short = (m.group(1) if m else name)[:40]
    print("%10.1f us  +%8.1f us  q%-3s s%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, st, short))
last = [l for l in open(os.path.join(root, "trace.log")) if l.startswith("{")]
if last:
    d = json.loads(last[-1]); print("bench line:", d["ms_per_step"], d["roofline"]["kernel_ms"])
PY
cat $REPO/gpurun_out/readstl_$TAG.txt
