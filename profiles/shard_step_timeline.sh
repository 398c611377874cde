#!/bin/bash
# kernel timeline of the sharded step (profiles/pack_abl_time.py: emitting scan + pack beside the next scan, four slots) under
# rocprofv3 --kernel-trace: start, duration, hardware queue and stream of every kernel of three steps, for each of the script's passes
# usage: bash profiles/shard_step_timeline.sh <tag> [ENV=val ...]  -> gpurun_out/shardtl_<tag>.txt
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/shardtl_$TAG
mkdir -p $OUT
for e in "$@"; do export "$e"; done
export PACK=${PACK:-1}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $REPO/profiles/pack_abl_time.py 24 > $OUT/trace.log 2>&1
python3 - $OUT > $REPO/gpurun_out/shardtl_$TAG.txt <<'PY'
import csv, glob, os, re, sys
root = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
scans = [i for i, r in enumerate(rows) if "ts_scan_tiles" in r[2] and r[1] - r[0] > float(os.environ.get("MIN_SCAN_US", "400")) * 1e3]
# passes: separated by gaps of more than 50 ms between scans
passes, cur = [], [scans[0]]
for a, b in zip(scans, scans[1:]):
    if rows[b][0] - rows[a][1] > 50e6:
        passes.append(cur); cur = []
    cur.append(b)
passes.append(cur)
for pi, ps in enumerate(passes):
    tail = ps[-24:]
    per = (rows[tail[-1]][0] - rows[tail[0]][0]) / (len(tail) - 1) / 1e6
    print("== pass %d: %d scans, last %d start every %.4f ms, mean scan %.4f ms" % (pi, len(ps), len(tail), per, sum(rows[i][1] - rows[i][0] for i in tail) / len(tail) / 1e6))
    lo, hi = tail[-6], tail[-3]
    t0 = rows[lo][0]
    for s, e, name, q, st in rows[lo:hi]:
        m = re.search(r"(ts_\w+|__amd\w+|\w+_kernel\w*)", name)
        print("%10.1f us  +%8.1f us  q%-3s s%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, st, (m.group(1) if m else name)[:40]))
print(open(os.path.join(root, "trace.log")).read()[-400:])
PY
