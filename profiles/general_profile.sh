#!/bin/bash
# rocprofv3 evidence for the GENERAL path's fused kernel (generic.hip) on the 3 Gb bench assembly,
# -p TTAGGG,TTAGG -w 1000 -s 500 -r -g -e -m -i (mixed lengths 5/6, 70 patterns): kernel trace + separate PMC passes.
# usage: bash profiles/general_profile.sh <tag>   -> gpurun_out/genprof_<tag>/summary.txt
set -e
TAG=${1:-r03}
REPO=$(pwd)
make -s -C $REPO/teloscope_amd/csrc && make -s -C $REPO/oracle
OUT=$REPO/gpurun_out/genprof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export TS_GEN_ONLY=${GEN_ONLY:-mixed_5_6} TS_TIMING=1 KERNEL_KEY=${KERNEL_KEY:-ts_general_fused}
ARGS="$REPO/profiles/general_path_rate.py 3.0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/pmc_sq -- python3 $ARGS > $OUT/pmc_sq.log 2>&1 || true
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ARGS > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ARGS > $OUT/pmc_write.log 2>&1
python3 - $OUT > $OUT/summary.txt <<'PY'
import csv, glob, os, sys
root = sys.argv[1]
KEY = os.environ.get("KERNEL_KEY", "ts_general_fused")        # ts_general_wide for the wide form (GEN_ONLY=wide_9_lengths)
def find(p): return sorted(glob.glob(os.path.join(root, "**", p), recursive=True))
print("general path, 3.0 Gb / 200 contigs, parameter set " + os.environ.get("TS_GEN_ONLY", "mixed_5_6") + " of profiles/general_path_rate.py; 4 calls of the host entry points (2 x blocks, 2 x match vectors), 13 groups of ~256 MB each")
for f in find("*kernel_trace.csv"):
    durs = {}
    for row in csv.DictReader(open(f)):
        durs.setdefault(row["Kernel_Name"], []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    print("== kernel trace:", os.path.relpath(f, root))
    for name, v in sorted(durs.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print("   %-70s n=%d total=%.2f ms avg=%.1f us" % (name[:70], len(v), sum(v) / 1e6, sum(v) / len(v) / 1e3))
    calls = 4
    fused = [v for k, v in durs.items() if KEY in k]
    if fused:
        tot = sum(fused[0]) / 1e6 / calls
        print("   %s per 3 Gb call: %.2f ms = %.1f Gbases/s resident (kernel alone)" % (KEY, tot, 3.0 / tot * 1e3))
    allk = sum(sum(v) for k, v in durs.items() if "ts_general" in k or "ts_tile" in k) / 1e6 / calls
    print("   all kernels of the path (fused + prefix sum + compaction) per call: %.2f ms = %.1f Gbases/s" % (allk, 3.0 / allk * 1e3))
tot = {}
for f in find("*counter_collection.csv"):
    acc = {}
    for row in csv.DictReader(open(f)):
        if KEY not in row.get("Kernel_Name", ""): continue
        acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    print("== counters (%s, SUM over the dispatches of one 3 Gb call):" % KEY, os.path.relpath(f, root))
    for k, v in sorted(acc.items()):
        print("   %-24s dispatches=%d sum/call=%.6g" % (k, len(v), sum(v) / 4))
        tot[k] = sum(v) / 4
if "FETCH_SIZE" in tot and "WRITE_SIZE" in tot:
    hbm = tot["FETCH_SIZE"] * 1024 * 2 + tot["WRITE_SIZE"] * 1024     # KB; gfx950: FETCH_SIZE counts 64 B requests as 32 (guide's correction)
    print("HBM bytes of " + KEY + " per 3 Gb call: fetch %.3f GB (x2 corrected) + write %.3f GB = %.3f GB = %.2f B/base"
          % (tot["FETCH_SIZE"] * 2048 / 1e9, tot["WRITE_SIZE"] * 1024 / 1e9, hbm / 1e9, hbm / 3.0e9))
PY
cat $OUT/summary.txt
grep "general path" $OUT/trace.log | tail -2
