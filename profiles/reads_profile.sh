#!/bin/bash
# rocprofv3 evidence for the read step (configs[3]: bench.py --reads at 5e5 reads per step): kernel trace + one PMC pass
# usage: bash profiles/reads_profile.sh <tag>  -> gpurun_out/readsprof_<tag>/summary.txt
set -e
TAG=${1:-r04}
REPO=$(pwd)
make -s -C $REPO/teloscope_amd/csrc && make -s -C $REPO/oracle
OUT=$REPO/gpurun_out/readsprof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --reads --n-reads 5e5 --no-cpu-baseline --steps 20 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
    --output-format csv -d $OUT/pmc -- python3 $ARGS > $OUT/pmc.log 2>&1 || true
python3 - $OUT > $OUT/summary.txt <<'PY'
import csv, glob, os, sys, json
root = sys.argv[1]
print("bench.py --reads --n-reads 5e5 --steps 20 --warmup 3 (configs[3] at 500 k HiFi-like reads per step, 7.5 Gb resident), rocprofv3 --kernel-trace")
for f in sorted(glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True)):
    durs = {}
    for row in csv.DictReader(open(f)):
        durs.setdefault(row["Kernel_Name"], []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for name, v in sorted(durs.items(), key=lambda kv: -sum(kv[1]))[:10]:
        big = sorted(v)[-20:]
        print("   %-72s n=%-4d total=%9.2f ms  mean of the 20 longest %.1f us" % (name[:72], len(v), sum(v) / 1e6, sum(big) / len(big) / 1e3))
last = [l for l in open(os.path.join(root, "trace.log")) if l.startswith("{")]
if last:
    d = json.loads(last[-1])
    print("bench line under the kernel trace:", json.dumps({k: d[k] for k in ("metric", "value", "unit", "ms_per_step", "roofline") if k in d}))
PY
python3 $REPO/profiles/kernel_counters.py $OUT/pmc >> $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
