#!/usr/bin/env python3
"""Per-kernel averages out of a rocprofv3 --kernel-trace --stats directory: kstat.py <dir> [name substring]"""
import csv, glob, os, sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else "ts_"
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if sub in r["Name"]]
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows:
        name = r["Name"].split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:48]
        print("%-48s calls %5s avg %9.1f us  min %9.1f  max %9.1f" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
