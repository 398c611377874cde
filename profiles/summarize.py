#!/usr/bin/env python3
"""Condenses a rocprofv3 output directory (kernel-trace stats + PMC passes) into a short text
summary: per-kernel average duration, and per-dispatch counter averages for ts_scan_tiles."""
import csv
import glob
import os
import sys

root = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


for f in find("*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, root))
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 8:
                print("  ", ",".join(row))
for f in find("*kernel_trace.csv"):
    durs = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            d = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
            durs.setdefault(name, []).append(d)
    print("== kernel trace:", os.path.relpath(f, root))
    for name, v in sorted(durs.items(), key=lambda kv: -sum(kv[1]))[:6]:
        print("   %-60s n=%d avg=%.1f us min=%.1f us max=%.1f us" % (name[:60], len(v), sum(v) / len(v) / 1e3,
                                                              min(v) / 1e3, max(v) / 1e3))
        if "ts_scan_tiles" in name and len(v) > 5:
            # bench.py runs 5 untimed warm-up launches first; its roofline uses the timed ones
            t = v[5:]
            print("   %-60s      timed launches only (first 5 dropped): n=%d avg=%.1f us" % ("", len(t), sum(t) / len(t) / 1e3))
for f in find("*counter_collection.csv"):
    acc = {}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            if "ts_scan_tiles" not in row.get("Kernel_Name", ""):
                continue
            acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    print("== counters (ts_scan_tiles, per dispatch avg):", os.path.relpath(f, root))
    for k, v in sorted(acc.items()):
        print("   %-28s n=%d avg=%.6g" % (k, len(v), sum(v) / len(v)))
