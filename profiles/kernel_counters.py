"""Per-kernel means of the counters in a rocprofv3 --pmc run's counter_collection.csv (the 20 largest dispatches of each of our
kernels: the full-size ones).  python profiles/kernel_counters.py <dir>"""
import collections
import csv
import glob
import sys

for f in glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(f) as fh:
        for r in csv.DictReader(fh):
            n = r["Kernel_Name"]
            k = "scan" if "ts_scan_tiles" in n else ("pred_long" if "predicate_long" in n else ("pred" if "terminal_predicate" in n else None))
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in ("pred", "pred_long", "scan"):
        for c, v in sorted(acc[k].items()):
            big = sorted(v)[-20:]
            print("%-10s %-24s n=%-5d mean of the 20 largest %.4g" % (k, c, len(v), sum(big) / len(big)))
