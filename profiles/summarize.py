#!/usr/bin/env python3
"""Condenses a rocprofv3 output directory (kernel-trace stats + PMC passes) into a short text
summary: per-kernel average duration, and per-dispatch counter averages for ts_scan_tiles."""
import csv
import glob
import os
import sys

root = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
TIMED = 50                          # bench.py's default --steps: its timed region is the LAST 50 scans of the run (before them:
                                    # the first / single-shot scans, W warm-ups + K steps of the unsettled protocol, the settling
                                    # scans and W warm-ups; after them, with --no-reads --no-e2e, nothing)


def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def variant(name):
    """The two builds of the scan kernel (kernels.hip, template EMIT): 'plain' leaves window records + match stream, 'emit' also
    the visible records and per-tile chain summaries (the scans of the line's scan_plus_block_calling sub-record)."""
    if "ts_scan_tiles" not in name:
        return None
    import re
    m = re.search(r"ts_scan_tiles<([^>]*)>", name)          # <FC_BYTES, PAIR_BYTES, WAVES_EU, EMIT[, FAST]>
    args = [a.strip() for a in m.group(1).split(",")] if m else []
    emit = args[3] if len(args) > 3 else ("true" if ", true>" in name else "false")
    return "emit" if emit in ("true", "1") else "plain"


for f in find("*kernel_stats.csv"):
    print("== kernel stats:", os.path.relpath(f, root))
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 8:
                print("  ", ",".join(row)[:200])
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
        if variant(name) == "plain" and len(v) > TIMED:      # the plain build: what `value` and `roofline` time
            # bench.py's roofline uses the timed launches only: the last TIMED of the run
            t = v[-TIMED:]
            print("   %-60s      timed launches only (the last %d of %d): n=%d avg=%.1f us" % ("", TIMED, len(v), len(t), sum(t) / len(t) / 1e3))
            # what bench.py itself measured in this very run (its JSON line is in trace.log): the two must agree; between
            # runs the plateau moves by a few per cent
            log = os.path.join(root, "trace.log")
            if os.path.exists(log):
                import re
                m = re.search(r'"kernel_ms": ([0-9.]+)', open(log, errors="replace").read())
                if m:
                    print("   %-60s      bench.py in the same process (HIP events, roofline.kernel_ms): %.1f us" % ("", float(m.group(1)) * 1e3))
for f in find("*counter_collection.csv"):
    acc = {"plain": {}, "emit": {}}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = variant(row.get("Kernel_Name", ""))
            if k:
                acc[k].setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for kind in ("plain", "emit"):
        if not acc[kind]:
            continue
        print("== counters (ts_scan_tiles, %s build, per dispatch avg over the full-size dispatches):" % kind, os.path.relpath(f, root))
        for k, v in sorted(acc[kind].items()):
            big = [x for x in v if x >= 0.5 * max(v)]                # (the sub-record's settling runs include small dispatches)
            print("   %-28s n=%d avg=%.6g" % (k, len(big), sum(big) / len(big)))

# HBM bytes per launch of ts_scan_tiles for bench.py's roofline.traffic: FETCH_SIZE and WRITE_SIZE (KB, separate --pmc
# passes; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream -> x2, WRITE_SIZE is exact for 16-byte
# streaming stores: /opt/skills/guides/MI355X_MICROARCH.md, HBM section), stamped with the hash of the kernel source
# they were measured for — bench.py reports them only while that hash still matches.
import hashlib
import json

fetch = write = None
emit_fetch = emit_write = None
for f in find("*counter_collection.csv"):
    acc = {"plain": {}, "emit": {}}
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = variant(row.get("Kernel_Name", ""))
            if k:
                acc[k].setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    if "FETCH_SIZE" in acc["plain"]:
        v = acc["plain"]["FETCH_SIZE"][5:] or acc["plain"]["FETCH_SIZE"]
        fetch = sum(v) / len(v)
    if "WRITE_SIZE" in acc["plain"]:
        v = acc["plain"]["WRITE_SIZE"][5:] or acc["plain"]["WRITE_SIZE"]
        write = sum(v) / len(v)
    if "FETCH_SIZE" in acc["emit"]:
        v = acc["emit"]["FETCH_SIZE"]
        v = [x for x in v if x >= 0.5 * max(v)]
        emit_fetch = sum(v) / len(v)
    if "WRITE_SIZE" in acc["emit"]:
        v = acc["emit"]["WRITE_SIZE"]
        v = [x for x in v if x >= 0.5 * max(v)]
        emit_write = sum(v) / len(v)
if fetch is not None and write is not None:
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(repo, "teloscope_amd", "csrc", "kernels.hip")
    out = {"kernel": "ts_scan_tiles", "workload": "configs[1] 3.0 Gb / 200 contigs, bench.py defaults (timed launches)",
           "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
           "correction": "gfx950: FETCH_SIZE reports half of a wide coalesced read stream -> x2 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE exact",
           "hbm_bytes_per_launch": int(round((2 * fetch + write) * 1024)),
           "emit_build_hbm_bytes_per_launch": int(round((2 * emit_fetch + emit_write) * 1024)) if emit_fetch is not None and emit_write is not None else None,
           "kernels_hip_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest(),
           "source": "separate rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of `python3 bench.py --no-cpu-baseline --no-e2e` (profiles/run_profile.sh)"}
    with open(os.path.join(root, "pmc_traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("== pmc_traffic.json:", json.dumps(out))
