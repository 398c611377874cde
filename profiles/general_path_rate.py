#!/usr/bin/env python3
"""Rate of the GENERAL path (generic.hip: parameter sets the tiled kernel does not take) on the bench assembly:
mixed-length pattern set -p TTAGGG,TTAGG (two matches per position possible), -w 1000 -s 500 -r -g -e -m -i, through
the host entry points (host ASCII in, host results out).  Run on the GPU box:
    TS_TIMING=1 python3 profiles/general_path_rate.py [gbases] [contigs]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
import teloscope_amd as ta  # noqa: E402
from teloscope_amd import _capi as K  # noqa: E402
from teloscope_amd.cli import parse_cli, user_input  # noqa: E402

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
ncontig = int(sys.argv[2]) if len(sys.argv) > 2 else 200
out = {}
for name, flags in (("mixed_5_6", "-p TTAGGG,TTAGG -w 1000 -s 500 -r -g -e -m -i"),
                    ("k14", "-c TTTAGGGTTTAGGG -x 1 -w 1000 -s 500 -r -g -e -m -i"),
                    ("wrapped_start_index", "-w 1000 -s 997 -r -g -e -m -i"),
                    # lengths 6 and 14 under w > s: a stream that is not in position order (written in push order by the device)
                    ("mixed_6_14", "-p TTAGGG,TTTAGGGTTTAGGG -x 0 -w 1000 -s 500 -r -g -e -m -i"),
                    # nine distinct lengths: beyond the table forms, the wide form (ts_general_wide)
                    ("wide_9_lengths", "-x 0 -p TTAG,TTAGG,TTAGGG,TTTAGGG,TTTTAGGG,TTAGGGTTA,TTAGGGTTAG,TTAGGGTTAGG,TTAGGGTTAGGG -w 1000 -s 500 -r -g -e -m -i")):
    if os.environ.get("TS_GEN_ONLY") and os.environ["TS_GEN_ONLY"] != name:
        continue
    opts = parse_cli("x.fa " + flags)
    tel = ta.Teloscope(user_input(opts, device=0))
    assert not tel.usesFastPath()
    L = K.lib()
    total = int(gb * 1e9)
    lens = bench.contig_lengths(total, ncontig, 42)
    offs, off = [], 0
    for n in lens:
        offs.append(off)
        off += (n + 15) & ~15
    dev = torch.device("cuda", 0)
    buf = torch.zeros(off + 4096, dtype=torch.uint8, device=dev)
    bench.fill_synthetic(buf, offs, lens, 42, dev)
    host = buf.cpu().numpy()
    del buf
    n = len(lens)
    segs = (K.SegmentIn * n)()
    for i in range(n):
        segs[i].seq = C.cast(C.c_void_p(host.ctypes.data + offs[i]), C.c_char_p)
        segs[i].len = lens[i]
        segs[i].abs_pos = 0
        segs[i].tips_only = 0
    res = {}
    for label, with_matches in (("blocks_windows_counts", False), ("with_match_vectors", True)):
        best, nm = None, 0
        for _ in range(2):
            o = (K.SegmentOut * n)()
            cnt = (K.SegmentCounts * n)()
            t0 = time.perf_counter()
            rc = L.ts_scan_segments(tel._ctx.ptr, segs, n, o) if with_matches else L.ts_scan_segments_blocks(tel._ctx.ptr, segs, n, o, cnt)
            dt = time.perf_counter() - t0
            assert rc == 0, tel._ctx.error()
            nm = int(sum(o[i].n_matches for i in range(n))) if with_matches else int(sum(c.n_matches for c in cnt))
            nw = int(sum(o[i].n_windows for i in range(n)))
            L.ts_free_segments(o, n)
            best = dt if best is None else min(best, dt)
        res[label] = {"seconds": round(best, 3), "gbases_per_s": round(total / best / 1e9, 3), "matches": nm, "windows": nw}
    out[name] = {"flags": flags, "patterns": len(tel.userInput.patternInfo), **res}
    tel.close()
print(json.dumps({"workload": "synthetic %.2f Gb / %d contigs (bench.py's assembly), general kernels through the host entry points" % (gb, ncontig),
                  "results": out}, indent=1))
