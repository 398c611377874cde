#!/usr/bin/env python3
"""FASTA text in host memory -> blocks + windows + counts through ts_scan_segments_blocks (TS_INPUT_TEXT_PIECES), against the same
bases joined; TS_TIMING=1 TS_STAGE_TIMING=1 print where the upload stage's time goes.  python3 profiles/text_in_rate.py [gbases]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import teloscope_amd as ta
from teloscope_amd import _capi as K
from teloscope_amd.cli import parse_cli, user_input

gb = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
tel = ta.Teloscope(user_input(parse_cli("x.fa " + bench.FLAGS), device=0))
L = K.lib()
lens = bench.contig_lengths(int(gb * 1e9), 200, 42)
offs, off = [], 0
for n_ in lens:
    offs.append(off)
    off += (n_ + 15) & ~15
dev = torch.device("cuda", 0)
buf = torch.zeros(off + 4096, dtype=torch.uint8, device=dev)
bench.fill_synthetic(buf, offs, lens, 42, dev)
host = buf.cpu().numpy()
del buf
n = len(lens)
total = int(sum(lens))
segs = (K.SegmentIn * n)()
for i in range(n):
    segs[i].seq = C.cast(C.c_void_p(host.ctypes.data + offs[i]), C.c_char_p)
    segs[i].len = lens[i]
width, lines_per_piece = 80, 200_000
texts, arrays, tsegs = [], [], (K.SegmentIn * n)()
for i in range(n):
    m = int(lens[i])
    bases = host[offs[i]:offs[i] + m]
    rows = m // width
    txt = np.empty(m + rows + (1 if m % width else 0), dtype=np.uint8)
    if rows:
        body = txt[:rows * (width + 1)].reshape(rows, width + 1)
        body[:, :width] = bases[:rows * width].reshape(rows, width)
        body[:, width] = 10
    if m % width:
        txt[rows * (width + 1):-1] = bases[rows * width:]
        txt[-1] = 10
    texts.append(txt)
    st, sb = lines_per_piece * (width + 1), lines_per_piece * width
    npieces = max(1, -(-len(txt) // st))
    arr = (K.TextPiece * npieces)()
    for q in range(npieces):
        a, b = q * st, min(len(txt), (q + 1) * st)
        arr[q].text = C.cast(C.c_void_p(txt.ctypes.data + a), C.c_char_p)
        arr[q].text_len = b - a
        arr[q].n_bases = min(m, (q + 1) * sb) - q * sb
    arrays.append(arr)
    tsegs[i].seq = C.cast(arr, C.c_char_p)
    tsegs[i].len = m
    tsegs[i].input_format = K.TS_INPUT_TEXT_PIECES
    tsegs[i].n_pieces = npieces
best = {}
for rep in range(int(os.environ.get("REPS", "6"))):
    for label, ss in (("joined bases", segs), ("FASTA text", tsegs)):
        res = (K.SegmentOut * n)()
        cnts = (K.SegmentCounts * n)()
        sys.stderr.write("== %s\n" % label)
        t0 = time.perf_counter()
        rc = L.ts_scan_segments_blocks(tel._ctx.ptr, ss, n, res, cnts)
        dt = time.perf_counter() - t0
        assert rc == 0, tel._ctx.error()
        L.ts_free_segments(res, n)
        best[label] = min(best.get(label, 1e9), dt)
        print("%-13s %.1f ms = %.1f Gbases/s" % (label, dt * 1e3, total / dt / 1e9), flush=True)
for label, dt in best.items():
    print("best of all: %-13s %.1f ms = %.1f Gbases/s" % (label, dt * 1e3, total / dt / 1e9), flush=True)
