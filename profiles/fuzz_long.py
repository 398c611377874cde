#!/usr/bin/env python3
"""One-off extended fuzz (not part of the test suite): the seeded parameter fuzz of
tests/test_gpu_parity.py::test_fuzz_parameters_and_degenerate_segments with many more iterations, plus
dense-repeat segments and the device-block / counts path.  python3 profiles/fuzz_long.py [iterations] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tests import harness as H
from tests import seqgen
from tests.backends import BLOCK_FIELDS, OracleBackend, ProductBackend, assert_segment_equal

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
motifs = ["TTAGGG", "TTAGG", "CCCTAAA", "TTTTAGGG", "TTAGGGG", "TCAGG", "AAAAAA", "ACACAC", "TTAGGC", "TTAG", "TTA", "CCTA"]
done = n_wide = 0
for it in range(iters):
    c = motifs[int(rng.integers(0, len(motifs)))]
    w = int(rng.choice([len(c), 17, 64, 100, 333, 500, 1000, 2000, 5000, 12000, 30000]))
    s = int(rng.integers(max(1, w // 40), w + 1)) if rng.random() < 0.5 else w
    tips = rng.random() < 0.2
    pat = ""
    wide = rng.random() < float(os.environ.get("TS_FUZZ_WIDE", "0.06"))
    if wide:                                     # beyond 8 lengths or 32 bases: the general path's wide form
        pool = [c, c[:4], c + c[:1], c + c[:2], c * 2, c * 2 + c[:3], c * 3, (c * 8)[:33], (c * 8)[:40], (c * 11)[:62], c[:3], c + "A", c * 4 + "T"]
        k = int(rng.integers(1, len(pool)))
        pick = sorted(set(pool[int(i)] for i in rng.choice(len(pool), size=k, replace=False)) | {(c * 8)[:int(rng.integers(33, 50))]}, key=len)
        pat = " -p %s" % ",".join(pick)
    elif rng.random() < 0.25:                    # a second motif of another length: the general path's mixed-length sets
        c2 = motifs[int(rng.integers(0, len(motifs)))]
        if len(c2) != len(c) and min(len(c), len(c2)) >= 4:
            pat = " -p %s,%s" % (c, c2)
    cli = "-c %s%s -x %d -w %d -s %d -t %d -k %d -d %d -l %d -y %.2f" % (
        c, pat, (0 if wide and rng.random() < 0.7 else int(rng.integers(0, 2 if pat else 3))), w, s, int(rng.choice([50, 300, 5000, 50000])),
        int(rng.choice([5, 20, 50])), int(rng.choice([10, 100, 500])), int(rng.choice([12, 60, 300])),
        float(rng.choice([0.3, 0.5, 0.9])))
    if not tips:
        cli += " " + " ".join(rng.choice(["-r", "-g", "-e", "-m", "-i"], size=3, replace=False)) + " -g"
    opts = H.parse_cli("x.fa " + cli)
    if len(c) > w or (wide and w < 64):
        continue
    prod, orac = ProductBackend(opts), OracleBackend(opts)
    if orac.ambiguous:
        orac = orac.with_ambiguous_orientation_from(prod.patterns)
    dense = (c.encode() * 4000)[:int(rng.integers(2000, 24000))]
    mixed = bytearray(seqgen.chromosome(rng, int(rng.integers(30000, 200000)), opts.canonical_fwd, opts.canonical_rev,
                                        telo_repeats=int(rng.integers(10, 900)), n_its=6, iupac=int(rng.integers(0, 8)),
                                        n_runs=0))
    a = int(rng.integers(0, len(mixed) - len(dense)))
    mixed[a:a + len(dense)] = dense
    segs = [(bytes(mixed), int(rng.integers(0, 10 ** 9))), (dense, 3),
            (seqgen.chromosome(rng, int(rng.integers(1, 70000)), opts.canonical_fwd, opts.canonical_rev, telo_repeats=60,
                               n_its=3, iupac=5), 0),
            (b"A" * int(rng.integers(1, 9000)), 9)]
    got = prod.scan_segments([(q, ap, tips) for q, ap in segs])
    for (q, ap), g in zip(segs, got):
        e = orac.scan_segment(q.upper(), ap, tips)
        assert_segment_equal(g, e, tips, ctx="fuzz %d cli=%r len=%d" % (it, cli, len(q)))
    if True:                                     # (every path: the general kernels' blocks are called on the device too where the stream's order allows)
        gb, counts = prod.teloscope.scanSegmentsBlocksOnly(segs, tipsOnly=tips, with_counts=True)
        for (q, ap), g, cnt in zip(segs, gb, counts):
            e = orac.scan_segment(q.upper(), ap, tips)
            for name, blk in (("terminal_blocks", g.terminalBlocks), ("interstitial_blocks", g.interstitialBlocks)):
                assert len(blk) == len(e[name]), (cli, name, len(q))
                for f in BLOCK_FIELDS:
                    assert np.array_equal(blk[f], e[name][f]), (cli, name, f, len(q))
            assert cnt[1] == len(e["fwd_matches"]) + len(e["rev_matches"]) and cnt[3] == len(e["fwd_matches"]), (cli, cnt)
    done += 1
    n_wide += int(wide)
    if done % 20 == 0:
        print("fuzz: %d parameter sets ok" % done, flush=True)
print("fuzz: all %d parameter sets equal the oracle (%d wide: beyond 8 lengths / 32 bases)" % (done, n_wide))
