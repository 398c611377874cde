#!/usr/bin/env python3
"""One-off fuzz of the shard results (not part of the test suite): random parameter sets x random segment layouts x random
numbers of parts; every part scanned, block-called and packed on its own restricted batch, the messages merged, the
result compared with the oracle segment by segment (windows, blocks, writer-visible records, counts).  A merge that
answers TS_SHARD_NEED_FULL is counted, not compared (the caller's fallback).  python3 profiles/fuzz_shards.py [iterations] [seed]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import teloscope_amd as ta
from teloscope_amd import _capi as K
from teloscope_amd.cli import parse_cli, user_input
from teloscope_amd.distributed import ShardPlan, finalize_shards, free_segments
from tests import seqgen
from tests.backends import OracleBackend, assert_visible_view_equal
from tests.test_gpu_shard_results import _fill, _pack_all_parts

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
dev = torch.device("cuda", 0)
done = need_full = grown = 0
for it in range(iters):
    tips = rng.random() < 0.15
    w = int(rng.choice([500, 1000, 2000, 3000]))
    s = int(rng.choice([w, w // 2, w // 4]))
    cli = "-c %s -x %d -t %d -k %d -d %d -l %d" % (rng.choice(["TTAGGG", "CCCTAAA", "TTAGG"]), int(rng.integers(0, 2)),
                                                    int(rng.choice([300, 2500, 20000, 50000])), int(rng.choice([20, 50, 120])),
                                                    int(rng.choice([50, 200])), int(rng.choice([30, 100, 500])))
    if not tips:
        cli += " -w %d -s %d -r -g -e -m -i" % (w, s)
    opts = parse_cli("x.fa " + cli)
    tel = ta.Teloscope(user_input(opts, device=0))
    if not tel.usesFastPath():
        continue
    orac = OracleBackend(opts)
    nseg = int(rng.integers(1, 9))
    lens = [int(rng.choice([0, 7, 999, 5000, 40_000, 200_000, 600_000, 1_500_000], p=[.05, .05, .1, .1, .2, .2, .2, .1])) for _ in range(nseg)]
    seqs = []
    for n in lens:
        if n == 0:
            seqs.append(b"")
            continue
        q = bytearray(seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev, telo_repeats=int(rng.integers(1, 600)),
                                        n_its=int(rng.integers(0, 8)), iupac=int(rng.integers(0, 6))))
        if n > 100_000 and rng.random() < 0.5:                         # an array of a few kb somewhere: may straddle a boundary
            ln = int(rng.integers(600, 6000))
            a = int(rng.integers(0, n - ln))
            q[a:a + ln] = (opts.canonical_rev.encode() * (ln // len(opts.canonical_rev) + 1))[:ln]
        seqs.append(bytes(q))
    abs_pos = [int(rng.integers(0, 10 ** 9)) for _ in lens]
    exp = [orac.scan_segment(q, abs_pos[i], tips) for i, q in enumerate(seqs)]
    for world in sorted(set(int(x) for x in rng.integers(1, 10, size=2))):
        plan = ShardPlan(tel, lens, abs_pos=abs_pos, tips_only=tips, world=world)
        buf = _fill(plan, seqs, dev)
        scale = 1
        while True:                                                    # (a message that overflows is packed again at a larger scale)
            try:
                msgs, _ = _pack_all_parts(plan, buf, dev, scale=scale)
                break
            except AssertionError as e:
                if "message overflow" not in str(e) or scale >= 64:
                    raise
                scale *= 4
                grown += 1
        rc, out, cnt = finalize_shards(plan, msgs)
        if rc == K.SHARD_NEED_FULL:
            need_full += 1
        else:
            assert rc == 0, (cli, world, rc, tel._ctx.error())
            for i in range(len(lens)):
                assert_visible_view_equal(ta.SegmentData(out[i], tips), exp[i], tips, cnt[i], "fuzz %d cli=%r world %d segment %d len %d" % (it, cli, world, i, lens[i]))
            free_segments(plan, out)
        plan.close()
    done += 1
    if done % 20 == 0:
        print("shard fuzz: %d layouts ok (%d merges asked for the full exchange)" % (done, need_full), flush=True)
print("shard fuzz: all %d layouts equal the oracle; %d merges asked for the full exchange, %d packs were repeated at a larger scale" % (done, need_full, grown))
