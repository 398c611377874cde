"""GPU parity of shard results (include/teloscan.h: ts_batch_restrict_shard / ts_batch_pack_shard /
ts_shards_finalize) and of ts_scan_segments_multi.

Every part of a plan is scanned on its own restricted batch — its owned tile range plus context tiles — calls its
blocks on the device and packs its message; the messages of all parts, merged on the host, must be what the oracle
says about every segment: windows (all fields), terminal and interstitial blocks, the match records a writer reads
(canonicalMatches, terminal nonCanonicalMatches) and the per-segment counts — for any number of parts, wherever the
boundaries fall.  Inputs the shards' assumptions do not hold for (a telomere longer than the context tiles, across a
boundary) must be REPORTED (TS_SHARD_NEED_FULL), never answered wrongly."""
import ctypes as C

import numpy as np
import pytest

from tests import harness as H
from tests import seqgen
from tests.backends import OracleBackend, assert_visible_view_equal

pytestmark = pytest.mark.gpu
HEADLINE = "-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -w 1000 -s 500 -r -g -e -m -i"


def _teloscope(cli, device=0):
    import teloscope_amd as ta
    from teloscope_amd.cli import parse_cli, user_input
    opts = parse_cli("x.fa " + cli)
    return opts, ta.Teloscope(user_input(opts, device=device))


def _pack_all_parts(plan, buf, dev, scale=1, emit=True):
    """Every part scanned and packed on its own restricted batch, as a rank would; returns the host messages.
    emit=False: the scan leaves no visible records / chain summaries (ts_batch_set_emit off) — the pack then takes both out
    of the match stream, as it does for results adopted from elsewhere."""
    import torch
    from teloscope_amd import _capi as K
    from teloscope_amd.distributed import PackedShard
    sptr = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    msgs, stats = [], []
    for p in range(plan.world):
        ps = PackedShard(plan, p, dev, slots=1, scale=scale)
        if not emit:
            assert K.lib().ts_batch_set_emit(ps.batches[0], 0) == 0
        local = buf[ps.info.input_begin:max(ps.info.input_end, ps.info.input_begin + 64)].clone()   # a rank holds only the bytes its range reads
        for _ in range(6):
            ps.scan_pack(local.data_ptr(), sptr, 0)
            st = ps.status(0)
            if st.flags & K.SHARD_OVERFLOW_SCAN:
                ps.sync(0)
            elif st.flags & (K.SHARD_OVERFLOW_VISIBLE | K.SHARD_OVERFLOW_BLOCKS):
                raise AssertionError("message overflow at this scale: visible %d / %d, blocks %d / %d"
                                     % (st.n_visible, st.visible_capacity, st.n_blocks, st.block_capacity))
            else:
                break
        msgs.append(ps.msgs[0].cpu().numpy().copy())
        stats.append(st)
        ps.close()
    return msgs, stats


def _fill(plan, seqs, dev):
    import torch
    buf = torch.zeros(int(plan.info.input_bytes), dtype=torch.uint8, device=dev)
    for off, s in zip(plan.segment_offsets(), seqs):
        if len(s):
            buf[off:off + len(s)] = torch.frombuffer(bytearray(s), dtype=torch.uint8).to(dev)
    return buf


CLIS = [HEADLINE, "-r -g -e -m -i", "-c CCCTAAA -w 2000 -s 1000 -r -g -e -m -i", HEADLINE + " -t 3000",
        "-r -i -t 2500 -k 120", "-r -w 700 -s 700 -t 1000", "", "-t 3000"]


@pytest.mark.parametrize("cli", CLIS)
def test_shard_messages_merge_to_the_oracle(cli):
    import torch
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.distributed import ShardPlan, finalize_shards, free_segments, shard_info
    dev = torch.device("cuda", 0)
    opts, tel = _teloscope(cli)
    rng = np.random.default_rng(len(cli) + 29)
    lens = [70001, 7, 250003, 0, 1999, 1_000_000, 16500, 333_333, 120_000]
    seqs = [seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev, n_its=4, iupac=2) if n else b"" for n in lens]
    abs_pos = [13 * i for i in range(len(lens))]
    tips = opts.ultra_fast
    orac = OracleBackend(opts)
    exp = [orac.scan_segment(sq, abs_pos[i], tips) for i, sq in enumerate(seqs)]
    split_inside = 0
    for world in (1, 2, 3, 5, 8):
        plan = ShardPlan(tel, lens, abs_pos=abs_pos, tips_only=tips, world=world)
        buf = _fill(plan, seqs, dev)
        msgs, _ = _pack_all_parts(plan, buf, dev)
        for p in range(world):
            si = shard_info(plan, p)
            split_inside += int(si.ext_begin != si.own_begin) + int(si.ext_end != si.own_end)
        rc, out, cnt = finalize_shards(plan, msgs)
        assert rc == 0, (world, rc, tel._ctx.error())
        for i in range(len(lens)):
            assert_visible_view_equal(ta.SegmentData(out[i], tips), exp[i], tips, cnt[i], "world %d segment %d" % (world, i))
        free_segments(plan, out)
        plan.close()
    if not tips and ("-t 3000" in cli or "-t 2500" in cli or "-t 1000" in cli):
        assert split_inside > 0, "no boundary fell inside a segment: the context tiles were never exercised"


def test_visible_record_regions_that_overflow_are_regrown(monkeypatch):
    """The emitting scan appends a wave's visible records to a region of its own; one that is too small (here: 64 records, by
    TS_VIS_CAP) must be REPORTED — TS_SHARD_F_SCAN_OVERFLOW in the message's header, nothing trusted — and ts_batch_sync must
    size the regions for the fullest wave and rescan, after which the merge equals the oracle."""
    import torch
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.distributed import PackedShard, ShardPlan, finalize_shards, free_segments
    monkeypatch.setenv("TS_VIS_CAP", "64")
    dev = torch.device("cuda", 0)
    opts, tel = _teloscope(HEADLINE + " -t 3000")
    rng = np.random.default_rng(3)
    lens = [900_000, 250_003]
    seqs = [seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev, telo_repeats=1500, n_its=5) for n in lens]
    plan = ShardPlan(tel, lens, world=2)
    buf = _fill(plan, seqs, dev)
    sptr = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    msgs, syncs = [], 0
    for p in range(2):
        ps = PackedShard(plan, p, dev, slots=1)
        local = buf[ps.info.input_begin:max(ps.info.input_end, ps.info.input_begin + 64)].clone()
        for _ in range(4):
            ps.scan_pack(local.data_ptr(), sptr, 0)
            if not (ps.status(0).flags & K.SHARD_OVERFLOW_SCAN):
                break
            ps.sync(0)
            syncs += 1
        msgs.append(ps.msgs[0].cpu().numpy().copy())
        ps.close()
    assert syncs >= 1, "a 64-record region did not overflow on a telomere: the path under test did not run"
    rc, out, cnt = finalize_shards(plan, msgs)
    assert rc == 0, tel._ctx.error()
    orac = OracleBackend(opts)
    for i, sq in enumerate(seqs):
        assert_visible_view_equal(ta.SegmentData(out[i], False), orac.scan_segment(sq, 0, False), False, cnt[i], "segment %d" % i)
    free_segments(plan, out)
    plan.close()


def test_kernel_messages_equal_the_messages_packed_from_oracle_results():
    """Byte level: what the kernels pack (header, per-segment entries, bit-packed windows, per-tile counts, visible records,
    blocks) against tests/shardpack.py's rendering of the oracle's results in the same layout."""
    import torch
    from tests import shardpack
    from teloscope_amd.distributed import ShardPlan
    dev = torch.device("cuda", 0)
    for cli in (HEADLINE + " -t 3000", "-r -i -w 700 -s 700 -t 1000"):
        opts, tel = _teloscope(cli)
        rng = np.random.default_rng(8)
        lens = [300_000, 0, 4999, 610_000, 45_000]
        seqs = [seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev, n_its=3, iupac=1) if n else b"" for n in lens]
        abs_pos = [7 * i for i in range(len(lens))]
        orac = OracleBackend(opts)
        exp = [orac.scan_segment(sq, abs_pos[i], False) for i, sq in enumerate(seqs)]
        for world in (1, 3, 4):
            plan = ShardPlan(tel, lens, abs_pos=abs_pos, world=world)
            msgs, _ = _pack_all_parts(plan, _fill(plan, seqs, dev), dev)
            for p in range(world):
                want = shardpack.pack_shard(plan, opts, p, exp)
                got = msgs[p]
                hg, hw = shardpack.read_header(got), shardpack.read_header(want)
                assert hg == hw, (cli, world, p, hg, hw)
                nseg, nown, nwin = int(hg["n_segs"]), int(hg["own_end"] - hg["own_begin"]), int(hg["n_windows"])
                from teloscope_amd.distributed import shard_info
                off = shardpack.sections(shard_info(plan, p), nseg, nown, nwin)
                assert np.array_equal(shardpack.read_segs(got), shardpack.read_segs(want)), (cli, world, p)
                for name, n in (("windows", nwin * int(hg["window_bytes"])), ("tilevis", nown * 2), ("visible", int(hg["n_visible"]) * int(hg["visible_bytes"]))):
                    assert np.array_equal(got[off[name]:off[name] + n], want[off[name]:off[name] + n]), (cli, world, p, name)
                nb = int(hg["n_blocks"])
                bg = np.frombuffer(got[off["blocks"]:off["blocks"] + nb * 64].tobytes(), dtype=shardpack.DEVBLOCK_DT)
                bw = np.frombuffer(want[off["blocks"]:off["blocks"] + nb * 64].tobytes(), dtype=shardpack.DEVBLOCK_DT)
                key = ("seg", "kind", "seq", "start")            # (the kernels append blocks in completion order)
                assert np.array_equal(np.sort(bg, order=key), np.sort(bw, order=key)), (cli, world, p)
            plan.close()


@pytest.mark.parametrize("cli", [HEADLINE + " -t 3000", "-r -g -e -m -i", "-r -i -w 700 -s 700 -t 1000 -k 120"])
def test_messages_do_not_depend_on_whether_the_scan_emits(cli):
    """ts_batch_set_emit: with it the pack copies the visible records the scan left and the interstitial search walks only
    the tiles the scan's chain summaries cannot rule out; without it both read the whole match stream.  The messages must be
    the same bytes either way (blocks: the same set — the kernels append them in completion order)."""
    import torch
    from tests import shardpack
    from teloscope_amd.distributed import ShardPlan, shard_info
    dev = torch.device("cuda", 0)
    opts, tel = _teloscope(cli)
    rng = np.random.default_rng(41)
    lens = [420_000, 12, 0, 90_001, 1_300_000, 33_000]
    seqs = [seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev, n_its=6, iupac=2) if n else b"" for n in lens]
    # a telomere-like array inside the big contig, away from its ends: chains of canonical matches across many tiles
    big = bytearray(seqs[4])
    big[600_000:640_000] = (opts.canonical_fwd.encode() * 7000)[:40_000]
    seqs[4] = bytes(big)
    for world in (1, 3):
        plan = ShardPlan(tel, lens, world=world)
        buf = _fill(plan, seqs, dev)
        with_emit, st1 = _pack_all_parts(plan, buf, dev, emit=True)
        without, st0 = _pack_all_parts(plan, buf, dev, emit=False)
        for p in range(world):
            a, b = with_emit[p], without[p]
            ha, hb = shardpack.read_header(a), shardpack.read_header(b)
            assert ha == hb, (cli, world, p, ha, hb)
            nb = int(ha["n_blocks"])
            nseg, nown, nwin = int(ha["n_segs"]), int(ha["own_end"] - ha["own_begin"]), int(ha["n_windows"])
            off = shardpack.sections(shard_info(plan, p), nseg, nown, nwin)
            assert np.array_equal(a[128:off["blocks"]], b[128:off["blocks"]]), (cli, world, p)
            key = ("seg", "kind", "seq", "start")
            ba = np.frombuffer(a[off["blocks"]:off["blocks"] + nb * 64].tobytes(), dtype=shardpack.DEVBLOCK_DT)
            bb = np.frombuffer(b[off["blocks"]:off["blocks"] + nb * 64].tobytes(), dtype=shardpack.DEVBLOCK_DT)
            assert np.array_equal(np.sort(ba, order=key), np.sort(bb, order=key)), (cli, world, p)
        assert sum(int(x.n_blocks) for x in st1) > 0
        plan.close()


def test_block_at_a_shard_boundary_and_a_telomere_longer_than_the_context():
    """(a) An interstitial block planted right across every boundary of a 4-way split: it starts in one shard's tiles and
    ends in the next one's — the owner of its start follows it into its context tiles.  (b) The same segment with a
    repeat array of 60 kb across a boundary: the chain outruns the context, and the merge must say NEED_FULL."""
    import torch
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.distributed import ShardPlan, finalize_shards, free_segments, shard_info
    dev = torch.device("cuda", 0)
    opts, tel = _teloscope(HEADLINE + " -t 3000")
    rng = np.random.default_rng(5)
    n = 1_200_000
    base = bytearray(seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev, n_its=2))
    plan = ShardPlan(tel, [n], world=4)
    tiles = plan.tiles
    cuts = [int(tiles["seg_offset"][shard_info(plan, p).own_begin]) for p in range(1, 4)]
    assert all(0 < c < n for c in cuts) and len(set(cuts)) == 3
    orac = OracleBackend(opts)
    for variant, length in (("block", 900), ("long", 60_000)):
        seq = bytearray(base)
        for c in cuts[:(3 if variant == "block" else 1)]:
            rep = (b"TTAGGG" * (length // 6 + 1))[:length]
            seq[c - length // 2:c - length // 2 + length] = rep
        seq = bytes(seq)
        buf = _fill(plan, [seq], dev)
        msgs, stats = _pack_all_parts(plan, buf, dev)
        rc, out, cnt = finalize_shards(plan, msgs)
        if variant == "block":
            assert rc == 0, tel._ctx.error()
            e = orac.scan_segment(seq, 0, False)
            assert len(e["interstitial_blocks"]) >= 3
            assert_visible_view_equal(ta.SegmentData(out[0], False), e, False, cnt[0], "blocks across boundaries")
            free_segments(plan, out)
        else:
            assert rc == K.SHARD_NEED_FULL, rc
            assert any(s.flags & K.SHARD_OUT_OF_CONTEXT for s in stats)
    plan.close()


def test_message_overflow_is_reported_and_a_larger_scale_fixes_it():
    """The variable sections of a message (visible records, blocks) are sized from the plan for ordinary sequence; an
    input with far more canonical matches than that — 800 kb of TTAGGG in one part's tiles, and a hundred short
    interstitial blocks — must come back with TS_SHARD_OVERFLOW_* set and the factor to grow by, merge as
    TS_SHARD_RETRY_GROW, and after ts_batch_set_shard_scale on every part merge to the oracle's result."""
    import torch
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.distributed import PackedShard, ShardPlan, finalize_shards, free_segments
    dev = torch.device("cuda", 0)
    opts, tel = _teloscope(HEADLINE + " -t 3000")
    rng = np.random.default_rng(21)
    n = 2_400_000
    seq = bytearray(seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev, n_its=2))
    seq[100_000:900_000] = (b"TTAGGG" * 133_334)[:800_000]
    for i in range(100):
        a = 1_300_000 + 3_000 * i
        seq[a:a + 120] = b"TTAGGG" * 20
    seq = bytes(seq)
    plan = ShardPlan(tel, [n], world=2)
    buf = _fill(plan, [seq], dev)
    sptr = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    shards = [PackedShard(plan, p, dev, slots=1) for p in range(2)]
    locals_ = [buf[s.info.input_begin:max(s.info.input_end, s.info.input_begin + 64)].clone() for s in shards]

    def pack_all():
        msgs, sts = [], []
        for s, loc in zip(shards, locals_):
            for _ in range(4):
                s.scan_pack(loc.data_ptr(), sptr, 0)
                st = s.status(0)
                if not (st.flags & K.SHARD_OVERFLOW_SCAN):
                    break
                s.sync(0)
            msgs.append(s.msgs[0].cpu().numpy().copy())
            sts.append(st)
        return msgs, sts

    msgs, sts = pack_all()
    assert any(st.flags & (K.SHARD_OVERFLOW_VISIBLE | K.SHARD_OVERFLOW_BLOCKS) for st in sts), [hex(st.flags) for st in sts]
    factor = max(int(st.scale_factor_needed) for st in sts)
    assert factor >= 2
    rc, out, cnt = finalize_shards(plan, msgs)
    assert rc == K.SHARD_RETRY_GROW, rc
    for s in shards:
        s.set_scale(factor)
    msgs, sts = pack_all()
    assert not any(st.flags & (K.SHARD_OVERFLOW_VISIBLE | K.SHARD_OVERFLOW_BLOCKS) for st in sts), [hex(st.flags) for st in sts]
    rc, out, cnt = finalize_shards(plan, msgs)
    assert rc == 0, (rc, tel._ctx.error())
    e = OracleBackend(opts).scan_segment(seq, 0, False)
    assert len(e["interstitial_blocks"]) >= 100
    assert_visible_view_equal(ta.SegmentData(out[0], False), e, False, cnt[0], "after growing the message")
    free_segments(plan, out)
    for s in shards:
        s.close()
    plan.close()


def test_scan_segments_multi_equals_oracle():
    """ts_scan_segments_multi over 1, 2 and 3 contexts (sharing this GPU): per segment == oracle, full scans and
    tips-only segments mixed in one call; parameter sets outside the tiled kernel (a mixed-length set, a nine-length one) have their
    segments dealt whole to the contexts."""
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.cli import parse_cli, user_input
    L = K.lib()
    rng = np.random.default_rng(77)
    for cli in (HEADLINE + " -t 3000", "-r -g -i", "-p TTAGGG,TTAGG -r -i -w 1000 -s 500",
                "-x 0 -p TTAG,TTAGG,TTAGGG,TTTAGGG,TTTTAGGG,TTAGGGTTA,TTAGGGTTAG,TTAGGGTTAGG,TTAGGGTTAGGG -g -i -w 1000 -s 500"):
        opts = parse_cli("x.fa " + cli)
        tels = [ta.Teloscope(user_input(opts, device=0)) for _ in range(3)]
        lens = [400_000, 0, 33, 150_000, 20_000, 600_001, 90_000]
        seqs = [seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev, n_its=3, iupac=1) if n else b"" for n in lens]
        tipsv = [False, False, False, True, False, False, True]
        n = len(lens)
        segs = (K.SegmentIn * n)()
        for i in range(n):
            segs[i].seq = seqs[i]
            segs[i].len = lens[i]
            segs[i].abs_pos = 1000 * i
            segs[i].tips_only = int(tipsv[i])
        orac = OracleBackend(opts)
        exp = [orac.scan_segment(seqs[i], 1000 * i, tipsv[i]) for i in range(n)]
        for nctx in (1, 2, 3):
            ctxs = (C.c_void_p * nctx)(*[t._ctx.ptr for t in tels[:nctx]])
            out = (K.SegmentOut * n)()
            cnt = (K.SegmentCounts * n)()
            rc = L.ts_scan_segments_multi(ctxs, nctx, segs, n, out, cnt)
            assert rc == 0, (cli, nctx, tels[0]._ctx.error())
            for i in range(n):
                assert_visible_view_equal(ta.SegmentData(out[i], tipsv[i]), exp[i], tipsv[i], cnt[i], "%s nctx %d segment %d" % (cli, nctx, i))
            L.ts_free_segments(out, n)


def test_eight_way_split_of_the_bench_assembly_merges_to_the_one_part_result():
    """configs[2] at its real size, split as 8 GPUs would split it: the bench's own assembly (3.0 Gb, 200 contigs: 250 Mb
    contigs spread over several parts, boundaries inside segments, context tiles) — every part scanned, block-called and
    packed on its own restricted batch (one after the other on this GPU), the 8 messages merged.  The oracle needs
    minutes for 3 Gb; the witness here is the ONE-part message of the same plan (no boundary, no context), whose kernels
    the small-size tests pin to the oracle: windows, blocks, visible records and counts must be identical for every
    segment, and the 8 messages together must stay under the 100 MB the design promises."""
    import torch
    import bench
    import teloscope_amd as ta
    from teloscope_amd.distributed import ShardPlan, finalize_shards, free_segments, shard_info
    dev = torch.device("cuda", 0)
    opts, tel = _teloscope(HEADLINE)
    total = 3_000_000_000
    lens = bench.contig_lengths(total, 200, 42)
    plan1 = ShardPlan(tel, lens, world=1)
    buf = torch.zeros(int(plan1.info.input_bytes), dtype=torch.uint8, device=dev)
    bench.fill_synthetic(buf, plan1.segment_offsets(), lens, 42, dev)
    results = {}
    for world in (1, 8):
        plan = plan1 if world == 1 else ShardPlan(tel, lens, world=world)
        msgs, stats = _pack_all_parts(plan, buf, dev)
        if world == 8:
            assert sum(len(m) for m in msgs[1:]) < 100e6
            inside = sum(int(shard_info(plan, p).ext_begin != shard_info(plan, p).own_begin) for p in range(world))
            assert inside >= 4, "boundaries did not fall inside segments"
        rc, out, cnt = finalize_shards(plan, msgs)
        assert rc == 0, (world, rc, tel._ctx.error())
        res = []
        for i in range(len(lens)):
            sd = ta.SegmentData(out[i], False)
            res.append((sd.windows.tobytes(), sd.terminalBlocks.tobytes(), sd.interstitialBlocks.tobytes(), sd._m.tobytes(),
                        (cnt[i].n_windows, cnt[i].n_matches, cnt[i].n_canonical, cnt[i].n_forward)))
        results[world] = res
        free_segments(plan, out)
        if world != 1:
            plan.close()
    plan1.close()
    assert sum(r[4][1] for r in results[1]) > 90_000_000
    for i, (a, b) in enumerate(zip(results[1], results[8])):
        for name, x, y in zip(("windows", "terminal blocks", "interstitial blocks", "visible records", "counts"), a, b):
            assert x == y, "segment %d: %s differ between the 1-part and the 8-part merge" % (i, name)


def test_scan_segments_multi_takes_the_single_context_path_when_the_shards_cannot_agree():
    """A 60 kb repeat array across the boundary between the two shards of a segment (longer than the context tiles): the
    merge says NEED_FULL and ts_scan_segments_multi answers from one context instead — the same result as the oracle's,
    never an error and never a wrong block."""
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.cli import parse_cli, user_input
    L = K.lib()
    opts = parse_cli("x.fa " + HEADLINE + " -t 3000")
    tels = [ta.Teloscope(user_input(opts, device=0)) for _ in range(2)]
    rng = np.random.default_rng(31)
    n = 1_200_000
    seq = bytearray(seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev, n_its=2))
    seq[570_000:630_000] = (b"TTAGGG" * 10_000)[:60_000]
    seq = bytes(seq)
    segs = (K.SegmentIn * 1)()
    segs[0].seq, segs[0].len, segs[0].abs_pos = seq, n, 0
    out, cnt = (K.SegmentOut * 1)(), (K.SegmentCounts * 1)()
    ctxs = (C.c_void_p * 2)(*[t._ctx.ptr for t in tels])
    assert L.ts_scan_segments_multi(ctxs, 2, segs, 1, out, cnt) == 0, tels[0]._ctx.error()
    e = OracleBackend(opts).scan_segment(seq, 0, False)
    assert any(b["block_len"] >= 59_000 for b in e["interstitial_blocks"])
    assert_visible_view_equal(ta.SegmentData(out[0], False), e, False, cnt[0], "fallback")
    L.ts_free_segments(out, 1)


def test_rank_exchange_carries_every_part_s_message_through_rccl():
    """The C-ABI's rank-form exchange (ts_exchange_*: one grouped ncclSend / ncclRecv round, librccl opened at run time) on
    the one rank this box has: every part's message of a 3-way split goes through RCCL (the loop-back form of
    ts_exchange_gather) into a receive buffer; the received bytes are the packed bytes, and merged they are the oracle's
    segments.  Argument errors are refused without touching the communicator."""
    import torch
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.distributed import PackedShard, ShardPlan, finalize_shards, free_segments
    dev = torch.device("cuda", 0)
    L = K.lib()
    opts, tel = _teloscope(HEADLINE + " -t 3000")
    rng = np.random.default_rng(77)
    lens = [250003, 70001, 0, 1_000_000, 16500]
    seqs = [seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev, n_its=4, iupac=2) if n else b"" for n in lens]
    orac = OracleBackend(opts)
    exp = [orac.scan_segment(sq, 11 * i, False) for i, sq in enumerate(seqs)]
    ident = (C.c_char * 128)()
    assert L.ts_exchange_unique_id(ident) == 0, L.ts_exchange_last_error()
    x = L.ts_exchange_create(tel._ctx.ptr, ident, 0, 1)
    assert x, tel._ctx.error()
    assert not L.ts_exchange_create(tel._ctx.ptr, ident, 1, 1)                 # rank outside the communicator
    plan = ShardPlan(tel, lens, abs_pos=[11 * i for i in range(len(lens))], tips_only=False, world=3)
    buf = _fill(plan, seqs, dev)
    stream = torch.cuda.current_stream()
    sptr = C.c_void_p(stream.cuda_stream)
    received = []
    for p in range(3):
        ps = PackedShard(plan, p, dev, slots=1)
        local = buf[ps.info.input_begin:max(ps.info.input_end, ps.info.input_begin + 64)].clone()
        ps.scan_pack(local.data_ptr(), sptr, 0)
        assert ps.status(0).flags == 0
        msg = ps.msgs[0]
        n = int(ps.info.msg_bytes)
        landing = torch.full((n,), 0xEE, dtype=torch.uint8, device=dev)
        recv = (C.c_void_p * 1)(landing.data_ptr())
        sizes = (C.c_uint64 * 1)(n)
        assert L.ts_exchange_gather(x, 0, C.c_void_p(msg.data_ptr()), n, recv, sizes, sptr) == 0, tel._ctx.error()
        stream.synchronize()
        assert torch.equal(landing, msg[:n]), "part %d: the message changed on its way through RCCL" % p
        received.append(landing.cpu().numpy().copy())
        # refused: a dst that is not a rank, the receiving rank without buffers, a size that disagrees with the message
        assert L.ts_exchange_gather(x, 1, C.c_void_p(msg.data_ptr()), n, recv, sizes, sptr) == K.TS_ERR_INVALID_ARG
        assert L.ts_exchange_gather(x, 0, C.c_void_p(msg.data_ptr()), n, None, None, sptr) == K.TS_ERR_INVALID_ARG
        bad = (C.c_uint64 * 1)(n - 1)
        assert L.ts_exchange_gather(x, 0, C.c_void_p(msg.data_ptr()), n, recv, bad, sptr) == K.TS_ERR_INVALID_ARG
        ps.close()
    rc, out, cnt = finalize_shards(plan, received)
    assert rc == 0, tel._ctx.error()
    for i in range(len(lens)):
        assert_visible_view_equal(ta.SegmentData(out[i], False), exp[i], False, cnt[i], "through RCCL, segment %d" % i)
    free_segments(plan, out)
    plan.close()
    L.ts_exchange_destroy(x)


def test_streams_on_one_queue_are_told_from_streams_that_run_side_by_side():
    """ts_streams_concurrent (include/teloscan.h): a stream shares its hardware queue with itself; distributed.concurrent_streams
    hands out streams that pairwise do not; and a pack on a stream of its own, with the library's side stream chosen against the
    scan and pack streams (shard.cpp: side_stream_for), leaves the same message as one on the scan stream."""
    import torch
    from teloscope_amd import _capi as K
    from teloscope_amd.distributed import PackedShard, ShardPlan, concurrent_streams
    dev = torch.device("cuda", 0)
    opts, tel = _teloscope(HEADLINE)
    L, ctx = K.lib(), tel._ctx.ptr
    s0 = torch.cuda.Stream(device=dev)
    assert L.ts_streams_concurrent(ctx, C.c_void_p(s0.cuda_stream), C.c_void_p(s0.cuda_stream)) == 0
    streams = concurrent_streams(tel, dev, 3, first=s0)
    assert len(streams) == 3 and streams[0] is s0 and len({s.cuda_stream for s in streams}) == 3
    for i in range(3):
        for j in range(3):
            if i != j:
                assert L.ts_streams_concurrent(ctx, C.c_void_p(streams[i].cuda_stream), C.c_void_p(streams[j].cuda_stream)) == 1
    rng = np.random.default_rng(77)
    lens = [60_000, 9_000, 150_000]
    seqs = [seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev, n_its=3, iupac=1) for n in lens]
    plan = ShardPlan(tel, lens, world=1)
    buf = _fill(plan, seqs, dev)
    msgs = []
    for pack_stream in (streams[0], streams[1], streams[2]):
        ps = PackedShard(plan, 0, dev, slots=1)
        with torch.cuda.stream(streams[0]):
            ps.scan(buf.data_ptr(), C.c_void_p(streams[0].cuda_stream), 0)
        ps.wait_scan(C.c_void_p(pack_stream.cuda_stream), 0)          # ts_batch_wait_scan: the pack stream behind the slot's scan
        ps.pack(C.c_void_p(pack_stream.cuda_stream), 0)
        torch.cuda.synchronize()
        assert not ps.status(0).flags
        msgs.append(ps.msgs[0].cpu().numpy().copy())
        ps.close()
    from tests import shardpack
    from teloscope_amd.distributed import shard_info
    h = shardpack.read_header(msgs[0])
    nb = int(h["n_blocks"])
    assert nb > 0
    off = shardpack.sections(shard_info(plan, 0), int(h["n_segs"]), int(h["own_end"] - h["own_begin"]), int(h["n_windows"]))
    plan.close()
    blocks = lambda m: np.sort(np.frombuffer(m[off["blocks"]:off["blocks"] + nb * 64].tobytes(), dtype=shardpack.DEVBLOCK_DT),
                               order=("seg", "kind", "seq", "start"))           # (appended in completion order)
    for m in msgs[1:]:
        assert shardpack.read_header(m) == h
        assert np.array_equal(msgs[0][128:off["blocks"]], m[128:off["blocks"]])
        assert np.array_equal(blocks(msgs[0]), blocks(m))


def test_one_scan_in_n_is_timed():
    """ts_batch_set_timing: with n = 3 the scans 0, 3, 6 of seven have a start event; ts_batch_info's kernel time is their mean, and
    the results do not depend on it."""
    import torch
    from teloscope_amd import _capi as K
    from teloscope_amd.distributed import ShardPlan
    dev = torch.device("cuda", 0)
    opts, tel = _teloscope(HEADLINE)
    L = K.lib()
    rng = np.random.default_rng(5)
    lens = [300_000, 41_000]
    seqs = [seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev, n_its=2, iupac=1) for n in lens]
    plan = ShardPlan(tel, lens, world=1)
    buf = _fill(plan, seqs, dev)
    sptr = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    info = K.BatchInfo()

    def scans(n):
        for _ in range(n):
            assert L.ts_batch_scan(plan.batch, C.c_void_p(buf.data_ptr()), sptr) == 0
        assert L.ts_batch_sync(plan.batch) == 0
        assert L.ts_batch_get_info(plan.batch, C.byref(info)) == 0
        return int(info.kernel_launches), float(info.avg_kernel_ms), int(info.n_matches)

    n_all, ms_all, matches = scans(4)
    assert n_all == 4 and ms_all > 0
    assert L.ts_batch_set_timing(plan.batch, 3) == 0
    n3, ms3, m3 = scans(7)                      # scans 4 .. 10 of the batch: 6 and 9 are multiples of 3
    assert n3 == 2 and ms3 > 0 and m3 == matches
    assert L.ts_batch_set_timing(plan.batch, 0) == 0
    n0, ms0, m0 = scans(3)
    assert (n0, ms0) == (n3, ms3) and m0 == matches          # nothing timed: the figures stay what they were
    plan.close()
