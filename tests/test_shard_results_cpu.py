"""The host side of a sharded scan, on CPU: the split of a plan into shards, the merge of the shards' messages
(ts_shards_finalize) and the exchange that carries them (teloscope_amd.distributed.ShardExchange over gloo).

No GPU here, so the messages are built from oracle results in the layout the HIP kernels produce (tests/shardpack.py);
tests/test_gpu_shard_results.py runs the same merge over the kernels' own messages.  On a planning-only context
(device = TS_DEVICE_NONE) the library plans and merges but never touches HIP."""
import os
import socket
import sys

import numpy as np
import pytest

from tests import seqgen, shardpack
from tests.backends import OracleBackend, assert_visible_view_equal

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADLINE = "-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -w 1000 -s 500 -r -g -e -m -i"
LENS = [70001, 7, 250003, 0, 1999, 400_000, 16500, 133_333]
ABS = [17 * i for i in range(len(LENS))]


def _plan(cli, world, lens=LENS, abs_pos=ABS):
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.cli import parse_cli, user_input
    from teloscope_amd.distributed import ShardPlan
    opts = parse_cli("x.fa " + cli)
    tel = ta.Teloscope(user_input(opts, device=K.DEVICE_NONE))          # planning only: no GPU here
    return opts, tel, ShardPlan(tel, lens, abs_pos=abs_pos, tips_only=opts.ultra_fast, world=world)


def _seqs(opts, lens=LENS, seed=41):
    rng = np.random.default_rng(seed)
    return [seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev, n_its=3, iupac=1) if n else b"" for n in lens]


def _oracle(opts, seqs, abs_pos=ABS):
    ob = OracleBackend(opts)
    return [ob.scan_segment(s, abs_pos[i], opts.ultra_fast) for i, s in enumerate(seqs)]


def test_partition_is_consecutive_balanced_and_keeps_clear_of_segment_ends():
    from teloscope_amd.distributed import lpt_partition, shard_info
    for cli, lens in ((HEADLINE + " -t 3000", [2_000_000, 30000, 1500, 0, 900_000, 5_000_001]), (HEADLINE, [3_000_000, 120_000, 7_500_000]),
                      ("-t 9000", [30000, 1500, 52000, 7, 999, 0, 20001, 12345, 64000])):
        for world in (1, 2, 3, 8):
            opts, tel, plan = _plan(cli, world, lens, None)
            tiles = plan.tiles
            assert plan.ranges[0].tile_begin == 0 and plan.ranges[-1].tile_end == plan.n_tiles
            for a, b in zip(plan.ranges, plan.ranges[1:]):
                assert a.tile_end == b.tile_begin and a.window_end == b.window_begin
            assert sum(r.bases for r in plan.ranges) == sum(int(t["owned_bases"]) for t in tiles)
            first = {}
            count = {}
            for t in range(plan.n_tiles):
                s = int(tiles["seg_index"][t])
                first.setdefault(s, t)
                count[s] = count.get(s, 0) + 1
            ctx = int(shard_info(plan, 0).context_tiles)
            tile_bases = int(tiles["owned_bases"].max())
            zone = -(-(opts.terminal_limit + 1) // tile_bases)
            for r in plan.ranges[1:]:
                b = int(r.tile_begin)
                if b in (0, plan.n_tiles):
                    continue
                s = int(tiles["seg_index"][b])
                f, e = first[s], first[s] + count[s]
                if opts.ultra_fast:
                    assert b == f                                # tips-only: a segment's two regions stay together
                else:
                    assert b == f or (b - f >= zone + ctx and e - b >= zone + ctx), (cli, world, b, f, e, zone, ctx)
            if not opts.ultra_fast and world > 1:
                bases = [int(r.bases) for r in plan.ranges]
                assert max(bases) - min(bases) <= 2 * (zone + ctx + 1) * tile_bases + 130_000, (cli, world, bases)
            # what a shard scans: its own tiles plus the context either side, inside the boundary segment
            for p in range(world):
                si = shard_info(plan, p)
                assert si.own_begin == plan.ranges[p].tile_begin and si.own_end == plan.ranges[p].tile_end
                assert si.own_begin - si.ext_begin in (0, ctx) and si.ext_end - si.own_end in (0, ctx)
                if si.own_end > si.own_begin:
                    assert (si.ext_begin == si.own_begin) == (si.own_begin == first[int(tiles["seg_index"][si.own_begin])])
            plan.close()
    rng = np.random.default_rng(3)
    lens = [int(x) for x in np.exp(rng.uniform(np.log(1e6), np.log(250e6), size=200))]
    for world in (1, 2, 4, 8):
        shards = lpt_partition(lens, world)
        assert sorted(i for s in shards for i in s) == list(range(200))


def test_bench_assembly_split_and_message_sizes():
    """configs[2]: the 3 Gb / 200-contig plan cut for 8 ranks — balance, and what the exchange carries."""
    import bench
    from teloscope_amd.distributed import shard_info
    lens = bench.contig_lengths(3_000_000_000, 200, 42)
    opts, tel, plan = _plan(bench.FLAGS, 8, lens, None)
    infos = [shard_info(plan, p) for p in range(8)]
    bases = [int(i.bases) for i in infos]
    assert sum(bases) == 3_000_000_000 and max(bases) - min(bases) < 400_000
    over_links = sum(int(i.msg_bytes) for i in infos[1:])
    assert over_links < 100_000_000, over_links                  # (round 2's full exchange: 245 MB)
    assert all(int(i.window_bytes) == 9 and int(i.visible_bytes) == 2 for i in infos)
    plan.close()


@pytest.mark.parametrize("cli", [HEADLINE + " -t 3000", "-r -g -i -t 2500 -k 120", "-r -w 700 -s 700", "-t 3000"])
def test_messages_merge_to_the_oracle(cli):
    import teloscope_amd as ta
    from teloscope_amd.distributed import finalize_shards, free_segments
    opts, tel, _ = _plan(cli, 1)
    seqs = _seqs(opts)
    exp = _oracle(opts, seqs)
    for world in (1, 2, 3, 5, 8):
        opts, tel, plan = _plan(cli, world)
        msgs = [shardpack.pack_shard(plan, opts, p, exp) for p in range(world)]
        rc, out, cnt = finalize_shards(plan, msgs)
        assert rc == 0, (world, rc, tel._ctx.error())
        for i in range(len(LENS)):
            assert_visible_view_equal(ta.SegmentData(out[i], opts.ultra_fast), exp[i], opts.ultra_fast, cnt[i], "world %d segment %d" % (world, i))
        free_segments(plan, out)
        plan.close()


def test_merge_reports_what_it_cannot_merge():
    from teloscope_amd import _capi as K
    from teloscope_amd.distributed import finalize_shards, free_segments
    cli = HEADLINE + " -t 3000"
    opts, tel, plan = _plan(cli, 3)
    exp = _oracle(opts, _seqs(opts))
    msgs = [shardpack.pack_shard(plan, opts, p, exp) for p in range(3)]
    assert any(int(shardpack.read_header(m)["ext_begin"]) != int(shardpack.read_header(m)["own_begin"]) for m in msgs)
    # flags a shard raises
    for flag, want in ((K.SHARD_OVERFLOW_SCAN, K.SHARD_RETRY_SYNC), (K.SHARD_OVERFLOW_VISIBLE, K.SHARD_RETRY_GROW),
                       (K.SHARD_OVERFLOW_BLOCKS, K.SHARD_RETRY_GROW), (K.SHARD_OUT_OF_CONTEXT, K.SHARD_NEED_FULL)):
        bad = list(msgs)
        bad[1] = shardpack.set_flags(msgs[1], flag)
        assert finalize_shards(plan, bad)[0] == want
    # a terminal block that reaches into a neighbour's tiles: the bounds the other shards assumed do not hold
    split = None
    for p, m in enumerate(msgs):
        h = shardpack.read_header(m)
        segs = shardpack.read_segs(m)
        for s in range(int(h["n_segs"])):
            if segs[s]["flags"] & shardpack.SEG_HAS_START and not segs[s]["flags"] & shardpack.SEG_HAS_END:
                split = (p, s)
    assert split is not None, "no segment is split over two shards in this plan"
    p, s = split
    bad = [m.copy() for m in msgs]
    segs = np.frombuffer(bad[p][128:128 + 64 * int(shardpack.read_header(bad[p])["n_segs"])], dtype=shardpack.SEG_DT).copy()
    segs[s]["fwd_boundary"] = 10 ** 9
    bad[p][128:128 + segs.nbytes] = segs.view(np.uint8)
    assert finalize_shards(plan, bad)[0] == K.SHARD_NEED_FULL
    # messages of another split are refused
    opts2, tel2, plan2 = _plan(cli, 2)
    with pytest.raises(K.TeloscanError):
        finalize_shards(plan2, msgs[:2])
    rc, out, _ = finalize_shards(plan, msgs)
    assert rc == 0
    free_segments(plan, out)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, cli, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from teloscope_amd.distributed import ShardExchange, free_segments
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    opts, tel, plan = _plan(cli, world)
    exp = _oracle(opts, _seqs(opts))
    exch = ShardExchange(plan, rank, torch.device("cpu"), dst=0, slots=2)
    mine = torch.from_numpy(shardpack.pack_shard(plan, opts, rank, exp))
    for slot in (0, 1):                                          # both buffer sets, as the bench's two slots do
        for w in exch.post(mine, slot):
            w.wait()
    action, factor, merged = exch.check(mine, 1)
    assert action == exch.OK and factor == 1
    # a rank whose message overflowed: every rank learns that the scale has to grow
    from teloscope_amd import _capi as K
    bad = torch.from_numpy(shardpack.set_flags(mine.numpy(), K.SHARD_OVERFLOW_VISIBLE)) if rank == world - 1 else mine
    for w in exch.post(bad, 0):
        w.wait()
    action2, factor2, merged2 = exch.check(bad, 0)
    assert action2 == exch.GROW and factor2 >= 2 and merged2 is None
    if rank == 0:
        rc, out, cnt = merged
        import teloscope_amd as ta
        res = []
        for i in range(len(LENS)):
            sd = ta.SegmentData(out[i], opts.ultra_fast)
            res.append((sd.windows, sd.terminalBlocks, sd.interstitialBlocks, sd._m, (cnt[i].n_windows, cnt[i].n_matches, cnt[i].n_canonical, cnt[i].n_forward)))
        free_segments(plan, out)
        q.put((res, exch.bytes_over_links))
    else:
        assert merged is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("cli,world", [(HEADLINE + " -t 3000", 2), (HEADLINE + " -t 3000", 3), ("-t 3000", 2)])
def test_exchange_over_gloo_ranks_merges_to_the_oracle(cli, world):
    import torch.multiprocessing as mp
    from tests.backends import BLOCK_FIELDS, WINDOW_FIELDS
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, cli, q)) for r in range(world)]
    for p in procs:
        p.start()
    res, over_links = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    opts, tel, plan = _plan(cli, world)
    exp = _oracle(opts, _seqs(opts))
    assert over_links > 0
    for i, (wins, term, its, m, cnt) in enumerate(res):
        e = exp[i]
        for f in WINDOW_FIELDS:
            assert np.array_equal(wins[f], e["windows"][f]), (i, f)
        for f in BLOCK_FIELDS:
            assert np.array_equal(term[f], e["terminal_blocks"][f]) and np.array_equal(its[f], e["interstitial_blocks"][f]), (i, f)
        vis = np.concatenate([e["canonical_matches"], e["non_canonical_matches"]])
        vis = vis[np.argsort(vis["position"], kind="stable")]
        assert np.array_equal(m["position"], vis["position"]), i
        if not opts.ultra_fast:
            assert cnt == (len(e["windows"]), len(e["all_matches"]), len(e["canonical_matches"]), len(e["fwd_matches"])), i
