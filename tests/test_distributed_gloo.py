"""The N > 1 path on CPU: two gloo ranks run the product's own shard / gather / merge code
(teloscope_amd.distributed: ShardPlan, gather_shards, decode_segments — the code bench.py runs at N > 1, on
a planning-only context) and the result on rank 0 must equal the single-process result bit for bit.

Without a GPU the tile results cannot come from the HIP kernels, so each rank computes the raw arrays of ITS
tile range — window records, tile directory entries, packed match records in tile order — from the CPU oracle,
in exactly the layout ts_scan_tiles + ts_batch_export leave on a device.  What is under test is the plan
split, the exchange and the reassembly, not the scan (tests/test_gpu_parity.py does that on the GPU, and
test_sharded_scan_equals_whole_scan there runs this same exchange over real kernel output)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LENS = [30000, 1500, 52000, 7, 999, 0, 20001, 12345, 64000]
CLI = {"windows": "x -w 1000 -s 500 -r -g -i", "tips": "x -t 9000"}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _seqs():
    from tests import seqgen
    return [seqgen.chromosome(np.random.default_rng(100 + i), n, n_its=2) if n else b"" for i, n in enumerate(LENS)]


def _plan(mode, world):
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.cli import parse_cli, user_input
    from teloscope_amd.distributed import ShardPlan
    opts = parse_cli(CLI[mode])
    tel = ta.Teloscope(user_input(opts, device=K.DEVICE_NONE))          # planning only: no GPU here
    return opts, ShardPlan(tel, LENS, abs_pos=[1000 * i for i in range(len(LENS))], tips_only=opts.ultra_fast, world=world)


def oracle_shard(plan, opts, rank, seqs):
    """Raw result arrays of the rank's tile range, from the oracle: (windows [n*8], stats [n*4], dense) as
    int32 views of the u32 values."""
    from tests.backends import OracleBackend
    ob = OracleBackend(opts)
    r = plan.ranges[rank]
    tiles = plan.tiles[r.tile_begin:r.tile_end]
    tips = plan.tips_only
    wins = np.zeros((0 if tips else r.window_end - r.window_begin, 8), dtype=np.uint32)
    stats = np.zeros((len(tiles), 4), dtype=np.uint32)
    dense = []
    cache = {}
    for ti, t in enumerate(tiles):
        si = int(t["seg_index"])
        if si not in cache:
            cache[si] = ob.scan_segment(seqs[si], plan.abs_pos[si], tips)
        res = cache[si]
        if tips:                                               # src/teloscope.cpp:566-570 fills fwd/rev only
            m = np.concatenate([res["fwd_matches"], res["rev_matches"]])
            m = m[np.argsort(m["position"], kind="stable")]
        else:
            m = res["all_matches"]
        rel = m["position"].astype(np.int64) - plan.abs_pos[si] - int(t["seg_offset"])
        sel = (rel >= 0) & (rel < int(t["owned_bases"]))
        rec = (rel[sel].astype(np.uint32) << 2) | (m["is_forward"][sel].astype(np.uint32) << 1) | m["is_canonical"][sel].astype(np.uint32)
        stats[ti] = [len(rec), int(m["is_canonical"][sel].sum()), int(m["is_forward"][sel].sum()), 0]
        dense.append(rec)
        if not tips:
            w0 = int(t["first_window"])
            seg_w0 = int(plan.tiles[plan.tiles["seg_index"] == si]["first_window"].min())
            for j in range(int(t["n_windows"])):
                ow = res["windows"][w0 - seg_w0 + j]
                wins[w0 - r.window_begin + j] = list(ow["nucleotide_counts"]) + [
                    ow["canonical_covered"], ow["non_canonical_covered"], ow["fwd_covered"], ow["rev_covered"]]
    dense = np.concatenate(dense) if dense else np.zeros(0, dtype=np.uint32)
    return wins.reshape(-1).view(np.int32), stats.reshape(-1).view(np.int32), dense.view(np.int32)


def _worker(rank, world, port, mode, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from teloscope_amd.distributed import gather_shards

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    opts, plan = _plan(mode, world)
    w, s, d = oracle_shard(plan, opts, rank, _seqs())
    spare = torch.zeros(17, dtype=torch.int32)                  # the export buffer is larger than what it holds
    got = gather_shards(plan, rank, torch.from_numpy(w.copy()), torch.from_numpy(s.copy()),
                        torch.cat([torch.from_numpy(d.copy()), spare]), len(d), dst=0)
    again = gather_shards(plan, rank, torch.from_numpy(w.copy()), torch.from_numpy(s.copy()),
                          torch.from_numpy(d.copy()), len(d), dst=0, async_op=True, directory_only=True).wait()
    assert plan.wire16_ok                                        # (the first exchange went over the wire as u16)
    plain = gather_shards(plan, rank, torch.from_numpy(w.copy()), torch.from_numpy(s.copy()),
                          torch.from_numpy(d.copy()), len(d), dst=0, wire16=False)
    if rank == 0:
        assert again.n_records == 0 and torch.equal(again.stats, got.stats)
        assert torch.equal(plain.windows, got.windows) and torch.equal(plain.stats, got.stats)
        assert plain.n_records == got.n_records and torch.equal(plain.dense[:plain.n_records], got.dense[:got.n_records])
        q.put((got.windows.numpy(), got.stats.numpy(), got.dense[:got.n_records].numpy(), got.counts))
    else:
        assert got is None and again is None and plain is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode,world", [("windows", 2), ("tips", 2), ("windows", 3)])
def test_gather_over_ranks_equals_single_process(mode, world):
    import torch.multiprocessing as mp
    from teloscope_amd.distributed import decode_segments
    from tests.backends import OracleBackend
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    windows, stats, dense, counts = q.get(timeout=240)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # (a) bit-equal to the single-process arrays
    opts, plan1 = _plan(mode, 1)
    seqs = _seqs()
    w1, s1, d1 = oracle_shard(plan1, opts, 0, seqs)
    assert np.array_equal(windows, w1) and np.array_equal(stats, s1) and np.array_equal(dense, d1)
    assert sum(counts) == len(d1) and min(counts) > 0
    # (b) and, decoded per segment in global order, to what the oracle says about every segment
    ob = OracleBackend(opts)
    for si, seg in enumerate(decode_segments(plan1, windows.view(np.uint32), stats.view(np.uint32), dense.view(np.uint32))):
        e = ob.scan_segment(seqs[si], plan1.abs_pos[si], opts.ultra_fast)
        if opts.ultra_fast:
            em = np.concatenate([e["fwd_matches"], e["rev_matches"]])
            em = em[np.argsort(em["position"], kind="stable")]
        else:
            em = e["all_matches"]
            assert np.array_equal(seg["windows"][:, 4], e["windows"]["canonical_covered"])
            assert np.array_equal(seg["windows"][:, :4], e["windows"]["nucleotide_counts"])
        assert np.array_equal(seg["matches"]["position"], em["position"]), si
        assert np.array_equal(seg["matches"]["is_forward"], em["is_forward"] != 0)
        assert np.array_equal(seg["matches"]["is_canonical"], em["is_canonical"] != 0)
