"""GPU parity of the pieces added in round 2:

  * tile-range shards: the parts of one plan scanned separately (ts_batch_restrict / bind_results / export),
    concatenated, adopted (ts_batch_adopt) — equal to the whole scan and to the oracle, for any number of parts;
  * bench.py --gpus 2 on one GPU (two rank processes, gloo): the assembled arrays equal a single-GPU scan;
  * full-size parity under the driver's eyes: bench.py --verify at configs[1] (3.0 Gb) and configs[4] (15 Gb),
    a segment compared record for record with the oracle behind 2^32 input bytes and > 65 536 tiles, and a slice
    around position 2^32 of one 4.5 Gb contig;
  * the read filter: pattern sets outside the tiled kernel (the path that used to relock its own mutex), and the
    read shard over several contexts.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import harness as H
from tests import seqgen
from tests.backends import (BLOCK_FIELDS, WINDOW_FIELDS, OracleBackend, OracleReadFilter, ProductBackend,
                            ProductReadFilter, assert_segment_equal, segment_as_dict)

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADLINE = "-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -w 1000 -s 500 -r -g -e -m -i"


def _teloscope(cli, device=0):
    import teloscope_amd as ta
    from teloscope_amd.cli import parse_cli, user_input
    opts = parse_cli("x.fa " + cli)
    return opts, ta.Teloscope(user_input(opts, device=device))


def _scan_parts(plan, buf, dev):
    """Every part of the plan scanned on its own restricted batch (as a rank would), the parts' arrays
    concatenated in rank order (what the gather does): (windows, stats, dense[:n], counts)."""
    import torch
    from teloscope_amd.distributed import HipShard
    stream = torch.cuda.current_stream()
    sptr = C.c_void_p(stream.cuda_stream)
    ws, ss, ds, counts = [], [], [], []
    for p in range(plan.world):
        r = plan.ranges[p]
        local = buf[r.input_begin:r.input_end].clone()         # a rank holds only the bytes its range reads
        hs = HipShard(plan, p, dev, slots=1)
        hs.scan(local.data_ptr(), sptr, 0)
        n = hs.finish(local.data_ptr(), sptr, 0)
        ws.append(hs.windows[0].clone()); ss.append(hs.stats[0].clone()); ds.append(hs.dense[0][:n].clone())
        counts.append(n)
        hs.close()
    return torch.cat(ws), torch.cat(ss), torch.cat(ds), counts


@pytest.mark.parametrize("cli", [HEADLINE, "-r -g -e -m -i", "-c CCCTAAA -w 2000 -s 1000 -r -g -e -m -i", "", "-t 3000"])
def test_sharded_scan_equals_whole_scan_and_oracle(cli):
    import torch
    from teloscope_amd import _capi as K
    from teloscope_amd.distributed import Assembled, ShardPlan, adopt
    dev = torch.device("cuda", 0)
    opts, tel = _teloscope(cli)
    rng = np.random.default_rng(len(cli) + 17)
    lens = [70001, 7, 250003, 0, 1999, 1_000_000, 16500, 333_333]
    seqs = [seqgen.chromosome(rng, n, opts.canonical_fwd, opts.canonical_rev, n_its=3, iupac=2) if n else b"" for n in lens]
    abs_pos = [11 * i for i in range(len(lens))]
    tips = opts.ultra_fast
    whole = None
    for world in (1, 2, 3, 5):
        plan = ShardPlan(tel, lens, abs_pos=abs_pos, tips_only=tips, world=world)
        buf = torch.zeros(int(plan.info.input_bytes), dtype=torch.uint8, device=dev)
        for off, s in zip(plan.segment_offsets(), seqs):
            if len(s):
                buf[off:off + len(s)] = torch.frombuffer(bytearray(s), dtype=torch.uint8).to(dev)
        w, s_, d, counts = _scan_parts(plan, buf, dev)
        if whole is None:
            whole = (w, s_, d)
        else:
            assert torch.equal(w, whole[0]) and torch.equal(s_, whole[1]) and torch.equal(d, whole[2]), world
        # adopt the concatenation and download: must be what the oracle says, segment by segment
        a = Assembled(w, s_, d if d.numel() else torch.zeros(1, dtype=torch.int32, device=dev), int(d.numel()), counts)
        b = adopt(plan, a)
        L = K.lib()
        out = (K.SegmentOut * len(lens))()
        assert L.ts_batch_download(b, None, out) == 0, tel._ctx.error()
        import teloscope_amd as ta
        orac = OracleBackend(opts)
        for i, sq in enumerate(seqs):
            got = segment_as_dict(ta.SegmentData(out[i], tips))
            assert_segment_equal(got, orac.scan_segment(sq, abs_pos[i], tips), tips, ctx="world %d segment %d" % (world, i))
        L.ts_free_segments(out, len(lens))
        out2 = (K.SegmentOut * len(lens))()
        assert L.ts_batch_download_blocks(b, out2) == 0, tel._ctx.error()
        for i, sq in enumerate(seqs):
            e = orac.scan_segment(sq, abs_pos[i], tips)
            g = ta.SegmentData(out2[i], tips)
            for f in BLOCK_FIELDS:
                assert np.array_equal(g.terminalBlocks[f], e["terminal_blocks"][f]), (world, i, f)
                assert np.array_equal(g.interstitialBlocks[f], e["interstitial_blocks"][f]), (world, i, f)
        L.ts_free_segments(out2, len(lens))
        L.ts_batch_destroy(b)
        plan.close()


def _bench(*args, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def test_bench_two_ranks_on_one_gpu_equals_single_gpu():
    """`python bench.py --gpus 2` spawns its own two ranks (they share this GPU, so the exchange runs over gloo):
    configs[2] in small — the same assembly, two shards, one message per rank — and what the messages merge to on
    rank 0 is compared with a single-GPU scan of the whole assembly: windows and blocks byte for byte, the visible match
    records, the per-segment counts (--verify)."""
    out = _bench("--gpus", "2", "--gbases", "0.3", "--contigs", "14", "--steps", "4", "--warmup", "2", "--verify")
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 4
    cfg = out["config"]
    assert len(cfg["bases_per_rank"]) == 2 and abs(cfg["bases_per_rank"][0] - cfg["bases_per_rank"][1]) <= 400_000
    x = cfg["exchange"]
    assert x["bytes_over_links_per_step"] == x["message_bytes_per_rank"][1] and x["window_bytes"] == 9
    assert all(v <= c for v, c in zip(x["visible_records_per_rank"], x["visible_capacity_per_rank"]))
    v = out["verify"]
    assert v["contigs_checked"] == 14 and v["matches_checked"] == cfg["matches"]
    assert v["sharded_equals_single_gpu"]["segments"] == 14 and v["sharded_equals_single_gpu"]["visible_matches"] == sum(x["visible_records_per_rank"])
    assert cfg["step_split"]["exchange_ms"] > 0


def test_bench_rehearses_the_eight_rank_exchange_shape_on_rccl(monkeypatch):
    """One rank on the real RCCL group, the sharded code path at full size (configs[1], 3 Gb) for 50 steps with --verify,
    posting per step what the ranks of an 8-rank job post between them: rank 0's grouped batch of seven receives (message
    sizes of the 8-way plan) and the seven sends, fourteen operations in one batch_isend_irecv, several steps in flight,
    from the pack streams.  The received bytes must be the bytes sent, and the merged writer view must still equal the
    single-GPU scan (windows, blocks, visible match records — compared at full size)."""
    monkeypatch.setenv("TS_BENCH_FORCE_STRONG", "1")
    monkeypatch.setenv("TS_BENCH_REHEARSE_WORLD", "8")
    out = _bench("--steps", "50", "--warmup", "3", "--verify", "--no-cpu-baseline", "--no-e2e", "--no-reads")
    cfg = out["config"]
    r = cfg["exchange"]["rehearsal"]
    assert r["of_world"] == 8 and r["posted_ops_per_step"] == 14 and r["received_equals_sent"] and len(r["message_bytes"]) == 7
    assert r["steps_posted"] >= 50 and r["bytes_per_step"] < 100_000_000
    v = out["verify"]
    assert v["windows_checked_field_by_field"] >= 10_000
    assert v["sharded_equals_single_gpu"]["visible_matches"] > 2_000_000       # compared, not skipped, at 3 Gb


def test_bench_full_exchange_two_ranks_on_one_gpu_equals_single_gpu():
    """Round 2's exchange (every record assembled on rank 0) stays available behind --full-exchange: it is what a batch
    takes when the shards' assumptions do not hold for its input."""
    out = _bench("--gpus", "2", "--gbases", "0.3", "--contigs", "14", "--steps", "4", "--warmup", "2", "--verify", "--full-exchange")
    cfg = out["config"]
    assert sum(cfg["records_per_rank"]) == cfg["matches"] == out["verify"]["sharded_equals_single_gpu"]["records"]
    assert cfg["summaries_only_variant"]["value"] > 0 and cfg["step_split"]["exchange_ms"] > 0


def test_bench_read_shard_over_two_ranks_equals_one_rank_and_the_oracle():
    """configs[3] in small (500 k reads, 7.5 Gb) through bench.py's own read-shard code: `--reads --gpus 2` (two rank
    processes sharing this GPU, the pass bytes gathered over gloo in input order) against `--reads` on one rank — the same
    pass bytes (the reference pins -j 1 == -j 8 the same way: validateFiles/fastq_subset_large_j1.tst / _j8.tst) — and both
    against the oracle's ReadTelomereFilter::matches on a sample from both ends of the read set (--verify)."""
    one = _bench("--reads", "--n-reads", "5e5", "--steps", "2", "--warmup", "1", "--verify", "--no-cpu-baseline")
    two = _bench("--reads", "--n-reads", "5e5", "--steps", "2", "--warmup", "1", "--verify", "--no-cpu-baseline", "--gpus", "2")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert one["config"]["pass_bytes_sha1"] == two["config"]["pass_bytes_sha1"]
    assert one["config"]["kept"] == two["config"]["kept"] > 2000
    for o in (one, two):
        assert o["verify"]["reads_checked_against_oracle"] > 100
        assert 0.02 < o["roofline"]["frac"] < 1.0 and o["roofline"]["algorithmic_bytes"] < 1.001 * o["config"]["bases"] / o["n_gpus"] + 1e6


def test_bench_reads_at_five_million_reads():
    """configs[3] at a tenth of its size — 5 M reads, 75 Gbases resident in HBM (BASELINE: "may be scaled to 5 M reads for
    CI") — through bench.py --reads on one GPU with --verify: every planted carrier kept, a sample from both ends of the read
    set equal to the oracle's ReadTelomereFilter::matches, and the roofline figure over the whole step (scan + predicate)."""
    out = _bench("--reads", "--n-reads", "5e6", "--steps", "2", "--warmup", "1", "--verify", "--no-cpu-baseline", timeout=1200)
    assert out["n_gpus"] == 1 and out["config"]["reads"] == 5_000_000
    assert out["config"]["bases"] > 70e9 and out["config"]["kept"] > 20_000
    assert out["verify"]["reads_checked_against_oracle"] > 100
    assert 0.3 < out["roofline"]["frac"] < 1.0


def test_bench_under_the_drivers_launcher():
    """The driver's own launch form for N > 1 — python -m torch.distributed.run --nproc-per-node N bench.py --gpus N, ranks
    from RANK / LOCAL_RANK / WORLD_SIZE in the environment — with two ranks on this one GPU (gloo): one JSON line from rank
    0, the merged result equal to a single-GPU scan."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--gbases", "0.3", "--contigs", "14", "--verify", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 3
    assert out["verify"]["sharded_equals_single_gpu"]["segments"] == 14


def test_bench_two_ranks_configs4_plant_equals_single_gpu():
    """configs[4] (15 Gb plant assembly, k = 7, w = 2000 s = 1000) as two ranks (gloo, sharing this GPU): the merged writer
    view of the two shards against a single-GPU scan of the whole assembly, at full size (--verify)."""
    out = _bench("--gpus", "2", "--verify", "--gbases", "15", "--contigs", "521", "--flags", "-c CCCTAAA -w 2000 -s 1000 -r -g -e -m -i",
                 "--steps", "2", "--warmup", "1", "--no-cpu-baseline", timeout=1500)
    assert out["n_gpus"] == 2 and out["config"]["bases"] == 15_000_000_000
    v = out["verify"]
    assert v["contigs_checked"] == 521 and v["matches_checked"] == out["config"]["matches"]
    assert v["sharded_equals_single_gpu"]["segments"] == 521 and v["sharded_equals_single_gpu"]["windows"] > 14_000_000


def test_default_line_carries_the_read_filter_and_the_launch_protocol():
    """What the driver's plain `python bench.py` prints, minus the slow CPU leg: the `reads` sub-record (configs[3] at
    500 k reads, roofline over the whole step by 1 B/base + 1 bit/read, oracle-checked sample), the single-shot figures
    beside the settled value, and the PCIe-inclusive legs incl. ts_scan_segments_multi."""
    out = _bench("--steps", "5", "--warmup", "2", "--no-cpu-baseline")
    lp = out["config"]["launch_protocol"]
    assert lp["single_shot_ms"] > 0 and lp["first_scan_ms_incl_module_load_and_allocation"] >= lp["single_shot_ms"] and lp["settled_ms_per_step"] == out["ms_per_step"]
    r = out["reads"]
    assert r["reads"] == 500_000 and r["kept"] >= r["planted_carriers"] > 2000 and r["oracle_checked_reads"] > 100
    assert abs(r["roofline"]["algorithmic_bytes"] - (r["bases"] + 62500)) <= 1 and 0.02 < r["roofline"]["frac"] < 1.0
    pc = out["pcie_inclusive"]
    assert pc["writer_view_multi"]["n_ctx"] >= 1 and pc["writer_view_multi"]["matches"] == out["config"]["matches"]
    assert 0 < pc["writer_view_multi"]["visible_matches"] < 0.05 * out["config"]["matches"]


def test_bench_verify_full_size_configs1():
    """configs[1] at its real size: 3.0 Gb, 200 contigs, 91.5 M matches — per-contig counts and nucleotide totals
    against the independent torch computation."""
    out = _bench("--verify", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-reads")
    assert out["config"]["bases"] == 3_000_000_000 and out["verify"]["contigs_checked"] == 200
    assert out["verify"]["matches_checked"] == out["config"]["matches"] > 90_000_000


def test_bench_verify_full_size_configs4_plant():
    """configs[4] on one GPU: 15 Gb, -c CCCTAAA -w 2000 -s 1000 (k = 7), ~15 M windows — the same properties."""
    out = _bench("--verify", "--gbases", "15", "--contigs", "521", "--flags", "-c CCCTAAA -w 2000 -s 1000 -r -g -e -m -i",
                 "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--no-reads", timeout=1500)
    assert out["config"]["bases"] == 15_000_000_000 and out["verify"]["contigs_checked"] == 521
    assert out["config"]["windows"] > 14_000_000 and out["verify"]["matches_checked"] == out["config"]["matches"]


def _device_random(n, dev, seed):
    import torch
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    for a in range(0, n, 1 << 28):
        b = min(n, a + (1 << 28))
        out[a:b] = lut[torch.randint(0, 4, (b - a,), dtype=torch.uint8, device=dev, generator=g).long()]
    return out


def test_segment_behind_4g_bytes_and_65536_tiles_record_for_record():
    """A 2 Mb segment placed behind a 4.4 Gb one in the same batch: its bytes start beyond input offset 2^32 and
    its tiles beyond index 65 536.  Windows, every match (position, orientation, canonical) in push order and the
    blocks of that segment equal the oracle's."""
    import torch
    from teloscope_amd import _capi as K
    from teloscope_amd.distributed import HipShard, ShardPlan
    import teloscope_amd as ta
    dev = torch.device("cuda", 0)
    opts, tel = _teloscope(HEADLINE)
    rng = np.random.default_rng(99)
    fixture = seqgen.chromosome(rng, 2_000_003, opts.canonical_fwd, opts.canonical_rev, n_its=6, iupac=3)
    lens = [4_400_000_000, len(fixture)]
    plan = ShardPlan(tel, lens, abs_pos=[0, 5_000_000_000], world=1)
    offs = plan.segment_offsets()
    tiles = plan.tiles
    first_tile = int(np.flatnonzero(tiles["seg_index"] == 1)[0])
    assert offs[1] > 2 ** 32 and first_tile > 65536
    buf = torch.empty(int(plan.info.input_bytes), dtype=torch.uint8, device=dev)
    buf[:lens[0]] = _device_random(lens[0], dev, 5)
    buf[offs[1]:offs[1] + lens[1]] = torch.frombuffer(bytearray(fixture), dtype=torch.uint8).to(dev)
    stream = torch.cuda.current_stream()
    sptr = C.c_void_p(stream.cuda_stream)
    hs = HipShard(plan, 0, dev, slots=1)
    hs.scan(buf.data_ptr(), sptr, 0)
    n = hs.finish(buf.data_ptr(), sptr, 0)
    stats = hs.stats[0].view(-1, 4)
    start = int(stats[:first_tile, 0].to(torch.int64).sum().item())
    recs = hs.dense[0][start:n].cpu().numpy().view(np.uint32)
    tcount = stats[first_tile:, 0].cpu().numpy().astype(np.int64)
    tbase = tiles["seg_offset"][first_tile:].astype(np.uint64)
    pos = 5_000_000_000 + np.repeat(tbase, tcount) + (recs >> 2).astype(np.uint64)
    e = OracleBackend(opts).scan_segment(fixture, 5_000_000_000, False)
    em = e["all_matches"]
    assert len(em) == len(recs) > 50_000
    assert np.array_equal(pos, em["position"])
    assert np.array_equal((recs & 2) != 0, em["is_forward"] != 0) and np.array_equal((recs & 1) != 0, em["is_canonical"] != 0)
    w0 = int(tiles["first_window"][first_tile])
    wins = hs.windows[0].view(-1, 8)[w0:].cpu().numpy().view(np.uint32)
    assert len(wins) == len(e["windows"])
    assert np.array_equal(wins[:, :4], e["windows"]["nucleotide_counts"])
    for col, f in zip(range(4, 8), ("canonical_covered", "non_canonical_covered", "fwd_covered", "rev_covered")):
        assert np.array_equal(wins[:, col], e["windows"][f]), f
    # blocks of the segment, called on the device over the same resident stream
    L = K.lib()
    out = (K.SegmentOut * 2)()
    b = hs.batches[0]
    assert L.ts_batch_sync(b) == 0 and L.ts_batch_download_blocks(b, out) == 0, tel._ctx.error()
    g = ta.SegmentData(out[1], False)
    assert len(g.terminalBlocks) == len(e["terminal_blocks"]) >= 2
    for f in BLOCK_FIELDS:
        assert np.array_equal(g.terminalBlocks[f], e["terminal_blocks"][f]), f
        assert np.array_equal(g.interstitialBlocks[f], e["interstitial_blocks"][f]), f
    for f in WINDOW_FIELDS:
        assert np.array_equal(g.windows[f], e["windows"][f]), f
    L.ts_free_segments(out, 2)
    hs.close()
    plan.close()


def test_slice_around_position_4g_of_one_contig():
    """One 4.5 Gb contig: the windows and matches of a 1 Mb slice around position 2^32 equal what the oracle gives
    for that slice scanned on its own (windows are independent: window j of the slice is window j0 + j of the
    contig when the slice starts at a multiple of the step)."""
    import torch
    from teloscope_amd.distributed import HipShard, ShardPlan
    dev = torch.device("cuda", 0)
    opts, tel = _teloscope(HEADLINE)
    n_total = 4_500_000_000
    plan = ShardPlan(tel, [n_total], world=1)
    buf = torch.empty(int(plan.info.input_bytes), dtype=torch.uint8, device=dev)
    buf[:n_total] = _device_random(n_total, dev, 6)
    s, w = opts.step, opts.window_size
    a = (2 ** 32 - 500_000) // s * s
    ln = 1_000_000
    rep = torch.frombuffer(bytearray(b"TTAGGG" * 2000 + b"CCCTAA" * 2000), dtype=torch.uint8).to(dev)
    buf[2 ** 32 - 9000:2 ** 32 - 9000 + rep.numel()] = rep                 # a dense stretch across the 2^32 boundary
    buf[2 ** 32 + 40_000:2 ** 32 + 40_100] = ord("N")
    sl = bytes(buf[a:a + ln].cpu().numpy().tobytes())
    stream = torch.cuda.current_stream()
    sptr = C.c_void_p(stream.cuda_stream)
    hs = HipShard(plan, 0, dev, slots=1)
    hs.scan(buf.data_ptr(), sptr, 0)
    n = hs.finish(buf.data_ptr(), sptr, 0)
    e = OracleBackend(opts).scan_segment(sl, a, False)
    nw = ln // s - 2                                                      # windows wholly inside the slice
    wins = hs.windows[0].view(-1, 8)[a // s:a // s + nw].cpu().numpy().view(np.uint32)
    assert np.array_equal(wins[:, :4], e["windows"]["nucleotide_counts"][:nw])
    for col, f in zip(range(4, 8), ("canonical_covered", "non_canonical_covered", "fwd_covered", "rev_covered")):
        assert np.array_equal(wins[:, col], e["windows"][f][:nw]), f
    tiles = plan.tiles
    tb = int(tiles["owned_bases"][0])
    t0, t1 = a // tb, (a + ln) // tb + 1
    stats = hs.stats[0].view(-1, 4)
    start = int(stats[:t0, 0].to(torch.int64).sum().item())
    cnt = stats[t0:t1, 0].cpu().numpy().astype(np.int64)
    recs = hs.dense[0][start:start + int(cnt.sum())].cpu().numpy().view(np.uint32)
    pos = np.repeat(tiles["seg_offset"][t0:t1].astype(np.uint64), cnt) + (recs >> 2).astype(np.uint64)
    hi = a + nw * s
    sel = (pos >= a) & (pos < hi)
    em = e["all_matches"]
    esel = em["position"] < hi
    assert sel.sum() == esel.sum() > 10_000
    assert np.array_equal(pos[sel], em["position"][esel])
    assert np.array_equal((recs[sel] & 2) != 0, em["is_forward"][esel] != 0)
    assert np.array_equal((recs[sel] & 1) != 0, em["is_canonical"][esel] != 0)
    hs.close()
    plan.close()


@pytest.mark.parametrize("cli", ["--fastq-subset -p TTAGGG,TTAGG", "--fastq-subset -c TTAGGGTTA -x 0 -l 30",
                                 "--fastq-subset -c AACCCTAACC -x 1",
                                 "--fastq-subset -x 0 -p TTAG,TTAGG,TTAGGG,TTTAGGG,TTTTAGGG,TTAGGGTTA,TTAGGGTTAG,TTAGGGTTAGG,TTAGGGTTAGGG",   # nine lengths: the wide form
                                 "--fastq-subset -c TTAGGG -x 0 -p TTAGGG," + "TTAGGG" * 7])
def test_read_filter_on_pattern_sets_outside_the_tiled_kernel(cli):
    """Mixed-length -p sets and 9-10 nt canonical motifs take the general kernels inside ts_filter_reads (the path
    that used to relock the context's call mutex and hang)."""
    opts = H.parse_cli(cli)
    rng = np.random.default_rng(7)
    unit_f, unit_r = opts.canonical_fwd.encode(), opts.canonical_rev.encode()
    reads = []
    for i in range(60):
        body = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=int(rng.integers(200, 4000))))
        if i % 3 == 0:
            body = unit_f * int(rng.integers(3, 40)) + body
        elif i % 3 == 1:
            body = body + unit_r * int(rng.integers(3, 40))
        reads.append(body)
    got = ProductReadFilter(opts).filter(reads)
    assert got == OracleReadFilter(opts).filter(reads)
    assert 0 < sum(got) < len(got)


def test_read_shard_over_two_contexts_equals_one():
    """ts_filter_reads_multi: the batch cut into consecutive shards of equal bases, one per context (here two
    contexts on one GPU), pass bytes in input order — equal to one context and to the oracle."""
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.cli import user_input
    opts = H.parse_cli("--fastq-subset -l 42")
    rng = np.random.default_rng(11)
    reads = []
    for i in range(400):
        n = int(rng.integers(500, 30000))
        s = seqgen.chromosome(rng, n, n_its=1) if i % 7 else seqgen.chromosome(rng, n)
        if i % 5 == 0:
            s = bytes(rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n))
        reads.append(s)
    exp = OracleReadFilter(opts).filter(reads)
    filters = [ta.ReadTelomereFilter(user_input(opts, device=0)) for _ in range(3)]
    L = K.lib()
    n = len(reads)
    arr = (C.c_char_p * n)(*reads)
    lens = (C.c_uint64 * n)(*[len(r) for r in reads])
    for nctx in (1, 2, 3):
        ctxs = (C.c_void_p * nctx)(*[f._ctx.ptr for f in filters[:nctx]])
        out = (C.c_uint8 * n)()
        assert L.ts_filter_reads_multi(ctxs, nctx, arr, lens, n, out) == 0
        assert [bool(x) for x in out] == exp, nctx
    assert 0 < sum(exp) < n


def test_taken_tiles_equal_dealt_tiles_and_survive_overflow():
    """A range above 64 Mb is scanned with the waves TAKING their tiles (ticket counters); below it, after an overflow of a
    wave's record region, and with TS_DEALT_TILES=1 the tiles are dealt round-robin.  Which wave scans a tile must not
    show in the results: window records, tile directory and the tile-ordered record stream are compared array for array —
    taken vs dealt, and a plan whose tiny match capacity forces overflow -> grow -> rescan vs both."""
    import torch
    from teloscope_amd import distributed as D
    dev = torch.device("cuda:0")
    opts, tel = _teloscope(HEADLINE)
    lens = [120_000_123, 60_000_001, 29_999_999, 777]
    plan = D.ShardPlan(tel, lens, world=1)
    offsets = plan.segment_offsets()
    buf = torch.zeros(int(plan.info.input_bytes), dtype=torch.uint8, device=dev)
    rng = np.random.default_rng(77)
    for i, (o, n) in enumerate(zip(offsets, lens)):
        buf[o:o + n] = _device_random(n, dev, 500 + i)
        tract = torch.from_numpy(seqgen.mutate(rng, seqgen.repeat_array("TTAGGG", 3000), 0.02).copy()).to(dev)
        m = min(n, tract.numel())
        buf[o:o + m] = tract[:m]                                  # a dense telomere at every segment start
    assert "TS_DEALT_TILES" not in os.environ
    taken = _scan_parts(plan, buf, dev)
    os.environ["TS_DEALT_TILES"] = "1"
    try:
        dealt = _scan_parts(plan, buf, dev)
    finally:
        del os.environ["TS_DEALT_TILES"]
    tight = D.ShardPlan(tel, lens, world=1, match_capacity=5000)     # far below the ~6.4 M records of this input
    regrown = _scan_parts(tight, buf, dev)
    assert taken[3] == dealt[3] == regrown[3] and taken[3][0] > 6_000_000
    for name, a, b, c in zip(("windows", "tile directory", "records"), taken, dealt, regrown):
        assert torch.equal(a, b), name + ": taken vs dealt"
        assert torch.equal(a, c), name + ": taken vs regrown after overflow"
    # and two segments against the oracle, so that "equal" is not "equally wrong"
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    w, s_, d, counts = taken
    b = D.adopt(plan, D.Assembled(w, s_, d, int(d.numel()), counts))
    L = K.lib()
    out = (K.SegmentOut * len(lens))()
    assert L.ts_batch_download(b, None, out) == 0, tel._ctx.error()
    orac = OracleBackend(opts)
    for i in (2, 3):
        seq = bytes(buf[offsets[i]:offsets[i] + lens[i]].cpu().numpy())
        assert_segment_equal(segment_as_dict(ta.SegmentData(out[i], False)), orac.scan_segment(seq, 0, False), False, ctx="segment %d" % i)
    L.ts_free_segments(out, len(lens))
    L.ts_batch_destroy(b)
    plan.close()
    tight.close()


def test_batches_of_different_geometries_alternate():
    """Three batches alive at once whose launches differ in everything the launch depends on — two workgroups of ten waves
    with 80 KB of LDS each (the headline patterns), one workgroup of sixteen with 160 KB (default flags: sparse matches), the
    2-bit-table kernel (k = 7) and a tips-only batch — scanned in turn, twice, out of order: the dynamic-LDS limit and the
    kernel build are properties of a launch's function, not of the last batch prepared.  Every scan of a batch must give
    the same arrays, and they are the whole-scan arrays of test_sharded_scan_equals_whole_scan_and_oracle's kind."""
    import torch
    from teloscope_amd import distributed as D
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream()
    sptr = C.c_void_p(stream.cuda_stream)
    lens = [3_000_017, 999, 1_500_000]
    clis = [HEADLINE, "-r -g -e -m -i", "-c CCCTAAA -w 2000 -s 1000 -r -g -e -m -i", "-t 3000"]
    keep, plans, shards, bufs = [], [], [], []
    for j, cli in enumerate(clis):
        opts, tel = _teloscope(cli)
        plan = D.ShardPlan(tel, lens, world=1)
        buf = torch.zeros(int(plan.info.input_bytes), dtype=torch.uint8, device=dev)
        for i, (o, n) in enumerate(zip(plan.segment_offsets(), lens)):
            buf[o:o + n] = _device_random(n, dev, 900 + 10 * j + i)
        keep.append(tel); plans.append(plan); bufs.append(buf)
        shards.append(D.HipShard(plan, 0, dev, slots=1))

    def scan(j):
        shards[j].scan(bufs[j].data_ptr(), sptr, 0)
        n = shards[j].finish(bufs[j].data_ptr(), sptr, 0)
        return shards[j].windows[0].clone(), shards[j].stats[0].clone(), shards[j].dense[0][:n].clone()

    first = [scan(j) for j in range(len(clis))]
    assert sum(int(f[2].numel()) for f in first) > 100_000
    for j in (2, 0, 3, 1, 0, 2):
        again = scan(j)
        for name, a, b in zip(("windows", "tile directory", "records"), first[j], again):
            assert torch.equal(a, b), "%s differ on a later scan of %r" % (name, clis[j])
    # and against fresh single-batch processes' worth of truth: a new plan, scanned alone
    for j in (0, 1):
        opts, tel = _teloscope(clis[j])
        alone = D.ShardPlan(tel, lens, world=1)
        w, s_, d, _ = _scan_parts(alone, bufs[j], dev)
        assert torch.equal(w, first[j][0]) and torch.equal(s_, first[j][1]) and torch.equal(d, first[j][2])
        alone.close()
    for sh, plan in zip(shards, plans):
        sh.close()
        plan.close()


def test_bind_thread_to_device_node():
    """ts_bind_thread_to_device on a real context: where the host publishes the device's NUMA node, a worker thread that calls
    it ends up on a subset of its former CPUs (and the call says 1); where it does not, the mask stays and the call says 0."""
    import threading
    from teloscope_amd import _capi as K
    opts, tel = _teloscope(HEADLINE)
    res = {}

    def worker():
        before = os.sched_getaffinity(0)
        res["rc"] = K.lib().ts_bind_thread_to_device(tel._ctx.ptr)
        res["before"], res["after"] = before, os.sched_getaffinity(0)
    t = threading.Thread(target=worker)
    t.start(); t.join()
    assert res["rc"] in (0, 1) and res["after"] <= res["before"] and len(res["after"]) > 0
    if res["rc"] == 0:
        assert res["after"] == res["before"] or len(res["after"]) < len(res["before"])


def test_read_pass_reports_a_scan_that_overflowed_its_record_regions():
    """ts_batch_read_pass on a scan whose record regions were too small judges nothing and raises the flag that
    ts_batch_read_pass_status reads; after ts_batch_sync (regrow + rescan) the same call gives the oracle's bits."""
    import torch
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.cli import user_input
    opts = H.parse_cli("--fastq-subset -l 42")
    rf = ta.ReadTelomereFilter(user_input(opts, device=0))
    L = K.lib()
    rng = np.random.default_rng(21)
    reads = [seqgen.chromosome(rng, int(rng.integers(3000, 20000)), n_its=1) for _ in range(300)]
    reads += [seqgen.repeat_array("TTAGGG", 2500).tobytes() for _ in range(40)]          # dense: 2 500 matches each
    n = len(reads)
    lens = (C.c_uint64 * n)(*[len(r) for r in reads])
    b = L.ts_batch_create(rf._ctx.ptr, lens, None, n, 1, 4096)                          # room for 4 096 records in all
    assert b
    info = K.BatchInfo()
    L.ts_batch_get_info(b, C.byref(info))
    dev = torch.device("cuda", 0)
    buf = torch.zeros(int(info.input_bytes), dtype=torch.uint8, device=dev)
    for i, r in enumerate(reads):
        off = int(L.ts_batch_segment_offset(b, i))
        buf[off:off + len(r)] = torch.frombuffer(bytearray(r), dtype=torch.uint8).to(dev)
    d_pass = torch.full((n + 16,), 7, dtype=torch.uint8, device=dev)
    sptr = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    flag = C.c_int(0)
    assert L.ts_batch_scan(b, C.c_void_p(buf.data_ptr()), sptr) == 0
    assert L.ts_batch_read_pass(b, C.c_void_p(d_pass.data_ptr()), sptr) == 0
    assert L.ts_batch_read_pass_status(b, C.byref(flag)) == 0 and flag.value == 1
    assert (d_pass[:n] == 7).all()                                                       # nothing was judged
    assert L.ts_batch_read_pass_status(b, C.byref(flag)) == 0 and flag.value == 0       # (read once)
    assert L.ts_batch_sync(b) == 0                                                       # regrows the regions, rescans
    assert L.ts_batch_read_pass(b, C.c_void_p(d_pass.data_ptr()), sptr) == 0
    assert L.ts_batch_read_pass_status(b, C.byref(flag)) == 0 and flag.value == 0
    assert [bool(x) for x in d_pass[:n].cpu().numpy()] == OracleReadFilter(opts).filter(reads)
    L.ts_batch_destroy(b)


@pytest.mark.parametrize("cli,gb", [(HEADLINE, 3.0), ("-c CCCTAAA -w 2000 -s 1000 -r -g -e -m -i", 1.0), ("-r -g -e -m -i", 1.0)])
def test_tiled_and_general_kernels_agree_at_full_size(cli, gb, monkeypatch):
    """Two independent device implementations of one parameter set at full size: the tiled kernel (closed form of the
    reference's carry loop, pair-table probes, a wave per tile of 23 windows) and the general kernels (the literal main /
    carry attribution of analyzeWindow, candidate lists, a workgroup per 4096 positions), forced onto the same set by
    TS_FORCE_GENERAL=1 — every window record (all fields, GC and entropy bit for bit), every terminal and interstitial
    block and the per-segment counts of bench.py's 3 Gb assembly must be equal.  Neither path shares a kernel with the
    other (kernels.hip + blockcall.hip's tiled record format against generic.hip + the general format)."""
    import torch
    import bench
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.cli import parse_cli, user_input
    L = K.lib()
    opts = parse_cli("x.fa " + cli)
    tel_t = ta.Teloscope(user_input(opts, device=0))
    assert tel_t.usesFastPath()
    monkeypatch.setenv("TS_FORCE_GENERAL", "1")
    tel_g = ta.Teloscope(user_input(opts, device=0))
    monkeypatch.delenv("TS_FORCE_GENERAL")
    assert not tel_g.usesFastPath()
    total = int(gb * 1e9)
    lens = bench.contig_lengths(total, 200, 42)
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
        segs[i].abs_pos = 1000 * i
    res = {}
    for name, tel in (("tiled", tel_t), ("general", tel_g)):
        o = (K.SegmentOut * n)()
        cnt = (K.SegmentCounts * n)()
        assert L.ts_scan_segments_blocks(tel._ctx.ptr, segs, n, o, cnt) == 0, tel._ctx.error()
        res[name] = (o, cnt)
    (ot, ct), (og, cg) = res["tiled"], res["general"]
    nwin = nblk = 0
    for i in range(n):
        assert (ct[i].n_windows, ct[i].n_matches, ct[i].n_canonical, ct[i].n_forward) == \
               (cg[i].n_windows, cg[i].n_matches, cg[i].n_canonical, cg[i].n_forward), "counts of segment %d" % i
        assert ot[i].n_windows == og[i].n_windows and ot[i].n_terminal_blocks == og[i].n_terminal_blocks and \
               ot[i].n_interstitial_blocks == og[i].n_interstitial_blocks, "sizes of segment %d" % i
        for field, k, size in (("windows", ot[i].n_windows, C.sizeof(K.Window)), ("terminal_blocks", ot[i].n_terminal_blocks, C.sizeof(K.Block)),
                               ("interstitial_blocks", ot[i].n_interstitial_blocks, C.sizeof(K.Block))):
            if k:
                a = C.string_at(C.cast(getattr(ot[i], field), C.c_void_p), k * size)
                b = C.string_at(C.cast(getattr(og[i], field), C.c_void_p), k * size)
                assert a == b, "%s of segment %d differ between the tiled and the general kernels" % (field, i)
        nwin += ot[i].n_windows
        nblk += ot[i].n_terminal_blocks + ot[i].n_interstitial_blocks
    assert nwin >= total // opts.step and nblk > 0
    L.ts_free_segments(ot, n)
    L.ts_free_segments(og, n)
    # ... and the five match vectors' source — every record, in the reference's push order — of the first forty contigs
    m = min(n, 40)
    mo = {}
    for name, tel in (("tiled", tel_t), ("general", tel_g)):
        o = (K.SegmentOut * m)()
        assert L.ts_scan_segments(tel._ctx.ptr, segs, m, o) == 0, tel._ctx.error()
        mo[name] = o
    nrec = 0
    for i in range(m):
        a, b = mo["tiled"][i], mo["general"][i]
        assert a.n_matches == b.n_matches, "match count of segment %d" % i
        if a.n_matches:
            ka = C.string_at(C.cast(a.matches, C.c_void_p), a.n_matches * C.sizeof(K.Match))
            kb = C.string_at(C.cast(b.matches, C.c_void_p), b.n_matches * C.sizeof(K.Match))
            assert ka == kb, "match records of segment %d differ between the tiled and the general kernels" % i
        nrec += a.n_matches
    assert nrec > 1000
    L.ts_free_segments(mo["tiled"], m)
    L.ts_free_segments(mo["general"], m)


@pytest.mark.parametrize("cli,gb", [("-p TTAGGG,TTAGG -w 1000 -s 500 -r -g -e -m -i", 1.0),
                                    ("-x 0 -p TTAG,TTAGG,TTAGGG,TTTAGGG,TTTTAGGG,TTAGGGTTA,TTAGGGTTAG,TTAGGGTTAGG,TTAGGGTTAGGG -w 1000 -s 500 -r -g -e -m -i", 0.5)])
def test_general_path_counts_equal_an_independent_count_at_scale(cli, gb):
    """The general kernels (list form: a mixed-length set; wide form: nine lengths) against a computation that shares nothing
    with them — torch: a rolling 2-bit code and one lookup table PER PATTERN LENGTH — on bench.py's synthetic assembly: per
    contig the number of matches, of canonical and of forward ones (for w > s every match that fits its segment is pushed
    exactly once), and the A / C / G / T totals summed over the windows that tile the contig."""
    import torch
    import bench
    import teloscope_amd as ta
    from teloscope_amd import _capi as K
    from teloscope_amd.cli import parse_cli, user_input
    L = K.lib()
    opts = parse_cli("x.fa " + cli)
    tel = ta.Teloscope(user_input(opts, device=0))
    assert not tel.usesFastPath()
    ui = tel.userInput
    total = int(gb * 1e9)
    lens = bench.contig_lengths(total, 200, 42)
    offs, off = [], 0
    for n_ in lens:
        offs.append(off)
        off += (n_ + 15) & ~15
    dev = torch.device("cuda", 0)
    buf = torch.zeros(off + 4096, dtype=torch.uint8, device=dev)
    bench.fill_synthetic(buf, offs, lens, 42, dev)
    host = buf.cpu().numpy()
    n = len(lens)
    segs = (K.SegmentIn * n)()
    for i in range(n):
        segs[i].seq = C.cast(C.c_void_p(host.ctypes.data + offs[i]), C.c_char_p)
        segs[i].len = lens[i]
    out = (K.SegmentOut * n)()
    cnt = (K.SegmentCounts * n)()
    assert L.ts_scan_segments_blocks(tel._ctx.ptr, segs, n, out, cnt) == 0, tel._ctx.error()
    # the independent side: a table per pattern length
    code_of = {"A": 0, "C": 1, "T": 2, "G": 3}
    by_len = {}
    for pat, fwd in ui.patternInfo:
        by_len.setdefault(len(pat), []).append((sum(code_of[ch] << (2 * i) for i, ch in enumerate(pat)), bool(fwd), pat in (ui.canonicalFwd, ui.canonicalRev)))
    tables = {}
    for k, pats in by_len.items():
        t = torch.zeros(3, 4 ** k, dtype=torch.bool)
        for x, fwd, can in pats:
            t[0, x], t[1, x], t[2, x] = True, fwd, can
        tables[k] = t.to(dev)
    lut = torch.full((256,), 4, dtype=torch.int64)
    for ch, c in code_of.items():
        lut[ord(ch)] = c
        lut[ord(ch.lower())] = c
    lut = lut.to(dev)
    step, window = ui.step, ui.windowSize
    assert window == 2 * step
    kmax = max(tables)
    for ci in range(n):
        nb = lens[ci]
        c = lut[buf[offs[ci]:offs[ci] + nb].long()]
        nuc = torch.bincount(c, minlength=5)[:4]
        tot = torch.zeros(3, dtype=torch.int64, device=dev)
        for k, t in tables.items():
            m = nb - k + 1
            if m <= 0:
                continue
            code = torch.zeros(m, dtype=torch.int64, device=dev)
            bad = torch.zeros(m, dtype=torch.bool, device=dev)
            for i in range(k):
                ci_ = c[i:i + m]
                code += (ci_ & 3) << (2 * i)
                bad |= ci_ == 4
            for f in range(3):
                tot[f] += (t[f][code] & ~bad).sum()
        nwin = -(-nb // step)
        assert (cnt[ci].n_windows, cnt[ci].n_matches, cnt[ci].n_forward, cnt[ci].n_canonical) == (nwin, int(tot[0]), int(tot[1]), int(tot[2])), \
            "contig %d: counts %s against the independent %s" % (ci, (cnt[ci].n_windows, cnt[ci].n_matches, cnt[ci].n_forward, cnt[ci].n_canonical), (nwin, tot.tolist()))
        w = np.ctypeslib.as_array(C.cast(out[ci].windows, C.POINTER(C.c_uint8)), shape=(nwin * C.sizeof(K.Window),)).view(
            np.dtype({"names": ["nuc"], "formats": [(np.uint32, 4)], "offsets": [K.Window.nucleotide_counts.offset], "itemsize": C.sizeof(K.Window)}))["nuc"]
        got = w[0::2].sum(axis=0, dtype=np.int64)                    # records are A C G T; codes A C T G
        assert got[[0, 1, 3, 2]].tolist() == nuc.tolist(), "contig %d: nucleotide totals" % ci
        del c
    assert kmax >= 6
    L.ts_free_segments(out, n)
