"""Multi-GPU form of the scan path: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on ROCm; "gloo" for CPU tests and for rehearsing several ranks on one GPU).

The reference parallelises by path — one thread-pool job per path (src/input.cpp:719-724), results merged
in seqPos order (sortBySeqPos, include/teloscope.h:262-266).  Here the unit is a TILE of the batch's plan
(a run of consecutive windows of one segment + its w-s halo; windows are independent):

  * every rank builds the SAME plan from the segment lengths (host-only, deterministic) and takes the p-th of
    `world` CONSECUTIVE tile ranges of equal bases (ts_batch_partition): a 250 Mb chromosome spreads over
    ranks, and no data-path collective is needed to scan;
  * a rank's results are three arrays — window records (8 x u32 per window), tile directory entries
    ({matches, canonical, forward, 0} x u32 per tile) and its packed match records as ONE stream in tile order
    (ts_batch_export) — and because the ranges are consecutive, the whole batch's arrays are their
    concatenation in rank order;
  * ONE exchange per scan puts them together on the destination rank: an all-gather of the record counts
    (8 B per rank) and one grouped round of send/recv (a Gatherv: 7 senders -> rank 0 over 7 distinct xGMI
    links) straight into their places in the destination's arrays.  The destination then adopts the arrays
    (ts_batch_adopt: a prefix sum over the tile counts rebuilds the directory) and holds exactly what a
    single-GPU scan of the whole batch holds: ts_batch_download / _download_blocks / _segment_summary apply.

`gather_shards` is backend-agnostic and is what bench.py runs at N > 1; tests/test_distributed_gloo.py drives
the same function on two gloo ranks (no GPU: the tile results there are computed by the test itself).
"""
import ctypes as C
import os
from typing import List, Optional, Sequence

import numpy as np

from . import _capi as K


def lpt_partition(lengths: Sequence[int], world_size: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of whole segments to ranks (deterministic): per rank, the
    ascending list of segment indices.  The tile-range split of ShardPlan balances better (a segment may
    spread over ranks); this remains for callers that must keep segments whole (e.g. one FASTA record per
    host upload)."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    load = [0] * world_size
    shards: List[List[int]] = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda j: (load[j], j))
        shards[r].append(i)
        load[r] += int(lengths[i])
    return [sorted(s) for s in shards]


class ShardPlan:
    """The plan of one batch (all segments) and its split into `world` consecutive tile ranges.
    Host-only: `teloscope` may sit on a planning-only context (UserInputTeloscope.device = DEVICE_NONE)."""

    def __init__(self, teloscope, seg_lens: Sequence[int], abs_pos: Optional[Sequence[int]] = None,
                 tips_only: bool = False, world: int = 1, match_capacity: int = 0):
        self.L = K.lib()
        self.teloscope = teloscope
        self.seg_lens = [int(x) for x in seg_lens]
        self.abs_pos = [int(x) for x in abs_pos] if abs_pos is not None else None
        self.tips_only = bool(tips_only)
        self.world = int(world)
        self.match_capacity = int(match_capacity)
        self.batch = self._create()
        self.info = K.BatchInfo()
        self.L.ts_batch_get_info(self.batch, C.byref(self.info))
        self.n_tiles = int(self.info.n_tiles)
        self.n_windows = 0 if self.tips_only else int(self.info.n_windows)
        # every value of the three result arrays fits 16 bits: the exchange may halve its bytes (ts_batch_wire16_ok)
        self.wire16_ok = bool(self.L.ts_batch_wire16_ok(self.batch))
        self.ranges = []
        for p in range(self.world):
            lo, hi = C.c_uint64(), C.c_uint64()
            rc = self.L.ts_batch_partition(self.batch, self.world, p, C.byref(lo), C.byref(hi))
            if rc != K.TS_OK:
                raise K.TeloscanError(rc, "ts_batch_partition failed")
            self.ranges.append(self.range_info(lo.value, hi.value))
        self._tiles = None

    def _create(self):
        n = len(self.seg_lens)
        lens = (C.c_uint64 * max(1, n))(*self.seg_lens)
        ab = (C.c_uint64 * max(1, n))(*self.abs_pos) if self.abs_pos is not None else None
        b = self.L.ts_batch_create(self.teloscope._ctx.ptr, lens, ab, n, int(self.tips_only), self.match_capacity)
        if not b:
            raise K.TeloscanError(K.TS_ERR_UNSUPPORTED, self.teloscope._ctx.error())
        return b

    def new_batch(self):
        """Another batch object over the same plan (a rank's shard, the destination's assembly)."""
        return self._create()

    def range_info(self, lo, hi):
        r = K.RangeInfo()
        rc = self.L.ts_batch_range_info(self.batch, lo, hi, C.byref(r))
        if rc != K.TS_OK:
            raise K.TeloscanError(rc, "ts_batch_range_info failed")
        return r

    @property
    def tiles(self):
        """numpy structured array of ts_tile_info for every tile of the plan."""
        if self._tiles is None:
            t = np.zeros(self.n_tiles, dtype=K.TILE_DT)
            if self.n_tiles:
                rc = self.L.ts_batch_get_tiles(self.batch, 0, self.n_tiles, t.ctypes.data)
                if rc != K.TS_OK:
                    raise K.TeloscanError(rc, "ts_batch_get_tiles failed")
            self._tiles = t
        return self._tiles

    def segment_offsets(self):
        return [int(self.L.ts_batch_segment_offset(self.batch, i)) for i in range(len(self.seg_lens))]

    def close(self):
        if self.batch:
            self.L.ts_batch_destroy(self.batch)
            self.batch = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Assembled:
    """The whole batch's result arrays on the destination rank (torch tensors, int32 = the bits of u32)."""

    def __init__(self, windows, stats, dense, n_records, counts):
        self.windows, self.stats, self.dense = windows, stats, dense
        self.n_records = int(n_records)
        self.counts = [int(c) for c in counts]              # records per rank
        self.landing = {}                                   # (rank, array) -> int16 buffer the wire format lands in


class GatherHandle:
    def __init__(self, works, finish, keep=None):
        self._works, self._finish, self.result = works, finish, None
        self._keep = keep                                   # tensors that must outlive the transfers

    def wait(self):
        for w in self._works:
            w.wait()
        self._works = []
        self._keep = None
        if self._finish is not None:
            self.result = self._finish()
            self._finish = None
        return self.result


def _widen_u16(plan, src16, dst32):
    """dst32[i] = the u16 in src16[i] (int16 / int32 tensors carrying the bits of u16 / u32)."""
    if dst32.is_cuda:
        import torch
        rc = plan.L.ts_wire_widen_u16(plan.teloscope._ctx.ptr, C.c_void_p(src16.data_ptr()), C.c_void_p(dst32.data_ptr()),
                                      src16.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != K.TS_OK:
            raise K.TeloscanError(rc, plan.teloscope._ctx.error())
    else:
        import torch
        torch.bitwise_and(src16.to(torch.int32), 0xFFFF, out=dst32)


def gather_shards(plan: ShardPlan, rank: int, windows, stats, dense, n_records: int, dst: int = 0,
                  group=None, out: Optional[Assembled] = None, async_op: bool = False, directory_only: bool = False,
                  wire16: Optional[bool] = None):
    """The one exchange of a sharded scan (a Gatherv to `dst`).

    windows / stats / dense: this rank's result arrays as flat int32 torch tensors — windows of its range
    (8 per window), tile directory entries of its range (4 per tile), and its tile-ordered record stream
    (the first n_records elements are sent).  On `dst` they may be views into `out` already (its own results
    are then in place and not copied).  Returns (on dst) an Assembled whose arrays hold the whole batch, None
    elsewhere; with async_op a GatherHandle whose wait() returns that.

    wire16 (default: whenever the plan allows, ShardPlan.wire16_ok): the arrays travel as u16 — every value fits
    (packed records are a tile-relative position < 2^14 plus two flag bits, tile counts at most a tile's bases, window
    fields at most pattern length x window) — and are widened where they land: half the bytes over a gather that is
    bound by the per-link xGMI rate into the destination.

    directory_only: the summaries-only variant — only the tile directory entries travel (16 B per tile: what the
    path summary report needs, per-segment match / canonical / forward counts, is a sum over a segment's
    tiles); window and match records stay on the rank that produced them.

    Works on any backend: tensors on the device for "nccl" (RCCL, straight between HBMs over xGMI), staged
    through host memory for "gloo"."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    assert world == plan.world, "the plan was split for a different number of ranks"
    backend = dist.get_backend(group)
    on_device = backend == "nccl"
    dev = windows.device
    wire_dev = dev if on_device else torch.device("cpu")
    if wire16 is None:
        wire16 = plan.wire16_ok
    wire16 = bool(wire16) and plan.wire16_ok
    r = plan.ranges[rank]
    assert windows.numel() == 8 * (0 if plan.tips_only else (r.window_end - r.window_begin))
    assert stats.numel() == 4 * (r.tile_end - r.tile_begin)

    # 1. record counts of all ranks (8 B each)
    if directory_only:
        counts = [0] * world
        n_records = 0
    else:
        mine = torch.tensor([int(n_records)], dtype=torch.int64, device=wire_dev)
        allc = torch.zeros(world, dtype=torch.int64, device=wire_dev)
        dist.all_gather(list(allc.split(1)), mine, group=group)   # views of one buffer: read back with ONE copy, not one per rank
        counts = [int(c) for c in allc.tolist()]

    def wire(t):
        if wire16:
            t = t.to(torch.int16)                            # keeps the low 16 bits: the whole value
        return t if t.device == wire_dev else t.to(wire_dev)

    def on_wire(t):
        # what is handed to send / recv: 16-bit integers are not a type the NCCL / RCCL process group transports
        # (ProcessGroupNCCL maps no int16), so they travel as their bytes
        return t.view(torch.uint8) if t.dtype == torch.int16 else t

    # 2. grouped send/recv: every rank's three arrays go to their places on dst
    if rank != dst:
        ops, keep = [], []
        if windows.numel() and not directory_only:
            keep.append(wire(windows))
            ops.append(dist.P2POp(dist.isend, on_wire(keep[-1]), dst, group=group, tag=1))
        if stats.numel():
            keep.append(wire(stats))
            ops.append(dist.P2POp(dist.isend, on_wire(keep[-1]), dst, group=group, tag=2))
        if n_records:
            keep.append(wire(dense[:n_records]))
            ops.append(dist.P2POp(dist.isend, on_wire(keep[-1]), dst, group=group, tag=3))
        works = dist.batch_isend_irecv(ops) if ops else []
        h = GatherHandle(works, None, keep)
        return h if async_op else h.wait()

    total = sum(counts)
    if out is None:
        out = Assembled(torch.empty(8 * plan.n_windows, dtype=torch.int32, device=dev),
                        torch.empty(4 * plan.n_tiles, dtype=torch.int32, device=dev),
                        torch.empty(max(total, 1), dtype=torch.int32, device=dev), 0, counts)
    if out.dense.numel() < total:
        grown = torch.empty(total + total // 8 + 1024, dtype=torch.int32, device=dev)
        out.dense = grown
    out.n_records, out.counts = total, counts
    rec_off = [0]
    for c in counts:
        rec_off.append(rec_off[-1] + c)

    def place(dst_t, src_t):                                   # dst's own arrays, unless they are already views of `out`
        if src_t.numel() and dst_t.data_ptr() != src_t.data_ptr():
            dst_t.copy_(src_t)

    ops, landing = [], []
    for p in range(world):
        rp = plan.ranges[p]
        w = out.windows[8 * rp.window_begin:8 * rp.window_end] if not plan.tips_only else out.windows[:0]
        s = out.stats[4 * rp.tile_begin:4 * rp.tile_end]
        d = out.dense[rec_off[p]:rec_off[p + 1]]
        if p == dst:
            if not directory_only:
                place(w, windows)
                place(d, dense[:counts[p]])
            place(s, stats)
            continue
        for t, tag in ((w, 1), (s, 2), (d, 3)):
            if not t.numel() or (directory_only and tag != 2):
                continue
            if on_device and not wire16:
                ops.append(dist.P2POp(dist.irecv, t, p, group=group, tag=tag))
                continue
            # a landing buffer (kept between exchanges): u16 on the wire, and / or host memory for gloo
            key = (p, tag, wire16)
            tmp = out.landing.get(key)
            if tmp is None or tmp.numel() < t.numel() or tmp.device != wire_dev:
                tmp = torch.empty(t.numel() + t.numel() // 8 + 16, dtype=torch.int16 if wire16 else torch.int32, device=wire_dev)
                out.landing[key] = tmp
            tmp = tmp[:t.numel()]
            ops.append(dist.P2POp(dist.irecv, on_wire(tmp), p, group=group, tag=tag))
            landing.append((t, tmp))
    works = dist.batch_isend_irecv(ops) if ops else []

    def finish():
        for t, tmp in landing:
            if tmp.device != t.device:
                tmp = tmp.to(t.device)
            if wire16:
                _widen_u16(plan, tmp, t)
            else:
                t.copy_(tmp)
        return out

    h = GatherHandle(works, finish)
    return h if async_op else h.wait()


def decode_segments(plan: ShardPlan, windows: np.ndarray, stats: np.ndarray, dense: np.ndarray):
    """Host view of a whole batch's raw arrays: per segment a dict with `windows` ([n, 8] u32: A, C, G, T,
    canonical/non-canonical/forward/reverse covered bases) and `matches` (structured: absolute position,
    forward, canonical) in position order."""
    tiles = plan.tiles
    stats = np.asarray(stats, dtype=np.uint32).reshape(-1, 4)
    dense = np.asarray(dense, dtype=np.uint32)
    wins = np.asarray(windows, dtype=np.uint32).reshape(-1, 8)
    counts = stats[:, 0].astype(np.int64)
    offs = np.concatenate(([0], np.cumsum(counts)))
    # absolute position of every record: segment abs_pos + tile offset + tile-relative position
    abs_pos = np.asarray(plan.abs_pos if plan.abs_pos is not None else [0] * len(plan.seg_lens), dtype=np.uint64)
    tile_base = abs_pos[tiles["seg_index"].astype(np.int64)] + tiles["seg_offset"]
    rec_tile = np.repeat(np.arange(len(tiles)), counts)
    n = int(offs[-1])
    pos = tile_base[rec_tile] + (dense[:n] >> 2).astype(np.uint64)
    fwd = (dense[:n] & 2) != 0
    can = (dense[:n] & 1) != 0
    seg_of_tile = tiles["seg_index"].astype(np.int64)
    out = []
    win_at = 0
    step = plan.teloscope.userInput.step
    for si, ln in enumerate(plan.seg_lens):
        tsel = np.flatnonzero(seg_of_tile == si)
        if len(tsel):
            a, b = int(offs[tsel[0]]), int(offs[tsel[-1] + 1])
        else:
            a = b = 0
        m = np.zeros(b - a, dtype=[("position", np.uint64), ("is_forward", np.bool_), ("is_canonical", np.bool_)])
        m["position"], m["is_forward"], m["is_canonical"] = pos[a:b], fwd[a:b], can[a:b]
        nwin = 0 if plan.tips_only else -(-ln // step)
        out.append(dict(windows=wins[win_at:win_at + nwin], matches=m))
        win_at += nwin
    return out


class HipShard:
    """One rank's shard on its GPU: a batch restricted to the rank's tile range, bound to torch-owned result
    buffers, with the tile-ordered export that feeds gather_shards.  `slots` independent buffer sets (each its
    own batch object) let the exchange of one scan overlap the next scan."""

    def __init__(self, plan: ShardPlan, rank: int, device, slots: int = 1, assembled: Optional[List[Assembled]] = None):
        import torch
        self.plan, self.rank, self.device = plan, rank, device
        self.L = plan.L
        self.r = plan.ranges[rank]
        r = self.r
        self.n_tiles = int(r.tile_end - r.tile_begin)
        self.n_windows = 0 if plan.tips_only else int(r.window_end - r.window_begin)
        self.input_bytes = int(r.input_end - r.input_begin)
        self.batches, self.windows, self.stats, self.dense, self.total = [], [], [], [], []
        cap = max(int(r.bases) // 4 + 4096, 4096)
        for j in range(slots):
            b = plan.new_batch()
            rc = self.L.ts_batch_restrict(b, r.tile_begin, r.tile_end)
            if rc != K.TS_OK:
                raise K.TeloscanError(rc, plan.teloscope._ctx.error())
            if assembled is not None:                          # the destination scans straight into the assembly
                a = assembled[j]
                w = a.windows[8 * r.window_begin:8 * r.window_end] if not plan.tips_only else a.windows[:0]
                s = a.stats[4 * r.tile_begin:4 * r.tile_end]
            else:
                w = torch.empty(8 * self.n_windows, dtype=torch.int32, device=device)
                s = torch.empty(4 * self.n_tiles, dtype=torch.int32, device=device)
            rc = self.L.ts_batch_bind_results(b, C.c_void_p(w.data_ptr()) if w.numel() else None,
                                              C.c_void_p(s.data_ptr()) if s.numel() else None)
            if rc != K.TS_OK:
                raise K.TeloscanError(rc, plan.teloscope._ctx.error())
            self.batches.append(b)
            self.windows.append(w)
            self.stats.append(s)
            self.dense.append(torch.empty(cap, dtype=torch.int32, device=device))
            self.total.append(torch.zeros(2, dtype=torch.int64, device=device))

    def scan(self, d_input, stream_ptr, slot=0):
        """Enqueue scan + tile-ordered export of this rank's range (asynchronous on the stream).
        d_input: device address of byte `input_begin` of the input layout."""
        b = self.batches[slot]
        rc = self.L.ts_batch_scan(b, C.c_void_p(d_input), stream_ptr)
        if rc == K.TS_OK:
            rc = self.L.ts_batch_export(b, C.c_void_p(self.dense[slot].data_ptr()), self.dense[slot].numel(),
                                        C.c_void_p(self.total[slot].data_ptr()), stream_ptr)
        if rc != K.TS_OK:
            raise K.TeloscanError(rc, self.plan.teloscope._ctx.error())

    def finish(self, d_input, stream_ptr, slot=0):
        """Wait for the slot's scan + export; returns its record count.  If the stream came out incomplete
        (a wave's region or the export buffer was too small) the batch is synced — which grows the regions and
        rescans — the export buffer grown, and the export repeated."""
        import torch
        # the read-back below is ordered on torch's CURRENT stream: it sees the export only if that is the stream the scan
        # and the export were enqueued on
        if int(getattr(stream_ptr, "value", stream_ptr) or 0) != int(torch.cuda.current_stream().cuda_stream):
            raise ValueError("HipShard.finish: stream_ptr must be torch's current stream (enter `with torch.cuda.stream(s)` first)")
        for _ in range(4):
            n, bad = (int(x) for x in self.total[slot].tolist())         # D2H of 16 bytes: waits for the stream
            if not bad:
                return n
            b = self.batches[slot]
            rc = self.L.ts_batch_sync(b)
            if rc != K.TS_OK:
                raise K.TeloscanError(rc, self.plan.teloscope._ctx.error())
            info = K.BatchInfo()
            self.L.ts_batch_get_info(b, C.byref(info))
            need = int(info.n_matches)
            if self.dense[slot].numel() < need:
                self.dense[slot] = torch.empty(need + need // 8 + 1024, dtype=torch.int32, device=self.device)
            rc = self.L.ts_batch_export(b, C.c_void_p(self.dense[slot].data_ptr()), self.dense[slot].numel(),
                                        C.c_void_p(self.total[slot].data_ptr()), stream_ptr)
            if rc != K.TS_OK:
                raise K.TeloscanError(rc, self.plan.teloscope._ctx.error())
        raise K.TeloscanError(K.TS_ERR_STATE, "tile-ordered export kept coming out incomplete")

    def kernel_ms(self, slot=0):
        """(avg HIP-event ms of the slot's scans since its last sync, launches averaged)."""
        b = self.batches[slot]
        rc = self.L.ts_batch_sync(b)
        if rc != K.TS_OK:
            raise K.TeloscanError(rc, self.plan.teloscope._ctx.error())
        info = K.BatchInfo()
        self.L.ts_batch_get_info(b, C.byref(info))
        return float(info.avg_kernel_ms), int(info.kernel_launches), info

    def close(self):
        for b in self.batches:
            self.L.ts_batch_destroy(b)
        self.batches = []


# ------------------------------------------------------------------------------------------- shard results
# The exchange above assembles every record on one rank (0.56 GB per 3 Gb scan): bound by the xGMI links into that
# rank.  What follows ships what the reference's writers read instead (include/teloscan.h, "shard results"): every
# rank calls its blocks on its own device and packs ONE message of a size both sides know from the plan —
# bit-packed window records, the visible match records, its blocks — so that a step needs no count exchange and no
# host synchronisation, and ~60 MB cross the links at 8 ranks instead of 245.

def shard_info(plan: ShardPlan, part: int, scale: int = 1, world: Optional[int] = None):
    info = K.ShardInfo()
    rc = plan.L.ts_batch_shard_info(plan.batch, world or plan.world, part, scale, C.byref(info))
    if rc != K.TS_OK:
        raise K.TeloscanError(rc, "ts_batch_shard_info failed")
    return info


def concurrent_streams(teloscope, device, n, first=None, tries=24):
    """`n` torch streams whose kernels run beside each other: HIP puts a process's streams on a few hardware queues (four by
    default) without saying which, and two streams on one queue run their kernels one after the other — a pack stream on the scan
    stream's queue turns "the pack runs beside the next scan" into "the next scan waits for the pack" (+0.1 ms per 3 Gb step).
    Streams are made until `n` are found that ts_streams_concurrent says do not share a queue (`first`, if given, is the first of
    them); when the queues run out before that, the rest are plain new streams.  ~1 ms per pair tried."""
    import torch
    L, ctx = K.lib(), teloscope._ctx.ptr
    chosen = [first if first is not None else torch.cuda.Stream(device=device)]
    seen = {chosen[0].cuda_stream}
    for _ in range(tries):
        if len(chosen) >= n:
            break
        s = torch.cuda.Stream(device=device)
        if s.cuda_stream in seen:
            continue
        seen.add(s.cuda_stream)
        ok = True
        for c in chosen:
            rc = L.ts_streams_concurrent(ctx, C.c_void_p(c.cuda_stream), C.c_void_p(s.cuda_stream))
            if rc < 0:
                raise K.TeloscanError(rc, teloscope._ctx.error())
            if rc == 0:
                ok = False
                break
        if ok:
            chosen.append(s)
    while len(chosen) < n:
        chosen.append(torch.cuda.Stream(device=device))
    return chosen


class PackedShard:
    """One rank's shard on its GPU (ts_batch_restrict_shard): scan + device block calling + packed message, all
    asynchronous on the caller's stream.  `slots` message buffers (each with its own batch object) let the
    transfer of one scan's message overlap the next scan."""

    def __init__(self, plan: ShardPlan, part: int, device, slots: int = 1, scale: int = 1):
        import torch
        self.plan, self.part, self.device, self.scale = plan, part, device, scale
        self.L = plan.L
        self.info = shard_info(plan, part, scale)
        self.batches, self.msgs = [], []
        for _ in range(slots):
            b = plan.new_batch()
            rc = self.L.ts_batch_restrict_shard(b, plan.world, part, scale)
            if rc != K.TS_OK:
                raise K.TeloscanError(rc, plan.teloscope._ctx.error())
            self.batches.append(b)
            self.msgs.append(torch.zeros(int(self.info.msg_bytes), dtype=torch.uint8, device=device))
            self._bind(len(self.batches) - 1)

    def _bind(self, slot):
        """The slot's scans pack their window records into the slot's message themselves (ts_batch_bind_shard_message;
        TS_SHARD_BIND=0 leaves that to the pack's own kernel: A/B measurements, and the tests compare the two)."""
        if os.environ.get("TS_SHARD_BIND", "1") == "0":
            return
        rc = self.L.ts_batch_bind_shard_message(self.batches[slot], C.c_void_p(self.msgs[slot].data_ptr()), self.msgs[slot].numel())
        if rc != K.TS_OK:
            raise K.TeloscanError(rc, self.plan.teloscope._ctx.error())

    def scan(self, d_input, stream_ptr, slot=0):
        """Enqueue the scan of the shard's tiles.  d_input: device address of byte `info.input_begin` of the input layout."""
        rc = self.L.ts_batch_scan(self.batches[slot], C.c_void_p(d_input), stream_ptr)
        if rc != K.TS_OK:
            raise K.TeloscanError(rc, self.plan.teloscope._ctx.error())

    def set_timing(self, every):
        """ts_batch_set_timing on every slot: every n-th scan of a slot is timed (0: none).  A rank's steps time one scan in a few —
        the start event of a timed scan is a packet on the scan's queue."""
        for b in self.batches:
            self.L.ts_batch_set_timing(b, every)

    def wait_scan(self, stream_ptr, slot=0):
        """`stream_ptr` waits for the slot's last scan (ts_batch_wait_scan: the library's own event behind the scan — an event of the
        caller's would be a second packet on the scan's queue)."""
        rc = self.L.ts_batch_wait_scan(self.batches[slot], stream_ptr)
        if rc != K.TS_OK:
            raise K.TeloscanError(rc, self.plan.teloscope._ctx.error())

    def pack(self, stream_ptr, slot=0):
        """Enqueue block calling + the packed message of the slot's last scan (the stream must be ordered behind that
        scan: the same stream, or one that waits for an event recorded after it — the kernels use no LDS and few
        registers, so on a stream of their own they run beside the next scan)."""
        rc = self.L.ts_batch_pack_shard(self.batches[slot], C.c_void_p(self.msgs[slot].data_ptr()), self.msgs[slot].numel(), stream_ptr)
        if rc != K.TS_OK:
            raise K.TeloscanError(rc, self.plan.teloscope._ctx.error())

    def scan_pack(self, d_input, stream_ptr, slot=0):
        self.scan(d_input, stream_ptr, slot)
        self.pack(stream_ptr, slot)

    def status(self, slot=0):
        """The message's header, read back (synchronises with the device)."""
        head = self.msgs[slot][:128].cpu().numpy()
        st = K.ShardStatus()
        rc = self.L.ts_shard_peek(head.ctypes.data, head.nbytes, C.byref(st))
        if rc != K.TS_OK:
            raise K.TeloscanError(rc, "ts_shard_peek failed")
        return st

    def sync(self, slot=0):
        """ts_batch_sync of the slot's batch: grows the record regions and rescans if the scan overflowed."""
        rc = self.L.ts_batch_sync(self.batches[slot])
        if rc != K.TS_OK:
            raise K.TeloscanError(rc, self.plan.teloscope._ctx.error())

    def set_scale(self, scale):
        """Larger variable sections (after an overflow); every rank and the receiver must use the same scale."""
        import torch
        self.scale = scale
        self.info = shard_info(self.plan, self.part, scale)
        for j, b in enumerate(self.batches):
            rc = self.L.ts_batch_set_shard_scale(b, scale)
            if rc != K.TS_OK:                                   # a silent failure would leave layout and buffer size disagreeing between ranks
                raise K.TeloscanError(rc, "ts_batch_set_shard_scale(%d) failed" % scale)
            self.msgs[j] = torch.zeros(int(self.info.msg_bytes), dtype=torch.uint8, device=self.device)
            self._bind(j)

    def kernel_ms(self, slot=0):
        b = self.batches[slot]
        self.sync(slot)
        info = K.BatchInfo()
        self.L.ts_batch_get_info(b, C.byref(info))
        return float(info.avg_kernel_ms), int(info.kernel_launches), info

    def close(self):
        for b in self.batches:
            self.L.ts_batch_destroy(b)
        self.batches = []


class ShardExchange:
    """The ONE exchange of a sharded scan in its shard-result form: every rank's message to `dst`.

    Both sides know every message's size from the plan (ts_batch_shard_info), so a step is one grouped round of
    send / recv — no count exchange, nothing read back to the host — and `check` (outside the timed region: once
    after the warm-up, once at the end) is where rank `dst` reads the headers, merges the messages
    (ts_shards_finalize) and tells everybody whether something has to change: a scan overflowed its record regions
    (sync + rescan), a message overflowed (a larger capacity scale on every rank), or the input is one the shards'
    assumptions do not hold for (the full exchange, gather_shards, is then the path).

    Works on any backend: device tensors for "nccl" (RCCL over xGMI), staged through host memory for "gloo"."""
    OK, SYNC, GROW, FULL = 0, 1, 2, 3

    def __init__(self, plan: ShardPlan, rank: int, device, dst: int = 0, group=None, slots: int = 2, scale: int = 1):
        import torch.distributed as dist
        self.plan, self.rank, self.device, self.dst, self.group, self.slots = plan, rank, device, dst, group, slots
        self.world = plan.world
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self._keep = {}
        self.set_scale(scale)

    def set_scale(self, scale):
        import torch
        self.scale = scale
        self.infos = [shard_info(self.plan, p, scale) for p in range(self.world)]
        wire_dev = self.device if self.backend == "nccl" else torch.device("cpu")
        self.recv = None
        if self.rank == self.dst:
            self.recv = [[torch.zeros(int(self.infos[p].msg_bytes), dtype=torch.uint8, device=wire_dev) if p != self.dst else None
                          for p in range(self.world)] for _ in range(self.slots)]

    @property
    def bytes_over_links(self):
        return sum(int(self.infos[p].msg_bytes) for p in range(self.world) if p != self.dst)

    def post(self, msg, slot=0):
        """Enqueue the transfer of this rank's message (`msg`: its device tensor, packed on the current stream);
        returns the works to wait for before the slot's buffers are reused."""
        import torch.distributed as dist
        if self.world == 1:
            return []
        ops = []
        if self.rank != self.dst:
            t = msg if self.backend == "nccl" else msg.cpu()
            self._keep[slot] = t                                # a staged host copy lives until its slot is posted again (several slots are in flight)
            ops.append(dist.P2POp(dist.isend, t, self.dst, group=self.group))
        else:
            for p in range(self.world):
                if p != self.dst:
                    ops.append(dist.P2POp(dist.irecv, self.recv[slot][p], p, group=self.group))
        return dist.batch_isend_irecv(ops) if ops else []

    def messages(self, own_msg, slot=0):
        """On dst: the messages of all parts of the slot as host numpy arrays (synchronises)."""
        assert self.rank == self.dst
        return [(own_msg if p == self.dst else self.recv[slot][p]).cpu().numpy() for p in range(self.world)]

    def check(self, own_msg, slot=0, finalize=True):
        """Collective.  Returns (action, factor, merged) — merged = (rc, out, counts) of ts_shards_finalize on dst when
        the messages were merged (the caller frees out with free_segments), else None."""
        import torch
        import torch.distributed as dist
        action, factor, merged = self.OK, 1, None
        if self.rank == self.dst:
            msgs = self.messages(own_msg, slot)
            flags = 0
            for m in msgs:
                st = K.ShardStatus()
                if self.plan.L.ts_shard_peek(m.ctypes.data, m.nbytes, C.byref(st)) != K.TS_OK:
                    raise K.TeloscanError(K.TS_ERR_STATE, "a received shard message has no valid header")
                flags |= st.flags
                factor = max(factor, int(st.scale_factor_needed))
            if flags & K.SHARD_OVERFLOW_SCAN:
                action = self.SYNC
            elif flags & (K.SHARD_OVERFLOW_VISIBLE | K.SHARD_OVERFLOW_BLOCKS):
                action, factor = self.GROW, max(2, factor)
            elif flags & K.SHARD_OUT_OF_CONTEXT:
                action = self.FULL
            elif finalize:
                merged = finalize_shards(self.plan, msgs)
                if merged[0] != 0:
                    action = {K.SHARD_RETRY_SYNC: self.SYNC, K.SHARD_RETRY_GROW: self.GROW, K.SHARD_NEED_FULL: self.FULL}[merged[0]]
                    merged = None
        if self.world > 1:
            wire_dev = self.device if self.backend == "nccl" else torch.device("cpu")
            t = torch.tensor([action, factor], dtype=torch.int64, device=wire_dev)
            dist.broadcast(t, src=self.dst, group=self.group)
            action, factor = (int(x) for x in t.tolist())
        return action, factor, merged


def finalize_shards(plan: ShardPlan, msgs, with_counts=True):
    """ts_shards_finalize: the messages of all parts (host memory: numpy uint8 arrays or bytes) -> (rc, out, counts);
    rc 0 = out[i] / counts[i] hold segment i (free out with free_segments), else K.SHARD_RETRY_SYNC / _GROW / NEED_FULL."""
    n = len(msgs)
    assert n == plan.world
    arrs = [np.frombuffer(m, dtype=np.uint8) if not isinstance(m, np.ndarray) else np.ascontiguousarray(m) for m in msgs]
    ptrs = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
    sizes = (C.c_uint64 * n)(*[a.nbytes for a in arrs])
    ns = len(plan.seg_lens)
    out = (K.SegmentOut * max(1, ns))()
    cnt = (K.SegmentCounts * max(1, ns))()
    rc = plan.L.ts_shards_finalize(plan.batch, ptrs, sizes, n, out, cnt if with_counts else None)
    if rc < 0:
        raise K.TeloscanError(rc, plan.teloscope._ctx.error())
    return rc, out, cnt


def free_segments(plan: ShardPlan, out):
    plan.L.ts_free_segments(out, len(plan.seg_lens))


def adopt(plan: ShardPlan, a: Assembled, stream_ptr=None):
    """A whole-plan batch holding the assembled results (ts_batch_adopt): block calling, downloads and the
    segment summary then work as after a single-GPU scan.  The caller destroys it (ts_batch_destroy) and
    keeps `a` alive meanwhile."""
    L = plan.L
    b = plan.new_batch()
    rc = L.ts_batch_adopt(b, C.c_void_p(a.windows.data_ptr()) if a.windows.numel() else None,
                          C.c_void_p(a.stats.data_ptr()), C.c_void_p(a.dense.data_ptr()), a.n_records, stream_ptr)
    if rc != K.TS_OK:
        L.ts_batch_destroy(b)
        raise K.TeloscanError(rc, plan.teloscope._ctx.error())
    return b


def gather_segment_summaries(local_summary, local_indices, n_total, dst=0, group=None):
    """The summaries-only variant: one fixed-size gather of per-segment {windows, matches, canonical, forward}
    (what the path summary report prints) to `dst`; window and match records stay where they were produced.

    local_summary: tensor [n_local, 4] int64 on the rank's device (cuda for nccl, cpu for gloo),
    rows in the order of local_indices.  Returns on dst a numpy array [n_total, 4] in global
    segment order, None elsewhere."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = local_summary.device
    n_local = torch.tensor([local_summary.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)              # 8 bytes per rank
    max_n = max(int(c.item()) for c in counts)
    # rows: [global index, windows, matches, canonical, forward]; padded to the largest shard
    msg = torch.full((max_n, 5), -1, dtype=torch.int64, device=dev)
    if local_summary.shape[0]:
        msg[:local_summary.shape[0], 0] = torch.as_tensor(list(local_indices), dtype=torch.int64, device=dev)
        msg[:local_summary.shape[0], 1:] = local_summary
    bufs = [torch.empty_like(msg) for _ in range(world)] if rank == dst else None
    dist.gather(msg, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    out = np.zeros((n_total, 4), dtype=np.int64)
    seen = np.zeros(n_total, dtype=bool)
    for b in bufs:
        a = b.cpu().numpy()
        a = a[a[:, 0] >= 0]
        out[a[:, 0]] = a[:, 1:]
        seen[a[:, 0]] = True
    if not seen.all():
        raise RuntimeError("segment summaries missing after gather: %s" % np.flatnonzero(~seen)[:8])
    return out
