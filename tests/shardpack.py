"""TEST INFRASTRUCTURE: a shard's result message built from ORACLE results, byte for byte in the layout
ts_batch_pack_shard produces (teloscope_amd/csrc/ts_internal.h: TsShardHeader, TsShardSeg, TsDevBlock; sections and
capacities from ts_batch_shard_info).

Without a GPU the tile results cannot come from the HIP kernels; what the CPU tests exercise with these messages is
the host side of a sharded scan — the split of the plan, ts_shards_finalize (merge + the checks of what the shards
assume about each other) and the exchange over torch.distributed — and on the GPU box the same function is the
expected value of the kernels' messages.

A shard's view, restated from shard.cpp / blockcall.hip:
  * it owns tiles [own_begin, own_end) and sees [ext_begin, ext_end);
  * terminal blocks of a segment belong to the shard that owns that end of it (forward walk: the first tile; reverse
    walk: the last tile); an interstitial block to the shard that owns the tile it starts in;
  * per segment with an owned tile: the bounds it used for the interstitial search (its own walk's, or the widest
    possible where another shard walks that end), the match counts of its owned tiles, the counts over the tiles it saw.
"""
import ctypes as C

import numpy as np

from teloscope_amd import _capi as K

MAGIC, VERSION = 0x44485354, 1
SEG_HAS_START, SEG_HAS_END, SEG_FWD_WALKED, SEG_REV_WALKED, SEG_CONTEXT = 1, 2, 4, 8, 16

HEADER_DT = np.dtype([("magic", "<u4"), ("version", "<u4"), ("part", "<u4"), ("n_parts", "<u4"),
                      ("own_begin", "<u8"), ("own_end", "<u8"), ("ext_begin", "<u8"), ("ext_end", "<u8"),
                      ("seg_begin", "<u8"), ("n_segs", "<u4"), ("flags", "<u4"), ("n_visible", "<u8"),
                      ("n_blocks", "<u4"), ("visible_bytes", "<u4"), ("visible_capacity", "<u8"),
                      ("block_capacity", "<u4"), ("window_bytes", "<u4"), ("n_windows", "<u8"), ("msg_bytes", "<u8"),
                      ("reserved", "<u8", 2)])
SEG_DT = np.dtype([("fwd_boundary", "<u8"), ("rev_boundary", "<u8"), ("n_matches", "<u8"), ("n_canonical", "<u8"),
                   ("n_forward", "<u8"), ("seen_matches", "<u8"), ("seen_forward", "<u8"), ("flags", "<u4"), ("reserved", "<u4")])
DEVBLOCK_DT = np.dtype([("start", "<u8"), ("block_len", "<u4"), ("block_counts", "<u4"), ("forward_count", "<u4"),
                        ("reverse_count", "<u4"), ("canonical_count", "<u4"), ("non_canonical_count", "<u4"),
                        ("total_covered", "<u4"), ("fwd_covered", "<u4"), ("can_covered", "<u4"), ("has_valid_or", "u1"),
                        ("is_longest", "u1"), ("block_label", "S1"), ("reserved", "u1"), ("seg", "<u4"), ("kind", "<u4"),
                        ("seq", "<u4"), ("pad", "<u4")])
assert HEADER_DT.itemsize == 128 and SEG_DT.itemsize == 64 and DEVBLOCK_DT.itemsize == 64


def _align16(x):
    return (x + 15) & ~15


def sections(info, n_segs, n_own_tiles, n_windows):
    """Byte offsets of a message's sections (shard_layout, shard.cpp)."""
    o = 128
    off = {"segs": o}
    o = _align16(o + n_segs * 64)
    off["windows"] = o
    o = _align16(o + n_windows * int(info.window_bytes))
    off["tilevis"] = o
    o = _align16(o + n_own_tiles * 2)
    off["visible"] = o
    o = _align16(o + int(info.visible_capacity) * int(info.visible_bytes))
    off["blocks"] = o
    o = _align16(o + int(info.block_capacity) * 64)
    assert o == int(info.msg_bytes), (o, int(info.msg_bytes))
    return off


def pack_shard(plan, opts, part, oracle_results, scale=1):
    """The message shard `part` of `plan` packs, from oracle_results[i] = OracleBackend.scan_segment of segment i."""
    from teloscope_amd.distributed import shard_info
    info = shard_info(plan, part, scale)
    tiles = plan.tiles
    tips = plan.tips_only
    tl = opts.terminal_limit
    k = len(opts.canonical_fwd) if hasattr(opts, "canonical_fwd") else None
    nown = int(info.own_end - info.own_begin)
    nseg = int(info.seg_end - info.seg_begin) if nown else 0
    nwin = int(info.window_end - info.window_begin)
    off = sections(info, nseg, nown, nwin)
    msg = np.zeros(int(info.msg_bytes), dtype=np.uint8)
    abs_pos = plan.abs_pos if plan.abs_pos is not None else [0] * len(plan.seg_lens)
    seg_first = {}
    seg_ntiles = {}
    for t in range(plan.n_tiles):
        s = int(tiles["seg_index"][t])
        seg_first.setdefault(s, t)
        seg_ntiles[s] = seg_ntiles.get(s, 0) + 1

    def seg_matches(si):
        r = oracle_results[si]
        if tips:                                               # src/teloscope.cpp:566-570 fills fwd/rev only
            m = np.concatenate([r["fwd_matches"], r["rev_matches"]])
            return m[np.argsort(m["position"], kind="stable")]
        return r["all_matches"]

    def tile_span(t):
        return int(tiles["seg_offset"][t]), int(tiles["seg_offset"][t]) + int(tiles["owned_bases"][t])

    segs = np.zeros(nseg, dtype=SEG_DT)
    blocks = []
    visible = []
    tilevis = np.zeros(nown, dtype="<u2")
    flags_any = 0
    for s in range(nseg):
        si = int(info.seg_begin) + s
        n = plan.seg_lens[si]
        if si not in seg_first:                                # a segment without tiles: nothing to report
            segs[s]["flags"] = SEG_HAS_START | SEG_HAS_END
            segs[s]["rev_boundary"] = 0
            continue
        f, e = seg_first[si], seg_first[si] + seg_ntiles[si]
        t0, t1 = max(f, int(info.ext_begin)), min(e, int(info.ext_end))
        o0, o1 = max(f, int(info.own_begin)), min(e, int(info.own_end))
        has_start, has_end = int(info.own_begin) <= f < int(info.own_end), int(info.own_begin) < e <= int(info.own_end)
        m = seg_matches(si)
        rel = m["position"].astype(np.int64) - abs_pos[si]
        seen = (rel >= tile_span(t0)[0]) & (rel < tile_span(t1 - 1)[1]) if t1 > t0 else np.zeros(len(m), bool)
        own = (rel >= tile_span(o0)[0]) & (rel < tile_span(o1 - 1)[1]) if o1 > o0 else np.zeros(len(m), bool)
        seen_n, seen_f = int(seen.sum()), int((m["is_forward"][seen] != 0).sum())
        r = oracle_results[si]
        tb = r["terminal_blocks"]
        fwd_walked = has_start and seen_f >= 2
        rev_walked = has_end and seen_n - seen_f >= 2
        p_blocks = tb[tb["block_label"] == b"p"]
        q_blocks = tb[tb["block_label"] == b"q"]
        fb = (int(p_blocks["start"][-1]) - abs_pos[si] + int(p_blocks["block_len"][-1])) if (fwd_walked and len(p_blocks)) else 0
        rb = (int(q_blocks["start"].min()) - abs_pos[si]) if (rev_walked and len(q_blocks)) else n
        flags = (SEG_HAS_START if has_start else 0) | (SEG_HAS_END if has_end else 0) | \
            (SEG_FWD_WALKED if fwd_walked else 0) | (SEG_REV_WALKED if rev_walked else 0)
        segs[s] = (fb, rb if seen_n >= 2 else 0, int(own.sum()), int((m["is_canonical"][own] != 0).sum()),
                   int((m["is_forward"][own] != 0).sum()), seen_n, seen_f, flags, 0)
        # terminal blocks: the owner of that end of the segment, in push order per direction
        for kind, lab, mine in ((0, b"p", fwd_walked), (1, b"q", rev_walked)):
            if not mine:
                continue
            for seq, b in enumerate(tb[tb["block_label"] == lab]):
                blocks.append((b, si, kind, seq))
        # interstitial blocks: the owner of the tile they start in
        if not tips and o1 > o0:
            lo, hi = tile_span(o0)[0], tile_span(o1 - 1)[1]
            for b in r["interstitial_blocks"]:
                if lo <= int(b["start"]) - abs_pos[si] < hi:
                    blocks.append((b, si, 2, 0))
        # visible records of the owned tiles: canonical, or inside the terminal zone (src/teloscope.cpp:451-459, :486-496)
        if not tips:
            term_end = n - tl if n > tl else 0
            for t in range(o0, o1):
                a, z = tile_span(t)
                sel = (rel >= a) & (rel < z)
                vis = sel & ((m["is_canonical"] != 0) | (rel <= tl) | (rel >= term_end))
                rec = ((rel[vis] - a).astype(np.uint32) << 2) | (m["is_forward"][vis] != 0).astype(np.uint32) << 1 | (m["is_canonical"][vis] != 0).astype(np.uint32)
                tilevis[t - int(info.own_begin)] = len(rec)
                visible.append(rec)
    visible = np.concatenate(visible) if visible else np.zeros(0, dtype=np.uint32)
    # ---- sections
    msg[off["segs"]:off["segs"] + nseg * 64] = segs.view(np.uint8)
    if nwin:
        kk = len(opts.canonical_fwd)
        nuc = bool(opts.out_gc or opts.out_entropy)
        B = max(int(opts.window_size).bit_length(), 1)
        wb = int(info.window_bytes)
        packed = np.zeros((nwin, wb), dtype=np.uint8)
        wi = 0
        for si in range(len(plan.seg_lens)):
            w = oracle_results[si]["windows"]
            if tips or not len(w):
                continue
            seg_w0 = int(tiles["first_window"][seg_first[si]])
            for j in range(len(w)):
                g = seg_w0 + j
                if not (int(info.window_begin) <= g < int(info.window_end)):
                    continue
                fields = ([int(x) for x in w["nucleotide_counts"][j]] if nuc else []) + \
                    [int(w["canonical_covered"][j]) // kk, int(w["non_canonical_covered"][j]) // kk, int(w["fwd_covered"][j]) // kk]
                v = 0
                for q, f in enumerate(fields):
                    v |= f << (q * B)
                packed[g - int(info.window_begin)] = np.frombuffer(v.to_bytes(wb, "little"), dtype=np.uint8)
                wi += 1
        assert wi == nwin
        msg[off["windows"]:off["windows"] + nwin * wb] = packed.reshape(-1)
    msg[off["tilevis"]:off["tilevis"] + nown * 2] = tilevis.view(np.uint8)
    vb = int(info.visible_bytes)
    if len(visible) <= int(info.visible_capacity):
        vis_arr = visible.astype("<u2" if vb == 2 else "<u4")
        msg[off["visible"]:off["visible"] + len(visible) * vb] = vis_arr.view(np.uint8)
    else:
        flags_any |= K.SHARD_OVERFLOW_VISIBLE
    if len(blocks) <= int(info.block_capacity):
        arr = np.zeros(len(blocks), dtype=DEVBLOCK_DT)
        for i, (b, si, kind, seq) in enumerate(blocks):
            for fld in ("start", "block_len", "block_counts", "forward_count", "reverse_count", "canonical_count",
                        "non_canonical_count", "total_covered", "fwd_covered", "can_covered", "has_valid_or"):
                arr[i][fld] = b[fld]
            arr[i]["block_label"] = b["block_label"]
            arr[i]["seg"], arr[i]["kind"], arr[i]["seq"] = si, kind, seq
        msg[off["blocks"]:off["blocks"] + len(blocks) * 64] = arr.view(np.uint8)
    else:
        flags_any |= K.SHARD_OVERFLOW_BLOCKS
    hdr = np.zeros(1, dtype=HEADER_DT)
    hdr[0] = (MAGIC, VERSION, part, plan.world, int(info.own_begin), int(info.own_end), int(info.ext_begin), int(info.ext_end),
              int(info.seg_begin) if nown else 0, nseg, flags_any, len(visible), len(blocks), vb, int(info.visible_capacity),
              int(info.block_capacity), int(info.window_bytes), nwin, int(info.msg_bytes), (0, scale))
    msg[:128] = hdr.view(np.uint8)
    return msg


def read_header(msg):
    return np.frombuffer(np.ascontiguousarray(msg[:128]), dtype=HEADER_DT)[0]


def read_segs(msg):
    h = read_header(msg)
    return np.frombuffer(np.ascontiguousarray(msg[128:128 + int(h["n_segs"]) * 64]), dtype=SEG_DT)


def set_flags(msg, flags):
    out = msg.copy()
    h = np.frombuffer(out[:128], dtype=HEADER_DT).copy()
    h[0]["flags"] = flags
    out[:128] = h.view(np.uint8)
    return out
