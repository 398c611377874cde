// shard.cpp — shard results: how several devices share one scan without shipping the match stream.
//
// The reference fans paths out to thread-pool workers and merges their PathData in seqPos order
// (src/input.cpp:719-733, include/teloscope.h:262-266); what its writers read of a path is its windows, its blocks,
// canonicalMatches and the terminal nonCanonicalMatches (src/teloscope.cpp:486-496, :700-868) — ~3 % of the match
// records.  Here the unit of work is a tile range of ONE plan (a 250 Mb contig spreads over devices), and a device
// hands over exactly that view:
//
//   ts_batch_partition        consecutive tile ranges of equal bases whose boundaries keep clear of the terminal zones
//                             of the segments (a boundary inside a segment lies at least zone + context tiles from both
//                             its ends), so that the device owning a segment's end can walk its terminal blocks alone
//   ts_batch_restrict_shard   the batch executes its owned range plus CONTEXT tiles either side inside a segment that
//                             continues on a neighbour: a chain of matches that starts in an owned tile is followed there
//   ts_batch_pack_shard       block calling on the device (blockcall.hip) + the packed message (shard.hip), all
//                             asynchronous on the caller's stream: no host synchronisation per scan
//   ts_shards_finalize        host: the messages of all parts -> SegmentData per segment, in input order; checks the
//                             assumptions the parts made about each other (the bounds of the interstitial search) and
//                             says TS_SHARD_NEED_FULL when one does not hold (a telomere longer than the context: the
//                             caller then takes the full path — every record to one device — for that batch)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "capi_internal.hpp"

// Measurement only (-DTS_PACK_ABL=bits: results are wrong): 1 no terminal walks, 2 no visible-record copy, 8 no interstitial search,
// 32 no header kernel — which of the pack's kernels costs the scan beside it what (profiles/r05_pack_abl.sh).
#ifndef TS_PACK_ABL
#define TS_PACK_ABL 0
#endif

namespace {

uint32_t bit_width_u32(uint32_t v) { uint32_t b = 0; while (v) { ++b; v >>= 1; } return b ? b : 1u; }

struct ShardGeom { uint64_t tile_bases; uint32_t ctx, zone, margin; };

// context tiles: enough that a record beyond them cannot chain (gap <= -k) to one in an owned tile across an empty
// context, and at least two; zone tiles: those holding positions [0, terminal_limit]
ShardGeom shard_geom(const ts_batch *b) {
    ShardGeom g{};
    const ts_params &P = b->ctx->params;
    g.tile_bases = std::max<uint64_t>((uint64_t)b->wpt * b->kp.s, 1);
    if (b->tips) { g.ctx = 0; g.zone = 0; g.margin = 0; return g; }
    g.ctx = (uint32_t)std::max<uint64_t>(2, ceil_div((uint64_t)P.max_match_dist + b->ctx->k + 1, g.tile_bases) + 1);
    g.zone = (uint32_t)std::min<uint64_t>(ceil_div((uint64_t)P.terminal_limit + 1, g.tile_bases), 0x3FFFFFFFull);
    g.margin = g.zone + g.ctx;
    return g;
}

// Boundaries of the split into n_parts: part p owns tiles [out[p], out[p + 1]).
void shard_boundaries(const ts_batch *b, uint32_t n_parts, std::vector<uint64_t> &out) {
    const ShardGeom g = shard_geom(b);
    const uint64_t nt = b->tiles.size();
    out.assign((size_t)n_parts + 1, nt);
    out[0] = 0;
    uint64_t total = 0;
    for (const TsTile &T : b->tiles) total += T.own_len;
    // where equal bases would cut: the first tile whose preceding owned bases reach q / n_parts of the total
    uint64_t acc = 0, t = 0;
    for (uint32_t q = 1; q < n_parts; ++q) {
        const unsigned __int128 target = (unsigned __int128)total * q;
        for (; t < nt; ++t) {
            if ((unsigned __int128)acc * n_parts >= target) break;
            acc += b->tiles[t].own_len;
        }
        uint64_t cut = t;
        if (cut > 0 && cut < nt) {
            const SegPlan &sp = b->segs[b->tiles[cut].seg];
            const uint64_t f = sp.first_tile, e = f + sp.n_tiles;
            if (cut != f) {
                uint64_t lo, hi;                                     // the allowed positions either side of the cut
                if (b->tips || e - f < 2ull * g.margin) { lo = f; hi = e; }
                else if (cut - f < g.margin) { lo = f; hi = f + g.margin; }
                else if (e - cut < g.margin) { lo = e - g.margin; hi = e; }
                else { lo = hi = cut; }
                cut = (cut - lo <= hi - cut) ? lo : hi;
            }
        }
        out[q] = std::max(cut, out[q - 1]);
    }
}

ShardRange shard_range(const ts_batch *b, uint32_t n_parts, uint32_t part) {
    std::vector<uint64_t> cut;
    shard_boundaries(b, n_parts, cut);
    const ShardGeom g = shard_geom(b);
    ShardRange r{};
    r.own_lo = cut[part]; r.own_hi = cut[part + 1];
    r.ext_lo = r.own_lo; r.ext_hi = r.own_hi;
    if (r.own_hi > r.own_lo) {
        const uint64_t nt = b->tiles.size();
        if (r.own_lo > 0 && b->segs[b->tiles[r.own_lo].seg].first_tile != r.own_lo) r.ext_lo = r.own_lo - g.ctx;
        if (r.own_hi < nt && b->segs[b->tiles[r.own_hi].seg].first_tile != r.own_hi) r.ext_hi = r.own_hi + g.ctx;
        r.seg_begin = b->tiles[r.own_lo].seg;
        r.n_segs = b->tiles[r.own_hi - 1].seg - r.seg_begin + 1;
    }
    return r;
}

// The context's side stream, on a hardware queue of its own where that can be had.  HIP maps the streams of a process onto a few
// hardware queues (four by default) and does not say which; kernels of two streams on one queue run one after the other.  A side
// stream on the SCAN stream's queue puts the terminal walks of step i (a latency chain: ~60-90 us) in front of the scan of step
// i + 1 — 0.95 ms per 3 Gb step instead of 0.85 (profiles/r05/shard_step_queues.txt); on a pack stream's queue it costs less.
// So every stream a pack meets (the batch's scan stream, the pack's own) is tried once against the side stream
// (ts_k_streams_concurrent: ~1 ms, and it waits for the work those streams hold); when they share a queue, up to eight fresh
// streams are tried for one that runs beside every stream seen so far.  TS_SIDE_PROBE=0 keeps the first stream.
int side_stream_for(ts_ctx *c, void *scan_stream, void *pack_stream) {
    std::lock_guard<std::mutex> lk(c->side_mtx);
    if (!c->side_stream) HIP_TRY(c, hipStreamCreateWithPriority(&c->side_stream, hipStreamNonBlocking, c->knobs.side_priority));
    if (!c->knobs.side_probe) return TS_OK;
    for (void *s : {scan_stream, pack_stream}) {
        if (!s || std::find(c->side_tried.begin(), c->side_tried.end(), s) != c->side_tried.end()) continue;
        c->side_tried.push_back(s);                          // (tried once, whatever comes of it)
        int ok = 0;
        if (ts_k_streams_concurrent(s, c->side_stream, &ok) != 0) return c->fail(TS_ERR_HIP, "stream probe failed");
        if (ok) continue;
        std::vector<hipStream_t> fresh;
        hipStream_t found = nullptr;
        for (int i = 0; i < 8 && !found; ++i) {
            hipStream_t cand = nullptr;
            if (hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, c->knobs.side_priority) != hipSuccess) break;
            fresh.push_back(cand);
            bool all = true;
            for (void *t : c->side_tried) {
                int k = 0;
                if (ts_k_streams_concurrent(t, cand, &k) != 0) { all = false; break; }
                if (!k) { all = false; break; }
            }
            if (all) found = cand;
        }
        for (hipStream_t f : fresh) if (f != found) (void)hipStreamDestroy(f);
        if (found) {
            (void)hipStreamSynchronize(c->side_stream);
            (void)hipStreamDestroy(c->side_stream);
            c->side_stream = found;
        }
    }
    return TS_OK;
}

uint64_t align16u(uint64_t v) { return (v + 15ull) & ~15ull; }

// Sizes of a shard's message.  The variable sections (visible records, blocks) get capacities estimated from the plan —
// canonical patterns per 4^k positions on random sequence, every position of a terminal zone, a block or two per segment —
// times `scale`; a message that overflows says so in its header and the caller packs again with a larger scale.
ShardLayout shard_layout(const ts_batch *b, const ShardRange &r, uint32_t scale) {
    const ts_ctx *c = b->ctx;
    const ts_params &P = c->params;
    ShardLayout L{};
    if (scale == 0) scale = 1;
    {
        uint64_t w0 = 0, w1 = 0;
        if (!b->tips && r.own_hi > r.own_lo) {
            w0 = b->tiles[r.own_lo].win_out;
            const TsTile &last = b->tiles[r.own_hi - 1];
            w1 = last.win_out + last.nwin;
        }
        L.n_windows = w1 - w0;
    }
    const bool nuc = P.out_gc || P.out_entropy;
    L.field_bits = bit_width_u32(P.window_size);
    L.window_bytes = b->tips ? 0u : (uint32_t)(((nuc ? 7u : 3u) * L.field_bits + 7u) / 8u);
    L.visible_bytes = ts_batch_wire16_ok(b) ? 2u : 4u;
    uint64_t own_bases = 0, zone_bases = 0;
    for (uint64_t t = r.own_lo; t < r.own_hi; ++t) {
        const TsTile &T = b->tiles[t];
        own_bases += T.own_len;
        const SegPlan &sp = b->segs[T.seg];
        const uint64_t rel0 = T.in_off - sp.in_off, rel1 = rel0 + T.own_len;
        const uint64_t hi_begin = sp.len > P.terminal_limit ? sp.len - P.terminal_limit : 0;
        if (rel0 <= P.terminal_limit || rel1 > hi_begin) zone_bases += T.own_len;
    }
    uint64_t ncanon = 0;
    for (const ts::Pattern &p : c->patterns) ncanon += p.is_canonical ? 1 : 0;
    const double d_canon = (double)std::max<uint64_t>(ncanon, 1) / (double)(1ull << (2 * std::min<uint32_t>(c->k, 16)));
    uint64_t vis = b->tips ? 0 : (uint64_t)((double)own_bases * d_canon * 1.5) + zone_bases / 8 + 65536;   // (every match of a terminal zone is visible: ~3 % of its bases, more inside a telomere)
    vis = std::min<uint64_t>(vis * scale, own_bases + 16);
    L.visible_capacity = (vis + 7ull) & ~7ull;
    L.block_capacity = (uint32_t)std::min<uint64_t>((16ull * r.n_segs + 256ull) * scale, 1u << 24);
    uint64_t o = sizeof(TsShardHeader);
    L.off_segs = o; o = align16u(o + r.n_segs * sizeof(TsShardSeg));
    L.off_windows = o; o = align16u(o + L.n_windows * L.window_bytes);
    L.off_tilevis = o; o = align16u(o + (r.own_hi - r.own_lo) * 2ull);
    L.off_visible = o; o = align16u(o + L.visible_capacity * L.visible_bytes);
    L.off_blocks = o; o = align16u(o + (uint64_t)L.block_capacity * sizeof(TsDevBlock));
    L.bytes = o;
    return L;
}

void fill_shard_info(const ts_batch *b, uint32_t n_parts, uint32_t part, uint32_t scale, ts_shard_info &o) {
    const ShardRange r = shard_range(b, n_parts, part);
    const ShardLayout L = shard_layout(b, r, scale);
    std::memset(&o, 0, sizeof o);
    o.n_parts = n_parts; o.part = part;
    o.own_begin = r.own_lo; o.own_end = r.own_hi; o.ext_begin = r.ext_lo; o.ext_end = r.ext_hi;
    ts_range_info own{}, ext{};
    ts_batch_range_info(b, r.own_lo, r.own_hi, &own);
    ts_batch_range_info(b, r.ext_lo, r.ext_hi, &ext);
    o.window_begin = own.window_begin; o.window_end = own.window_end;
    o.input_begin = ext.input_begin; o.input_end = ext.input_end;
    o.bases = own.bases;
    o.seg_begin = r.seg_begin; o.seg_end = r.seg_begin + r.n_segs;
    o.msg_bytes = L.bytes;
    o.visible_capacity = L.visible_capacity; o.block_capacity = L.block_capacity;
    o.window_bytes = L.window_bytes; o.visible_bytes = L.visible_bytes;
    o.context_tiles = shard_geom(b).ctx;
}

// f(i) for i in [0, n) on up to max_threads host threads
template <typename F>
void parallel_for(size_t n, unsigned max_threads, F &&f) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nt = (unsigned)std::min<size_t>({(size_t)max_threads, (size_t)hw, n});
    if (nt <= 1) { for (size_t i = 0; i < n; ++i) f(i); return; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t)
        pool.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < n;) f(i); });
    for (std::thread &th : pool) th.join();
}

}  // namespace

void ts_shard_boundaries(const ts_batch *b, uint32_t n_parts, std::vector<uint64_t> &out) { shard_boundaries(b, n_parts, out); }

extern "C" {

int ts_batch_shard_info(const ts_batch *b, uint32_t n_parts, uint32_t part, uint32_t scale, ts_shard_info *out) {
    if (!b || !out || !n_parts || part >= n_parts) return TS_ERR_INVALID_ARG;
    fill_shard_info(b, n_parts, part, scale, *out);
    return TS_OK;
}

int ts_batch_restrict_shard(ts_batch *b, uint32_t n_parts, uint32_t part, uint32_t scale) {
    if (!b || !n_parts || part >= n_parts) return TS_ERR_INVALID_ARG;
    if (b->allocated || b->scanned) return b->ctx->fail(TS_ERR_STATE, "ts_batch_restrict_shard after the batch was used on the device");
    const ShardRange r = shard_range(b, n_parts, part);
    int rc = ts_batch_restrict(b, r.ext_lo, r.ext_hi);
    if (rc != TS_OK) return rc;
    (void)ts_batch_set_emit(b, 1);                       // the pack takes the visible records and the chain summaries from the scan
    b->shard_parts = n_parts; b->shard_part = part; b->shard_scale = scale ? scale : 1;
    b->own_lo = r.own_lo; b->own_hi = r.own_hi;
    b->shard_r = r;
    b->shard_L = shard_layout(b, r, b->shard_scale);
    return TS_OK;
}

int ts_batch_set_shard_scale(ts_batch *b, uint32_t scale) {
    if (!b || !b->shard_parts || !scale) return TS_ERR_INVALID_ARG;
    b->shard_scale = scale;
    b->shard_L = shard_layout(b, b->shard_r, scale);
    b->bound_msg = nullptr;                               // (a message of the new size is bound anew)
    return TS_OK;
}

int ts_batch_bind_shard_message(ts_batch *b, void *d_msg, uint64_t msg_bytes) {
    if (!b) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    if (!b->shard_parts) return c->fail(TS_ERR_STATE, "ts_batch_bind_shard_message needs ts_batch_restrict_shard first");
    if (d_msg && msg_bytes < b->shard_L.bytes) return c->fail(TS_ERR_INVALID_ARG, "ts_batch_bind_shard_message: message buffer smaller than ts_batch_shard_info says");
    b->bound_msg = d_msg;
    return TS_OK;
}

int ts_batch_pack_shard(ts_batch *b, void *d_msg, uint64_t msg_bytes, void *stream) {
    if (!b || !d_msg) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    DEVICE_TRY(c);
    if (!b->shard_parts) return c->fail(TS_ERR_STATE, "ts_batch_pack_shard needs ts_batch_restrict_shard first");
    if (!b->scanned || b->dense) return c->fail(TS_ERR_STATE, "ts_batch_pack_shard needs a scanned batch");
    const ShardRange &r = b->shard_r;
    const ShardLayout &L = b->shard_L;
    if (msg_bytes < L.bytes) return c->fail(TS_ERR_INVALID_ARG, "ts_batch_pack_shard: message buffer smaller than ts_batch_shard_info says");
    // (a scan that packed its window records into a bound message left no 8 x u32 records for the pack's own kernel)
    if (b->msg_windows && b->msg_windows != d_msg)
        return c->fail(TS_ERR_STATE, "ts_batch_pack_shard: the scan packed its window records into the message bound by ts_batch_bind_shard_message; pack into that buffer");
    hipStream_t st = (hipStream_t)stream;
    const uint32_t ns = (uint32_t)r.n_segs;
    const uint32_t nown = (uint32_t)(r.own_hi - r.own_lo);
    // per-segment table of the kernels, uploaded once (the plan does not change)
    if (!b->d_shard_segs.p) {
        std::vector<TsShardSegIn> tab(std::max<uint32_t>(ns, 1));
        for (uint32_t i = 0; i < ns; ++i) {
            const SegPlan &sp = b->segs[r.seg_begin + i];
            TsShardSegIn &S = tab[i];
            S.in_off = sp.in_off; S.len = sp.len; S.abs_pos = sp.abs_pos;
            const uint64_t f = sp.first_tile, e = f + sp.n_tiles;
            const uint64_t t0 = std::min(std::max(f, r.ext_lo), r.ext_hi), t1 = std::max(std::min(e, r.ext_hi), t0);
            const uint64_t o0 = std::min(std::max(f, r.own_lo), r.own_hi), o1 = std::max(std::min(e, r.own_hi), o0);
            S.t0 = (uint32_t)(t0 - r.ext_lo); S.t1 = (uint32_t)(t1 - r.ext_lo);
            S.lo_rel = 0; S.hi_rel = sp.len;
            if (t1 > t0) {
                S.lo_rel = b->tiles[t0].in_off - sp.in_off;
                S.hi_rel = b->tiles[t1 - 1].in_off - sp.in_off + b->tiles[t1 - 1].own_len;
            }
            S.o0 = (uint32_t)(o0 - r.ext_lo); S.o1 = (uint32_t)(o1 - r.ext_lo);
            S.flags = 0;
            if (sp.n_tiles == 0 || (f >= r.own_lo && f < r.own_hi)) S.flags |= TS_SEG_F_HAS_START;
            if (sp.n_tiles == 0 || (e > r.own_lo && e <= r.own_hi)) S.flags |= TS_SEG_F_HAS_END;
            S.seg = (uint32_t)(r.seg_begin + i);
        }
        HIP_TRY(c, c->pool.take(tab.size() * sizeof(TsShardSegIn), b->d_shard_segs));
        HIP_TRY(c, hipMemcpy(b->d_shard_segs.p, tab.data(), tab.size() * sizeof(TsShardSegIn), hipMemcpyHostToDevice));
        HIP_TRY(c, c->pool.take((size_t)std::max<uint32_t>(ns, 1) * (16 + 40), b->d_shard_bounds));     // bounds, then the per-segment sums
        HIP_TRY(c, c->pool.take((size_t)ts_k_shard_tmp_bytes(nown), b->d_shard_tmp));
    }
    // the list of chains the interstitial screening hands to its evaluation kernel, sized with the blocks (its counter lives in
    // a reserved word of the header: cleared with it, overwritten by the header kernel at the end)
    const uint32_t cand_cap = 2u * L.block_capacity + 256u;
    if (b->d_shard_cand.bytes < (size_t)cand_cap * 8 + 16) {
        c->pool.give(std::move(b->d_shard_cand));
        HIP_TRY(c, c->pool.take((size_t)cand_cap * 8 + 16, b->d_shard_cand));
    }
    { const int rc = side_stream_for(c, b->last_stream, stream); if (rc != TS_OK) return rc; }
    if (!b->ev_fork) {
        HIP_TRY(c, hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
    }
    unsigned char *msg = (unsigned char *)d_msg;
    // everything the pack's kernels accumulate into, zeroed by ONE kernel: the message's header + per-segment entries, the
    // per-segment sums, the counter of the screening's tile list and the counter of the visible-record copy's list
    const uint32_t ns_z = std::max<uint32_t>((uint32_t)b->shard_r.n_segs, 1);
    const bool from_scan_z = b->chain_valid() && !b->tips;
    if (from_scan_z && b->d_scan_tmp.bytes < ((size_t)b->range_tiles() + 2) * 4) {
        c->pool.give(std::move(b->d_scan_tmp));
        HIP_TRY(c, c->pool.take(((size_t)b->range_tiles() + 2) * 4, b->d_scan_tmp));
    }
    (void)ns_z;                                          // (the per-segment counts accumulate in the message's own entries, zeroed with the header)
    if (ts_k_launch_zero(msg, L.off_windows & ~3ull, nullptr, 0,
                         from_scan_z ? b->d_scan_tmp.p : nullptr, 4,
                         (from_scan_z && nown) ? ts_k_shard_big_counter(b->d_shard_tmp.p, nown) : nullptr, 4, st) != 0)
        return c->fail(TS_ERR_HIP, "zeroing kernel launch failed");
    const ts_params &P = c->params;
    TsBlockCallParams Q{};
    Q.tiles = (const TsTile *)b->d_tiles.p;
    Q.tile_off = (const unsigned long long *)b->d_tile_off.p;
    Q.tile_stats = b->stats_ptr();
    Q.matches = b->records_ptr();
    Q.rec16 = b->records16() ? 1u : 0u;
    Q.blocks = (TsDevBlock *)(msg + L.off_blocks);
    Q.n_blocks = &((TsShardHeader *)msg)->n_blocks;
    Q.block_cap = L.block_capacity;
    Q.n_cand = (uint32_t *)&((TsShardHeader *)msg)->reserved[0];
    Q.cand = (uint32_t *)b->d_shard_cand.p;
    Q.cand_cap = cand_cap;
    Q.terminal_limit = P.terminal_limit; Q.max_match_dist = P.max_match_dist;
    Q.min_block_len = P.min_block_len; Q.max_block_dist = P.max_block_dist;
    Q.min_block_counts = P.min_block_counts; Q.min_block_density = P.min_block_density;
    Q.k = c->k; Q.its_min_len = (uint32_t)(uint16_t)(2 * c->bp.first_pattern_len);
    TsShardPackParams K{};
    K.tiles = Q.tiles; K.tile_off = Q.tile_off; K.tile_stats = Q.tile_stats; K.matches = Q.matches; K.rec16 = Q.rec16;
    K.windows = b->windows_ptr();
    K.wave_fill = (const uint32_t *)b->d_fill.p;
    K.region_cap = b->region_cap; K.nwaves = b->total_waves;
    K.own0 = (uint32_t)(r.own_lo - r.ext_lo); K.own1 = (uint32_t)(r.own_hi - r.ext_lo);
    K.win_lo = b->win_lo;
    if (!b->tips && nown) {
        K.own_win0 = b->tiles[r.own_lo].win_out;
        K.own_win1 = b->tiles[r.own_hi - 1].win_out + b->tiles[r.own_hi - 1].nwin;
    }
    K.segs = (const TsShardSegIn *)b->d_shard_segs.p;
    K.n_segs = ns;
    K.terminal_limit = P.terminal_limit; K.k = c->k; K.nuc_on = (P.out_gc || P.out_entropy) ? 1u : 0u;
    K.field_bits = L.field_bits;
    K.msg = msg;
    K.off_segs = L.off_segs; K.off_windows = L.off_windows; K.off_tilevis = L.off_tilevis; K.off_visible = L.off_visible;
    K.off_blocks = L.off_blocks;
    // with the scan's own visible records and chain summaries (kp.emit) neither the counting pass nor the interstitial pass
    // reads the match stream
    const bool from_scan = b->chain_valid() && !b->tips;
    if (from_scan) {
        K.chain = (const uint32_t *)b->d_chain.p;
        K.vis_src = b->d_vis.p; K.vis_src_wide = b->kp.vis_wide; K.vis_cap = b->vis_cap;
        if (b->d_scan_tmp.bytes < ((size_t)b->range_tiles() + 2) * 4) {           // the screening's list of tiles
            c->pool.give(std::move(b->d_scan_tmp));
            HIP_TRY(c, c->pool.take(((size_t)b->range_tiles() + 2) * 4, b->d_scan_tmp));
        }
    }
    TsShardHeader H{};
    H.magic = TS_SHARD_MAGIC; H.version = TS_SHARD_VERSION;
    H.part = b->shard_part; H.n_parts = b->shard_parts;
    H.own_begin = r.own_lo; H.own_end = r.own_hi; H.ext_begin = r.ext_lo; H.ext_end = r.ext_hi;
    H.seg_begin = r.seg_begin; H.n_segs = ns;
    H.visible_bytes = L.visible_bytes; H.visible_capacity = L.visible_capacity; H.block_capacity = L.block_capacity;
    H.window_bytes = L.window_bytes; H.n_windows = L.n_windows; H.msg_bytes = L.bytes;
    H.reserved[1] = b->shard_scale;                       // the capacity scale the sections were laid out with (any value >= 1)
    // visible records per owned tile and their places; then the terminal walks and the interstitial pass, which also
    // writes the visible records (it reads the whole stream anyway); then the window records and the header
    // The terminal walks (one latency-bound wave per segment end, ~60 us whatever the shard's size) run on a stream of their
    // own beside the counting, the prefix sum and the window packing; the interstitial pass needs both (the bounds, and
    // where every tile's visible records go) and joins them.
    HIP_TRY(c, hipEventRecord(b->ev_fork, st));
    HIP_TRY(c, hipStreamWaitEvent(c->side_stream, b->ev_fork, 0));
    if (!(TS_PACK_ABL & 1) && ts_k_launch_terminal(&Q, (const TsShardSegIn *)b->d_shard_segs.p, ns, (uint32_t)r.seg_begin, (uint32_t)b->range_tiles(),
                             (unsigned long long *)b->d_shard_bounds.p, (TsShardSeg *)(msg + L.off_segs), c->side_stream) != 0)
        return c->fail(TS_ERR_HIP, "terminal block kernel launch failed");
    HIP_TRY(c, hipEventRecord(b->ev_join, c->side_stream));
    TsVisibleOut vis{};
    if (from_scan) {
        if (!(TS_PACK_ABL & 2) && ts_k_launch_shard_visible(&K, &H, b->d_shard_tmp.p, 1, stream) != 0)
            return c->fail(TS_ERR_HIP, "visible-record kernel launch failed");
    } else if (ts_k_launch_shard_count(&K, &H, b->d_shard_tmp.p, b->tips ? 0 : 1, &vis, stream) != 0)
        return c->fail(TS_ERR_HIP, "shard count kernel launch failed");
    if (!b->msg_windows && ts_k_launch_shard_windows(&K, &H, stream) != 0)        // (unless the scan packed them: ts_batch_bind_shard_message)
        return c->fail(TS_ERR_HIP, "window packing kernel launch failed");
    if (ts_k_launch_shard_overflow(&K, stream) != 0)
        return c->fail(TS_ERR_HIP, "overflow check kernel launch failed");
    // the per-segment counts of the message: added up by the screening kernel of the interstitial search when that runs from
    // the scan's chain summaries, by a kernel of their own otherwise (a tips-only shard; results without summaries)
    if (!from_scan && ts_k_launch_segment_sums(&Q, (const TsShardSegIn *)b->d_shard_segs.p, ns, (uint32_t)r.seg_begin, (uint32_t)b->range_tiles(),
                                               nullptr, (TsShardSeg *)(msg + L.off_segs), 1, stream) != 0)
        return c->fail(TS_ERR_HIP, "segment sums kernel launch failed");
    HIP_TRY(c, hipStreamWaitEvent(st, b->ev_join, 0));
    if (!(TS_PACK_ABL & 8) && !b->tips && ts_k_launch_interstitial(&Q, (const TsShardSegIn *)b->d_shard_segs.p, ns, (uint32_t)r.seg_begin, (uint32_t)b->range_tiles(),
                                             (const unsigned long long *)b->d_shard_bounds.p, (TsShardSeg *)(msg + L.off_segs), &vis,
                                             from_scan ? K.chain : nullptr, from_scan ? (uint32_t *)b->d_scan_tmp.p : nullptr, 1, stream) != 0)
        return c->fail(TS_ERR_HIP, "interstitial block kernel launch failed");
    if (!(TS_PACK_ABL & 32) && ts_k_launch_shard_pack(&K, &H, b->d_shard_tmp.p, b->tips ? 0 : 1, stream) != 0)
        return c->fail(TS_ERR_HIP, "shard pack kernel launch failed");
    return TS_OK;
}

int ts_shard_peek(const void *msg, uint64_t msg_bytes, ts_shard_status *out) {
    if (!msg || !out || msg_bytes < sizeof(TsShardHeader)) return TS_ERR_INVALID_ARG;
    const TsShardHeader *H = (const TsShardHeader *)msg;
    if (H->magic != TS_SHARD_MAGIC || H->version != TS_SHARD_VERSION) return TS_ERR_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    out->part = H->part; out->n_parts = H->n_parts; out->flags = H->flags;
    out->n_visible = H->n_visible; out->visible_capacity = H->visible_capacity;
    out->n_blocks = H->n_blocks; out->block_capacity = H->block_capacity;
    out->msg_bytes = H->msg_bytes;
    uint32_t need = 1;
    while (H->visible_capacity && H->n_visible > H->visible_capacity * need) need *= 2;
    while (H->block_capacity && (uint64_t)H->n_blocks > (uint64_t)H->block_capacity * need) need *= 2;
    out->scale_factor_needed = need;
    return TS_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------ finalize (host)
namespace {

struct PartView {
    const unsigned char *msg;
    const TsShardHeader *H;
    const TsShardSeg *segs;
    ShardRange r;
    ShardLayout L;
    std::vector<uint64_t> vis_off;          // per owned tile: index of its first visible record
};

inline uint32_t visible_rec(const PartView &pv, uint64_t i) {
    return pv.H->visible_bytes == 2u ? ((const uint16_t *)(pv.msg + pv.L.off_visible))[i]
                                     : ((const uint32_t *)(pv.msg + pv.L.off_visible))[i];
}

}  // namespace

extern "C" int ts_shards_finalize(const ts_batch *b, const void *const *msgs, const uint64_t *msg_bytes, uint32_t n_parts,
                                  ts_segment_out *out, ts_segment_counts *counts) {
    if (!b || !msgs || !msg_bytes || !n_parts || !out) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    const ts_params &P = c->params;
    const size_t ns = b->segs.size();
    for (size_t i = 0; i < ns; ++i) {
        std::memset(&out[i], 0, sizeof out[i]);
        if (counts) counts[i] = ts_segment_counts{b->tips ? 0 : b->segs[i].n_windows, 0, 0, 0};
    }
    // ---- the parts' headers against the plan
    std::vector<PartView> parts(n_parts);
    uint32_t flags_any = 0;
    for (uint32_t p = 0; p < n_parts; ++p) {
        PartView &pv = parts[p];
        if (!msgs[p] || msg_bytes[p] < sizeof(TsShardHeader)) return c->fail(TS_ERR_INVALID_ARG, "ts_shards_finalize: missing message");
        pv.msg = (const unsigned char *)msgs[p];
        pv.H = (const TsShardHeader *)pv.msg;
        const TsShardHeader &H = *pv.H;
        if (H.magic != TS_SHARD_MAGIC || H.version != TS_SHARD_VERSION || H.n_parts != n_parts || H.part != p)
            return c->fail(TS_ERR_INVALID_ARG, "ts_shards_finalize: message " + std::to_string(p) + " is not part " + std::to_string(p) + " of " + std::to_string(n_parts));
        pv.r = shard_range(b, n_parts, p);
        // the scale the sender laid its sections out with travels in the header (reserved[1]); what it implies is checked below
        const uint32_t scale = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(H.reserved[1], 1), 1u << 24);
        pv.L = shard_layout(b, pv.r, scale);
        if (pv.L.visible_capacity != H.visible_capacity || pv.L.block_capacity != H.block_capacity)
            return c->fail(TS_ERR_INVALID_ARG, "ts_shards_finalize: message " + std::to_string(p) + " was packed with capacities that do not match its scale");
        if (H.own_begin != pv.r.own_lo || H.own_end != pv.r.own_hi || H.ext_begin != pv.r.ext_lo || H.ext_end != pv.r.ext_hi ||
            H.n_segs != pv.r.n_segs || (H.n_segs && H.seg_begin != pv.r.seg_begin) || H.msg_bytes != pv.L.bytes || msg_bytes[p] < pv.L.bytes ||
            H.window_bytes != pv.L.window_bytes || H.visible_bytes != pv.L.visible_bytes || H.n_windows != pv.L.n_windows)
            return c->fail(TS_ERR_INVALID_ARG, "ts_shards_finalize: message " + std::to_string(p) + " was packed for a different plan or split");
        pv.segs = (const TsShardSeg *)(pv.msg + pv.L.off_segs);
        flags_any |= H.flags;
        // counts a message reports beyond what its sections hold are only legitimate together with the overflow flag that
        // asks for a larger message; anything else (a truncated or corrupt message, a header kernel that did not run) is refused
        // here, before any count is used as a length
        const bool vis_over = H.n_visible > H.visible_capacity, blk_over = H.n_blocks > H.block_capacity;
        if ((vis_over && !(H.flags & TS_SHARD_F_VISIBLE_OVERFLOW)) || (blk_over && !(H.flags & TS_SHARD_F_BLOCK_OVERFLOW)) ||
            (!blk_over && pv.L.off_blocks + (uint64_t)H.n_blocks * sizeof(TsDevBlock) > msg_bytes[p]) ||
            (!vis_over && pv.L.off_visible + H.n_visible * H.visible_bytes > msg_bytes[p]))
            return c->fail(TS_ERR_INVALID_ARG, "ts_shards_finalize: message " + std::to_string(p) + " reports more records or blocks than it holds");
    }
    if (flags_any & TS_SHARD_F_SCAN_OVERFLOW) { c->fail(TS_OK, "a shard's scan overflowed its record regions: ts_batch_sync, then pack again"); return TS_SHARD_RETRY_SYNC; }
    if (flags_any & (TS_SHARD_F_VISIBLE_OVERFLOW | TS_SHARD_F_BLOCK_OVERFLOW)) { c->fail(TS_OK, "a shard's message overflowed: pack again with a larger scale"); return TS_SHARD_RETRY_GROW; }
    if (flags_any & TS_SHARD_F_CONTEXT) { c->fail(TS_OK, "a chain of matches or a terminal walk ran out of a shard's context tiles"); return TS_SHARD_NEED_FULL; }

    // ---- per segment: who owns tiles of it, the counts, and the checks on what the parts assumed about each other
    for (size_t si = 0; si < ns; ++si) {
        const SegPlan &sp = b->segs[si];
        if (!sp.n_tiles) continue;
        struct Own { uint32_t p; const TsShardSeg *e; };
        std::vector<Own> owners;
        for (uint32_t p = 0; p < n_parts; ++p) {
            const PartView &pv = parts[p];
            if (!pv.r.n_segs || si < pv.r.seg_begin || si >= pv.r.seg_begin + pv.r.n_segs) continue;
            owners.push_back({p, &pv.segs[si - pv.r.seg_begin]});
        }
        if (owners.empty()) return c->fail(TS_ERR_STATE, "ts_shards_finalize: a segment is owned by no part");
        uint64_t total = 0, ncan = 0, nfwd = 0;
        for (const Own &o : owners) { total += o.e->n_matches; ncan += o.e->n_canonical; nfwd += o.e->n_forward; }
        if (counts) { counts[si].n_matches = total; counts[si].n_canonical = ncan; counts[si].n_forward = nfwd; }
        if (owners.size() == 1) {
            if ((owners[0].e->flags & (TS_SEG_F_HAS_START | TS_SEG_F_HAS_END)) != (TS_SEG_F_HAS_START | TS_SEG_F_HAS_END))
                return c->fail(TS_ERR_STATE, "ts_shards_finalize: a segment's only owner does not hold both its ends");
            continue;
        }
        const Own &first = owners.front(), &last = owners.back();
        if (!(first.e->flags & TS_SEG_F_HAS_START) || !(last.e->flags & TS_SEG_F_HAS_END))
            return c->fail(TS_ERR_STATE, "ts_shards_finalize: the owners of a split segment do not hold its ends");
        // the walks ran iff the reference's would have (src/teloscope.cpp:646-651 gate them on the whole lists' sizes)
        const bool fwd_ok = ((first.e->flags & TS_SEG_F_FWD_WALKED) != 0) == (nfwd >= 2);
        const bool rev_ok = ((last.e->flags & TS_SEG_F_REV_WALKED) != 0) == (total - nfwd >= 2);
        // every other part searched [0, n) resp. up to n for interstitial blocks: right iff the real bounds lie outside
        // everything that part looked at
        const uint64_t fb = first.e->fwd_boundary, rb = (last.e->flags & TS_SEG_F_REV_WALKED) ? last.e->rev_boundary : sp.len;
        bool bounds_ok = true;
        for (const Own &o : owners) {
            const PartView &pv = parts[o.p];
            const uint64_t t0 = std::max<uint64_t>(sp.first_tile, pv.r.ext_lo), t1 = std::min<uint64_t>(sp.first_tile + sp.n_tiles, pv.r.ext_hi);
            if (t1 <= t0) continue;
            const uint64_t lo_rel = b->tiles[t0].in_off - sp.in_off;
            const uint64_t hi_rel = b->tiles[t1 - 1].in_off - sp.in_off + b->tiles[t1 - 1].own_len;
            if (o.p != first.p && fb > lo_rel) bounds_ok = false;
            if (o.p != last.p && rb < hi_rel) bounds_ok = false;
        }
        if (!fwd_ok || !rev_ok || !bounds_ok) {
            c->fail(TS_OK, "segment " + std::to_string(si) + ": a terminal block reaches beyond the tiles its owner keeps clear of the split");
            return TS_SHARD_NEED_FULL;
        }
    }

    // ---- allocate: windows, visible matches
    std::vector<uint64_t> seg_nvis(ns, 0);
    for (uint32_t p = 0; p < n_parts; ++p) {
        PartView &pv = parts[p];
        const uint64_t nown = pv.r.own_hi - pv.r.own_lo;
        const uint16_t *tv = (const uint16_t *)(pv.msg + pv.L.off_tilevis);
        pv.vis_off.assign(nown + 1, 0);
        for (uint64_t i = 0; i < nown; ++i) {
            pv.vis_off[i + 1] = pv.vis_off[i] + tv[i];
            seg_nvis[b->tiles[pv.r.own_lo + i].seg] += tv[i];
        }
        if (!b->tips && pv.vis_off[nown] != pv.H->n_visible)
            return c->fail(TS_ERR_STATE, "ts_shards_finalize: a message's per-tile counts do not add up to its record count");
    }
    int rc = TS_OK;
    for (size_t si = 0; si < ns && rc == TS_OK; ++si) {
        const SegPlan &sp = b->segs[si];
        if (!b->tips && sp.n_windows) {
            out[si].windows = (ts_window *)std::malloc(sp.n_windows * sizeof(ts_window));
            if (!out[si].windows) rc = c->fail(TS_ERR_ALLOC, "out of host memory");
            out[si].n_windows = sp.n_windows;
        }
        if (rc == TS_OK && seg_nvis[si]) {
            out[si].matches = (ts_match *)std::malloc(seg_nvis[si] * sizeof(ts_match));
            if (!out[si].matches) rc = c->fail(TS_ERR_ALLOC, "out of host memory");
            out[si].n_matches = seg_nvis[si];
        }
    }
    if (rc != TS_OK) { ts_free_segments(out, ns); return rc; }
    // index (within its segment's array) of the first visible record of the first owned tile of every (part, segment)
    // = the visible records of the segment's tiles owned by earlier parts
    std::vector<std::vector<uint64_t>> seg_base(n_parts);
    {
        std::vector<uint64_t> run(ns, 0);
        for (uint32_t p = 0; p < n_parts; ++p) {
            const PartView &pv = parts[p];
            seg_base[p].assign(pv.r.n_segs, 0);
            for (uint64_t s = 0; s < pv.r.n_segs; ++s) seg_base[p][s] = run[pv.r.seg_begin + s];
            const uint64_t nown = pv.r.own_hi - pv.r.own_lo;
            for (uint64_t i = 0; i < nown; ++i) run[b->tiles[pv.r.own_lo + i].seg] += pv.vis_off[i + 1] - pv.vis_off[i];
        }
    }
    // ---- fill, in pieces spread over the host threads: windows [a, z) of a part, or its owned tiles [a, z)
    struct Piece { uint32_t p; bool windows; uint64_t a, z; };
    std::vector<Piece> pieces;
    constexpr uint64_t kWinPiece = 1u << 13, kTilePiece = 256;
    for (uint32_t p = 0; p < n_parts; ++p) {
        const PartView &pv = parts[p];
        for (uint64_t a = 0; a < pv.L.n_windows; a += kWinPiece) pieces.push_back({p, true, a, std::min<uint64_t>(pv.L.n_windows, a + kWinPiece)});
        if (!b->tips && pv.H->n_visible)
            for (uint64_t a = 0; a < pv.r.own_hi - pv.r.own_lo; a += kTilePiece) pieces.push_back({p, false, a, std::min<uint64_t>(pv.r.own_hi - pv.r.own_lo, a + kTilePiece)});
    }
    const bool nuc = P.out_gc || P.out_entropy;
    const uint16_t klen = (uint16_t)c->k;
    unsigned nthreads = 16;
    if (const char *e = getenv("TS_HOST_THREADS")) { const int n = atoi(e); if (n > 0) nthreads = (unsigned)n; }
    parallel_for(pieces.size(), nthreads, [&](size_t pi) {
        const Piece &pc = pieces[pi];
        const PartView &pv = parts[pc.p];
        if (pc.windows) {
            const uint64_t w0 = b->tiles[pv.r.own_lo].win_out;               // plan index of the part's first window
            const uint32_t B = pv.L.field_bits, wb = pv.L.window_bytes;
            const uint64_t mask = B >= 64 ? ~0ull : ((1ull << B) - 1ull);
            // the segment of window w0 + a, then walk
            size_t si = b->tiles[pv.r.own_lo].seg;
            while (si + 1 < ns && (b->segs[si].n_windows == 0 || w0 + pc.a >= b->segs[si].win_base + b->segs[si].n_windows)) ++si;
            for (uint64_t i = pc.a; i < pc.z; ++i) {
                const uint64_t wi = w0 + i;
                while (b->segs[si].n_windows == 0 || wi >= b->segs[si].win_base + b->segs[si].n_windows) ++si;
                const SegPlan &sp = b->segs[si];
                const uint64_t kwin = wi - sp.win_base;
                const unsigned char *src = pv.msg + pv.L.off_windows + i * wb;
                unsigned __int128 v = 0;
                for (uint32_t q = 0; q < wb; ++q) v |= (unsigned __int128)src[q] << (8u * q);
                uint32_t f[7] = {0, 0, 0, 0, 0, 0, 0};
                const uint32_t nf = nuc ? 7u : 3u;
                for (uint32_t q = 0; q < nf; ++q) f[nuc ? q : q + 4u] = (uint32_t)((uint64_t)(v >> (q * B)) & mask);
                ts_window &w = out[si].windows[kwin];
                std::memset(&w, 0, sizeof w);
                const uint64_t ws = kwin * P.step;
                w.window_start = sp.abs_pos + ws;
                w.current_window_size = (uint32_t)std::min<uint64_t>(P.window_size, sp.len - ws);
                if (nuc) for (int q = 0; q < 4; ++q) w.nucleotide_counts[q] = f[q];
                if (P.out_gc) w.gc_content = ts::gc_content(w.nucleotide_counts, w.current_window_size);
                if (P.out_entropy) w.shannon_entropy = ts::shannon_entropy_memo(w.nucleotide_counts, w.current_window_size, c->entropy_term);
                w.canonical_covered = f[4] * klen;
                w.non_canonical_covered = f[5] * klen;
                w.fwd_covered = f[6] * klen;
                w.rev_covered = (f[4] + f[5] - f[6]) * klen;
            }
            return;
        }
        for (uint64_t i = pc.a; i < pc.z; ++i) {
            const uint64_t cnt = pv.vis_off[i + 1] - pv.vis_off[i];
            if (!cnt) continue;
            const TsTile &T = b->tiles[pv.r.own_lo + i];
            const SegPlan &sp = b->segs[T.seg];
            const uint64_t rel0 = T.in_off - sp.in_off;
            const uint64_t term_end = sp.len > P.terminal_limit ? sp.len - P.terminal_limit : 0;
            // where the tile's records go: records of the segment's earlier tiles of this part precede them
            uint64_t at = seg_base[pc.p][T.seg - pv.r.seg_begin];
            const uint64_t first_own = std::max<uint64_t>(sp.first_tile, pv.r.own_lo);
            at += pv.vis_off[i] - pv.vis_off[first_own - pv.r.own_lo];
            ts_match *m = out[T.seg].matches + at;
            for (uint64_t q = 0; q < cnt; ++q) {
                const uint32_t rec = visible_rec(pv, pv.vis_off[i] + q);
                const uint64_t rel = rel0 + (rec >> 2);
                std::memset(&m[q], 0, sizeof m[q]);
                m[q].position = sp.abs_pos + rel;
                m[q].match_size = klen;
                m[q].flags = (uint8_t)(((rec & 2u) ? TS_MATCH_FORWARD : 0u) | ((rec & 1u) ? TS_MATCH_CANONICAL : 0u) |
                                       ((rel <= P.terminal_limit || rel >= term_end) ? TS_MATCH_TERMINAL : 0u));
            }
        }
    });
    // ---- blocks: all parts' lists, ordered like a single device's (segment; terminal blocks in push order — forward
    //      walk, then reverse walk — interstitial blocks by start)
    std::vector<TsDevBlock> blocks;
    for (const PartView &pv : parts) {
        const TsDevBlock *src = (const TsDevBlock *)(pv.msg + pv.L.off_blocks);
        for (uint32_t q = 0; q < pv.H->n_blocks; ++q)           // a block names a segment of its part's range, or the message is corrupt
            if (src[q].seg < pv.r.seg_begin || src[q].seg >= pv.r.seg_begin + pv.r.n_segs) {
                ts_free_segments(out, ns);
                return c->fail(TS_ERR_INVALID_ARG, "ts_shards_finalize: a block names a segment outside its part's range");
            }
        blocks.insert(blocks.end(), src, src + pv.H->n_blocks);
    }
    std::sort(blocks.begin(), blocks.end(), [](const TsDevBlock &x, const TsDevBlock &y) {
        if (x.seg != y.seg) return x.seg < y.seg;
        const uint32_t kx = x.kind == 2 ? 1 : 0, ky = y.kind == 2 ? 1 : 0;
        if (kx != ky) return kx < ky;
        if (kx) return x.start < y.start;
        return x.kind != y.kind ? x.kind < y.kind : x.seq < y.seq;
    });
    size_t bi = 0;
    for (size_t si = 0; si < ns; ++si) {
        size_t nterm = 0, nits = 0;
        const size_t b0 = bi;
        while (bi < blocks.size() && blocks[bi].seg == si) { (blocks[bi].kind == 2 ? nits : nterm)++; ++bi; }
        auto fill = [&](ts_block *&dst, uint64_t &n, size_t count, bool its) -> bool {
            n = count; dst = nullptr;
            if (!count) return true;
            dst = (ts_block *)std::malloc(count * sizeof(ts_block));
            if (!dst) return false;
            size_t at = 0;
            for (size_t q = b0; q < bi; ++q)
                if ((blocks[q].kind == 2) == its) std::memcpy(&dst[at++], &blocks[q], sizeof(ts_block));
            return true;
        };
        if (!fill(out[si].terminal_blocks, out[si].n_terminal_blocks, nterm, false) ||
            !fill(out[si].interstitial_blocks, out[si].n_interstitial_blocks, nits, true)) {
            ts_free_segments(out, ns);
            return c->fail(TS_ERR_ALLOC, "out of host memory");
        }
    }
    if (bi != blocks.size()) { ts_free_segments(out, ns); return c->fail(TS_ERR_STATE, "ts_shards_finalize: a block names a segment outside the batch"); }
    return TS_OK;
}
