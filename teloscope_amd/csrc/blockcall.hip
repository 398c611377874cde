// blockcall.hip — telomere block calling on the device, from the packed match stream (gfx950).
//
// Device form of Teloscope::getTerminalBlocks (src/teloscope.cpp:29-176) and
// getInterstitialBlocks (:179-256), the O(matches) step that follows the scan inside
// scanSegment (:642-657).  With it only blocks (a few per segment) have to leave the GPU instead
// of the whole match stream.
//
//   ts_terminal_blocks      one wave per segment and direction: the reference's two-phase walk (chain matches
//                           <= -k apart inside the terminal zone, keep dense canonical sub-blocks,
//                           merge sub-blocks <= -d apart, keep >= -l) over the forward list from
//                           the start and the reverse list from the end, 64 records per step in parallel
//                           (prefix maximum for the predecessor, ballot of chain heads, one scalar step per
//                           chain); emits the blocks and the two boundaries that fence the interstitial search.
//   ts_interstitial_blocks  one wave per tile, sparse: an interstitial block needs >= 4
//                           canonical matches, and canonical matches are ~2 % of the stream, so the canonical
//                           records of the tile are compacted first and only the FIRST canonical match of a
//                           chain ("leader") walks its chain (matches <= -k apart inside [fwdBoundary,
//                           revBoundary)) and evaluates the reference's filters.
//
// Records are addressed through the tile directory {tile_off, tile_stats}; tiles of one segment
// are consecutive and position-ordered, so prev/next step across tile boundaries.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ts_internal.h"

namespace {

typedef unsigned long long u64;

struct Cursor { uint32_t t, i; };            // record i of tile t

struct SegView {
    const TsTile *tiles;
    const u64 *tile_off;
    const uint32_t *tile_stats;
    const uint32_t *matches;
    uint32_t t0, t1;                         // tiles of the segment (a shard: those of them it scanned)
    u64 base;                                // in_off of the segment
    // a shard's view may be CLIPPED: the segment goes on to the left of t0 / to the right of t1 on a neighbour.
    // lo_rel / hi_rel: segment-relative position of the first base of tile t0 / behind the last base of tile t1 - 1.
    bool open_l, open_r;
    u64 lo_rel, hi_rel;

    __device__ uint32_t count(uint32_t t) const { return tile_stats[4u * t]; }
    __device__ uint32_t rec(Cursor c) const { return matches[tile_off[c.t] + c.i]; }
    __device__ u64 pos(Cursor c, uint32_t r) const { return tiles[c.t].in_off - base + (r >> 2); }
    __device__ bool next(Cursor &c) const {
        if (c.i + 1u < count(c.t)) { ++c.i; return true; }
        for (uint32_t t = c.t + 1u; t < t1; ++t)
            if (count(t)) { c.t = t; c.i = 0; return true; }
        return false;
    }
    __device__ bool prev(Cursor &c) const {
        if (c.i > 0u) { --c.i; return true; }
        for (uint32_t t = c.t; t > t0; --t)
            if (count(t - 1u)) { c.t = t - 1u; c.i = count(t - 1u) - 1u; return true; }
        return false;
    }
    __device__ bool first(Cursor &c) const {
        for (uint32_t t = t0; t < t1; ++t)
            if (count(t)) { c.t = t; c.i = 0; return true; }
        return false;
    }
    __device__ bool last(Cursor &c) const {
        for (uint32_t t = t1; t > t0; --t)
            if (count(t - 1u)) { c.t = t - 1u; c.i = count(t - 1u) - 1u; return true; }
        return false;
    }
};

struct Chain {                               // running chain of matches (startNewBlock / extend)
    u64 start, end, prev;
    uint32_t counts, fwd, canon, cov, fwd_cov, can_cov;
    __device__ void begin(u64 p, uint32_t r, uint32_t k) {
        start = p; end = p + k; prev = p; counts = 1;
        fwd = (r >> 1) & 1u; canon = r & 1u; cov = k; fwd_cov = fwd * k; can_cov = canon * k;
    }
    __device__ void add(u64 p, uint32_t r, uint32_t k) {
        ++counts; const uint32_t f = (r >> 1) & 1u, c = r & 1u;
        fwd += f; canon += c; cov += k; fwd_cov += f * k; can_cov += c * k; prev = p;
    }
    __device__ void to_block(TsDevBlock &b) const {
        b.start = start; b.block_len = (uint32_t)(end - start); b.block_counts = counts;
        b.forward_count = fwd; b.reverse_count = counts - fwd; b.canonical_count = canon;
        b.non_canonical_count = counts - canon; b.total_covered = cov; b.fwd_covered = fwd_cov;
        b.can_covered = can_cov; b.has_valid_or = 1; b.is_longest = 0; b.block_label = 0; b.reserved = 0;
    }
};

__device__ void emit_block(const TsBlockCallParams &Q, TsDevBlock &b, uint32_t seg, uint32_t kind, uint32_t seq,
                           u64 abs_pos) {
    const uint32_t slot = atomicAdd(Q.n_blocks, 1u);
    if (slot >= Q.block_cap) return;                       // overflow: the host sees n_blocks > cap
    b.start += abs_pos;
    b.seg = seg; b.kind = kind; b.seq = seq; b.pad = 0;
    Q.blocks[slot] = b;
}

// Wave-wide inclusive prefix maximum in 6 DPP steps (row_shr 1/2/4/8 inside each row of 16, then row_bcast:15
// into rows 1,3 and row_bcast:31 into rows 2,3); lanes outside a shift read 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_max(uint32_t v) {
    const uint32_t o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
    return v > o ? v : o;
}
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t v) {
    v = dpp_max<0x111, 0xf>(v);
    v = dpp_max<0x112, 0xf>(v);
    v = dpp_max<0x114, 0xf>(v);
    v = dpp_max<0x118, 0xf>(v);
    v = dpp_max<0x142, 0xa>(v);
    v = dpp_max<0x143, 0xc>(v);
    return v;
}

// One direction of getTerminalBlocks for one segment; returns the boundary.  Run by a whole wave, 64 records
// per step IN PARALLEL: lane i holds the i-th record of the batch in walk order (ascending for the forward list
// from the start, descending for the reverse list from the end); a prefix maximum gives every record of the
// wanted orientation its predecessor in the walk, a ballot marks those that open a new chain (gap > -k), and the
// state machine — wave-uniform, scalar — steps once per CHAIN (counts are popcounts of ballots), not once per
// record: a telomere is one chain of thousands of matches.  Lane 0 writes the blocks.
__device__ u64 terminal_direction(const TsBlockCallParams &Q, const SegView &V, uint32_t seg, u64 n, u64 abs_pos,
                                  bool from_start, uint32_t &seq, uint32_t lane, bool &out_of_context) {
    u64 boundary = from_start ? 0 : n;                     // segment-relative
    Chain ch; bool open = false;
    ch.start = ch.end = ch.prev = 0; ch.counts = ch.fwd = ch.canon = ch.cov = ch.fwd_cov = ch.can_cov = 0;
    TsDevBlock cur; bool have_cur = false;
    cur.start = 0; cur.block_len = 0;
    auto close_block = [&]() {
        if (cur.block_len >= Q.min_block_len) {
            cur.block_label = from_start ? 'p' : 'q';
            const u64 rel_start = cur.start, rel_end = rel_start + cur.block_len;
            const u64 left = rel_start, right = rel_end <= n ? n - rel_end : 0;
            cur.has_valid_or = from_start ? (left <= right) : (left >= right);
            boundary = from_start ? cur.start + cur.block_len : cur.start;
            if (lane == 0) {
                TsDevBlock out = cur;
                emit_block(Q, out, seg, from_start ? 0u : 1u, seq, abs_pos);
            }
            ++seq;
        }
    };
    auto close_sub = [&]() {
        const float need = Q.min_block_density * (float)(ch.end - ch.start);
        if (ch.counts >= Q.min_block_counts && ch.canon > 0u && (float)ch.can_cov >= need) {
            TsDevBlock sb; ch.to_block(sb);
            if (!have_cur) { cur = sb; have_cur = true; }
            else {
                const u64 gap = from_start ? sb.start - (cur.start + cur.block_len)
                                           : cur.start - (sb.start + sb.block_len);
                if (gap <= Q.max_block_dist) {
                    if (from_start) cur.block_len = (uint32_t)((sb.start + sb.block_len) - cur.start);
                    else { cur.block_len = (uint32_t)((cur.start + cur.block_len) - sb.start); cur.start = sb.start; }
                    cur.block_counts += sb.block_counts; cur.forward_count += sb.forward_count;
                    cur.reverse_count += sb.reverse_count; cur.canonical_count += sb.canonical_count;
                    cur.non_canonical_count += sb.non_canonical_count; cur.total_covered += sb.total_covered;
                    cur.fwd_covered += sb.fwd_covered; cur.can_covered += sb.can_covered;
                } else { close_block(); cur = sb; }
            }
        }
        open = false;
    };
    bool stop = false;
    const uint32_t ntile = V.t1 - V.t0;
    for (uint32_t ti = 0; ti < ntile && !stop; ++ti) {
        const uint32_t t = from_start ? V.t0 + ti : V.t1 - 1u - ti;
        const uint32_t cnt = V.count(t);
        if (cnt == 0u) continue;
        const u64 off = V.tile_off[t];
        const u64 tile_rel = V.tiles[t].in_off - V.base;
        for (uint32_t b0 = 0; b0 < cnt && !stop; b0 += 64u) {
            const uint32_t nb = cnt - b0 < 64u ? cnt - b0 : 64u;
            const uint32_t idx = from_start ? b0 + lane : cnt - 1u - b0 - lane;        // walk order
            const uint32_t rec = lane < nb ? V.matches[off + idx] : 0u;
            const bool sel = lane < nb && (((rec >> 1) & 1u) != 0u) == from_start;      // forward list from the start, reverse from the end
            u64 rem = __ballot(sel);
            if (rem == 0ull) continue;
            const uint32_t p32 = rec >> 2;                 // tile-relative (a batch lies in one tile), < 2^30
            // predecessor in the walk among the wanted records of this batch: prefix maximum of position + 1
            // (ascending walk) or of ~position (descending walk: the maximum of ~p is the minimum of p), one lane down
            const uint32_t incl = wave_scan_max(sel ? (from_start ? p32 + 1u : ~p32) : 0u);
            const uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x138, 0xf, 0xf, false);   // wave_shr:1
            const uint32_t gap_in = from_start ? p32 - (before - 1u) : ~before - p32;
            const uint32_t first_lane = (uint32_t)__builtin_ctzll(rem);
            const u64 first_pos = tile_rel + (uint32_t)__builtin_amdgcn_readlane((int)p32, (int)first_lane);
            const bool first_head = !open || (from_start ? first_pos - ch.prev : ch.prev - first_pos) > Q.max_match_dist;
            const bool head = sel && (before == 0u ? first_head : gap_in > Q.max_match_dist);
            const u64 heads = __ballot(head), canon = __ballot(sel && (rec & 1u));
            while (rem) {                                  // one step per chain (or per piece of a chain that spans batches)
                const uint32_t l0 = (uint32_t)__builtin_ctzll(rem);
                if ((heads >> l0) & 1ull) {
                    if (open) close_sub();
                    const u64 p0 = tile_rel + (uint32_t)__builtin_amdgcn_readlane((int)p32, (int)l0);
                    const bool in_zone = n <= Q.terminal_limit ? true
                                       : (from_start ? p0 < Q.terminal_limit : p0 >= n - Q.terminal_limit);
                    if (!in_zone) { stop = true; break; }
                    ch.start = p0; ch.end = p0 + Q.k; ch.prev = p0;
                    ch.counts = ch.fwd = ch.canon = ch.cov = ch.fwd_cov = ch.can_cov = 0;
                    open = true;
                }
                const u64 later = l0 < 63u ? heads & ~((2ull << l0) - 1ull) : 0ull;      // heads after l0
                const u64 run = later ? rem & ((1ull << (uint32_t)__builtin_ctzll(later)) - 1ull) : rem;
                const uint32_t nrun = (uint32_t)__popcll(run), ncan = (uint32_t)__popcll(run & canon);
                const uint32_t nfwd = from_start ? nrun : 0u;                              // the list has one orientation
                ch.counts += nrun; ch.fwd += nfwd; ch.canon += ncan;
                ch.cov += nrun * Q.k; ch.fwd_cov += nfwd * Q.k; ch.can_cov += ncan * Q.k;
                const uint32_t last_lane = 63u - (uint32_t)__builtin_clzll(run);
                ch.prev = tile_rel + (uint32_t)__builtin_amdgcn_readlane((int)p32, (int)last_lane);
                if (from_start) ch.end = ch.prev + Q.k; else ch.start = ch.prev;
                rem &= ~run;
            }
        }
    }
    // A clipped view that the walk used up without meeting a chain head outside the terminal zone: the records the
    // walk would have read next are on a neighbour.  Harmless when they cannot matter — the open chain is too far from
    // the clip for any of them to extend it, and the clip lies outside the zone, so the next head ends the walk —
    // else the segment is reported (TS_SEG_F_CONTEXT) and takes the full path.
    if (!stop && (from_start ? V.open_r : V.open_l)) {
        const bool zone_clipped = n > Q.terminal_limit && (from_start ? V.hi_rel < Q.terminal_limit : V.lo_rel > n - Q.terminal_limit);
        const bool chain_clipped = open && (from_start ? ch.prev + Q.max_match_dist >= V.hi_rel : ch.prev <= V.lo_rel + Q.max_match_dist);
        if (zone_clipped || chain_clipped) out_of_context = true;
    }
    if (open) close_sub();
    if (have_cur) close_block();
    return boundary;
}

// the shard's view of segment S (TsShardSegIn): its scanned tiles, clipped where the segment continues on a neighbour
__device__ __forceinline__ SegView seg_view(const TsBlockCallParams &Q, const TsShardSegIn &S) {
    SegView V{Q.tiles, Q.tile_off, Q.tile_stats, Q.matches, S.t0, S.t1, S.in_off,
              !(S.flags & TS_SEG_F_HAS_START), !(S.flags & TS_SEG_F_HAS_END), 0, S.len};
    if (S.t1 > S.t0) {
        V.lo_rel = Q.tiles[S.t0].in_off - S.in_off;
        V.hi_rel = Q.tiles[S.t1 - 1u].in_off - S.in_off + Q.tiles[S.t1 - 1u].own_len;
    }
    return V;
}

__global__ void ts_terminal_blocks(const TsBlockCallParams Q, const TsShardSegIn *segs, uint32_t nseg, u64 *bounds,
                                   TsShardSeg *seg_out) {
    // one workgroup of four waves per segment: all threads add up the segment's tile counts (a 250 Mb
    // contig has ~35 k tiles), then wave 0 walks the forward list from the start while wave 1 walks
    // the reverse list from the end.  A shard (seg_out != nullptr) walks a direction only when it owns that end of
    // the segment; the bounds of the other end are the widest possible, which the receiver checks against what
    // the shard that did walk it reports (shard.cpp: finalize).
    __shared__ u64 part[5][256];
    __shared__ uint32_t walk_flags;
    const uint32_t si = blockIdx.x;
    if (si >= nseg) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const TsShardSegIn S = segs[si];
    const SegView V = seg_view(Q, S);
    const u64 n = S.len;
    u64 total = 0, nfwd = 0, own = 0, own_can = 0, own_fwd = 0;
    for (uint32_t t = V.t0 + threadIdx.x; t < V.t1; t += blockDim.x) {
        const uint4 st = *(const uint4 *)&V.tile_stats[4ull * t];
        total += st.x; nfwd += st.z;
        if (t >= S.o0 && t < S.o1) { own += st.x; own_can += st.y; own_fwd += st.z; }
    }
    part[0][threadIdx.x] = total; part[1][threadIdx.x] = nfwd;
    part[2][threadIdx.x] = own; part[3][threadIdx.x] = own_can; part[4][threadIdx.x] = own_fwd;
    if (threadIdx.x == 0) walk_flags = 0;
    __syncthreads();
    for (uint32_t o = 128; o >= 1; o >>= 1) {
        if (threadIdx.x < o)
            for (int f = 0; f < 5; ++f) part[f][threadIdx.x] += part[f][threadIdx.x + o];
        __syncthreads();
    }
    total = part[0][0]; nfwd = part[1][0];
    uint32_t seq = 0;                                      // blocks are ordered by (direction, seq)
    if (wave == 0) {
        u64 fb = 0;
        bool ooc = false;
        const bool walk = (S.flags & TS_SEG_F_HAS_START) && nfwd >= 2;
        if (walk) fb = terminal_direction(Q, V, S.seg, n, S.abs_pos, true, seq, lane, ooc);
        if (lane == 0) {
            bounds[2ull * si] = fb;
            atomicOr(&walk_flags, (walk ? TS_SEG_F_FWD_WALKED : 0u) | (ooc ? TS_SEG_F_CONTEXT : 0u));
        }
    } else if (wave == 1) {
        u64 rb = n;
        bool ooc = false;
        const bool walk = (S.flags & TS_SEG_F_HAS_END) && total - nfwd >= 2;
        if (walk) rb = terminal_direction(Q, V, S.seg, n, S.abs_pos, false, seq, lane, ooc);
        if (lane == 0) {
            bounds[2ull * si + 1] = total >= 2 ? rb : 0;       // 0 disables the interstitial search
            atomicOr(&walk_flags, (walk ? TS_SEG_F_REV_WALKED : 0u) | (ooc ? TS_SEG_F_CONTEXT : 0u));
        }
    }
    if (!seg_out) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        TsShardSeg o;
        o.fwd_boundary = bounds[2ull * si];
        o.rev_boundary = bounds[2ull * si + 1];
        o.n_matches = part[2][0]; o.n_canonical = part[3][0]; o.n_forward = part[4][0];
        o.seen_matches = total; o.seen_forward = nfwd;
        o.flags = (S.flags & (TS_SEG_F_HAS_START | TS_SEG_F_HAS_END)) | walk_flags;
        o.reserved = 0;
        seg_out[si] = o;
    }
}

// computeBlockLabel, include/teloscope.h:217-222
__device__ char its_label(uint32_t fwd_count, uint32_t counts) {
    const float ratio = ((float)fwd_count * 100.0f) / (float)counts;
    if (ratio > 66.6f) return 'p';
    if (ratio < 33.3f) return 'q';
    return 'b';
}

// A segment view that serves one tile's records from LDS (they were loaded with one coalesced sweep)
// and everything else from global memory: chains rarely leave the tile they start in.
struct TileView {
    SegView V;
    uint32_t tile, cnt;                      // the cached tile and its record count
    u64 tile_rel;                            // segment-relative position of the tile's first base
    const uint32_t *cache;                   // LDS copy of the tile's records (nullptr: not cached)

    __device__ uint32_t count(uint32_t t) const { return t == tile ? cnt : V.count(t); }
    __device__ uint32_t rec(Cursor c) const { return (cache && c.t == tile) ? cache[c.i] : V.rec(c); }
    __device__ u64 pos(Cursor c, uint32_t r) const { return c.t == tile ? tile_rel + (r >> 2) : V.pos(c, r); }
    __device__ bool next(Cursor &c) const {
        if (c.i + 1u < count(c.t)) { ++c.i; return true; }
        for (uint32_t t = c.t + 1u; t < V.t1; ++t)
            if (V.count(t)) { c.t = t; c.i = 0; return true; }
        return false;
    }
    __device__ bool prev(Cursor &c) const {
        if (c.i > 0u) { --c.i; return true; }
        for (uint32_t t = c.t; t > V.t0; --t)
            if (V.count(t - 1u)) { c.t = t - 1u; c.i = V.count(t - 1u) - 1u; return true; }
        return false;
    }
};

#define TS_ITS_CAND  256                     // canonical candidates of a tile kept in LDS (~2 % of its records)
#define TS_ITS_CACHE 1024                    // records of a tile kept in LDS (a tile of 13.5 kb holds ~400 at 3 % density)

__global__ void ts_interstitial_blocks(const TsBlockCallParams Q, const TsShardSegIn *segs, uint32_t seg_base,
                                       const u64 *bounds, uint32_t ntiles, TsShardSeg *seg_out) {
    // one wave per tile, lanes over its records.  A block belongs to the tile its chain STARTS in: a shard runs
    // this over every tile it scanned (a chain that starts in an owned tile may have its first canonical match, the
    // one that walks it, in the context) and emits the blocks that start in an owned tile.
    __shared__ uint32_t cache_all[4][TS_ITS_CACHE];
    __shared__ uint16_t cand_all[4][TS_ITS_CAND];
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t tile = blockIdx.x * (blockDim.x >> 6) + wave;
    if (tile >= ntiles) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t si = Q.tiles[tile].seg - seg_base;
    const u64 fb = bounds[2ull * si], rb = bounds[2ull * si + 1];
    if (rb == 0 || fb >= rb) return;
    if (Q.tile_stats[4u * tile + 1u] == 0u) return;       // no canonical match in the tile: nothing can lead a block
    const TsShardSegIn S = segs[si];
    TileView V{seg_view(Q, S), tile, 0, 0, nullptr};
    V.cnt = V.V.count(tile);
    V.tile_rel = Q.tiles[tile].in_off - V.V.base;
    const uint32_t cnt = V.cnt;
    if (cnt <= TS_ITS_CACHE) {
        const uint32_t *src = Q.matches + Q.tile_off[tile];
        for (uint32_t i = lane; i < cnt; i += 64u) cache_all[wave][i] = src[i];
        V.cache = cache_all[wave];
        __builtin_amdgcn_wave_barrier();
    }
    // Only canonical matches inside the interstitial range can lead a block, and they are ~2 % of the records:
    // their indices are compacted first (ballot + rank) so that the chain walks below run on dense lanes
    // instead of one or two lanes per 64 records.
    uint32_t ncand = 0;
    bool compacted = cnt <= TS_ITS_CACHE;
    if (compacted) {
        for (uint32_t i0 = 0; i0 < cnt; i0 += 64u) {
            const uint32_t i = i0 + lane;
            bool cand = false;
            if (i < cnt) {
                const uint32_t r = V.cache[i];
                const u64 p = V.tile_rel + (r >> 2);
                cand = (r & 1u) && p >= fb && p < rb;
            }
            const u64 m = __ballot(cand);
            const uint32_t slot = ncand + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if (cand && slot < TS_ITS_CAND) cand_all[wave][slot] = (uint16_t)i;
            ncand += (uint32_t)__popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
        if (ncand > TS_ITS_CAND) compacted = false;        // a tile full of canonical repeats: every record is tried
    }
    const uint32_t nwork = compacted ? ncand : cnt;
    bool ooc = false;
    for (uint32_t w = lane; w < nwork; w += 64u) {
        const uint32_t i = compacted ? (uint32_t)cand_all[wave][w] : w;
        Cursor c{tile, i};
        const uint32_t r = V.rec(c);
        if (!(r & 1u)) continue;                           // only canonical matches can lead a block
        const u64 p = V.pos(c, r);
        if (p < fb || p >= rb) continue;
        // walk left: not the leader if an earlier canonical match is in the same chain
        Cursor s = c; u64 sp = p; bool leader = true;
        for (Cursor q = c;;) {
            if (!V.prev(q)) {
                // The view's left edge, with the segment going on behind it: the chain's start is not known here.  For a
                // candidate in a context tile that is its owner's business (the left neighbour sees the same records and
                // more); for one in an owned tile the chain spans the whole left context — reported.
                if (V.V.open_l && sp <= V.V.lo_rel + Q.max_match_dist && V.V.lo_rel > fb) {
                    leader = false;
                    if (tile >= S.o0 && tile < S.o1) ooc = true;
                }
                break;
            }
            const uint32_t rq = V.rec(q);
            const u64 pq = V.pos(q, rq);
            if (pq < fb || sp - pq > Q.max_match_dist) break;
            if (rq & 1u) { leader = false; break; }
            s = q; sp = pq;
        }
        if (!leader) continue;
        if (s.t < S.o0 || s.t >= S.o1) continue;          // the chain starts in a context tile: its owner emits the block
        // s is the chain's first match: walk the whole chain
        Chain ch;
        ch.begin(sp, V.rec(s), Q.k);
        bool ended = false;
        for (Cursor q = s; V.next(q);) {
            const uint32_t rq = V.rec(q);
            const u64 pq = V.pos(q, rq);
            if (pq >= rb || pq - ch.prev > Q.max_match_dist) { ended = true; break; }
            ch.end = pq + Q.k;
            ch.add(pq, rq, Q.k);
        }
        if (!ended && V.V.open_r && ch.prev + Q.max_match_dist >= V.V.hi_rel && V.V.hi_rel < rb) { ooc = true; continue; }
        const uint32_t blen = (uint32_t)(ch.end - ch.start);
        const char lab = its_label(ch.fwd, ch.counts);
        if (blen >= Q.its_min_len && ch.canon >= 4u && !(lab == 'b' && ch.fwd < 2u && (ch.counts - ch.fwd) < 2u)) {
            TsDevBlock b; ch.to_block(b);
            b.block_label = lab;
            emit_block(Q, b, S.seg, 2u, 0u, S.abs_pos);
        }
    }
    if (seg_out && __ballot(ooc) != 0ull && lane == 0) atomicOr(&seg_out[si].flags, TS_SEG_F_CONTEXT);
}

}  // namespace

int ts_k_launch_block_call(const TsBlockCallParams *Q, const TsShardSegIn *segs, uint32_t nseg, uint32_t seg_base,
                           uint32_t ntiles, unsigned long long *bounds, TsShardSeg *seg_out, int with_its, void *stream) {
    if (nseg == 0) return 0;
    hipLaunchKernelGGL(ts_terminal_blocks, dim3(nseg), dim3(256), 0, (hipStream_t)stream, *Q, segs, nseg, bounds, seg_out);
    if (with_its && ntiles)
        hipLaunchKernelGGL(ts_interstitial_blocks, dim3((ntiles + 3u) / 4u), dim3(256), 0, (hipStream_t)stream, *Q,
                           segs, seg_base, (const u64 *)bounds, ntiles, seg_out);
    return (int)hipGetLastError();
}
