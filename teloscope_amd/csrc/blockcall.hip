// blockcall.hip — telomere block calling on the device, from the packed match stream (gfx950).
//
// Device form of Teloscope::getTerminalBlocks (src/teloscope.cpp:29-176) and
// getInterstitialBlocks (:179-256), the O(matches) step that follows the scan inside
// scanSegment (:642-657).  With it only blocks (a few per segment) have to leave the GPU instead
// of the whole match stream.
//
//   ts_segment_sums         per segment: record totals over its scanned and over its owned tiles (atomics, no LDS)
//   ts_terminal_blocks      one wave per segment and direction: the reference's two-phase walk (chain matches
//                           <= -k apart inside the terminal zone, keep dense canonical sub-blocks,
//                           merge sub-blocks <= -d apart, keep >= -l) over the forward list from
//                           the start and the reverse list from the end, 64 records per step in parallel
//                           (prefix maximum for the predecessor, ballot of chain heads, sub-blocks screened lane-parallel,
//                           record rows prefetched eight ahead); emits the blocks and the two boundaries that fence the
//                           interstitial search.
//   ts_interstitial_blocks  one wave per tile, 64 records per step: a ballot of chain heads cuts the records into
//                           chains, a running count of canonical matches SCREENS them (a block needs four), the
//                           few chains that pass are listed; for a shard the tile's writer-visible records leave in
//                           the same pass.
//   ts_interstitial_evaluate one wave per listed chain: the exact walk and the reference's filters.
//
// None of these kernels uses LDS: they run beside the next scan, whose persistent workgroups hold every CU's LDS.
//
// Records are addressed through the tile directory {tile_off, tile_stats}; tiles of one segment
// are consecutive and position-ordered, so prev/next step across tile boundaries.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ts_internal.h"

// ONE-WAVE WORKGROUPS.  These kernels run beside the persistent scan kernel of the next batch, whose two workgroups of ten
// waves leave a CU's SIMDs with 6, 6, 4 and 4 waves of 80 registers: 32 free registers on two of the four SIMDs.  A
// workgroup's waves go round the SIMDs, so a 256-thread workgroup of a kernel with more than 32 registers per thread cannot
// be placed on ANY CU while the scan is resident — kernels launched with 256 threads ended the microsecond the scan beside
// them did (profiles/r04/pack_beside_scan_trace.txt) — while a single wave goes where there is room.  None of these kernels
// uses LDS or a barrier: the workgroup size is free.  (Raising their wave priority above the scan's changed nothing.)
constexpr unsigned kSideWg = 64;

namespace {

typedef unsigned long long u64;

// Two record formats.  The tiled kernel's (uniform length Q.k): position << 2 | forward << 1 | canonical.  The general
// kernels' (generic.hip; Q.gen_lens != 0: up to eight pattern lengths, six bits each): position << 5 | length index << 2 |
// canonical << 1 | forward.  Every load goes through rec_norm, which hands the code below the first form, and rec_len.
// record i of the stream `matches` (the tiled kernel's may be 16 bits each: Q.rec16 — uniform, so the branch is scalar)
__device__ __forceinline__ uint32_t rec_at(const TsBlockCallParams &Q, const uint32_t *matches, u64 i) {
    return Q.rec16 ? (uint32_t)((const uint16_t *)matches)[i] : matches[i];
}
// MODE 1 (the instantiations the general path launches for what MODE 0 does not take): additionally the WIDE form's records
// (position << 8 | length index << 2 | canonical << 1 | forward, lengths in Q.wide_len) and streams in the reference's push order
// (Q.unordered: generic.hip, ts_general_compact_push).  MODE 0 is the code the tiled kernel's batches and shards run.
template <int MODE>
__device__ __forceinline__ uint32_t rec_norm(const TsBlockCallParams &Q, uint32_t raw) {
    if (MODE == 1 && Q.wide) return ((raw >> 8) << 2) | ((raw & 1u) << 1) | ((raw >> 1) & 1u);
    return Q.gen_lens ? ((raw >> 5) << 2) | ((raw & 1u) << 1) | ((raw >> 1) & 1u) : raw;
}
// (wide: lane i of the wave holds the length of index i — wlen_lane, loaded once per wave — and a record's length is a shuffle,
// not a load behind the load of the record: the walks are chains of latencies; call it from uniform control flow)
template <int MODE>
__device__ __forceinline__ uint32_t rec_len(const TsBlockCallParams &Q, uint32_t raw, uint32_t wlen_lane) {
    if (MODE == 1 && Q.wide) return (uint32_t)__shfl((int)wlen_lane, (int)((raw >> 2) & 63u));
    return Q.gen_lens ? (uint32_t)(Q.gen_lens >> (6u * ((raw >> 2) & 7u))) & 63u : Q.k;
}
// sum of v over the lanes of `mask` (wave-uniform result)
__device__ __forceinline__ uint32_t masked_sum(uint32_t v, u64 mask) {
    uint32_t x = ((mask >> (threadIdx.x & 63u)) & 1ull) ? v : 0u;
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

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
};

struct Chain {                               // running chain of matches (startNewBlock / extend)
    u64 start, end, prev;
    uint32_t counts, fwd, canon, cov, fwd_cov, can_cov;
    __device__ void to_block(TsDevBlock &b) const {
        b.start = start; b.block_len = (uint32_t)(end - start); b.block_counts = counts;
        b.forward_count = fwd; b.reverse_count = counts - fwd; b.canonical_count = canon;
        b.non_canonical_count = counts - canon; b.total_covered = cov; b.fwd_covered = fwd_cov;
        b.can_covered = can_cov; b.has_valid_or = 1; b.is_longest = 0; b.block_label = 0; b.reserved = 0;
    }
};

__device__ __forceinline__ void emit_block(const TsBlockCallParams &Q, TsDevBlock &b, uint32_t seg, uint32_t kind, uint32_t seq,
                           u64 abs_pos, uint32_t seq_hi = 0u) {
    const uint32_t slot = atomicAdd(Q.n_blocks, 1u);
    if (slot >= Q.block_cap) return;                       // overflow: the host sees n_blocks > cap
    b.start += abs_pos;
    b.seg = seg; b.kind = kind; b.seq = seq; b.pad = seq_hi;
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
// (forced inline: as a function of its own it took the kernel's parameters by reference — the whole parameter block went to scratch,
// every pointer out of it became a flat address, and the eight rows it fetches ahead were waited for behind the first scratch reload)
template <int MODE>
__device__ __forceinline__ u64 terminal_direction(const TsBlockCallParams &Q, const SegView &V, uint32_t seg, u64 n, u64 abs_pos,
                                  bool from_start, uint32_t &seq, uint32_t lane, bool &out_of_context) {
    u64 boundary = from_start ? 0 : n;                     // segment-relative
    const uint32_t wlen_lane = (MODE == 1 && Q.wide) ? Q.wide_len[lane] : 0u;     // (the table holds 64 entries)
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
    // The walk is one chain of dependent steps, and what it waits for is memory: the directory entries of the next tile are
    // requested while this one is walked, and eight batches of 64 records together (a telomere is thirty batches long).
    auto tile_at = [&](uint32_t ti) { return from_start ? V.t0 + ti : V.t1 - 1u - ti; };
    uint32_t n_cnt = ntile ? V.count(tile_at(0)) : 0u;
    u64 n_off = ntile ? V.tile_off[tile_at(0)] : 0ull, n_in = ntile ? V.tiles[tile_at(0)].in_off : 0ull;
    for (uint32_t ti = 0; ti < ntile && !stop; ++ti) {
        const uint32_t cnt = n_cnt;
        const u64 off = n_off;
        const u64 tile_rel = n_in - V.base;
        // the terminal zone's edge, relative to the tile: a head is in the zone iff p32 < zone32 (walk from the start) or
        // p32 >= zone32 (walk from the end); a segment no longer than the limit is zone as a whole
        uint32_t zone32;
        {
            const u64 edge = from_start ? (u64)Q.terminal_limit : n - Q.terminal_limit;
            if (n <= Q.terminal_limit) zone32 = from_start ? 0xFFFFFFFFu : 0u;
            else zone32 = edge <= tile_rel ? 0u : (edge - tile_rel > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(edge - tile_rel));
        }
        if (ti + 1u < ntile) {
            const uint32_t t2 = tile_at(ti + 1u);
            n_cnt = V.count(t2); n_off = V.tile_off[t2]; n_in = V.tiles[t2].in_off;
        }
        if (cnt == 0u) continue;
        for (uint32_t b4 = 0; b4 < cnt && !stop; b4 += 512u) {
          uint32_t recs[8];
#pragma unroll
          for (uint32_t q = 0; q < 8u; ++q) {
              const uint32_t bq = b4 + 64u * q;
              const uint32_t idx = from_start ? bq + lane : cnt - 1u - bq - lane;      // walk order
              recs[q] = bq + lane < cnt ? rec_at(Q, V.matches, off + idx) : 0u;      // (raw: normalised where a row is taken up)
          }
#pragma unroll
          for (uint32_t q = 0; q < 8u; ++q) {
            const uint32_t b0 = b4 + 64u * q;
            if (b0 >= cnt || stop) break;
            const uint32_t nb = cnt - b0 < 64u ? cnt - b0 : 64u;
            const uint32_t lenv = rec_len<MODE>(Q, recs[q], wlen_lane);       // this lane's match length (Q.k for the tiled kernel's records)
            const uint32_t rec = rec_norm<MODE>(Q, recs[q]);
            const bool uni = Q.gen_lens == 0ull;             // uniform length: covered bases = records x k
            const bool sel = lane < nb && (((rec >> 1) & 1u) != 0u) == from_start;      // forward list from the start, reverse from the end
            u64 rem = __ballot(sel);
            if (rem == 0ull) continue;
            const uint32_t p32 = rec >> 2;                 // tile-relative (a batch lies in one tile), < 2^30
            // predecessor in the walk among the wanted records of this batch: prefix maximum of position + 1
            // (ascending walk) or of ~position (descending walk: the maximum of ~p is the minimum of p), one lane down
            uint32_t before;
            if (MODE == 1 && Q.unordered) {
                // a stream in push order: the predecessor is the wanted record before this one AS THE STREAM LIES (the nearest
                // wanted lane below), whatever its position — one that lies behind this record's makes the gap wrap, and the
                // chain breaks there, as the reference's unsigned subtraction does (src/teloscope.cpp:60-75)
                const u64 below_me = rem & (lane ? (1ull << lane) - 1ull : 0ull);
                const int pl = below_me ? 63 - (int)__builtin_clzll(below_me) : 0;
                const uint32_t v = (uint32_t)__shfl((int)(from_start ? p32 + 1u : ~p32), pl);
                before = below_me ? v : 0u;
            } else {
                const uint32_t incl = wave_scan_max(sel ? (from_start ? p32 + 1u : ~p32) : 0u);
                before = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x138, 0xf, 0xf, false);   // wave_shr:1
            }
            const uint32_t gap_in = from_start ? p32 - (before - 1u) : ~before - p32;
            const uint32_t first_lane = (uint32_t)__builtin_ctzll(rem);
            const u64 first_pos = tile_rel + (uint32_t)__builtin_amdgcn_readlane((int)p32, (int)first_lane);
            const bool first_head = !open || (from_start ? first_pos - ch.prev : ch.prev - first_pos) > Q.max_match_dist;
            const bool head = sel && (before == 0u ? first_head : gap_in > Q.max_match_dist);
            const u64 heads = __ballot(head), canon = __ballot(sel && (rec & 1u));
            // The walk ends at the first head outside the terminal zone; ahead of it every head closes the chain before it.
            // In the zone's random stretch a batch holds ~15 chains and a sub-block is one in a hundred of them (it needs
            // min_block_counts records and a canonical one): every head lane counts its own chain, and the scalar state
            // machine steps only through the chains that pass those two tests — the others leave no trace in it.
            const u64 Z = __ballot(head && !(from_start ? p32 < zone32 : p32 >= zone32));
            const uint32_t stop_lane = Z ? (uint32_t)__builtin_ctzll(Z) : 64u;
            const u64 live = stop_lane < 64u ? (1ull << stop_lane) - 1ull : ~0ull;
            const u64 h_all = heads & (stop_lane < 64u ? live | (1ull << stop_lane) : ~0ull), h_live = heads & live;
            rem &= live;
            const uint32_t first = h_all ? (uint32_t)__builtin_ctzll(h_all) : 64u;
            const u64 pre = rem & (first < 64u ? (1ull << first) - 1ull : ~0ull);
            if (open && pre) {                             // the chain carried into the batch takes the records ahead of the first head
                const uint32_t nrun = (uint32_t)__popcll(pre), ncan = (uint32_t)__popcll(pre & canon);
                const uint32_t nfwd = from_start ? nrun : 0u;                              // the list has one orientation
                ch.counts += nrun; ch.fwd += nfwd; ch.canon += ncan;
                const uint32_t cov_pre = uni ? nrun * Q.k : masked_sum(lenv, pre), can_pre = uni ? ncan * Q.k : masked_sum(lenv, pre & canon);
                ch.cov += cov_pre; ch.fwd_cov += from_start ? cov_pre : 0u; ch.can_cov += can_pre;
                const int last_pre = 63 - (int)__builtin_clzll(pre);
                ch.prev = tile_rel + (uint32_t)__builtin_amdgcn_readlane((int)p32, last_pre);
                if (from_start) ch.end = ch.prev + (uint32_t)__builtin_amdgcn_readlane((int)lenv, last_pre); else ch.start = ch.prev;
            }
            if (open && h_all) close_sub();
            if (h_live) {
                const u64 above = lane < 63u ? h_all >> (lane + 1u) : 0ull;
                const uint32_t next = above ? lane + 1u + (uint32_t)__builtin_ctzll(above) : 64u;   // the next head's lane
                const u64 lo_me = lane ? (1ull << lane) - 1ull : 0ull;
                const u64 mine = rem & (next < 64u ? (1ull << next) - 1ull : ~0ull) & ~lo_me;
                const uint32_t nrun_l = (uint32_t)__popcll(mine), ncan_l = (uint32_t)__popcll(mine & canon);
                const uint32_t last_l = mine ? 63u - (uint32_t)__builtin_clzll(mine) : lane;
                const uint32_t p_last_l = (uint32_t)__shfl((int)p32, (int)last_l);
                const uint32_t len_last_l = (uint32_t)__shfl((int)lenv, (int)last_l);           // length of the chain's last record in walk order
                u64 sub = __ballot(((h_live >> lane) & 1ull) && next < 64u && nrun_l >= Q.min_block_counts && ncan_l > 0u);
                while (sub) {                              // the complete chains that may be sub-blocks, in walk order
                    const int i = (int)__builtin_ctzll(sub);
                    sub &= sub - 1ull;
                    const u64 p_head = tile_rel + (uint32_t)__builtin_amdgcn_readlane((int)p32, i);
                    const u64 p_end = tile_rel + (uint32_t)__builtin_amdgcn_readlane((int)p_last_l, i);
                    const uint32_t nrun = (uint32_t)__builtin_amdgcn_readlane((int)nrun_l, i);
                    const uint32_t ncan = (uint32_t)__builtin_amdgcn_readlane((int)ncan_l, i);
                    const uint32_t nfwd = from_start ? nrun : 0u;
                    ch.prev = p_end;
                    // (a chain's end = its last record in POSITION order + that record's length: the walk's last record from the
                    // start, its first from the end)
                    if (from_start) { ch.start = p_head; ch.end = p_end + (uint32_t)__builtin_amdgcn_readlane((int)len_last_l, i); }
                    else { ch.start = p_end; ch.end = p_head + (uint32_t)__builtin_amdgcn_readlane((int)lenv, i); }
                    ch.counts = nrun; ch.fwd = nfwd; ch.canon = ncan;
                    if (uni) { ch.cov = nrun * Q.k; ch.can_cov = ncan * Q.k; }
                    else {
                        const u64 mine_i = ((u64)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(mine >> 32), i) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)mine, i);
                        ch.cov = masked_sum(lenv, mine_i); ch.can_cov = masked_sum(lenv, mine_i & canon);
                    }
                    ch.fwd_cov = from_start ? ch.cov : 0u;
                    open = true;
                    close_sub();
                }
                if (stop_lane == 64u) {                    // the last head's chain stays open
                    const uint32_t top = 63u - (uint32_t)__builtin_clzll(h_live);
                    const u64 tail = rem & ~(top ? (1ull << top) - 1ull : 0ull);
                    const uint32_t nrun = (uint32_t)__popcll(tail), ncan = (uint32_t)__popcll(tail & canon);
                    const uint32_t nfwd = from_start ? nrun : 0u;
                    const u64 p_head = tile_rel + (uint32_t)__builtin_amdgcn_readlane((int)p32, (int)top);
                    const int last_tail = 63 - (int)__builtin_clzll(tail);
                    ch.prev = tile_rel + (uint32_t)__builtin_amdgcn_readlane((int)p32, last_tail);
                    if (from_start) { ch.start = p_head; ch.end = ch.prev + (uint32_t)__builtin_amdgcn_readlane((int)lenv, last_tail); }
                    else { ch.start = ch.prev; ch.end = p_head + (uint32_t)__builtin_amdgcn_readlane((int)lenv, (int)top); }
                    ch.counts = nrun; ch.fwd = nfwd; ch.canon = ncan;
                    ch.cov = uni ? nrun * Q.k : masked_sum(lenv, tail); ch.can_cov = uni ? ncan * Q.k : masked_sum(lenv, tail & canon);
                    ch.fwd_cov = from_start ? ch.cov : 0u;
                    open = true;
                }
            }
            if (stop_lane < 64u) stop = true;
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
    return SegView{Q.tiles, Q.tile_off, Q.tile_stats, Q.matches, S.t0, S.t1, S.in_off,
                   !(S.flags & TS_SEG_F_HAS_START), !(S.flags & TS_SEG_F_HAS_END), S.lo_rel, S.hi_rel};
}

// Per segment: match / forward counts over all its tiles the batch scanned, and match / canonical / forward counts over its
// OWNED tiles — five u64 per segment: in `sums` (zero at launch), or, for a shard, straight in the segment's entry of the message
// (seg_out, zeroed with the header).  One thread per tile; a wave whose tiles all belong to one segment (nearly every wave) adds
// up first and issues five atomics.  No LDS (see exchange.hip).  Called by every thread of the wave (t >= ntiles: nothing to add).
__device__ __forceinline__ void add_segment_sums(const TsBlockCallParams &Q, const TsShardSegIn *segs, uint32_t seg_base, uint32_t t,
                                                 uint32_t ntiles, u64 *sums, TsShardSeg *seg_out) {
    uint32_t si = 0xFFFFFFFFu;
    u64 v[5] = {0, 0, 0, 0, 0};
    if (t < ntiles) {
        si = Q.tiles[t].seg - seg_base;
        const uint4 st = *(const uint4 *)&Q.tile_stats[4ull * t];
        const TsShardSegIn S = segs[si];
        if (t >= S.t0 && t < S.t1) { v[0] = st.x; v[1] = st.z; }
        if (t >= S.o0 && t < S.o1) { v[2] = st.x; v[3] = st.y; v[4] = st.z; }
    }
    // where field f of segment s goes: sums[5 s + f], or the u64 of the segment's TsShardSeg that holds it ({fwd_boundary, rev_boundary,
    // n_matches, n_canonical, n_forward, seen_matches, seen_forward}: f = 0, 1 are the seen counts, 2 .. 4 the owned ones)
    u64 *const base = seg_out ? (u64 *)seg_out : sums;
    const uint32_t stride = seg_out ? (uint32_t)(sizeof(TsShardSeg) / 8u) : 5u;
    auto slot = [&](int f) -> uint32_t { return seg_out ? (f < 2 ? 5u + (uint32_t)f : (uint32_t)f) : (uint32_t)f; };
    const uint32_t s0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)si);
    if (__ballot(si != s0) == 0ull) {
        if (s0 != 0xFFFFFFFFu) {
            for (int f = 0; f < 5; ++f) {
                u64 x = v[f];
                for (int o = 32; o >= 1; o >>= 1) {
                    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)x, o), hi = (uint32_t)__shfl_xor((int)(uint32_t)(x >> 32), o);
                    x += ((u64)hi << 32) | lo;
                }
                if ((threadIdx.x & 63u) == 0u && x) atomicAdd(base + (u64)s0 * stride + slot(f), x);
            }
        }
    } else if (si != 0xFFFFFFFFu) {
        for (int f = 0; f < 5; ++f)
            if (v[f]) atomicAdd(base + (u64)si * stride + slot(f), v[f]);
    }
}

__global__ __launch_bounds__(kSideWg)
void ts_segment_sums(const TsBlockCallParams Q, const TsShardSegIn *segs, uint32_t seg_base, uint32_t ntiles, u64 *sums, TsShardSeg *seg_out) {
    add_segment_sums(Q, segs, seg_base, blockIdx.x * blockDim.x + threadIdx.x, ntiles, sums, seg_out);
}

// Does the segment's view hold at least two records of a kind (which: 0 any, 1 forward, 2 reverse)?  The gates of
// scanSegment's block calling (src/teloscope.cpp:646-653: fwdMatches.size() >= 2, revMatches.size() >= 2, allMatches.size()
// >= 2), answered by the wave that needs it from the tile directory — 64 tiles per step, from the end the walk starts at,
// until it knows: one step on any sequence with matches in it.  (Round 4 had a kernel add up every segment's totals first:
// 28 us on the walks' critical path.)
__device__ __forceinline__ bool view_has_two(const TsBlockCallParams &Q, const SegView &V, int which, bool from_end, uint32_t lane) {
    const uint32_t nt = V.t1 - V.t0;
    uint32_t acc = 0;
    for (uint32_t b0 = 0; b0 < nt && acc < 2u; b0 += 64u) {
        uint32_t v = 0;
        if (b0 + lane < nt) {
            const uint32_t t = from_end ? V.t1 - 1u - (b0 + lane) : V.t0 + b0 + lane;
            const uint4 st = *(const uint4 *)&Q.tile_stats[4ull * t];
            v = which == 0 ? st.x : which == 1 ? st.z : st.x - st.z;
        }
        const u64 two = __ballot(v >= 2u), one = __ballot(v == 1u);
        acc = two ? 2u : acc + (uint32_t)__popcll(one);
    }
    return acc >= 2u;
}

#ifndef TS_TERMINAL_PRIO
#define TS_TERMINAL_PRIO 0
#endif
template <int MODE>
__global__ __launch_bounds__(kSideWg)
void ts_terminal_blocks(const TsBlockCallParams Q, const TsShardSegIn *segs, uint32_t nseg, u64 *bounds, TsShardSeg *seg_out) {
    // two independent waves per segment: wave 0 walks the forward list from the start, wave 1 the reverse list from the end.
    // A shard (seg_out != nullptr) walks a direction only when it owns that end of the segment; the bounds of the other end
    // are the widest possible, which the receiver checks against what the shard that did walk it reports (shard.cpp:
    // finalize).  No LDS, no barrier: the waves share nothing (see exchange.hip for why that matters).
    // (one-wave workgroups, two per segment: see kSideWg; blockIdx is uniform, so the walk's state machine — ballots, chain
    // state, the branch on the direction — stays in scalar registers)
    // (the per-segment counts of a shard's message are added up elsewhere: ts_chain_screen / ts_segment_sums)
    const uint32_t si = blockIdx.x >> 1;
    if (si >= nseg) return;
#if TS_TERMINAL_PRIO
    __builtin_amdgcn_s_setprio(TS_TERMINAL_PRIO);
#endif
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x & 1u;
    const TsShardSegIn S = segs[si];
    const SegView V = seg_view(Q, S);
    const u64 n = S.len;
    uint32_t seq = 0;                                      // blocks are ordered by (direction, seq)
    if (wave == 0) {
        u64 fb = 0;
        bool ooc = false;
        const bool walk = (S.flags & TS_SEG_F_HAS_START) && view_has_two(Q, V, 1, false, lane);
        if (walk) fb = terminal_direction<MODE>(Q, V, S.seg, n, S.abs_pos, true, seq, lane, ooc);
        if (lane == 0) {
            bounds[2ull * si] = fb;
            if (seg_out) {
                seg_out[si].fwd_boundary = fb;
                atomicOr(&seg_out[si].flags, (S.flags & (TS_SEG_F_HAS_START | TS_SEG_F_HAS_END)) | (walk ? TS_SEG_F_FWD_WALKED : 0u) | (ooc ? TS_SEG_F_CONTEXT : 0u));
            }
        }
    } else {
        u64 rb = n;
        bool ooc = false;
        const bool two_rev = view_has_two(Q, V, 2, true, lane);
        const bool walk = (S.flags & TS_SEG_F_HAS_END) && two_rev;
        if (walk) rb = terminal_direction<MODE>(Q, V, S.seg, n, S.abs_pos, false, seq, lane, ooc);
        const bool two_any = two_rev || view_has_two(Q, V, 0, true, lane);
        if (lane == 0) {
            const u64 rb_out = two_any ? rb : 0;           // 0 disables the interstitial search
            bounds[2ull * si + 1] = rb_out;
            if (seg_out) {
                seg_out[si].rev_boundary = rb_out;
                atomicOr(&seg_out[si].flags, (walk ? TS_SEG_F_REV_WALKED : 0u) | (ooc ? TS_SEG_F_CONTEXT : 0u));
            }
        }
    }
}

// computeBlockLabel, include/teloscope.h:217-222
__device__ char its_label(uint32_t fwd_count, uint32_t counts) {
    const float ratio = ((float)fwd_count * 100.0f) / (float)counts;
    if (ratio > 66.6f) return 'p';
    if (ratio < 33.3f) return 'q';
    return 'b';
}

// The value of the lane below (lane 0: 0) by DPP wave_shr:1.  The empty asm keeps it a v_mov_b32_dpp: folded into the
// subtraction that follows (v_subrev_u32_dpp v, x, x wave_shr:1, what the DPP combiner makes of it) it came back wrong on gfx950.
__device__ __forceinline__ uint32_t lane_below(uint32_t v) {
    uint32_t r = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false);
    asm volatile("" : "+v"(r));
    return r;
}

__device__ __forceinline__ u64 low_bits(uint32_t n) { return n >= 64u ? ~0ull : ((1ull << n) - 1ull); }

// The chain that starts at record i0 of tile t0, walked record by record (64 per step) to its end — the first record more
// than -k behind its predecessor, the first at or behind rb, or the end of the view — and, if it passes the reference's
// filters (src/teloscope.cpp:206-233), its block.  Run by ts_interstitial_evaluate for the few chains the screening
// kernel lists as holding four canonical matches; whole wave, lane 0 writes.  (Called from inside the screening kernel it
// cost that kernel its registers: 99 VGPRs and a scratch frame for the call, 0.95 ms instead of 0.44.)
template <int MODE>
__device__ __forceinline__ void its_evaluate(const TsBlockCallParams &Q, const TsShardSegIn &S, uint32_t t0, uint32_t i0, u64 rb) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wlen_lane = (MODE == 1 && Q.wide) ? Q.wide_len[lane] : 0u;
    u64 start = 0, prev = 0;
    uint32_t counts = 0, fwd = 0, canon = 0;
    uint32_t last_len = Q.k, cov = 0, fwd_cov = 0, can_cov = 0;       // (general records: length of the chain's last record, covered bases)
    bool first = true, closed = false;
    for (uint32_t t = t0; t < S.t1 && !closed; ++t) {
        const uint32_t cnt = Q.tile_stats[4u * t];
        if (cnt == 0u) continue;
        const u64 rel = Q.tiles[t].in_off - S.in_off;
        const u64 src = Q.tile_off[t];                     // (index of the tile's first record: rec_at reads either width)
        const uint32_t rb32 = rb <= rel ? 0u : (rb - rel > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)(rb - rel));
        for (uint32_t b0 = (t == t0 ? i0 : 0u); b0 < cnt && !closed; b0 += 64u) {
            const uint32_t nvalid = cnt - b0 < 64u ? cnt - b0 : 64u;
            const u64 VALID = low_bits(nvalid);
            const uint32_t raw = lane < nvalid ? rec_at(Q, Q.matches, src + b0 + lane) : 0u;
            const uint32_t r = rec_norm<MODE>(Q, raw), lenv = rec_len<MODE>(Q, raw, wlen_lane);
            const uint32_t p32 = r >> 2;
            const uint32_t below = lane_below(p32);
            // records that end the chain: out of range, or too far behind their predecessor (lane 0: the last record of the
            // step before; the chain's own first record never ends it)
            u64 E = (__ballot(p32 >= rb32) | __ballot(p32 - below > Q.max_match_dist)) & VALID & ~1ull;
            const u64 p0 = rel + (uint32_t)__builtin_amdgcn_readfirstlane((int)p32);
            if (!first && ((uint32_t)__builtin_amdgcn_readfirstlane((int)p32) >= rb32 || p0 - prev > Q.max_match_dist)) E |= 1ull;
            if (first && (uint32_t)__builtin_amdgcn_readfirstlane((int)p32) >= rb32) E |= 1ull;
            const uint32_t stop = E ? (uint32_t)__builtin_ctzll(E) : nvalid;
            const u64 take = low_bits(stop) & VALID;
            if (take) {
                if (first) { start = p0; first = false; }
                counts += (uint32_t)__popcll(take);
                fwd += (uint32_t)__popcll(__ballot((r & 2u) != 0u) & take);
                canon += (uint32_t)__popcll(__ballot((r & 1u) != 0u) & take);
                const int last_take = 63 - (int)__builtin_clzll(take);
                prev = rel + (uint32_t)__builtin_amdgcn_readlane((int)p32, last_take);
                last_len = (uint32_t)__builtin_amdgcn_readlane((int)lenv, last_take);
                if (Q.gen_lens) {                           // covered bases: the records' own lengths
                    cov += masked_sum(lenv, take);
                    fwd_cov += masked_sum(lenv, __ballot((r & 2u) != 0u) & take);
                    can_cov += masked_sum(lenv, __ballot((r & 1u) != 0u) & take);
                }
            }
            if (E) closed = true;
        }
    }
    if (first) return;
    const uint32_t blen = (uint32_t)(prev + last_len - start);
    if (canon < 4u || blen < Q.its_min_len) return;
    const char lab = its_label(fwd, counts);
    if (lab == 'b' && fwd < 2u && (counts - fwd) < 2u) return;
    if (lane != 0u) return;
    TsDevBlock b;
    b.start = start; b.block_len = blen; b.block_counts = counts;
    b.forward_count = fwd; b.reverse_count = counts - fwd; b.canonical_count = canon;
    b.non_canonical_count = counts - canon;
    b.total_covered = Q.gen_lens ? cov : counts * Q.k; b.fwd_covered = Q.gen_lens ? fwd_cov : fwd * Q.k;
    b.can_covered = Q.gen_lens ? can_cov : canon * Q.k; b.has_valid_or = 1; b.is_longest = 0; b.block_label = lab; b.reserved = 0;
    // (MODE 1: the stream index of the chain's first record orders the blocks as the reference's walk emits them — over a stream in
    // push order a later chain may start ahead of an earlier one, or at the same position)
    const u64 order = MODE == 1 ? Q.tile_off[t0] + i0 : 0ull;
    emit_block(Q, b, S.seg, 2u, (uint32_t)order, S.abs_pos, (uint32_t)(order >> 32));
}

// getInterstitialBlocks (src/teloscope.cpp:179-256) over the packed match stream, and — for a shard — the visible
// match records of the tile on the way (the same records are in registers).
//
// One wave per tile, 64 records per step, no LDS.  A record opens a chain iff it is the first in [fwdBoundary,
// revBoundary) or lies more than -k behind its predecessor; a ballot of those heads cuts the 64 records into chains.
// Almost no chain holds the four canonical matches a block needs, so the pass only SCREENS: all it keeps of the open
// chain is where it starts and how many canonical matches the stream held before it (a running count); a chain that
// closes with four or more is walked again, exactly, by its_evaluate — a few hundred times per genome.  The kernel is
// bound by scalar issue (one scalar ALU per CU), and this is what a step costs least: one ballot of gap tests, one of
// canonical flags, a dozen scalar bit operations.  A block belongs to the tile its chain STARTS in: records ahead of a
// tile's first head belong to a chain of an earlier tile and are skipped; the chain that is open when the tile's
// records end is followed (counting canonical matches) through the tiles behind until a head closes it.
//
// Measured on the 91.5 M records of configs[1]: profiles/r03/its_kernel_variants.txt.
template <int MODE>
__device__ __forceinline__ void its_tile(const TsBlockCallParams &Q, const TsShardSegIn *segs, uint32_t seg_base,
                                         const u64 *bounds, uint32_t ntiles, TsShardSeg *seg_out, const TsVisibleOut &W,
                                         const uint32_t tile) {
    const uint32_t lane = threadIdx.x & 63u;
    // A wave's time is also a chain of memory round trips: everything that does not depend on something else is
    // requested together — first the directory entries of the tile and its two neighbours, then the segment's entry, the
    // visible offsets, the record ahead of the tile, the tile's first eight batches and the first batch of the tile behind.
    const TsTile T = Q.tiles[tile];
    const uint32_t cnt = Q.tile_stats[4u * tile];
    const u64 off = Q.tile_off[tile];
    const uint32_t tp = tile ? tile - 1u : 0u;
    const u64 prev_in_off = Q.tiles[tp].in_off;
    const uint32_t prev_cnt = Q.tile_stats[4u * tp];
    const u64 prev_off = Q.tile_off[tp];
    const uint32_t tn = tile + 1u < ntiles ? tile + 1u : tile;    // the tile behind: four in five chains run on into it
    const u64 next_in_off = Q.tiles[tn].in_off;
    const uint32_t next_cnt = Q.tile_stats[4u * tn];
    const u64 next_off = Q.tile_off[tn];
    const uint32_t si = T.seg - seg_base;
    const TsShardSegIn S = segs[si];
    if (tile < S.o0 || tile >= S.o1) return;              // a context tile: its chains and its records are its owner's
    const u64 fb = bounds[2ull * si], rb = bounds[2ull * si + 1];
    // A stream in push order (MODE 1, Q.unordered): the search range is a range of stream INDICES — from where the reference's
    // lower_bound lands to the first record at or behind revBoundary as the stream lies (ts_its_range) — not of positions.
    const bool unord = MODE == 1 && Q.unordered != 0u;
    const u64 it_idx = unord ? Q.its_range[2ull * si] : 0ull, end_idx = unord ? Q.its_range[2ull * si + 1] : 0ull;
    const bool its_on = unord ? it_idx < end_idx : !(rb == 0 || fb >= rb);
    // visible records of this tile (a shard's message): where they go, and whether there are any
    u64 vis_at = 0;
    bool vis_on = false;
    if (W.off) {
        const uint32_t i = tile - W.own0;
        vis_at = W.off[i];
        vis_on = W.off[i + 1] > vis_at && W.off[W.own1 - W.own0] <= W.capacity;
    }
    if ((!its_on && !vis_on) || cnt == 0u) return;
    const u64 tile_rel = T.in_off - S.in_off;
    const u64 src = off;                                   // (index of the tile's first record)
    const u64 z_lo = W.terminal_limit, z_hi = S.len > W.terminal_limit ? S.len - W.terminal_limit : 0ull;
    constexpr uint32_t kGroup = 8;                         // batches of 64 records requested together
    uint32_t recs[kGroup];
#pragma unroll
    for (uint32_t q = 0; q < kGroup; ++q) recs[q] = 64u * q + lane < cnt ? rec_norm<MODE>(Q, rec_at(Q, Q.matches, src + 64u * q + lane)) : 0u;
    const uint32_t next_rec = (tn != tile && lane < next_cnt) ? rec_norm<MODE>(Q, rec_at(Q, Q.matches, next_off + lane)) : 0u;

    // the record ahead of the tile's first one (head test of that record), as a position relative to the tile (negative)
    bool has_last = false;
    int last_rel = 0;
    bool ooc = false;
    const bool open_l = !(S.flags & TS_SEG_F_HAS_START), open_r = !(S.flags & TS_SEG_F_HAS_END);
    if (its_on) {
        bool found = false;
        u64 last_pos = 0;
        if (tile > S.t0 && prev_cnt) {                     // the usual case: the tile before holds it
            const uint32_t r = rec_norm<MODE>(Q, rec_at(Q, Q.matches, prev_off + prev_cnt - 1u));
            last_pos = prev_in_off - S.in_off + (r >> 2);
            found = true;
        } else {
            for (uint32_t t = tile; t > S.t0; --t) {
                const uint32_t c = Q.tile_stats[4u * (t - 1u)];
                if (c) {
                    const uint32_t r = rec_norm<MODE>(Q, rec_at(Q, Q.matches, Q.tile_off[t - 1u] + c - 1u));
                    last_pos = Q.tiles[t - 1u].in_off - S.in_off + (r >> 2);
                    found = true;
                    break;
                }
            }
        }
        has_last = found && last_pos >= fb;                // (a record ahead of the search range does not chain)
        // (farther than 2^30 ahead is as good as 2^30: -k is 16 bits)
        last_rel = tile_rel - last_pos > 0x40000000ull ? -0x40000000 : -(int)(uint32_t)(tile_rel - last_pos);
        if (unord) {
            // the record ahead is the stream's record off - 1, in range iff that index is; it may lie BEHIND this tile's first
            // base (a record that moved into the tile before): the gap to it is then negative — wraps, in the reference — and
            // the first record opens a chain whatever the distance
            has_last = found && off > it_idx;
            if (last_pos >= tile_rel) last_rel = last_pos - tile_rel > 0x3FFFFFFFull ? 0x3FFFFFFF : (int)(uint32_t)(last_pos - tile_rel);
        }
        if (!found && open_l && S.lo_rel > fb) {
            // nothing in the whole left context: a record further left is too far to chain if the context is wider than -k
            // (it is, shard.cpp sizes it so) — unless this tile's first record sits right at the view's edge
            const u64 p0 = tile_rel + ((uint32_t)__builtin_amdgcn_readfirstlane((int)recs[0]) >> 2);
            if (p0 <= S.lo_rel + Q.max_match_dist) ooc = true;
        }
    }
    // the open chain: the index of its first record in this tile, and the stream's canonical count ahead of it
    bool open = false, finished = false;                   // finished: the search range ended inside this tile
    uint32_t idx0 = 0, c0 = 0, ccum = 0;
    uint32_t vis_done = 0;
    // Everything a record is compared with, relative to the tile and in 32 bits (a tile's positions are < 2^16); and what
    // holds for the whole tile is decided once: nearly every tile lies wholly inside the search range and wholly outside
    // the terminal zone, where "in range" is "valid" and "visible" is "canonical".
    const u64 tile_end = tile_rel + T.own_len;
    auto rel32 = [&](u64 x) -> uint32_t { return x <= tile_rel ? 0u : (x - tile_rel > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)(x - tile_rel)); };
    const uint32_t fb32 = rel32(fb), rb32 = rel32(rb);
    const bool all_in = unord ? (off >= it_idx && off + cnt <= end_idx) : (fb <= tile_rel && rb >= tile_end);
    const bool vis_all = tile_end - 1 <= z_lo || tile_rel >= z_hi, vis_canon = tile_rel > z_lo && tile_end <= z_hi;
    const uint32_t zlo32 = z_lo < tile_rel ? 0u : rel32(z_lo), zhi32 = rel32(z_hi);
    const bool zlo_none = z_lo < tile_rel;                 // no position of the tile is <= z_lo
    const int kdist = (int)Q.max_match_dist;
    // a chain with four canonical matches: listed for ts_interstitial_evaluate (past the list's capacity it is only counted:
    // the evaluation kernel then reports an overflow, and the caller comes back with a longer list)
    auto candidate = [&](uint32_t i0) {
        if (lane == 0u) {
            const uint32_t slot = atomicAdd(Q.n_cand, 1u);
            if (slot < Q.cand_cap) { Q.cand[2u * slot] = tile; Q.cand[2u * slot + 1u] = i0; }
        }
    };
    auto batch = [&](uint32_t b0, uint32_t r) {
        const uint32_t nvalid = cnt - b0 < 64u ? cnt - b0 : 64u;
        const u64 VALID = low_bits(nvalid);
        const uint32_t p32 = r >> 2;
        const u64 CAN = __ballot((r & 1u) != 0u) & VALID;
        if (vis_on) {
            u64 m;
            if (vis_all) m = VALID;
            else if (vis_canon) m = CAN;
            else m = __ballot((r & 1u) || (!zlo_none && p32 <= zlo32) || p32 >= zhi32) & VALID;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            if ((m >> lane) & 1ull) {
                if (W.rec_bytes == 2u) ((uint16_t *)W.dst)[vis_at + vis_done + rank] = (uint16_t)r;
                else ((uint32_t *)W.dst)[vis_at + vis_done + rank] = r;
            }
            vis_done += (uint32_t)__popcll(m);
        }
        if (!its_on || finished) return;
        u64 R;
        bool past;
        if (unord && !all_in) {
            const u64 g0 = off + b0;                        // stream index of lane 0's record
            const uint32_t lo = it_idx > g0 ? (it_idx - g0 > 64ull ? 64u : (uint32_t)(it_idx - g0)) : 0u;
            const uint32_t hi = end_idx > g0 ? (end_idx - g0 > 64ull ? 64u : (uint32_t)(end_idx - g0)) : 0u;
            R = low_bits(hi) & ~low_bits(lo) & VALID;
            past = end_idx < g0 + nvalid;
        } else {
            R = all_in ? VALID : __ballot(p32 >= fb32 && p32 < rb32) & VALID;
            past = all_in ? false : (__ballot(p32 >= rb32) & VALID) != 0ull;   // records at or behind revBoundary: the search ends here
        }
        if (R != 0ull) {
            // a record opens a chain iff its predecessor is out of range or more than -k ahead of it: the lane below, or for
            // lane 0 the last record of the batch / tile before
            const uint32_t below = lane_below(p32);
            u64 H = R & (~(R << 1) | __ballot(p32 - below > Q.max_match_dist));
            // (unsigned: a predecessor that lies behind the record — a stream in push order — is farther than any -k, as the reference's
            // wrapped gap is; in position order the difference is never negative)
            const int gap0 = (int)(uint32_t)__builtin_amdgcn_readfirstlane((int)p32) - last_rel;
            if ((R & 1ull) && has_last && (MODE == 1 ? (uint32_t)gap0 <= (uint32_t)kdist : gap0 <= kdist)) H &= ~1ull;
            const u64 Cn = CAN & R;
            if (H != 0ull) {
                const uint32_t first_head = (uint32_t)__builtin_ctzll(H), top_head = 63u - (uint32_t)__builtin_clzll(H);
                const u64 ahead = low_bits(first_head), below_top = low_bits(top_head);
                // the chain carried into this batch ends at the first head
                if (open && ccum + (uint32_t)__popcll(Cn & ahead) - c0 >= 4u) candidate(idx0);
                // the chains that start and end inside this batch: only when they hold four canonical matches between them
                // does every head lane look at its own (its lane range runs to the next head)
                const u64 inner = Cn & ~ahead & (past ? ~0ull : below_top);
                if (__popcll(inner) >= 4) {
                    const u64 above = lane < 63u ? H >> (lane + 1u) : 0ull;
                    const uint32_t next = above ? lane + 1u + (uint32_t)__builtin_ctzll(above) : 64u;   // the next head's lane
                    const bool mine4 = __popcll(Cn & low_bits(next) & ~low_bits(lane)) >= 4;
                    u64 todo = __ballot(((H >> lane) & 1ull) && (lane != top_head || past) && mine4);
                    for (; todo; todo &= todo - 1ull) candidate(b0 + (uint32_t)__builtin_ctzll(todo));
                }
                open = !past;                                             // the last head's chain stays open
                idx0 = b0 + top_head;
                c0 = ccum + (uint32_t)__popcll(Cn & below_top);
            }
            ccum += (uint32_t)__popcll(Cn);
            has_last = true;
            last_rel = (int)(uint32_t)__builtin_amdgcn_readlane((int)p32, 63 - (int)__builtin_clzll(R));
        }
        if (past) {
            if (open && ccum - c0 >= 4u) candidate(idx0);
            open = false;
            finished = true;
        }
    };
    // The same step for the tiles that lie wholly inside the search range and wholly outside the terminal zone — all but a
    // few per segment — where every valid record is in range and the visible ones are the canonical ones: straight-line
    // scalar code (selects, no branch but the two rare ones), because scalar issue is what bounds this kernel.
    auto fast_batch = [&](uint32_t b0, uint32_t r) {
        const uint32_t nvalid = cnt - b0 < 64u ? cnt - b0 : 64u;
        const u64 VALID = ~0ull >> (64u - nvalid);
        const uint32_t p32 = r >> 2;
        const u64 Cn = __ballot((r & 1u) != 0u) & VALID;
        if (vis_on) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(Cn >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Cn, 0u));
            if (r & 1u) {
                if (W.rec_bytes == 2u) ((uint16_t *)W.dst)[vis_at + vis_done + rank] = (uint16_t)r;
                else ((uint32_t *)W.dst)[vis_at + vis_done + rank] = r;
            }
            vis_done += (uint32_t)__popcll(Cn);
        }
        if (!its_on) return;
        const uint32_t below = lane_below(p32);
        u64 H = VALID & (__ballot(p32 - below > Q.max_match_dist) | 1ull);
        const int gap0 = (int)(uint32_t)__builtin_amdgcn_readfirstlane((int)p32) - last_rel;
        H &= ~(u64)((has_last && (MODE == 1 ? (uint32_t)gap0 <= (uint32_t)kdist : gap0 <= kdist)) ? 1u : 0u);
        const bool any = H != 0ull;
        const u64 ahead = (H - 1ull) & ~H;                                // the lanes ahead of the first head (all of them when there is none)
        const uint32_t top_head = any ? 63u - (uint32_t)__builtin_clzll(H) : 0u;
        const u64 below_top = (1ull << top_head) - 1ull;
        const bool cand_carry = open && any && ccum + (uint32_t)__popcll(Cn & ahead) - c0 >= 4u;
        const bool cand_inner = any && __popcll(Cn & ~ahead & below_top) >= 4;
        if (cand_carry || cand_inner) {                                   // rare: a chain with four canonical matches ends in this batch
            if (cand_carry) candidate(idx0);
            if (cand_inner) {
                const u64 above = lane < 63u ? H >> (lane + 1u) : 0ull;
                const uint32_t next = above ? lane + 1u + (uint32_t)__builtin_ctzll(above) : 64u;
                const bool mine4 = __popcll(Cn & low_bits(next) & ~low_bits(lane)) >= 4;
                u64 todo = __ballot(((H >> lane) & 1ull) && lane != top_head && mine4);
                for (; todo; todo &= todo - 1ull) candidate(b0 + (uint32_t)__builtin_ctzll(todo));
            }
        }
        open = any ? true : open;
        idx0 = any ? b0 + top_head : idx0;
        c0 = any ? ccum + (uint32_t)__popcll(Cn & below_top) : c0;
        ccum += (uint32_t)__popcll(Cn);
        has_last = true;
        last_rel = (int)(uint32_t)__builtin_amdgcn_readlane((int)p32, (int)nvalid - 1);
    };
    if (all_in && (!vis_on || vis_canon)) {
#pragma unroll
        for (uint32_t q = 0; q < kGroup; ++q)
            if (64u * q < cnt) fast_batch(64u * q, recs[q]);
        for (uint32_t b0 = 64u * kGroup; b0 < cnt; b0 += 64u)           // a dense tile: the rest, batch by batch
            fast_batch(b0, b0 + lane < cnt ? rec_norm<MODE>(Q, rec_at(Q, Q.matches, src + b0 + lane)) : 0u);
    } else {
#pragma unroll
        for (uint32_t q = 0; q < kGroup; ++q)
            if (64u * q < cnt) batch(64u * q, recs[q]);
        for (uint32_t b0 = 64u * kGroup; b0 < cnt; b0 += 64u)
            batch(b0, b0 + lane < cnt ? rec_norm<MODE>(Q, rec_at(Q, Q.matches, src + b0 + lane)) : 0u);
    }
    // the chain that is still open belongs to this tile: follow it through the tiles behind until a head closes it
    if (its_on && open && !finished) {
        bool closed = false;
        uint32_t canon = ccum - c0;
        u64 last_abs = tile_rel + (uint32_t)last_rel;                      // (the open chain's last record lies in this tile)
        for (uint32_t t = tile + 1u; t < S.t1 && !closed; ++t) {
            const bool pre_fetched = t == tn;
            const uint32_t c2 = pre_fetched ? next_cnt : Q.tile_stats[4u * t];
            if (c2 == 0u) continue;
            const u64 rel2 = (pre_fetched ? next_in_off : Q.tiles[t].in_off) - S.in_off;
            const u64 src2 = pre_fetched ? next_off : Q.tile_off[t];
            const uint32_t rb2 = rb <= rel2 ? 0u : (rb - rel2 > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)(rb - rel2));
            for (uint32_t b0 = 0; b0 < c2 && !closed; b0 += 64u) {
                const uint32_t r = (pre_fetched && b0 == 0u) ? next_rec : (b0 + lane < c2 ? rec_norm<MODE>(Q, rec_at(Q, Q.matches, src2 + b0 + lane)) : 0u);
                const uint32_t nvalid = c2 - b0 < 64u ? c2 - b0 : 64u;
                const u64 VALID = low_bits(nvalid);
                const uint32_t p32 = r >> 2;
                u64 R = __ballot(p32 < rb2) & VALID;                     // (p >= fb: behind a record that was)
                if (unord) {                                             // by stream index: up to the range's end
                    const u64 g0 = src2 + b0;
                    R = low_bits(end_idx > g0 ? (end_idx - g0 > 64ull ? 64u : (uint32_t)(end_idx - g0)) : 0u) & VALID;
                }
                if (R != 0ull) {
                    const uint32_t below = lane_below(p32);
                    u64 H = R & (__ballot(p32 - below > Q.max_match_dist) | 1ull);
                    if ((R & 1ull) && rel2 + (uint32_t)__builtin_amdgcn_readfirstlane((int)p32) - last_abs <= Q.max_match_dist) H &= ~1ull;
                    const u64 pre = R & low_bits(H ? (uint32_t)__builtin_ctzll(H) : 64u);
                    if (pre) {
                        canon += (uint32_t)__popcll(__ballot((r & 1u) != 0u) & pre);
                        last_abs = rel2 + (uint32_t)__builtin_amdgcn_readlane((int)p32, 63 - (int)__builtin_clzll(pre));
                    }
                    if (H != 0ull) closed = true;
                }
                if (R != VALID) closed = true;                           // a record at or behind revBoundary
            }
        }
        // the view ended first: the segment's end — or a neighbour's tiles, where the chain may go on
        if (!closed && open_r && last_abs + Q.max_match_dist >= S.hi_rel && S.hi_rel < rb) ooc = true;
        else if (canon >= 4u) candidate(idx0);
    }
    if (seg_out && ooc && lane == 0) atomicOr(&seg_out[si].flags, TS_SEG_F_CONTEXT);
}

// every tile of the range: results that carry no chain summaries (adopted from elsewhere, TS_EMIT=0), and the shard pack
// that takes the visible records from the match stream
template <int MODE>
__global__ __launch_bounds__(kSideWg)
void ts_interstitial_blocks(const TsBlockCallParams Q, const TsShardSegIn *segs, uint32_t seg_base,
                            const u64 *bounds, uint32_t ntiles, TsShardSeg *seg_out, const TsVisibleOut W) {
    // (readfirstlane: the compiler cannot know that threadIdx.x >> 6 is the same in all lanes of a wave; told so, it keeps the
    // tile's directory entries, the bounds and every ballot in scalar registers and branches instead of predicating)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t tile = blockIdx.x * (blockDim.x >> 6) + wave;
    if (tile >= ntiles) return;
    its_tile<MODE>(Q, segs, seg_base, bounds, ntiles, seg_out, W, tile);
}

// ---- the interstitial search from the scan's own chain summaries (TsTileChain, written by ts_scan_tiles with P.emit)
//
// ts_chain_screen: ONE THREAD per tile decides, from 16 bytes per tile, whether a chain that STARTS in the tile can hold
// the four canonical matches a block needs (getInterstitialBlocks, src/teloscope.cpp:206): a chain between two internal
// heads (the scan flagged it), the chain of the tile's first record when that record opens one, or the chain open at the
// tile's end followed through the summaries of the tiles behind.  The test errs on the side of listing — counts are upper
// bounds where the search range ends inside a tile — because every listed tile is then walked by its_tile above, record by
// record, exactly as before.  Tiles the two terminal boundaries cut, and tiles at the edge of a shard's view (where
// its_tile decides whether a chain ran out of context), are always listed.  On random sequence about one tile in a
// thousand is listed: the pass reads 3 MB instead of the 0.37 GB of records.
__global__ __launch_bounds__(kSideWg)
void ts_chain_screen(const TsBlockCallParams Q, const uint32_t *chain, const TsShardSegIn *segs, uint32_t seg_base,
                     const u64 *bounds, uint32_t ntiles, uint32_t *work, uint32_t *n_work, TsShardSeg *seg_out) {
    const uint32_t tile = blockIdx.x * blockDim.x + threadIdx.x;
    // (a shard: the per-segment counts of its message, from the same directory rows this thread reads anyway)
    if (seg_out) add_segment_sums(Q, segs, seg_base, tile, ntiles, nullptr, seg_out);
    bool list = false;
    if (tile < ntiles) {
        const uint32_t cnt = Q.tile_stats[4u * tile];
        const TsTile T = Q.tiles[tile];
        const uint32_t si = T.seg - seg_base;
        const TsShardSegIn S = segs[si];
        const u64 fb = bounds[2ull * si], rb = bounds[2ull * si + 1];
        const bool its_on = !(rb == 0 || fb >= rb);
        if (cnt != 0u && its_on && tile >= S.o0 && tile < S.o1) {
            const u64 tile_rel = T.in_off - S.in_off, tile_end = tile_rel + T.own_len;
            const uint2 c = *(const uint2 *)&chain[4ull * tile];
            const uint32_t p_first = c.x & 0xFFFFu, p_last = c.x >> 16;
            const uint32_t A = c.y & 0x7FFFu, Z = (c.y >> 16) & 0x7FFFu;
            const bool heads = (c.y & TS_CHAIN_HEADS) != 0u, inner = (c.y & TS_CHAIN_INNER) != 0u;
            const bool open_l = !(S.flags & TS_SEG_F_HAS_START), open_r = !(S.flags & TS_SEG_F_HAS_END);
            if (!(fb <= tile_rel && rb >= tile_end)) {
                // a boundary inside the tile: exact walk, unless no record of it can lie in [fb, rb)
                list = tile_rel + p_last >= fb && tile_rel + p_first < rb;
            } else if (inner) {
                list = true;
            } else {
                // does the tile's first record open a chain?  (the record before it: in the nearest tile ahead that holds one; an
                // empty tile wider than -k in between settles it)
                bool first_head = true;
                bool found = false;
                for (uint32_t t = tile; t > S.t0; --t) {
                    const uint32_t c2 = Q.tile_stats[4u * (t - 1u)];
                    if (c2 == 0u) {
                        if (Q.tiles[t - 1u].own_len > Q.max_match_dist) { found = true; break; }     // too far whatever lies ahead
                        continue;
                    }
                    const u64 last_pos = Q.tiles[t - 1u].in_off - S.in_off + (chain[4ull * (t - 1u)] >> 16);
                    found = true;
                    if (last_pos >= fb && tile_rel + p_first - last_pos <= Q.max_match_dist) first_head = false;
                    break;
                }
                // nothing in the whole left context of a clipped view: its_tile decides whether that is out of context
                if (!found && open_l && S.lo_rel > fb) list = true;
                if (first_head && heads && A >= 4u) list = true;
                if (heads || first_head) {
                    // the chain open at the tile's end — from its last internal head, or from its first record — through the tiles behind
                    uint32_t canon = Z;
                    u64 last_abs = tile_rel + p_last;
                    bool closed = false;
                    for (uint32_t t = tile + 1u; t < S.t1 && !closed; ++t) {
                        const uint32_t c2 = Q.tile_stats[4u * t];
                        if (c2 == 0u) {
                            if (Q.tiles[t].own_len > Q.max_match_dist) closed = true;
                            continue;
                        }
                        const uint2 d = *(const uint2 *)&chain[4ull * t];
                        const u64 rel2 = Q.tiles[t].in_off - S.in_off;
                        if (rel2 + (d.x & 0xFFFFu) - last_abs > Q.max_match_dist) { closed = true; break; }
                        canon += d.y & 0x7FFFu;
                        if (d.y & TS_CHAIN_HEADS) { closed = true; break; }
                        last_abs = rel2 + (d.x >> 16);
                    }
                    if (canon >= 4u || (!closed && open_r)) list = true;
                }
            }
        }
    }
    const u64 m = __ballot(list);
    if (m == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == 0u) base = atomicAdd(n_work, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (list) work[base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u))] = tile;
}

// the listed tiles, one wave each (the list holds at most every tile of the range)
__global__ __launch_bounds__(kSideWg)
void ts_interstitial_listed(const TsBlockCallParams Q, const TsShardSegIn *segs, uint32_t seg_base, const u64 *bounds,
                            uint32_t ntiles, TsShardSeg *seg_out, const uint32_t *work, const uint32_t *n_work) {
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const uint32_t n = *n_work;
    const TsVisibleOut W{};
    for (uint32_t i = blockIdx.x * (blockDim.x >> 6) + wave; i < n; i += gridDim.x * (blockDim.x >> 6))
        its_tile<0>(Q, segs, seg_base, bounds, ntiles, seg_out, W, (uint32_t)__builtin_amdgcn_readfirstlane((int)work[i]));
}

// The listed chains, one wave each: walked exactly, filtered, written.  A list that overflowed is reported through the
// block counter (more blocks than the buffer holds = "come back with more room": the callers' existing path).
template <int MODE>
__global__ __launch_bounds__(kSideWg)
void ts_interstitial_evaluate(const TsBlockCallParams Q, const TsShardSegIn *segs, uint32_t seg_base, const u64 *bounds) {
    const uint32_t n_all = *Q.n_cand;
    const uint32_t n = n_all < Q.cand_cap ? n_all : Q.cand_cap;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (blockIdx.x == 0 && threadIdx.x == 0 && n_all > Q.cand_cap) atomicAdd(Q.n_blocks, Q.block_cap + 1u);
    for (uint32_t i = blockIdx.x * (blockDim.x >> 6) + wave; i < n; i += gridDim.x * (blockDim.x >> 6)) {
        const uint32_t tile = Q.cand[2u * i], i0 = Q.cand[2u * i + 1u];
        const uint32_t si = Q.tiles[tile].seg - seg_base;
        const TsShardSegIn S = segs[si];
        its_evaluate<MODE>(Q, S, tile, i0, bounds[2ull * si + 1]);
    }
}

// The interstitial search range of a stream in push order, per segment, as stream indices {first, end}: `first` is where
// std::lower_bound(allMatches, fwdBoundary) lands — the bisection restated probe by probe, because on a stream that is not quite
// sorted its result depends on which records it looks at (src/teloscope.cpp:188-191; libstdc++: half = len >> 1, mid = first +
// half) — and `end` the first record from there on that lies at or behind revBoundary (:235 ends the loop at it).  {0, 0}: no
// search (the gates of scanSegment, :646-653, or an empty range).  One wave per segment; a probe finds its record's tile by a
// 64-way search of the tile offsets.
__global__ __launch_bounds__(kSideWg)
void ts_its_range(const TsBlockCallParams Q, const TsShardSegIn *segs, uint32_t nseg, const u64 *bounds, u64 *range) {
    const uint32_t si = blockIdx.x;
    if (si >= nseg) return;
    const uint32_t lane = threadIdx.x & 63u;
    const TsShardSegIn S = segs[si];
    const u64 fb = bounds[2ull * si], rb = bounds[2ull * si + 1];
    const u64 g_lo = Q.tile_off[S.t0], g_hi = Q.tile_off[S.t1], n = g_hi - g_lo;
    u64 it = 0, end = 0;
    // the last tile of [t0, t1) whose `what` (non-decreasing over the tiles) is <= v; t0 if none
    auto last_tile_le = [&](auto what, u64 v) -> uint32_t {
        uint32_t lo = S.t0, hi = S.t1;
        while (hi - lo > 1u) {
            const uint32_t step = (hi - lo + 63u) / 64u;
            const uint32_t t = lo + lane * step;
            const u64 m = __ballot(t < hi && (lane == 0u || what(t) <= v));
            const uint32_t top = 63u - (uint32_t)__builtin_clzll(m);
            lo += top * step;
            hi = lo + step < hi ? lo + step : hi;
        }
        return lo;
    };
    auto pos_at = [&](u64 g) -> u64 {                      // segment-relative position of stream record g (wave-uniform)
        const uint32_t t = last_tile_le([&](uint32_t x) { return Q.tile_off[x]; }, g);
        return Q.tiles[t].in_off - S.in_off + (rec_norm<1>(Q, rec_at(Q, Q.matches, g)) >> 2);
    };
    if (rb != 0ull && n != 0ull && S.t1 > S.t0) {
        u64 first = 0, len = n;
        while (len > 0ull) {
            const u64 half = len >> 1, mid = first + half;
            if (pos_at(g_lo + mid) < fb) { first = mid + 1ull; len = len - half - 1ull; }
            else len = half;
        }
        if (first < n && pos_at(g_lo + first) < rb) {
            it = g_lo + first;
            end = g_hi;
            // records of the tiles ahead of the one that holds position rb - 64 all lie ahead of rb (a tile's records start
            // before the next tile's 63rd base): the scan for the first record at or behind rb begins there
            const u64 from = rb >= 64ull ? rb - 64ull : 0ull;
            const uint32_t tq = last_tile_le([&](uint32_t x) { return Q.tiles[x].in_off - S.in_off; }, from);
            bool found = false;
            for (uint32_t t = tq; t < S.t1 && !found; ++t) {
                const u64 o = Q.tile_off[t], o1 = Q.tile_off[t + 1u];
                if (o1 <= it) continue;
                const u64 rel = Q.tiles[t].in_off - S.in_off;
                for (u64 g = o; g < o1 && !found; g += 64ull) {
                    const bool valid = g + lane < o1 && g + lane >= it;
                    const u64 p = valid ? rel + (rec_norm<1>(Q, rec_at(Q, Q.matches, g + lane)) >> 2) : 0ull;
                    const u64 m = __ballot(valid && p >= rb);
                    if (m) { end = g + (uint32_t)__builtin_ctzll(m); found = true; }
                }
            }
        }
    }
    if (lane == 0u) { range[2ull * si] = it; range[2ull * si + 1] = end; }
}

}  // namespace

namespace {
struct ZeroJobs { void *p[4]; unsigned long long n[4]; };     // byte counts, multiples of 4, pointers 4-byte aligned
__global__ __launch_bounds__(kSideWg)
void ts_zero_ranges(const ZeroJobs J) {
    for (int q = 0; q < 4; ++q) {
        uint32_t *dst = (uint32_t *)J.p[q];
        const u64 n = J.n[q] >> 2;
        for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) dst[i] = 0u;
    }
}
}  // namespace

// Up to four ranges zeroed by ONE kernel (a pack accumulates into its header, the per-segment sums and two list counters: four
// memsets were four launches on its chain of latencies).  n: bytes, multiples of 4; null pointers are skipped.
int ts_k_launch_zero(void *p0, unsigned long long n0, void *p1, unsigned long long n1, void *p2, unsigned long long n2,
                     void *p3, unsigned long long n3, void *stream) {
    ZeroJobs J{{p0, p1, p2, p3}, {p0 ? n0 : 0, p1 ? n1 : 0, p2 ? n2 : 0, p3 ? n3 : 0}};
    unsigned long long most = 0;
    for (int q = 0; q < 4; ++q) most = J.n[q] > most ? J.n[q] : most;
    const unsigned grid = (unsigned)((most / 4 + 255) / 256 < 1 ? 1 : ((most / 4 + 255) / 256 > 64 ? 64 : (most / 4 + 255) / 256));
    hipLaunchKernelGGL(ts_zero_ranges, dim3(grid * 4u), dim3(kSideWg), 0, (hipStream_t)stream, J);
    return (int)hipGetLastError();
}

// The two halves of block calling, for a caller that runs the terminal walks on a stream of their own (shard.cpp): the
// walks are one latency-bound wave per segment end, and nothing but the interstitial search waits for them.
int ts_k_launch_terminal(const TsBlockCallParams *Q, const TsShardSegIn *segs, uint32_t nseg, uint32_t seg_base, uint32_t ntiles,
                         unsigned long long *bounds, TsShardSeg *seg_out, void *stream) {
    (void)seg_base; (void)ntiles;
    if (nseg == 0) return 0;
    if (Q->unordered || Q->wide)
        hipLaunchKernelGGL(ts_terminal_blocks<1>, dim3(2u * nseg), dim3(kSideWg), 0, (hipStream_t)stream, *Q, segs, nseg, bounds, seg_out);
    else
        hipLaunchKernelGGL(ts_terminal_blocks<0>, dim3(2u * nseg), dim3(kSideWg), 0, (hipStream_t)stream, *Q, segs, nseg, bounds, seg_out);
    return (int)hipGetLastError();
}

// Per-segment totals {seen matches, seen forward, owned matches, owned canonical, owned forward}: into sums (5 x u64 per
// segment, zeroed here unless prezeroed) or, seg_out != nullptr, into the segments' entries of a shard's message (zeroed by
// the caller).  (A pack that screens by the scan's chain summaries does not need this: ts_chain_screen adds them up.)
int ts_k_launch_segment_sums(const TsBlockCallParams *Q, const TsShardSegIn *segs, uint32_t nseg, uint32_t seg_base, uint32_t ntiles,
                             unsigned long long *sums, TsShardSeg *seg_out, int prezeroed, void *stream) {
    if (nseg == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    if (!seg_out && !prezeroed) {
        hipError_t e = hipMemsetAsync(sums, 0, (size_t)nseg * 40, st);
        if (e != hipSuccess) return (int)e;
    }
    if (ntiles)
        hipLaunchKernelGGL(ts_segment_sums, dim3((ntiles + kSideWg - 1u) / kSideWg), dim3(kSideWg), 0, st, *Q, segs, seg_base, ntiles, sums, seg_out);
    return (int)hipGetLastError();
}

int ts_k_launch_interstitial(const TsBlockCallParams *Q, const TsShardSegIn *segs, uint32_t nseg, uint32_t seg_base, uint32_t ntiles,
                             const unsigned long long *bounds, TsShardSeg *seg_out, const TsVisibleOut *vis,
                             const uint32_t *chain, uint32_t *work, int prezeroed, void *stream) {
    if (nseg == 0 || ntiles == 0) return 0;
    TsVisibleOut W{};
    if (vis) W = *vis;
    if (chain && work && !W.off) {
        // from the scan's chain summaries: screen (a thread per tile), then the exact walk of the listed tiles only.
        // work[0] = the list's counter (zeroed here unless the caller did), work[1 ..] = the list
        if (!prezeroed) {
            hipError_t e = hipMemsetAsync(work, 0, 4, (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(ts_chain_screen, dim3((ntiles + kSideWg - 1u) / kSideWg), dim3(kSideWg), 0, (hipStream_t)stream, *Q, chain,
                           segs, seg_base, (const u64 *)bounds, ntiles, work + 1, work, seg_out);
        hipLaunchKernelGGL(ts_interstitial_listed, dim3(512), dim3(kSideWg), 0, (hipStream_t)stream, *Q, segs, seg_base,
                           (const u64 *)bounds, ntiles, seg_out, (const uint32_t *)(work + 1), (const uint32_t *)work);
    } else if (Q->unordered || Q->wide) {
        // the general path's wide records and push-ordered streams (MODE 1); the search range by stream index first
        if (Q->unordered) {
            if (!Q->its_range || seg_base != 0u) return (int)hipErrorInvalidValue;
            hipLaunchKernelGGL(ts_its_range, dim3(nseg), dim3(kSideWg), 0, (hipStream_t)stream, *Q, segs, nseg, (const u64 *)bounds,
                               (u64 *)Q->its_range);
        }
        hipLaunchKernelGGL(ts_interstitial_blocks<1>, dim3(ntiles), dim3(kSideWg), 0, (hipStream_t)stream, *Q,
                           segs, seg_base, (const u64 *)bounds, ntiles, seg_out, W);
        hipLaunchKernelGGL(ts_interstitial_evaluate<1>, dim3(1024), dim3(kSideWg), 0, (hipStream_t)stream, *Q, segs, seg_base,
                           (const u64 *)bounds);
        return (int)hipGetLastError();
    } else {
        hipLaunchKernelGGL(ts_interstitial_blocks<0>, dim3(ntiles), dim3(kSideWg), 0, (hipStream_t)stream, *Q,
                           segs, seg_base, (const u64 *)bounds, ntiles, seg_out, W);
    }
    hipLaunchKernelGGL(ts_interstitial_evaluate<0>, dim3(1024), dim3(kSideWg), 0, (hipStream_t)stream, *Q, segs, seg_base,
                       (const u64 *)bounds);
    return (int)hipGetLastError();
}

int ts_k_launch_block_call(const TsBlockCallParams *Q, const TsShardSegIn *segs, uint32_t nseg, uint32_t seg_base,
                           uint32_t ntiles, unsigned long long *bounds, TsShardSeg *seg_out, int with_its,
                           const TsVisibleOut *vis, unsigned long long *sums, const uint32_t *chain, uint32_t *work, void *stream) {
    int e = ts_k_launch_terminal(Q, segs, nseg, seg_base, ntiles, bounds, seg_out, stream);
    if (e == 0 && sums) e = ts_k_launch_segment_sums(Q, segs, nseg, seg_base, ntiles, sums, nullptr, 0, stream);
    if (e == 0 && (with_its || (vis && vis->off))) e = ts_k_launch_interstitial(Q, segs, nseg, seg_base, ntiles, bounds, seg_out, vis, chain, work, 0, stream);
    return e;
}
