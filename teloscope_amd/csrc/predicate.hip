// predicate.hip — gfx950 kernels of the read predicate: "does this segment have a terminal telomere block?" decided on
// the device from the packed match stream that ts_scan_tiles (kernels.hip) leaves behind.  Kept in a file of its own so
// that work on it does not touch the scan kernel's source (whose hash stamps the profiled HBM traffic).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ts_internal.h"

namespace {

typedef unsigned long long u64;

// ---------------------------------------------------------------------------------------
// ts_terminal_predicate: "does this segment have a terminal telomere block?" decided on the
// device from the packed match stream — Teloscope::getTerminalBlocks (src/teloscope.cpp:29-176)
// for both orientations, reduced to whether outBlocks would be non-empty.  One thread per
// segment (one wave for a read with a long match list, see pred_scan_wave) walks its tiles' records (ascending for the forward list, descending for the reverse
// list) through the same two-phase state machine: chain matches <= -k apart into sub-blocks,
// keep those with >= minBlockCounts matches, a canonical match and canonical density >= -y,
// merge kept sub-blocks <= -d apart, pass if a merged block is >= -l long.  This is what turns
// ReadTelomereFilter::matches (src/read-filter.cpp:37-45) into one byte per read off the device.
struct PredState {
    bool in_block, have_cur, pass;
    u64 bstart, bend, prev;                 // running sub-block
    uint32_t counts, canon, can_cov;
    u64 cstart, clen;                       // running merged block
};

__device__ __forceinline__ void pred_close_sub(PredState &st, const TsPredParams &Q, bool from_start) {
    const float need = Q.min_block_density * (float)(st.bend - st.bstart);
    if (st.counts >= Q.min_block_counts && st.canon > 0u && (float)st.can_cov >= need) {
        const u64 sstart = st.bstart, slen = (uint32_t)(st.bend - st.bstart);
        if (!st.have_cur) {
            st.cstart = sstart; st.clen = slen; st.have_cur = true;
        } else {
            const u64 gap = from_start ? sstart - (st.cstart + st.clen) : st.cstart - (sstart + slen);
            if (gap <= Q.max_block_dist) {
                if (from_start) st.clen = (uint32_t)((sstart + slen) - st.cstart);
                else { st.clen = (uint32_t)((st.cstart + st.clen) - sstart); st.cstart = sstart; }
            } else {
                if (st.clen >= Q.min_block_len) st.pass = true;
                st.cstart = sstart; st.clen = slen;
            }
        }
    }
    st.in_block = false;
}

// returns false when the walk must stop (a match outside the terminal zone while no chain is open)
__device__ __forceinline__ bool pred_feed(PredState &st, const TsPredParams &Q, bool from_start, u64 pos,
                                          bool canonical, u64 seg_len) {
    if (st.in_block) {
        const u64 gap = from_start ? pos - st.prev : st.prev - pos;
        if (gap <= Q.max_match_dist) {
            if (from_start) st.bend = pos + Q.k; else st.bstart = pos;
            st.counts++; st.canon += canonical; st.can_cov += canonical ? Q.k : 0u;
            st.prev = pos;
            return true;
        }
        pred_close_sub(st, Q, from_start);
    }
    const bool in_zone = seg_len <= Q.terminal_limit ? true
                       : (from_start ? pos < Q.terminal_limit : pos >= seg_len - Q.terminal_limit);
    if (!in_zone) return false;
    st.bstart = pos; st.bend = pos + Q.k; st.prev = pos;
    st.counts = 1; st.canon = canonical; st.can_cov = canonical ? Q.k : 0u;
    st.in_block = true;
    return true;
}

// One orientation of the walk, one thread per read.  Records are visited in walk order (ascending for the
// forward list, descending for the reverse list).  Every lane walks its own read, so its loads are scattered and the
// kernel is bound by the number of load INSTRUCTIONS (the address unit takes one lane's address per cycle): records are
// therefore fetched as aligned 16-byte blocks — the four-record blocks that cover the tile's records, the ends
// masked — two blocks at a time with the next two already requested (round 2; dword loads before: 4x the instructions).
// A block that would reach outside [matches, matches + nrec_limit) is read record by record.
constexpr uint32_t kPredBlocks = 2;      // 16-byte blocks a thread requests at a time (and as many again in flight; four measured the same)
template <bool FROM_START, typename REC = uint32_t>
__device__ __forceinline__ bool pred_walk(const TsTile *tiles, const u64 *tile_off, const uint32_t *tile_stats,
                                          const uint32_t *matches, u64 nrec_limit, uint32_t t0, uint32_t t1, u64 base, u64 n,
                                          const TsPredParams &Q) {
    constexpr uint32_t RPB = 16u / (uint32_t)sizeof(REC);     // records per 16-byte block (REC: see pred_walk_read)
    PredState st = {};
    bool go = true;
    for (uint32_t tt = 0; tt < t1 - t0 && go && !st.pass; ++tt) {
        const uint32_t t = FROM_START ? t0 + tt : t1 - 1u - tt;
        const uint32_t cnt = tile_stats[4u * t];
        if (cnt == 0u) continue;
        const u64 rel0 = tiles[t].in_off - base;
        const u64 off = tile_off[t];
        const REC *r = (const REC *)matches + off;
        const uint32_t m = (uint32_t)(((uintptr_t)r / sizeof(REC)) & (RPB - 1u));   // records of the first block that are not ours
        const uint32_t nb = (m + cnt + RPB - 1u) / RPB;                  // blocks covering records 0 .. cnt-1: record i = entry m + i
        const uint4 *ra = (const uint4 *)(r - m);
        const bool inside = off >= m && off - m + (u64)RPB * nb <= nrec_limit;
        auto blk = [&](uint32_t bi) -> uint4 {             // bi-th block in walk order (clamped: loads are unconditional)
            const uint32_t qi = bi < nb ? bi : nb - 1u;
            const uint32_t q = FROM_START ? qi : nb - 1u - qi;
            if (inside) return ra[q];
            uint32_t e[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (uint32_t j = 0; j < RPB; ++j) {
                const uint32_t i = RPB * q + j - m;        // wraps below the first record
                const uint32_t x = i < cnt ? (uint32_t)r[i] : 0u;
                if (sizeof(REC) == 4) e[j & 3u] = x; else e[j >> 1] |= x << (16u * (j & 1u));
            }
            return make_uint4(e[0], e[1], e[2], e[3]);
        };
        auto feed_block = [&](uint32_t bi, const uint4 &v) {
            if (bi >= nb) return;
            const uint32_t q = FROM_START ? bi : nb - 1u - bi;
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t jj = 0; jj < RPB; ++jj) {
                const uint32_t j = FROM_START ? jj : RPB - 1u - jj;
                const uint32_t ej = sizeof(REC) == 4 ? w[j & 3u] : (w[j >> 1] >> (16u * (j & 1u))) & 0xFFFFu;
                const uint32_t i = RPB * q + j - m;
                if (i < cnt && go && ((ej & 2u) != 0u) == FROM_START)
                    go = pred_feed(st, Q, FROM_START, rel0 + (ej >> 2), ej & 1u, n);
            }
        };
        uint4 v[kPredBlocks], w[kPredBlocks];
#pragma unroll
        for (uint32_t j = 0; j < kPredBlocks; ++j) v[j] = blk(j);
        for (uint32_t bi = 0; bi < nb && go; bi += kPredBlocks) {
#pragma unroll
            for (uint32_t j = 0; j < kPredBlocks; ++j) w[j] = blk(bi + kPredBlocks + j);
#pragma unroll
            for (uint32_t j = 0; j < kPredBlocks; ++j) feed_block(bi + j, v[j]);
#pragma unroll
            for (uint32_t j = 0; j < kPredBlocks; ++j) v[j] = w[j];
        }
    }
    if (st.in_block) pred_close_sub(st, Q, FROM_START);
    if (st.have_cur && st.clen >= Q.min_block_len) st.pass = true;
    return st.pass;
}

// The walk of a segment that is terminal zone as a whole (n <= terminal_limit: every read), one thread per segment:
// BOTH lists in ONE ascending pass over the records, 32-bit positions, the state machine cut down to what such a
// segment needs.  Nothing stops the walk early there, and chaining (gap <= -k), the sub-block filters and the merging
// (gap <= -d) are symmetric, so the reverse list gives the same blocks walked from the start as from the end (the
// whole-wave walk of long reads below relies on the same fact).  A chain is {first position, last position, matches,
// canonical matches}: its end is last + k and its canonical coverage canonical x k, as pred_feed accumulates them; a
// kept chain is merged into the running block exactly like pred_close_sub does (same u32 / float expressions).
// pred_walk<> (64-bit, zone rule, either direction) costs ~4x the instructions per record; it stays for segments longer
// than the terminal limit.
struct ReadChain { uint32_t first, last, counts, canon; };            // the open chain of one list (counts == 0: none)
struct ReadBlock { uint32_t cstart, clen; bool have_cur, pass; };      // the running merged block of one list

// a kept chain [sstart, sstart + slen) into its list's running block: pred_close_sub's merge, same u32 expressions
__device__ __forceinline__ void read_block_merge(ReadBlock &b, const TsPredParams &Q, uint32_t sstart, uint32_t slen) {
    if (!b.have_cur) { b.cstart = sstart; b.clen = slen; b.have_cur = true; }
    else if (sstart - (b.cstart + b.clen) <= Q.max_block_dist) b.clen = sstart + slen - b.cstart;   // (wraps to "far" when chains overlap)
    else { if (b.clen >= Q.min_block_len) b.pass = true; b.cstart = sstart; b.clen = slen; }
}

// is the chain kept when it closes?  enough matches, a canonical one, the density (pred_close_sub's float expression)
__device__ __forceinline__ bool read_chain_kept(const ReadChain &c, const TsPredParams &Q) {
    const uint32_t slen = c.last + Q.k - c.first;
    return (c.counts >= Q.min_block_counts) & (c.canon > 0u) & ((float)(c.canon * Q.k) >= Q.min_block_density * (float)slen);
}

// One record into the chain of ITS list, without a branch on the record: every lane walks its own read, so a branch
// taken by any lane is paid by all sixty-four (the branchy form cost ~100 executed instructions per record, and the
// predicate of 500 k reads as many wave-instructions as two thirds of a 3 Gb scan).  The record's list is selected by
// v_cndmask (four chain fields in, four out per list); only a chain that is KEPT when it closes takes a branch, into
// the merge: one record in a few thousand outside telomeres.
__device__ __forceinline__ void read_feed(ReadChain &cf, ReadChain &cr, ReadBlock &bf, ReadBlock &br, const TsPredParams &Q,
                                          bool sel, bool fwd, uint32_t pos, uint32_t canonical) {
    ReadChain c;
    c.first = fwd ? cf.first : cr.first; c.last = fwd ? cf.last : cr.last;
    c.counts = fwd ? cf.counts : cr.counts; c.canon = fwd ? cf.canon : cr.canon;
    const bool open = c.counts != 0u;
    const bool ext = sel & open & (pos - c.last <= Q.max_match_dist);
    const bool kept = sel & open & !ext & read_chain_kept(c, Q);
    if (kept) {
        const uint32_t slen = c.last + Q.k - c.first;
        if (fwd) read_block_merge(bf, Q, c.first, slen); else read_block_merge(br, Q, c.first, slen);
    }
    c.first = ext ? c.first : pos;
    c.counts = ext ? c.counts + 1u : 1u;
    c.canon = ext ? c.canon + canonical : canonical;
    const bool wf = sel & fwd, wr = sel & !fwd;
    cf.first = wf ? c.first : cf.first; cf.last = wf ? pos : cf.last; cf.counts = wf ? c.counts : cf.counts; cf.canon = wf ? c.canon : cf.canon;
    cr.first = wr ? c.first : cr.first; cr.last = wr ? pos : cr.last; cr.counts = wr ? c.counts : cr.counts; cr.canon = wr ? c.canon : cr.canon;
}

template <uint32_t NB, bool PADDED, typename REC = uint32_t>   // NB: 16-byte blocks a thread requests at a time (and as many again in flight);
                                                // PADDED: the records lie in a 16-byte aligned buffer with 16 bytes of slack;
                                                // REC: uint16_t for a read batch whose scan left 16-bit records (TsScanParams.rec16):
                                                // a block then holds eight records — half the load instructions, half the bytes
__device__ __forceinline__ bool pred_walk_read(const TsTile *tiles, const u64 *tile_off, const uint32_t *tile_stats,
                                               const uint32_t *matches, u64 nrec_limit, uint32_t t0, uint32_t t1, u64 base,
                                               const TsPredParams &Q, bool walk_fwd, bool walk_rev) {
    constexpr uint32_t RPB = 16u / (uint32_t)sizeof(REC);     // records per 16-byte block
    ReadChain cf = {}, cr = {};                 // (a list of fewer than two matches is not walked: walk_fwd / walk_rev)
    ReadBlock bf = {}, br = {};
    for (uint32_t t = t0; t < t1 && !bf.pass && !br.pass; ++t) {
        const uint32_t cnt = tile_stats[4u * t];
        if (cnt == 0u) continue;
        const uint32_t rel0 = (uint32_t)(tiles[t].in_off - base);
        const u64 off = tile_off[t];
        const REC *r = (const REC *)matches + off;
        const uint32_t m = (uint32_t)(((uintptr_t)r / sizeof(REC)) & (RPB - 1u));   // see pred_walk: aligned 16-byte blocks, record i = entry m + i
        const uint32_t nb = (m + cnt + RPB - 1u) / RPB;
        const uint4 *ra = (const uint4 *)(r - m);
        const bool inside = PADDED || (off >= m && off - m + (u64)RPB * nb <= nrec_limit);
        auto blk = [&](uint32_t bi) -> uint4 {
            const uint32_t q = bi < nb ? bi : nb - 1u;
            if (PADDED || inside) return ra[q];
            uint32_t e[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (uint32_t j = 0; j < RPB; ++j) {
                const uint32_t i = RPB * q + j - m;
                const uint32_t x = i < cnt ? (uint32_t)r[i] : 0u;
                if (sizeof(REC) == 4) e[j & 3u] = x; else e[j >> 1] |= x << (16u * (j & 1u));
            }
            return make_uint4(e[0], e[1], e[2], e[3]);
        };
        auto feed_block = [&](uint32_t bi, const uint4 &v) {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (uint32_t j = 0; j < RPB; ++j) {
                const uint32_t ej = sizeof(REC) == 4 ? w[j & 3u] : (w[j >> 1] >> (16u * (j & 1u))) & 0xFFFFu;
                const uint32_t i = RPB * bi + j - m;         // wraps below the first record, runs past the last one
                const bool valid = i < cnt, fwd = (ej & 2u) != 0u;
                const uint32_t pos = rel0 + (ej >> 2), can = ej & 1u;
                read_feed(cf, cr, bf, br, Q, valid & (fwd ? walk_fwd : walk_rev), fwd, pos, can);
            }
        };
        uint4 v[NB], w[NB];
#pragma unroll
        for (uint32_t j = 0; j < NB; ++j) v[j] = blk(j);
        for (uint32_t bi = 0; bi < nb; bi += NB) {
#pragma unroll
            for (uint32_t j = 0; j < NB; ++j) w[j] = blk(bi + NB + j);
#pragma unroll
            for (uint32_t j = 0; j < NB; ++j) if (bi + j < nb) feed_block(bi + j, v[j]);
#pragma unroll
            for (uint32_t j = 0; j < NB; ++j) v[j] = w[j];
        }
    }
    if (cf.counts != 0u && read_chain_kept(cf, Q)) read_block_merge(bf, Q, cf.first, cf.last + Q.k - cf.first);
    if (bf.have_cur && bf.clen >= Q.min_block_len) bf.pass = true;
    if (cr.counts != 0u && read_chain_kept(cr, Q)) read_block_merge(br, Q, cr.first, cr.last + Q.k - cr.first);
    if (br.have_cur && br.clen >= Q.min_block_len) br.pass = true;
    return bf.pass || br.pass;
}

// Wave-wide inclusive prefix maximum in 6 DPP steps (row_shr 1/2/4/8 inside each row of 16, then row_bcast:15 into rows
// 1,3 and row_bcast:31 into rows 2,3); lanes outside a shift read 0.
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

// A long match list walked by a whole wave, 64 records per step IN PARALLEL (all arguments wave-uniform).
// Valid when the whole segment is terminal zone (n <= terminal_limit: every read): then the walk never stops
// early, and the two-phase state machine gives the same answer in either direction, so both lists are taken in
// ascending order.  Per batch: the lanes hold the records; a prefix maximum gives every selected record its
// predecessor, a ballot marks the records that start a new sub-block (gap > -k), and the scalar state machine
// then steps once per SUB-BLOCK (counts by popcount of ballots) instead of once per record — a telomeric read
// is one sub-block of thousands of matches.
template <bool FWD_LIST, typename REC = uint32_t>
__device__ __forceinline__ bool pred_scan_wave(const TsTile *tiles, const u64 *tile_off, const uint32_t *tile_stats,
                               const uint32_t *matches, uint32_t t0, uint32_t t1, u64 base,
                               const TsPredParams &Q, uint32_t lane) {
    PredState st = {};
    bool have_prev = false;
    u64 prev = 0;                                            // last selected position so far
    for (uint32_t t = t0; t < t1 && !st.pass; ++t) {
        const uint32_t cnt = tile_stats[4u * t];
        if (cnt == 0u) continue;
        const u64 rel0 = tiles[t].in_off - base;
        const REC *r = (const REC *)matches + tile_off[t];
        for (uint32_t b0 = 0; b0 < cnt; b0 += 64u) {
            const uint32_t nb = cnt - b0 < 64u ? cnt - b0 : 64u;
            const uint32_t rec = lane < nb ? (uint32_t)r[b0 + lane] : 0u;
            const bool sel = lane < nb && (((rec & 2u) != 0u) == FWD_LIST);
            u64 rem = __ballot(sel);
            if (rem == 0ull) continue;
            const uint32_t p32 = rec >> 2;                   // tile-relative position (a batch lies in one tile)
            // predecessor among the selected records of this batch: prefix maximum of (position + 1), one lane down
            const uint32_t incl = wave_scan_max(sel ? p32 + 1u : 0u);
            const uint32_t before = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x138, 0xf, 0xf, false);   // wave_shr:1
            const uint32_t first_lane = (uint32_t)__builtin_ctzll(rem);
            const u64 first_pos = rel0 + (uint32_t)__builtin_amdgcn_readlane((int)p32, (int)first_lane);
            const bool first_head = !have_prev || first_pos - prev > Q.max_match_dist;
            const bool head = sel && (before == 0u ? first_head : p32 - (before - 1u) > Q.max_match_dist);
            const u64 heads = __ballot(head), canon = __ballot(sel && (rec & 1u));
            while (rem) {                                    // one step per run of chained records
                const uint32_t l0 = (uint32_t)__builtin_ctzll(rem);
                if ((heads >> l0) & 1ull) {
                    if (st.in_block) { st.can_cov = st.canon * Q.k; pred_close_sub(st, Q, true); }
                    st.bstart = rel0 + (uint32_t)__builtin_amdgcn_readlane((int)p32, (int)l0);
                    st.counts = 0; st.canon = 0; st.in_block = true;
                }
                const u64 later = l0 < 63u ? heads & ~((2ull << l0) - 1ull) : 0ull;      // heads after l0
                const u64 run = later ? rem & ((1ull << (uint32_t)__builtin_ctzll(later)) - 1ull) : rem;
                st.counts += (uint32_t)__popcll(run);
                st.canon += (uint32_t)__popcll(run & canon);
                const uint32_t last_lane = 63u - (uint32_t)__builtin_clzll(run);
                prev = rel0 + (uint32_t)__builtin_amdgcn_readlane((int)p32, (int)last_lane);
                st.bend = prev + Q.k;
                have_prev = true;
                rem &= ~run;
            }
        }
    }
    if (st.in_block) { st.can_cov = st.canon * Q.k; pred_close_sub(st, Q, true); }
    if (st.have_cur && st.clen >= Q.min_block_len) st.pass = true;
    return st.pass;
}


// One thread per segment.  A read with a long match list (a telomeric read has thousands of chained matches) would leave
// one lane running alone for hundreds of microseconds: such reads are only listed here — long_list[atomicAdd(long_count)]
// — and walked by ts_terminal_predicate_long, one WAVE per listed read, so that they spread over the whole device instead
// of queueing up behind each other in the waves that happen to hold several of them (0.5 % of the reads are long in
// configs[3]: the fullest of 7 800 waves held five, and the kernel took as long as that wave: 544 us per 500 k reads).
// "Long" is relative to what the reads around it carry: more than twice the mean list of the wave's 64 reads (and more than
// Q.long_list = 128).  A launch lasts as long as its slowest wave, i.e. as the longest list a lane walks: with a fixed 384
// against ~70 records per ordinary read a launch took 310 us, four times the walk of an ordinary wave.  Which kernel walks
// a read does not show in the result.
// READS: every segment of the batch is terminal zone as a whole and the records lie in the batch's own padded regions (the
// host checks both): the kernel then holds only the lean
// one-pass walk — 40-odd VGPRs instead of 63, eight waves per SIMD, and room for four blocks in flight per thread.  The walk
// is a chain of dependent scattered loads (a launch of 280 waves takes 150 us: its latency, not its work), so what counts is
// how many loads a thread has in flight and whether all reads of a batch are resident at once.
#ifndef TS_PRED_WAVES
#define TS_PRED_WAVES 8
#endif
#ifndef TS_PRED_NB
#define TS_PRED_NB 4u
#endif
template <bool READS, typename REC = uint32_t>
__global__ __launch_bounds__(64, TS_PRED_WAVES)
void ts_terminal_predicate(const TsTile *tiles, const u64 *tile_off, const uint32_t *tile_stats,
                           const uint32_t *matches, const u64 nrec_limit, const uint32_t *seg_first_tile,
                           const u64 *seg_in_off, const u64 *seg_len, uint32_t nseg,
                           const TsPredParams Q, unsigned char *pass, uint32_t *long_list, uint32_t *long_count,
                           const uint32_t *overflow) {
    // a wave's record region overflowed in the scan (ts_pred_guard): the directory promises records that were never
    // stored — nothing is judged, the flag stays up for ts_batch_read_pass_status
    if (*overflow) return;
    const uint32_t lane = threadIdx.x;                       // one wave per workgroup
    const uint32_t si = blockIdx.x * 64u + lane;
    const bool live = si < nseg;
    const uint32_t t0 = live ? seg_first_tile[si] : 0u, t1 = live ? seg_first_tile[si + 1] : 0u;
    const u64 n = live ? seg_len[si] : 0ull, base = live ? seg_in_off[si] : 0ull;
    u64 total = 0, nfwd = 0;
    for (uint32_t t = t0; t < t1; ++t) { total += tile_stats[4u * t]; nfwd += tile_stats[4u * t + 2u]; }
    // the wave's mean list length (all 64 lanes take part; a lane beyond the batch counts as an empty read)
    uint32_t sum = (uint32_t)(total < 0xFFFFFu ? total : 0xFFFFFu);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += (uint32_t)__shfl_xor((int)sum, d, 64);
    if (!live) return;
    const uint32_t long_from = Q.long_list > sum / 32u ? Q.long_list : sum / 32u;       // twice the mean
    // (the whole-wave walk assumes that the whole segment is terminal zone: every read; otherwise one thread walks it)
    if (total > long_from && n <= Q.terminal_limit) {
        long_list[atomicAdd(long_count, 1u)] = si;
        return;
    }
    bool ok = false;
    if (READS || n <= Q.terminal_limit) {
        if (nfwd >= 2 || total - nfwd >= 2)
            ok = pred_walk_read<READS ? TS_PRED_NB : 2u, READS, REC>(tiles, tile_off, tile_stats, matches, nrec_limit, t0, t1, base, Q, nfwd >= 2, total - nfwd >= 2);
    } else if (!READS) {
        if (nfwd >= 2)                                      // forward list, from the segment start
            ok = pred_walk<true, REC>(tiles, tile_off, tile_stats, matches, nrec_limit, t0, t1, base, n, Q);
        if (!ok && total - nfwd >= 2)                       // reverse list, from the segment end
            ok = pred_walk<false, REC>(tiles, tile_off, tile_stats, matches, nrec_limit, t0, t1, base, n, Q);
    }
    pass[si] = ok ? 1 : 0;
}


// ---------------------------------------------------------------------------------------
// ts_read_predicate_canon: the same answer from a twentieth of the records.  A chain of matches can only become a sub-block
// if it holds a canonical match (finalizeSubBlock, src/teloscope.cpp:75-92: canonicalCount > 0), and a chain that does not
// become one leaves no trace in what follows (phase 2 merges the KEPT sub-blocks, :129-174).  A read batch's scan leaves the
// index of every canonical record among its tile's records (kernels.hip, P.emit == 2): a thread per read visits those —
// seven per 15 kb read of random sequence under the default 38 patterns, of 140 records — and builds, around each one that
// no earlier chain of its list has covered, the chain it belongs to: back and forth over the neighbouring records of the
// same orientation while they are at most -k apart (a record of either orientation more than -k away ends the search:
// positions ascend).  A kept chain is merged into its list's running block exactly as the walk above does it
// (read_chain_kept, read_block_merge: the same u32 / float expressions), in the same order — chains of one list are
// disjoint and are met in position order.  Long lists go to ts_terminal_predicate_long, as before.
struct ReadCursor { uint32_t t, i; };          // a record of the read: tile, index among the tile's records
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));      // four records behind a dword-aligned address

__global__ __launch_bounds__(64, TS_PRED_WAVES)
void ts_read_predicate_canon(const TsTile *tiles, const u64 *tile_off, const uint32_t *tile_stats, const uint32_t *matches,
                             const u64 nrec_limit, const uint32_t *chain, const uint16_t *vis, const uint32_t *seg_first_tile,
                             const u64 *seg_in_off, uint32_t nseg, const TsPredParams Q, unsigned char *pass, uint32_t *long_list,
                             uint32_t *long_count, const uint32_t *overflow) {
    if (*overflow) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t si = blockIdx.x * 64u + lane;
    const bool live = si < nseg;
    const uint32_t t0 = live ? seg_first_tile[si] : 0u, t1 = live ? seg_first_tile[si + 1] : 0u;
    const u64 base = live ? seg_in_off[si] : 0ull;
    u64 total = 0, nfwd = 0;
    for (uint32_t t = t0; t < t1; ++t) { total += tile_stats[4u * t]; nfwd += tile_stats[4u * t + 2u]; }
    uint32_t sum = (uint32_t)(total < 0xFFFFFu ? total : 0xFFFFFu);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sum += (uint32_t)__shfl_xor((int)sum, d, 64);
    if (!live) return;
    const uint32_t long_from = Q.long_list > sum / 32u ? Q.long_list : sum / 32u;       // twice the wave's mean (see ts_terminal_predicate)
    if (total > long_from) {
        long_list[atomicAdd(long_count, 1u)] = si;
        return;
    }
    const bool walk_list[2] = {total - nfwd >= 2, nfwd >= 2};           // [reverse, forward]: a list of fewer than two matches is not walked
    ReadBlock blk[2] = {};
    ReadCursor covered[2] = {{t0, 0u}, {t0, 0u}};                         // per list: the first record no evaluated chain has reached
    auto rec_at = [&](ReadCursor c) -> uint32_t { return matches[tile_off[c.t] + c.i]; };
    auto pos_of = [&](ReadCursor c, uint32_t r) -> uint32_t { return (uint32_t)(tiles[c.t].in_off - base) + (r >> 2); };
    auto step_back = [&](ReadCursor &c) -> bool {                        // to the record before; false at the read's first record
        if (c.i) { --c.i; return true; }
        for (uint32_t t = c.t; t > t0; --t) {
            const uint32_t n = tile_stats[4u * (t - 1u)];
            if (n) { c.t = t - 1u; c.i = n - 1u; return true; }
        }
        return false;
    };
    auto step_on = [&](ReadCursor &c) -> bool {                          // to the record behind; past the read's last record: {t1, 0}, false
        if (c.i + 1u < tile_stats[4u * c.t]) { ++c.i; return true; }
        for (uint32_t t = c.t + 1u; t < t1; ++t)
            if (tile_stats[4u * t]) { c.t = t; c.i = 0u; return true; }
        c.t = t1; c.i = 0u;
        return false;
    };
    auto before = [](ReadCursor a, ReadCursor b) { return a.t < b.t || (a.t == b.t && a.i < b.i); };
    bool ok = false;
    for (uint32_t t = t0; t < t1 && !ok; ++t) {
        const uint32_t nv = tile_stats[4u * t + 3u];
        if (nv == 0u) continue;
        const u64 voff = ((u64)chain[4u * t + 3u] << 32) | chain[4u * t + 2u];
        const uint32_t cnt = tile_stats[4u * t];
        const u64 off = tile_off[t];
        const uint32_t rel0 = (uint32_t)(tiles[t].in_off - base);
        uint32_t idx_next = vis[voff];                                     // (the next index is asked for a record ahead of its use)
        for (uint32_t j = 0; j < nv && !ok; ++j) {
            const uint32_t i = idx_next;
            if (j + 1u < nv) idx_next = vis[voff + j + 1u];
            const ReadCursor c = {t, i};
            // The record and its neighbourhood — four records either side, two 16-byte loads in one round trip: on random sequence
            // a record's neighbours of either orientation are 100 bases apart and -k is 50, so the search for the chain's ends
            // nearly always stops inside it.  Entries outside the tile's records are masked (their tile's neighbour is looked at
            // record by record below, like a chain that runs out of the neighbourhood).
            const u64 g = off + i;
            uint32_t e[8];
            if (g >= 4u && g + 4u <= nrec_limit) {
                const u32x4_a4 wa = *(const u32x4_a4 *)(matches + g - 4u), wb = *(const u32x4_a4 *)(matches + g);
                e[0] = wa.x; e[1] = wa.y; e[2] = wa.z; e[3] = wa.w; e[4] = wb.x; e[5] = wb.y; e[6] = wb.z; e[7] = wb.w;
            } else {
#pragma unroll
                for (uint32_t q = 0; q < 8u; ++q) e[q] = (i + q >= 4u && i + q - 4u < cnt) ? matches[off + i + q - 4u] : 0u;
            }
            const uint32_t r = e[4];
            const uint32_t o = (r >> 1) & 1u;                              // the record's list
            if (!walk_list[o] || before(c, covered[o])) continue;          // inside a chain that was evaluated already
            ReadChain ch = {rel0 + (r >> 2), rel0 + (r >> 2), 1u, 1u};
            // back through the neighbourhood, nearest first; nb: records looked at, far_b: the search met a record too far ahead
            uint32_t nb = 0;
            bool far_b = false, go = true;
#pragma unroll
            for (uint32_t d = 1; d <= 4u; ++d) {
                const uint32_t x = e[4u - d], px = rel0 + (x >> 2);
                const bool in = go & (i >= d);                             // (the tile holds a record there)
                const bool far = in & (ch.first - px > Q.max_match_dist);
                const bool take = in & !far & (((x >> 1) & 1u) == o);
                ch.first = take ? px : ch.first; ch.counts += take ? 1u : 0u; ch.canon += take ? (x & 1u) : 0u;
                nb += in ? 1u : 0u;
                far_b |= far;
                go = in & !far;
            }
            if (!far_b) {                                                  // the neighbourhood (or the tile) ended first: record by record from there
                ReadCursor b = {t, i - nb};
                while (step_back(b)) {
                    const uint32_t rb = rec_at(b), pb = pos_of(b, rb);
                    if (ch.first - pb > Q.max_match_dist) break;          // (any orientation: whatever lies further ahead is further away)
                    if (((rb >> 1) & 1u) != o) continue;
                    ch.first = pb; ++ch.counts; ch.canon += rb & 1u;
                }
            }
            // forth; e_after: the record behind the chain's last one ({t1, 0} when that is the read's last record)
            uint32_t nf = 0, last_at = 0;                                  // records looked at; the neighbourhood offset (1..3) of the chain's last record so far
            bool far_f = false;
            go = true;
#pragma unroll
            for (uint32_t d = 1; d <= 3u; ++d) {
                const uint32_t x = e[4u + d], px = rel0 + (x >> 2);
                const bool in = go & (i + d < cnt);
                const bool far = in & (px - ch.last > Q.max_match_dist);
                const bool take = in & !far & (((x >> 1) & 1u) == o);
                ch.last = take ? px : ch.last; ch.counts += take ? 1u : 0u; ch.canon += take ? (x & 1u) : 0u;
                last_at = take ? d : last_at;
                nf += in ? 1u : 0u;
                far_f |= far;
                go = in & !far;
            }
            ReadCursor e_after = {t, i + last_at};
            if (e_after.i + 1u < cnt) ++e_after.i; else (void)step_on(e_after);   // (the tile's own count is at hand: no load on the usual path)
            if (!far_f) {
                ReadCursor f = {t, i + nf};
                while (step_on(f)) {
                    const uint32_t rf = rec_at(f), pf = pos_of(f, rf);
                    if (pf - ch.last > Q.max_match_dist) break;
                    if (((rf >> 1) & 1u) != o) continue;
                    ch.last = pf; ++ch.counts; ch.canon += rf & 1u;
                    e_after = f;
                    (void)step_on(e_after);
                }
            }
            covered[o] = e_after;
            if (read_chain_kept(ch, Q)) {
                read_block_merge(blk[o], Q, ch.first, ch.last + Q.k - ch.first);
                ok = blk[o].pass;
            }
        }
    }
    for (int o = 0; o < 2; ++o) if (blk[o].have_cur && blk[o].clen >= Q.min_block_len) ok = true;
    pass[si] = ok ? 1 : 0;
}

// One wave per listed read (grid-stride over the list), 64 records per step in parallel (pred_scan_wave).
template <typename REC>
__global__ __launch_bounds__(64)
void ts_terminal_predicate_long(const TsTile *tiles, const u64 *tile_off, const uint32_t *tile_stats,
                                const uint32_t *matches, const uint32_t *seg_first_tile, const u64 *seg_in_off,
                                const TsPredParams Q, unsigned char *pass, const uint32_t *long_list, const uint32_t *long_count) {
    const uint32_t lane = threadIdx.x;
    const uint32_t count = *long_count;
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {
        const uint32_t si = long_list[i];                    // wave-uniform
        const uint32_t t0 = seg_first_tile[si], t1 = seg_first_tile[si + 1];
        const u64 base = seg_in_off[si];
        u64 total = 0, nfwd = 0;
        for (uint32_t t = t0; t < t1; ++t) { total += tile_stats[4u * t]; nfwd += tile_stats[4u * t + 2u]; }
        bool ok = false;
        if (nfwd >= 2)
            ok = pred_scan_wave<true, REC>(tiles, tile_off, tile_stats, matches, t0, t1, base, Q, lane);
        if (!ok && total - nfwd >= 2)
            ok = pred_scan_wave<false, REC>(tiles, tile_off, tile_stats, matches, t0, t1, base, Q, lane);
        if (lane == 0) pass[si] = ok ? 1 : 0;
    }
}


// Raises *flag (sticky) when a wave needed more records than its region holds: with tiles taken on demand the per-wave
// fill differs from launch to launch, so a launch that fitted says nothing about the next.
// (vis_cap != 0: the scan also left the canonical records' indices, a region per wave — wave_fill[nwaves + w] of them)
__global__ void ts_pred_guard(const uint32_t *wave_fill, uint32_t region_cap, uint32_t nwaves, uint32_t *flag, uint32_t vis_cap) {
    bool over = false;
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < nwaves; w += gridDim.x * blockDim.x)
        over |= wave_fill[w] > region_cap || (vis_cap != 0u && wave_fill[nwaves + w] > vis_cap);
    if (over) atomicOr(flag, 1u);
}

}  // namespace

int ts_k_launch_predicate(const TsTile *tiles, const unsigned long long *tile_off, const uint32_t *tile_stats,
                          const uint32_t *matches, unsigned long long nrec_limit, const uint32_t *seg_first_tile,
                          const unsigned long long *seg_in_off, const unsigned long long *seg_len, uint32_t nseg,
                          const TsPredParams *Q, unsigned char *pass, uint32_t *long_list, uint32_t *long_count, int all_terminal,
                          const uint32_t *wave_fill, uint32_t region_cap, uint32_t nwaves, uint32_t *overflow,
                          const uint32_t *chain, const void *canon_idx, uint32_t vis_cap, int rec16, void *stream) {
    if (nseg == 0) return 0;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(long_count, 0, 4, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(ts_pred_guard, dim3(8), dim3(256), 0, st, wave_fill, region_cap, nwaves, overflow, chain ? vis_cap : 0u);
    if (all_terminal && chain && canon_idx)
        hipLaunchKernelGGL(ts_read_predicate_canon, dim3((nseg + 63u) / 64u), dim3(64), 0, st, tiles, tile_off, tile_stats, matches, (u64)nrec_limit, chain,
                           (const uint16_t *)canon_idx, seg_first_tile, seg_in_off, nseg, *Q, pass, long_list, long_count, (const uint32_t *)overflow);
    else if (all_terminal && rec16)                           // (16-bit records: TsScanParams.rec16)
        hipLaunchKernelGGL((ts_terminal_predicate<true, uint16_t>), dim3((nseg + 63u) / 64u), dim3(64), 0, st,
                           tiles, tile_off, tile_stats, matches, (u64)nrec_limit, seg_first_tile, seg_in_off, seg_len, nseg, *Q, pass,
                           long_list, long_count, (const uint32_t *)overflow);
    else if (rec16)
        hipLaunchKernelGGL((ts_terminal_predicate<false, uint16_t>), dim3((nseg + 63u) / 64u), dim3(64), 0, st,
                           tiles, tile_off, tile_stats, matches, (u64)nrec_limit, seg_first_tile, seg_in_off, seg_len, nseg, *Q, pass,
                           long_list, long_count, (const uint32_t *)overflow);
    else if (all_terminal)
        hipLaunchKernelGGL((ts_terminal_predicate<true>), dim3((nseg + 63u) / 64u), dim3(64), 0, st,
                           tiles, tile_off, tile_stats, matches, (u64)nrec_limit, seg_first_tile, seg_in_off, seg_len, nseg, *Q, pass,
                           long_list, long_count, (const uint32_t *)overflow);
    else
        hipLaunchKernelGGL((ts_terminal_predicate<false>), dim3((nseg + 63u) / 64u), dim3(64), 0, st,
                           tiles, tile_off, tile_stats, matches, (u64)nrec_limit, seg_first_tile, seg_in_off, seg_len, nseg, *Q, pass,
                           long_list, long_count, (const uint32_t *)overflow);
    const uint32_t grid = nseg < 8192u ? nseg : 8192u;      // waves of the second kernel: it strides over the list
    if (rec16)
        hipLaunchKernelGGL(ts_terminal_predicate_long<uint16_t>, dim3(grid), dim3(64), 0, st,
                           tiles, tile_off, tile_stats, matches, seg_first_tile, seg_in_off, *Q, pass,
                           (const uint32_t *)long_list, (const uint32_t *)long_count);
    else
        hipLaunchKernelGGL(ts_terminal_predicate_long<uint32_t>, dim3(grid), dim3(64), 0, st,
                           tiles, tile_off, tile_stats, matches, seg_first_tile, seg_in_off, *Q, pass,
                           (const uint32_t *)long_list, (const uint32_t *)long_count);
    return (int)hipGetLastError();
}

