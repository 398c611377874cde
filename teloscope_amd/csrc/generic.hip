// generic.hip — the GENERAL path of libteloscan (gfx950): every parameter set the tiled kernel in kernels.hip
// does not take.  That kernel covers uniform-length pattern sets (3 <= k <= 8) under the geometry where the
// reference's per-window carry loop has a closed form; what is left — mixed-length pattern sets (several matches per
// position), long patterns (up to the 63 bases a ts_pattern holds), and window/step pairs where the reference's `uint32` start index wraps
// (src/teloscope.cpp:413-415) — is restated here LITERALLY (the main / carry attribution of analyzeWindow, not its
// closed form), for a whole batch of segments at a time and with nothing but ordering work left to the host:
//
//   ts_general_fused_list  the LIST form (sets of up to 8 lengths of up to 32 bases whose lists fit LDS): persistent workgroups
//                       stride over tiles of 4096 positions; bases staged into LDS as 2-bit planes; ONE LDS probe per position
//                       from a 6-mer table that holds a bit per pattern length; candidates go to per-wave lists in LDS, and
//                       everything per match — flags (exact tables for lengths <= 6, a binary search above), push test, window
//                       shares, record — runs a lane per candidate on full wavefronts; window records a lane per record.
//   ts_general_fused    the position-strided form of the same pass (tiny steps, pattern lists too large for LDS, dense
//                       tiles whose candidate lists spill).
//   ts_general_wide     the WIDE form: what the two above do not take — up to 63 lengths of up to 63 bases (128-bit codes, a
//                       mask of matched lengths per position).
//   ts_general_compact  the tiles' slots into one dense tile-ordered stream (after a prefix sum over the counts);
//   ts_general_compact_push  the same with the records in the reference's push order (mixed lengths under w > s).
//
// Traffic: 1 B/base in, 32 B/window and 2 x 4 B/match out.  (Rounds 1-2 ran three kernels around a 4 B/base match mask
// in HBM: ~13 B/base.)  Blocks are called on the device (blockcall.hip) for every set: where the reference's calling
// order is not position order, ts_general_compact_push writes the dense stream in that order.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "ts_internal.h"

namespace {

typedef unsigned long long u64;

constexpr uint32_t kTile = TS_GENERAL_TILE;           // positions per tile
constexpr uint32_t kHalo = 32;                        // bases staged beyond it (longest pattern <= 32)
constexpr uint32_t kMaxLdsPatterns = 4096;       // (9 bytes of LDS each: the list form then keeps three workgroups per CU)

// nuc: the A C T G (code order) counts of a lane as four 8-bit fields (a lane visits at most 4096 / 64 positions of a tile:
// no field overflows) — indexed by a shift, where an array indexed by the code went to scratch memory (6 GB of write
// traffic per 3 Gb); then canonical, non-canonical, forward, reverse covered
struct Acc { uint32_t nuc, can, non, fwd, rev; };

__device__ __forceinline__ uint32_t wave_total(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// Inclusive prefix sum over the wave's lanes.
__device__ __forceinline__ uint32_t wave_inclusive(uint32_t v, uint32_t lane) {
#pragma unroll
    for (uint32_t o = 1; o < 64u; o <<= 1) {
        const uint32_t u = (uint32_t)__shfl_up((int)v, (int)o, 64);
        if (lane >= o) v += u;
    }
    return v;
}

// Per tile, what the push test needs of the segment's geometry — computed once (two 64-bit divisions), so that the
// test itself is 32-bit arithmetic relative to the tile.
struct PushGeom {
    u64 P0, n;
    uint32_t s, w, ov, start_index;
    u64 N0;              // ov == 0: n - k0 s      (k0 = P0 / s)
    uint32_t r0;         // ov == 0: P0 - k0 s
    uint32_t r1;         // ov > 0:  Xb - k1 s     (Xb = max(P0, ov) - ov, k1 = Xb / s)
    uint32_t D1;         // ov > 0:  P0 - k1 s     (< w + s)
    uint32_t dsub;       // ov > 0:  what to take off (P0-relative end) to get x - Xb: 0 if P0 > ov else ov - P0
    u64 k0, k1;          // the quotients above
    u64 kP0;             // P0 / s, and
    uint32_t rP0;        // P0 - kP0 s   (both cases)
};

__device__ __forceinline__ PushGeom push_geom(u64 P0, u64 n, const TsGenericGeom &Q) {
    PushGeom g{};
    g.P0 = P0; g.n = n; g.s = Q.s; g.w = Q.w; g.ov = Q.w - Q.s;
    const uint32_t t1 = Q.s - Q.longest, t2 = g.ov - Q.longest;       // uint32 on purpose (src/teloscope.cpp:413-415)
    g.start_index = t1 < t2 ? t1 : t2;
    if (g.ov == 0u) {
        const u64 k0 = P0 / Q.s;
        g.r0 = (uint32_t)(P0 - k0 * Q.s);
        g.N0 = n - k0 * Q.s;
        g.k0 = k0; g.kP0 = k0; g.rP0 = g.r0;
    } else {
        const u64 Xb = P0 > g.ov ? P0 - g.ov : 0ull;
        const u64 k1 = Xb / Q.s;
        g.r1 = (uint32_t)(Xb - k1 * Q.s);
        g.D1 = (uint32_t)(P0 - k1 * Q.s);
        g.dsub = P0 > g.ov ? 0u : (uint32_t)(g.ov - P0);
        g.k1 = k1;
        g.kP0 = P0 / Q.s;
        g.rP0 = (uint32_t)(P0 - g.kP0 * Q.s);
    }
    return g;
}

// Is the match (tile-relative position j, length l) pushed to the reference's match vectors by a full scan?
// (src/teloscope.cpp:485: by the window whose own scan sees it with j >= overlap, or always in window 0 / when
// windows do not overlap; restated from the window loop's index arithmetic, uint32 wrap included.)
// (*rec: the window whose own scan pushes it — the record its covered bases are added to as well)
__device__ __forceinline__ bool full_scan_pushes(uint32_t j, uint32_t l, const PushGeom &g, u64 *rec) {
    if (g.ov == 0u) {
        const uint32_t x = g.r0 + j;
        const uint32_t dk = x / g.s;
        const u64 left = g.N0 - (u64)dk * g.s;
        const uint32_t cws = left < g.w ? (uint32_t)left : g.w;
        *rec = g.k0 + dk;
        return (x - dk * g.s) + l <= cws;                   // may not cross its only window's end
    }
    const u64 e = g.P0 + j + l - 1u;
    *rec = 0;
    if (e < (g.n < g.w ? g.n : (u64)g.w)) return true;      // window 0 scans everything it holds
    const uint32_t xr = g.r1 + (j + l - 1u - g.dsub);       // (e - ov) relative to k1 s   [e >= w here]
    const uint32_t dk = xr / g.s;                           // the one window with j >= overlap: k = k1 + dk
    const long long diff = (long long)g.D1 + (long long)j - (long long)((u64)dk * g.s);
    *rec = g.k1 + dk;
    return diff >= 0 && (u64)diff >= g.start_index;
}

// code (0..3) and validity of tile position q from the packed planes
__device__ __forceinline__ uint32_t plane_code(const uint32_t *codes2, uint32_t q) { return (codes2[q >> 4] >> (2u * (q & 15u))) & 3u; }
__device__ __forceinline__ uint32_t plane_invalid(const uint32_t *inval, uint32_t q) { return (inval[q >> 5] >> (q & 31u)) & 1u; }

// What one analyzeWindow() call over window `kw` adds either to its own record (carry == false: bases with
// i >= overlap, matches with j >= overlap, or everything for window 0 / overlap == 0) or to the NEXT window's record
// (carry == true: i >= step) — restricted to the positions [P0, P0 + ntile) the tile holds in LDS; the wave's lanes
// stride over them.  (src/teloscope.cpp:387-534, the index arithmetic in uint32 as there.)
__device__ __forceinline__ void window_tile_part(const uint32_t *codes2, const uint32_t *inval, const uint32_t *mask,
                                                 const TsGenericPatterns &G, const TsGenericGeom &Q, u64 n, u64 kw, bool carry,
                                                 u64 P0, uint32_t ntile, uint32_t lane, Acc &a) {
    const u64 wstart = kw * Q.s;
    const uint32_t cws = (uint32_t)((n - wstart) < Q.w ? (n - wstart) : Q.w);
    const uint32_t ov = Q.w - Q.s;
    const bool always_main = (ov == 0 || wstart == 0);
    // uint32 arithmetic on purpose (wraps when the longest pattern exceeds step or overlap)
    const uint32_t t1 = Q.s - Q.longest, t2 = ov - Q.longest;
    uint32_t start_index = always_main ? 0u : (t1 < t2 ? t1 : t2);
    if (carry && start_index < Q.s) start_index = Q.s;               // the carry only takes i >= step
    if (start_index >= cws) return;
    const u64 lo = wstart + start_index, hi = wstart + cws;           // segment-relative positions the call visits
    const uint32_t qlo = lo > P0 ? (uint32_t)(lo - P0 < ntile ? lo - P0 : ntile) : 0u;
    const uint32_t qhi = hi > P0 ? (uint32_t)(hi - P0 < ntile ? hi - P0 : ntile) : 0u;
    const uint32_t ioff = (uint32_t)(P0 - wstart);                    // i = q + ioff (mod 2^32: i < cws fits)
    const bool all_nuc = carry || always_main;
    for (uint32_t q = qlo + lane; q < qhi; q += 64u) {
        const uint32_t i = q + ioff;
        if (Q.nuc_on) {
            if (plane_invalid(inval, q)) continue;
            if (all_nuc || i >= ov) a.nuc += 1u << (8u * plane_code(codes2, q));
        }
        const uint32_t m = mask[q] & 0xFFFFFFu;
        if (!m) continue;
        for (uint32_t li = 0; li < G.nlen; ++li) {
            const uint32_t b = (m >> (3u * li)) & 7u;
            if (!(b & 1u)) continue;
            const uint32_t l = G.len[li];
            const uint32_t j = i + l - 1u;
            if (j >= cws) continue;                                 // scanLimit: may not cross the window end
            if (!carry && !(always_main || j >= ov)) continue;
            if (b & 4u) a.can += l; else a.non += l;
            if (b & 2u) a.fwd += l; else a.rev += l;
        }
    }
}

constexpr uint32_t kCodeWords = (kTile + kHalo) / 16u + 4u;          // 2-bit plane, dwords (+ slack: three are read per position)
constexpr uint32_t kInvalWords = (kTile + kHalo) / 32u + 3u;         // validity plane

// ONE pass per tile of 4096 positions of one scanned region (round 3; rounds 1-2 wrote a 4 B/base match mask to HBM
// and read it back three times).  The workgroup
//   1. stages the tile's bases into LDS as a 2-bit plane + a validity plane (coalesced 16-byte loads, a thread packs
//      its 16 bases into one dword) beside the pattern lists (per length: ascending 2-bit codes + {forward,
//      canonical}) and one prefix bitmap per length (which min(l, 6)-mers start a pattern: most positions stop there);
//   2. matches: a position takes its next 32 bases out of the plane with one funnel shift, and for every length looks
//      the l-mer up (bitmap, then binary search in LDS); the result — 3 bits {match, forward, canonical} per length —
//      stays in LDS;
//   3. window records: a wave per (window, part) the tile's positions contribute to, lanes striding over the tile's
//      share of the bases analyzeWindow visits for the window's own record and for the carry into the next one, the
//      eight counters reduced by DPP and ADDED (atomics; the records are zeroed first) to the window's record — a window
//      is the sum of the shares of the one or two tiles it spans;
//   4. match records: the matches the reference pushes (a full scan keeps a match only if some window's own scan
//      pushes it, src/teloscope.cpp:485; a tips-only scan keeps the region's), in position then length order, into the
//      tile's own slot of `records` (slot_cap entries; a tile with more says so in *overflow and the group runs again
//      with slots that cannot overflow), count in tile_stats[4 t].
// Record: (tile-relative position << 5) | (length index << 2) | canonical << 1 | forward.
// Traffic: 1 B/base in, 32 B/window and 4 B/match out.
__global__ __launch_bounds__(256)
void ts_general_fused(const unsigned char *in, const TsGeneralTile *tiles, uint32_t ntiles, const u64 *seg_len,
                      const u64 *seg_win_base, const TsGenericPatterns G, const TsGenericGeom Q, int tips, uint32_t slot_cap,
                      uint32_t lds_patterns, uint32_t *tile_stats, uint32_t *records, uint32_t *win_out, uint32_t *overflow) {
    extern __shared__ __align__(16) unsigned char lds[];
    // layout: mask u32[kTile] | pcode u64[lds_patterns] | bitmap u32[8][128] | part u32[8] | codes2 | inval | pflag u8[lds_patterns]
    uint32_t *mask = (uint32_t *)lds;
    u64 *pcode = (u64 *)(lds + kTile * 4u);
    uint32_t *bitmap = (uint32_t *)(lds + kTile * 4u + (size_t)lds_patterns * 8u);
    uint32_t *part = bitmap + 8u * 128u;
    uint32_t *codes2 = part + 8;
    uint32_t *inval = codes2 + kCodeWords;
    unsigned char *pflag = (unsigned char *)(inval + kInvalWords);
    if (blockIdx.x >= ntiles) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t npat = G.first[G.nlen];
    const bool lds_lists = lds_patterns != 0u;
    for (uint32_t i = tid; i < 8u * 128u; i += 256u) bitmap[i] = 0u;
    if (lds_lists)
        for (uint32_t i = tid; i < npat; i += 256u) { pcode[i] = G.codes[i]; pflag[i] = G.flags[i]; }
    const TsGeneralTile T = tiles[blockIdx.x];
    if (Q.abl & 8u) { if (tid == 0u) *(uint4 *)&tile_stats[4ull * blockIdx.x] = make_uint4(0u, 0u, 0u, 0u); return; }
    // 1. stage: bases [0, avail) of the tile (avail = what lies between its start and the region end, at most
    // kTile + kHalo); layout offsets are 16-byte aligned per segment, tiles start at multiples of kTile inside it.
    // Thread t packs bases [16 t, 16 t + 16) (and the halo's); positions beyond avail are invalid.
    const uint32_t avail = T.avail;
    const unsigned char *src = in + T.in_off;
    for (uint32_t i = tid * 16u; i < kCodeWords * 16u; i += 256u * 16u) {
        uint32_t cw = 0, iv = 0xFFFFu;
        if (i < avail) {
            uint32_t d[4];
            if (((uintptr_t)(src + i) & 15u) == 0u && i + 16u <= avail) {
                const uint4 v = *(const uint4 *)(src + i);
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            } else {
                for (uint32_t q = 0; q < 4u; ++q) {
                    uint32_t x = 0;
                    for (uint32_t r = 0; r < 4u; ++r) x |= (i + 4u * q + r < avail ? (uint32_t)src[i + 4u * q + r] : 0u) << (8u * r);
                    d[q] = x;
                }
            }
            iv = 0;
#pragma unroll
            for (uint32_t q = 0; q < 16u; ++q) {
                uint32_t c = (d[q >> 2] >> (8u * (q & 3u))) & 0xFFu;
                if (Q.fold) c &= 0xDFu;
                const uint32_t code = (c >> 1) & 3u;                               // A 0, C 1, T 2, G 3
                cw |= code << (2u * q);
                iv |= (c != ((0x47544341u >> (8u * code)) & 0xFFu) ? 1u : 0u) << q; // 'A' 'C' 'T' 'G' by code
            }
        }
        codes2[i >> 4] = cw;
        ((unsigned short *)inval)[i >> 4] = (unsigned short)iv;
    }
    __syncthreads();
    for (uint32_t li = 0; li < G.nlen; ++li) {
        const uint32_t q = G.len[li] < 6u ? G.len[li] : 6u;
        for (uint32_t i = G.first[li] + tid; i < G.first[li + 1]; i += 256u) {
            const uint32_t pre = (uint32_t)(lds_lists ? pcode[i] : G.codes[i]) & ((1u << (2u * q)) - 1u);
            atomicOr(&bitmap[li * 128u + (pre >> 5)], 1u << (pre & 31u));
        }
    }
    __syncthreads();
    // 2. matches
    for (uint32_t j = tid; j < T.n; j += 256u) {
        if (Q.abl & 1u) { mask[j] = 0; continue; }
        // the next 32 bases of position j: three plane dwords funnelled by 2 (j mod 16) bits; their validity bits alike
        const uint32_t wd = j >> 4, sh = 2u * (j & 15u);
        const uint32_t c0 = codes2[wd], c1 = codes2[wd + 1u], c2 = codes2[wd + 2u];
        const u64 code64 = (u64)__funnelshift_r(c0, c1, sh) | ((u64)__funnelshift_r(c1, c2, sh) << 32);
        const uint32_t vd = j >> 5;
        const uint32_t inv32 = __funnelshift_r(inval[vd], inval[vd + 1u], j & 31u);
        uint32_t out = 0;
        for (uint32_t li = 0; li < G.nlen; ++li) {
            const uint32_t l = G.len[li];
            if (j + l > avail) break;                // lengths ascend; a match may not cross the region end
            if (inv32 & (l >= 32u ? 0xFFFFFFFFu : ((1u << l) - 1u))) break;   // a non-ACGT base kills this and every longer pattern
            const u64 code = l >= 32u ? code64 : (code64 & ((1ull << (2u * l)) - 1ull));
            const uint32_t q = l < 6u ? l : 6u;
            const uint32_t pre = (uint32_t)code & ((1u << (2u * q)) - 1u);
            if (!((bitmap[li * 128u + (pre >> 5)] >> (pre & 31u)) & 1u)) continue;
            uint32_t lo = G.first[li], hi = G.first[li + 1];
            const uint32_t end = hi;
            while (lo < hi) {                        // binary search in the sorted code list of this length
                const uint32_t mid = (lo + hi) >> 1;
                const u64 cm = lds_lists ? pcode[mid] : G.codes[mid];
                if (cm < code) lo = mid + 1; else hi = mid;
            }
            if (lo < end && (lds_lists ? pcode[lo] : G.codes[lo]) == code)
                out |= (1u | ((uint32_t)(lds_lists ? pflag[lo] : G.flags[lo]) << 1)) << (3u * li);
        }
        mask[j] = out;
    }
    __syncthreads();
    const u64 n = seg_len[T.seg];
    const u64 P0 = T.seg_rel;
    // 3. window records (full scans)
    if (!tips && T.n && !(Q.abl & 2u)) {
        const u64 nwin = (n + Q.s - 1u) / Q.s;
        const u64 kw_lo = P0 >= Q.w ? (P0 - Q.w) / Q.s + 1u : 0u;             // first call whose window reaches the tile
        u64 kw_hi = (P0 + T.n - 1u) / Q.s;                                    // last call that starts inside it
        if (kw_hi >= nwin) kw_hi = nwin - 1u;
        const bool carries = Q.w != Q.s;
        // one item per window RECORD the tile's positions add to: the main part of its own call and the carry of the
        // call before it.  Everything a record holds comes from the positions [R s, R s + w): a record whose span lies
        // inside the tile is STORED (this workgroup is its only writer), one that spans two tiles is added to atomically
        // (device-scope atomics are served at the memory side: eight of them per item were 6 GB of write traffic per 3 Gb)
        u64 rec_hi = kw_hi + (carries ? 1u : 0u);
        if (rec_hi >= nwin) rec_hi = nwin - 1u;
        uint32_t *const wrec = win_out + seg_win_base[T.seg] * 8ull;
        for (u64 R = kw_lo + wave; R <= rec_hi; R += 4u) {
            Acc a = {0, 0, 0, 0, 0};
            window_tile_part(codes2, inval, mask, G, Q, n, R, false, P0, T.n, lane, a);
            if (carries && R > 0u) window_tile_part(codes2, inval, mask, G, Q, n, R - 1u, true, P0, T.n, lane, a);
            // two fields per reduction (a tile's share of a window is at most 4096 bases: 16 bits hold it)
            const uint32_t tAT = wave_total((a.nuc & 0xFFu) | ((a.nuc >> 16) & 0xFFu) << 16);        // A | T << 16
            const uint32_t tCG = wave_total(((a.nuc >> 8) & 0xFFu) | ((a.nuc >> 24) & 0xFFu) << 16);  // C | G << 16
            const uint32_t tcan = wave_total(a.can), tnon = wave_total(a.non), tfwd = wave_total(a.fwd), trev = wave_total(a.rev);
            // A C G T, then the four covered counters
            const uint32_t mine = lane == 0u ? (tAT & 0xFFFFu) : lane == 1u ? (tCG & 0xFFFFu) : lane == 2u ? (tCG >> 16) : lane == 3u ? (tAT >> 16)
                                : lane == 4u ? tcan : lane == 5u ? tnon : lane == 6u ? tfwd : trev;
            const u64 span_lo = R * Q.s;
            const u64 span_hi = span_lo + Q.w < n ? span_lo + Q.w : n;
            const bool sole = span_lo >= P0 && span_hi <= P0 + T.n;
            if (lane < 8u) {
                if (sole) wrec[R * 8ull + lane] = mine;
                else if (mine) atomicAdd(&wrec[R * 8ull + lane], mine);
            }
        }
    }
    // 4. match records: wave v owns the 1024 consecutive positions [1024 v, 1024 v + 1024)
    if (Q.abl & 4u) { if (tid == 0u) *(uint4 *)&tile_stats[4ull * blockIdx.x] = make_uint4(0u, 0u, 0u, 0u); return; }
    const PushGeom pg = push_geom(P0, n, Q);
    uint32_t wave_cnt = 0;
    for (uint32_t r = 0; r < 16u; ++r) {
        const uint32_t j = wave * 1024u + r * 64u + lane;
        const uint32_t v = j < T.n ? (mask[j] & 0xFFFFFFu) : 0u;
        uint32_t keep = 0;
        if (v) {
            u64 unused_rec;
            for (uint32_t li = 0; li < G.nlen; ++li)
                if (((v >> (3u * li)) & 1u) && (tips || full_scan_pushes(j, G.len[li], pg, &unused_rec))) keep |= 1u << li;
            mask[j] = v | (keep << 24);
        }
        wave_cnt += (uint32_t)__popc(keep);
    }
    wave_cnt = wave_total(wave_cnt);
    if (lane == 0u) part[wave] = wave_cnt;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (uint32_t v = 0; v < 4u; ++v) { if (v < wave) base += part[v]; total += part[v]; }
    if (tid == 0u) {
        *(uint4 *)&tile_stats[4ull * blockIdx.x] = make_uint4(total, 0u, 0u, 0u);
        if (total > slot_cap) atomicOr(overflow, 1u);
    }
    if (total > slot_cap || wave_cnt == 0u) return;
    uint32_t *dst = records + (u64)blockIdx.x * slot_cap;
    for (uint32_t r = 0; r < 16u; ++r) {
        const uint32_t j = wave * 1024u + r * 64u + lane;
        const uint32_t v = j < T.n ? mask[j] : 0u;
        const uint32_t keep = v >> 24;
        if (__ballot(keep != 0u) == 0ull) continue;
        const uint32_t c = (uint32_t)__popc(keep);
        const uint32_t incl = wave_inclusive(c, lane);
        uint32_t at = base + incl - c;
        for (uint32_t li = 0; li < G.nlen; ++li)
            if ((keep >> li) & 1u) dst[at++] = (j << 5) | (li << 2) | ((v >> (3u * li + 1u)) & 3u);
        base += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
}

constexpr uint32_t kWaccMax = 256;          // window records a tile may add to on the list path
constexpr uint32_t kListWave = 1024;        // candidate entries a wave's 1024 positions may produce on the list path (u16 each:
                                            // position in the wave's range << 6 | length index << 3 | pushed << 2 | canonical << 1 | forward)
constexpr uint32_t kCumWords = 260;         // per-dword nucleotide prefix sums of a tile (256 + the end sentinel), list path
constexpr uint32_t kShortLen = 6;           // pattern lengths up to this have exact tables in LDS (their l-mers index them)

// Inclusive prefix sum over the wave's lanes by DPP (row shifts, then the two row broadcasts: lane 63 holds the total).
__device__ __forceinline__ uint32_t wave_inclusive_dpp(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return v;
}

// sixteen doubled 2-bit codes (ASCII & 6), one per byte of t[0..3] -> one dword, base i at bits 2i..2i+1
__device__ __forceinline__ uint32_t pack16(const uint32_t t[4]) {
    const uint32_t b0 = __builtin_amdgcn_udot4(t[0], 0x40100401u, 0u, false);
    const uint32_t b1 = __builtin_amdgcn_udot4(t[1], 0x40100401u, 0u, false);
    const uint32_t b2 = __builtin_amdgcn_udot4(t[2], 0x40100401u, 0u, false);
    const uint32_t b3 = __builtin_amdgcn_udot4(t[3], 0x40100401u, 0u, false);
    return ((b0 | (b1 << 8)) >> 1) | ((b2 | (b3 << 8)) << 15);
}

// x / s for the x this kernel divides (x < s + 4200): one multiply by ceil-ish(2^32 / s) when s <= 8192 (exact for
// x < 2^32 / s), a compare above that (x < 2 s).  s >= 2 wherever the list form runs.
__device__ __forceinline__ uint32_t div_step(uint32_t x, uint32_t s, uint32_t magic) {
    return s > 8192u ? (x >= s ? 1u : 0u) : __umulhi(x, magic);
}

// full_scan_pushes with the division above (same arithmetic otherwise)
__device__ __forceinline__ bool full_scan_pushes_m(uint32_t j, uint32_t l, const PushGeom &g, uint32_t magic, u64 *rec) {
    if (g.ov == 0u) {
        const uint32_t x = g.r0 + j;
        const uint32_t dk = div_step(x, g.s, magic), dks = dk * g.s;
        const u64 left = g.N0 - dks;
        const uint32_t cws = left < g.w ? (uint32_t)left : g.w;
        *rec = g.k0 + dk;
        return (x - dks) + l <= cws;                        // may not cross its only window's end
    }
    const u64 e = g.P0 + j + l - 1u;
    *rec = 0;
    if (e < (g.n < g.w ? g.n : (u64)g.w)) return true;      // window 0 scans everything it holds
    const uint32_t xr = g.r1 + (j + l - 1u - g.dsub);       // (e - ov) relative to k1 s   [e >= w here]
    const uint32_t dk = div_step(xr, g.s, magic);           // the one window with j >= overlap: k = k1 + dk
    const long long diff = (long long)g.D1 + (long long)j - (long long)((u64)dk * g.s);
    *rec = g.k1 + dk;
    return diff >= 0 && (u64)diff >= g.start_index;
}

// Global stores and return-less atomics the compiler does not see (round 5; kernels.hip has the same helpers and the full story):
// on gfx9 loads and stores share vmcnt, so with a store pending the compiler waits for the NEXT TILE'S PREFETCH with vmcnt(0) at the
// first register it reuses — in the list kernel a few instructions behind the prefetch's issue.  Nothing in the kernel loads what it
// stored.
__device__ __forceinline__ void gen_store(uint32_t *p, uint32_t v) { asm volatile("global_store_dword %0, %1, off" :: "v"(p), "v"(v)); }
__device__ __forceinline__ void gen_store(uint4 *p, uint4 v) {
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 0" :: "v"(p), "v"(d));       // (s_nop: the data of a store of more than 8 bytes is read a cycle late)
}
__device__ __forceinline__ void gen_add(uint32_t *p, uint32_t v) { asm volatile("global_atomic_add %0, %1, off" :: "v"(p), "v"(v)); }
__device__ __forceinline__ void gen_or(uint32_t *p, uint32_t v) { asm volatile("global_atomic_or %0, %1, off" :: "v"(p), "v"(v)); }

// The same pass in its LIST form (round 3; restructured in round 4): what the kernel above does per position — flag
// lookup, push test, window shares — is done here per CANDIDATE, on full wavefronts, and a position costs ONE LDS probe.
// PERSISTENT workgroups (a few per CU) stride over the tiles: the pattern tables are built once per workgroup, not per tile.
//   0.  once: cand6   a byte per 6-mer: bit li set when its first min(l, 6) bases begin a pattern of length index li — exact
//               for l <= 6 (a pattern shorter than six sets the bit under every extension), a filter for longer ones;
//             sflag   per length <= 6: {forward, canonical} of the l-mer, 2 bits each — those lengths need no search at all;
//   per tile:
//   1'. stage (SWAR: v_perm / v_dot4, four bases per instruction; the next tile's bases are requested before this one's
//       are consumed), and per plane dword the nucleotide counts of its 16 positions, prefix-summed over the tile (DPP);
//   2'. candidates: a lane owns SIXTEEN CONSECUTIVE positions (one plane dword + the next), takes their 6-mers with one
//       v_bfe / v_alignbit each and reads cand6 once per position; the wave appends its candidates (position, length
//       index), in order, to its own list in LDS after ONE prefix sum over the lanes' counts.  Non-ACGT bases and the
//       region's end are handled on a wave-uniform slow path (the bits of lengths that do not fit are dropped);
//   3'. a lane per candidate: flags from sflag, or the binary search for l > 6; the push test (divisions by the step as
//       one multiply); the covered bases of the match ADDED to the accumulators of the window records it belongs to — the
//       pushing window's own record (analyzeWindow's main part) and the record after every call that carries it;
//   4'. window records: a LANE per record — the nucleotides of the (up to two) position ranges a record collects are
//       differences of the prefix sums at their ends; stored (two 16-byte stores) when the tile holds the whole record,
//       added otherwise;
//   5'. match records: the pushed candidates, in list order (= position then length order), into the tile's slot.
// Taken when a tile adds to at most kWaccMax window records (decided on the host from w and s) and the pattern lists fit
// LDS; a wave whose list overflows (more than one candidate per position: dense repeats under a mixed-length set) raises
// overflow bit 1 and the group runs again with the kernel above.
#ifndef TS_GEN_WAVES
#define TS_GEN_WAVES 5
#endif
__global__ __launch_bounds__(256, TS_GEN_WAVES)        // waves per SIMD = workgroups per CU.  74 VGPRs and nothing spilled since round 5 (scalar descriptor loads); five per CU still measure best: six +1.5 %, seven +8 % (profiles/r05/general_list_waits.txt; round 4, at 96 VGPRs with two spilled: profiles/r04/general_occupancy.txt)
void ts_general_fused_list(const unsigned char *in, const TsGeneralTile *tiles, uint32_t ntiles, const u64 *seg_len,
                           const u64 *seg_win_base, const u64 *seg_nwin, const TsGenericPatterns G, const TsGenericGeom Q, int tips, uint32_t slot_cap,
                           uint32_t lds_patterns, uint32_t nshort, uint32_t *tile_stats, uint32_t *records, uint32_t *win_out, uint32_t *overflow) {
    extern __shared__ __align__(16) unsigned char lds[];
    // layout: list u16[4][kListWave] | pcode u64[lds_patterns] | cand6 u8[4096] | sflag u8[nshort][1024] | wacc u32[kWaccMax][4] |
    //         part u32[8] | first u32[12] | wtot u32[2][4] | wbase u32[2][8] | codes2 | inval | valid2 | cum u32[2][kCumWords] | pflag
    unsigned short *list_all = (unsigned short *)lds;
    u64 *pcode = (u64 *)(lds + 4u * kListWave * 2u);
    uint32_t *cand6w = (uint32_t *)(lds + 4u * kListWave * 2u + (((size_t)lds_patterns * 8u + 15u) & ~(size_t)15u));   // (16-byte aligned: wacc is zeroed and read as uint4)
    const unsigned char *cand6 = (const unsigned char *)cand6w;
    uint32_t *sflagw = cand6w + 1024u;
    uint32_t *wacc = sflagw + nshort * 256u;                               // [kWaccMax][4]: canonical, non-canonical, forward, reverse covered
    uint32_t *part = wacc + kWaccMax * 4u;
    uint32_t *firstl = part + 8;
    uint32_t *wtot = firstl + 12;                                          // per wave: nucleotide totals of its 64 plane dwords {V | G << 16, C | T << 16}
    uint32_t *wbase = wtot + 8;                                            // the same summed over the waves before (entries 0..4, twice)
    uint32_t *codes2 = wbase + 16;
    uint32_t *inval = codes2 + kCodeWords;
    uint32_t *valid2 = inval + kInvalWords;
    uint32_t *cumVG = valid2 + kCodeWords;                                 // per plane dword: counts of the dwords before it IN ITS WAVE
    uint32_t *cumCT = cumVG + kCumWords;
    unsigned char *pflag = (unsigned char *)(cumCT + kCumWords);
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    unsigned short *const list = list_all + wave * kListWave;
    if (blockIdx.x >= ntiles) return;
    const uint32_t npat = G.first[G.nlen];
    // 0. the tables, once per workgroup
    for (uint32_t i = tid; i < 1024u + nshort * 256u; i += 256u) cand6w[i] = 0u;       // (cand6 and sflag are adjacent)
    for (uint32_t i = tid; i < npat; i += 256u) { pcode[i] = G.codes[i]; pflag[i] = G.flags[i]; }
    if (tid == 0u) {
#pragma unroll
        for (uint32_t q = 0; q < 9u; ++q) firstl[q] = G.first[q];
        cumVG[256] = 0u; cumCT[256] = 0u;
    }
    // the pattern lengths as six bits each (uniform), for the per-candidate pass
    u64 lens64 = 0;
#pragma unroll
    for (uint32_t li = 0; li < 8u; ++li) lens64 |= (u64)(li < G.nlen ? G.len[li] : 0u) << (6u * li);
    __syncthreads();
    // every pattern once (the first of equal codes, as the search finds it)
    for (uint32_t li = 0; li < G.nlen; ++li) {
        const uint32_t l = G.len[li], q = l < kShortLen ? l : kShortLen, ext_bits = 2u * (kShortLen - q);
        const uint32_t f0 = G.first[li], cnt = G.first[li + 1] - f0;
        for (uint32_t x = tid; x < (cnt << ext_bits); x += 256u) {
            const uint32_t i = f0 + (x >> ext_bits);
            const u64 code = pcode[i];
            if (i > f0 && pcode[i - 1u] == code) continue;
            const uint32_t idx = ((uint32_t)code & ((1u << (2u * q)) - 1u)) | ((x & ((1u << ext_bits) - 1u)) << (2u * q));
            atomicOr(&cand6w[idx >> 2], (1u << li) << (8u * (idx & 3u)));
            if (l <= kShortLen && (x & ((1u << ext_bits) - 1u)) == 0u)
                atomicOr(&sflagw[li * 256u + ((uint32_t)code >> 4)], (uint32_t)(pflag[i] & 3u) << (2u * ((uint32_t)code & 15u)));
        }
    }
    const uint32_t fold_mask = Q.fold ? 0xDFDFDFDFu : 0xFFFFFFFFu;
    const uint32_t magic = Q.s_magic;
    const bool carries = Q.w != Q.s;
    const uint32_t ov = Q.w - Q.s;
    // the next 32 bases of tile position j
    auto bases_at = [&](uint32_t j) -> u64 {
        const uint32_t wd = j >> 4, sh = 2u * (j & 15u);
        const uint32_t c0 = codes2[wd], c1 = codes2[wd + 1u], c2 = codes2[wd + 2u];
        return (u64)__funnelshift_r(c0, c1, sh) | ((u64)__funnelshift_r(c1, c2, sh) << 32);
    };
    // nucleotide counts of the tile positions [0, q), q <= 4096: {V | G << 16, C | T << 16} (V: valid bases)
    auto prefix_counts = [&](uint32_t q, uint32_t &VG, uint32_t &CT) {
        const uint32_t d = q >> 4, m = (1u << (2u * (q & 15u))) - 1u;
        const uint32_t v = valid2[d] & m, w = codes2[d], lo = w & v, hi = (w >> 1) & v;
        const uint32_t g = (uint32_t)__popc(lo & hi);
        VG = cumVG[d] + wbase[d >> 6] + ((uint32_t)__popc(v) | (g << 16));
        CT = cumCT[d] + wbase[8u + (d >> 6)] + (((uint32_t)__popc(lo) - g) | (((uint32_t)__popc(hi) - g) << 16));
    };
    // a thread stages the plane dword tid (bases [16 tid, 16 tid + 16)); threads 0..5 also the halo's dwords 256 + tid.  The
    // layout's regions start on 16-byte boundaries and tiles at multiples of 4096 inside them (the host checks): aligned loads;
    // the layout keeps 64 readable bytes behind the last region.
    auto load16 = [&](const TsGeneralTile &T, uint32_t i) -> uint4 {
        return i < T.avail ? *(const uint4 *)(in + T.in_off + i) : make_uint4(0u, 0u, 0u, 0u);
    };
    auto stage16 = [&](const uint4 v, uint32_t i, uint32_t avail, uint32_t &vg, uint32_t &ct) {
        uint32_t cw = 0, iv = 0xFFFFu;
        if (i < avail) {
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
            // ASCII & 6 = twice the code (A 0, C 2, T 4, G 6) and the v_perm selector of the letter that code stands for; a byte
            // that is not that letter (up to case folding) is invalid
            uint32_t t[4], b4[4];
#pragma unroll
            for (uint32_t q = 0; q < 4u; ++q) {
                t[q] = d[q] & 0x06060606u;
                const uint32_t e = __builtin_amdgcn_perm(0xFF47FF54u, 0xFF43FF41u, t[q]);
                const uint32_t dd = (d[q] & fold_mask) ^ e;
                const uint32_t nz = ((((dd & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | dd) & 0x80808080u) >> 7;
                b4[q] = __builtin_amdgcn_udot4(nz, 0x08040201u, 0u, false);
            }
            cw = pack16(t);
            iv = b4[0] | (b4[1] << 4) | (b4[2] << 8) | (b4[3] << 12);
            if (i + 16u > avail) iv = (iv | (~0u << (avail - i))) & 0xFFFFu;       // bases behind the region's end
        }
        codes2[i >> 4] = cw;
        ((unsigned short *)inval)[i >> 4] = (unsigned short)iv;
        uint32_t ok = ~iv & 0xFFFFu;
        ok = (ok | (ok << 8)) & 0x00FF00FFu; ok = (ok | (ok << 4)) & 0x0F0F0F0Fu;
        ok = (ok | (ok << 2)) & 0x33333333u; ok = (ok | (ok << 1)) & 0x55555555u;
        valid2[i >> 4] = ok;
        const uint32_t lo = cw & ok, hi = (cw >> 1) & ok, g = (uint32_t)__popc(lo & hi);
        vg = (uint32_t)__popc(ok) | (g << 16);
        ct = ((uint32_t)__popc(lo) - g) | (((uint32_t)__popc(hi) - g) << 16);
    };
    // tile descriptors by scalar loads (the table is written by the host only): as vector loads they sat behind a vmcnt wait in
    // front of the very prefetch they address
    typedef const TsGeneralTile __attribute__((address_space(4))) *ConstTile;
    const ConstTile ctiles = (ConstTile)(uintptr_t)tiles;
    auto tile_at = [&](uint32_t t) {
        TsGeneralTile X;
        X.in_off = ctiles[t].in_off; X.seg_rel = ctiles[t].seg_rel; X.k_p0 = ctiles[t].k_p0;
        X.n = ctiles[t].n; X.avail = ctiles[t].avail; X.seg = ctiles[t].seg; X.r_p0 = ctiles[t].r_p0;
        return X;
    };
    typedef const u64 __attribute__((address_space(4))) *ConstU64;        // (the per-segment tables too: host-written, uniform reads)
    const ConstU64 c_seg_len = (ConstU64)(uintptr_t)seg_len, c_seg_win_base = (ConstU64)(uintptr_t)seg_win_base, c_seg_nwin = (ConstU64)(uintptr_t)seg_nwin;
    uint32_t tile = blockIdx.x;
    TsGeneralTile T = tile_at(tile);
    uint4 v_main = load16(T, tid * 16u), v_halo = tid < kCodeWords - 256u ? load16(T, 4096u + tid * 16u) : make_uint4(0u, 0u, 0u, 0u);
    for (;;) {
        // 1'. stage
        const uint32_t avail = T.avail;
        {
            uint32_t vg, ct, hv = 0, hc = 0;
            for (uint32_t i = tid; i < kWaccMax; i += 256u) *(uint4 *)&wacc[4u * i] = make_uint4(0u, 0u, 0u, 0u);
            stage16(v_main, tid * 16u, avail, vg, ct);
            if (tid < kCodeWords - 256u) stage16(v_halo, 4096u + tid * 16u, avail, hv, hc);
            (void)hv; (void)hc;
            // the wave's prefix sums of the counts (each field <= 4096: the 16-bit halves do not carry into each other)
            const uint32_t ivg = wave_inclusive_dpp(vg), ict = wave_inclusive_dpp(ct);
            cumVG[tid] = ivg - vg; cumCT[tid] = ict - ct;
            if (lane == 63u) { wtot[wave] = ivg; wtot[4u + wave] = ict; }
        }
        // the next tile's bases are on their way while this one is worked on
        const uint32_t tile_next = tile + gridDim.x;
        const bool more = tile_next < ntiles;
        TsGeneralTile Tn = T;
        if (more) {
            Tn = tile_at(tile_next);
            v_main = load16(Tn, tid * 16u);
            if (tid < kCodeWords - 256u) v_halo = load16(Tn, 4096u + tid * 16u);
        }
        __syncthreads();
        if (tid < 5u) {                                                    // sums over the waves before (entry 4: the whole tile)
            uint32_t a = 0, b = 0;
            for (uint32_t v = 0; v < tid; ++v) { a += wtot[v]; b += wtot[4u + v]; }
            wbase[tid] = a; wbase[8u + tid] = b;
        }
        // 2'. candidates: wave v owns the 1024 consecutive positions [1024 v, 1024 v + 1024), lane L the sixteen from 16 L on
        uint32_t ncand = 0;                                                    // (wave-uniform)
        bool spilled = false;
        if (!(Q.abl & 16u)) {
            const uint32_t wd0 = wave * 64u + lane, j0 = wd0 * 16u;
            const uint32_t c0 = codes2[wd0], c1 = codes2[wd0 + 1u];
            uint32_t acc[4] = {0u, 0u, 0u, 0u};                                // a byte of length bits per position
#pragma unroll
            for (uint32_t i = 0; i < 16u; ++i) {
                const uint32_t idx = (i <= 10u ? (c0 >> (2u * i)) : __builtin_amdgcn_alignbit(c1, c0, 2u * i)) & 0xFFFu;
                acc[i >> 2] |= (uint32_t)cand6[idx] << (8u * (i & 3u));
            }
            // non-ACGT bases within reach of the wave's positions, or the region's end: lengths that do not fit are dropped
            const unsigned short *inval16 = (const unsigned short *)inval;
            const u64 iv48 = (u64)inval16[wd0] | ((u64)inval16[wd0 + 1u] << 16) | ((u64)inval16[wd0 + 2u] << 32);
            const bool slow = wave * 1024u + 1024u + 32u > avail || __builtin_amdgcn_ballot_w64(iv48 != 0ull) != 0ull;
            if (slow) {
#pragma unroll 1
                for (uint32_t i = 0; i < 16u; ++i) {
                    const uint32_t ivi = (uint32_t)(iv48 >> i), j = j0 + i;
                    const uint32_t nvalid = ivi ? (uint32_t)__builtin_ctz(ivi) : 32u;
                    const uint32_t rem = avail > j ? avail - j : 0u;
                    const uint32_t maxlen = nvalid < rem ? nvalid : rem;
                    uint32_t cnt = 0;
#pragma unroll
                    for (uint32_t li = 0; li < 8u; ++li) cnt += ((uint32_t)(lens64 >> (6u * li)) & 63u) - 1u < maxlen ? 1u : 0u;   // 1 <= len <= maxlen (len 0: unused slot)
                    const uint32_t keep = ~(0xFFu << cnt) & 0xFFu;
                    const uint32_t r = i >> 2, sh = 8u * (i & 3u);
                    // (acc is indexed by constants only: no scratch)
                    if (r == 0u) acc[0] &= ~(0xFFu << sh) | (keep << sh);
                    else if (r == 1u) acc[1] &= ~(0xFFu << sh) | (keep << sh);
                    else if (r == 2u) acc[2] &= ~(0xFFu << sh) | (keep << sh);
                    else acc[3] &= ~(0xFFu << sh) | (keep << sh);
                }
            }
            const uint32_t c = (uint32_t)(__popc(acc[0]) + __popc(acc[1]) + __popc(acc[2]) + __popc(acc[3]));
            const uint32_t incl = wave_inclusive_dpp(c);
            ncand = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            if (ncand > kListWave) { spilled = true; ncand = 0; }
            else {
                uint32_t at = incl - c;
#pragma unroll
                for (uint32_t r = 0; r < 4u; ++r)
                    for (uint32_t m = acc[r]; m; m &= m - 1u) {
                        const uint32_t b = (uint32_t)__builtin_ctz(m);
                        list[at++] = (unsigned short)(((lane * 16u + 4u * r + (b >> 3)) << 6) | ((b & 7u) << 3));
                    }
            }
        }
        if (spilled && lane == 0u) gen_or(overflow, 2u);
        const u64 n = c_seg_len[T.seg];
        const u64 P0 = T.seg_rel;
        const u64 N1 = n - P0;                                                  // bases from the tile's first to the segment's end
        // window geometry of the tile: the calls whose windows reach it, the records it adds to.  No 64-bit division: the
        // host hands over P0 = k_p0 s + r_p0 per tile and w = cw s + rw per call, the rest are sums and compares
        const bool win_on = !tips && T.n;
        u64 nwin = 0, kw_lo = 0, kw_hi = 0, rec_hi = 0;
        if (win_on) {
            nwin = c_seg_nwin[T.seg];
            kw_lo = P0 >= Q.w ? T.k_p0 + 1u - Q.cw - (Q.rw > T.r_p0 ? 1u : 0u) : 0u;     // first call whose window reaches the tile: (P0 - w) / s + 1
            kw_hi = T.k_p0 + div_step(T.r_p0 + T.n - 1u, Q.s, magic);         // last call that starts inside it
            if (kw_hi >= nwin) kw_hi = nwin - 1u;
            rec_hi = kw_hi + (carries ? 1u : 0u);
            if (rec_hi >= nwin) rec_hi = nwin - 1u;
            if (rec_hi - kw_lo >= kWaccMax) { if (tid == 0u) gen_or(overflow, 2u); spilled = true; }     // (the host sizes this out: never)
        }
        PushGeom pg{};
        {
            pg.P0 = P0; pg.n = n; pg.s = Q.s; pg.w = Q.w; pg.ov = ov;
            const uint32_t t1 = Q.s - Q.longest, t2 = ov - Q.longest;        // uint32 on purpose (src/teloscope.cpp:413-415)
            pg.start_index = t1 < t2 ? t1 : t2;
            pg.kP0 = T.k_p0; pg.rP0 = T.r_p0;
            if (ov == 0u) {
                pg.k0 = T.k_p0; pg.r0 = T.r_p0;
                pg.N0 = N1 + T.r_p0;                                          // n - k0 s
            } else if (P0 > ov) {
                // P0 - ov = (k_p0 - cw + 1) s + (r_p0 - rw)
                const bool borrow = T.r_p0 < Q.rw;
                pg.k1 = T.k_p0 + 1u - Q.cw - (borrow ? 1u : 0u);
                pg.r1 = T.r_p0 - Q.rw + (borrow ? Q.s : 0u);
                pg.D1 = pg.r1 + ov;                                           // P0 - k1 s
                pg.dsub = 0u;
            } else {
                pg.k1 = 0u; pg.r1 = 0u; pg.D1 = (uint32_t)P0; pg.dsub = (uint32_t)(ov - P0);
            }
        }
        // 3'. a lane per candidate
        uint32_t npush = 0;
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (uint32_t e0 = 0; e0 < ncand && !spilled && !(Q.abl & 32u); e0 += 64u) {
            const uint32_t e = e0 + lane;
            bool pushed = false;
            if (e < ncand) {
                uint32_t ent = list[e];
                const uint32_t j = wave * 1024u + (ent >> 6), li = (ent >> 3) & 7u, l = (uint32_t)(lens64 >> (6u * li)) & 63u;
                const u64 code64 = bases_at(j);
                const u64 code = l >= 32u ? code64 : (code64 & ((1ull << (2u * l)) - 1ull));
                bool found = true;
                uint32_t fl;
                if (l <= kShortLen) {                                           // exact tables: a candidate is a match
                    const uint32_t cidx = (uint32_t)code;
                    fl = (sflagw[li * 256u + (cidx >> 4)] >> (2u * (cidx & 15u))) & 3u;
                } else {
                    uint32_t lo = firstl[li], hi = firstl[li + 1u];
                    const uint32_t end = hi;
                    while (lo < hi) {                    // binary search in the sorted code list of this length
                        const uint32_t mid = (lo + hi) >> 1;
                        if (pcode[mid] < code) lo = mid + 1; else hi = mid;
                    }
                    found = lo < end && pcode[lo] == code;
                    fl = found ? pflag[lo] : 0u;                                // bit0 forward, bit1 canonical
                }
                if (found) {
                    u64 rec = 0;
                    pushed = tips || full_scan_pushes_m(j, l, pg, magic, &rec);
                    if (win_on) {
                        const uint32_t f_can = (fl & 2u) ? 0u : 1u, f_fwd = (fl & 1u) ? 2u : 3u;
                        // the window that pushes a match counts it in its own record (analyzeWindow's main part) ...
                        if (pushed) {
                            uint32_t *a4 = wacc + (uint32_t)(rec - kw_lo) * 4u;
                            atomicAdd(a4 + f_can, l);
                            atomicAdd(a4 + f_fwd, l);
                        }
                        // ... and every call kw that meets it at i >= step (at or behind its own start index) carries it into
                        // record kw + 1 as long as it ends inside that call's window: kw = p / s - 1 downwards, i grows by s
                        if (carries) {
                            const uint32_t x = pg.rP0 + j, dkp = div_step(x, Q.s, magic);
                            u64 kw = pg.kP0 + dkp;
                            uint32_t i = x - dkp * Q.s;                          // (< w + s while the loop runs)
                            while (kw > 0u) {
                                --kw; i += Q.s;
                                if (i + l - 1u >= Q.w) break;                   // ends behind every further window too
                                const uint32_t si = kw == 0u ? Q.s : (pg.start_index > Q.s ? pg.start_index : Q.s);
                                if (i < si) continue;
                                const u64 left = N1 - j + i;                    // n - kw s
                                const u64 cws = left < Q.w ? left : Q.w;
                                if (i + l - 1u < cws && kw + 1u <= rec_hi) {
                                    uint32_t *a4 = wacc + (uint32_t)(kw + 1u - kw_lo) * 4u;
                                    atomicAdd(a4 + f_can, l);
                                    atomicAdd(a4 + f_fwd, l);
                                }
                            }
                        }
                    }
                    ent |= fl & 3u;
                }
                list[e] = (unsigned short)(pushed ? (ent | 4u) : 0u);
            }
            npush += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(pushed));
        }
        if (lane == 0u) { part[wave] = npush; part[4u + wave] = spilled ? 1u : 0u; }
        __syncthreads();
        uint32_t base = 0, total = 0;
        for (uint32_t v = 0; v < 4u; ++v) { if (v < wave) base += part[v]; total += part[v]; }
        // did any wave's list spill?  then nothing of this tile counts: the group runs again with the kernel above
        const bool any_spill = (part[4] | part[5] | part[6] | part[7]) != 0u;
        if (tid == 0u) {
            gen_store((uint4 *)&tile_stats[4ull * tile], make_uint4(total, 0u, 0u, 0u));
            if (total > slot_cap) gen_or(overflow, 1u);
        }
        // 4'. window records: thread r takes record kw_lo + r.  Positions relative to the tile's first base, 32-bit signed (the
        // host keeps w below 2^28 on this path; a segment end farther away than that is as good as infinitely far)
        if (win_on && !any_spill && !(Q.abl & 64u) && (u64)tid <= rec_hi - kw_lo) {
            const u64 R = kw_lo + tid;
            const int32_t rel = (int32_t)((long long)(kw_lo * Q.s) - (long long)P0) + (int32_t)(tid * Q.s);   // R s - P0
            const int32_t N1c = N1 > 0x3FFF0000ull ? 0x3FFF0000 : (int32_t)N1;
            const int32_t left = N1c - rel;                                  // n - R s (>= 1: R < nwin)
            const int32_t tn = (int32_t)T.n;
            uint32_t VG = 0, CT = 0;
            auto add_range = [&](int32_t a, int32_t b) {                     // tile positions [a, b) clipped to the tile
                const uint32_t qa = (uint32_t)(a < 0 ? 0 : (a > tn ? tn : a)), qb = (uint32_t)(b < 0 ? 0 : (b > tn ? tn : b));
                if (qa >= qb) return;
                uint32_t va, ca, vb, cb;
                prefix_counts(qa, va, ca);
                prefix_counts(qb, vb, cb);
                VG += vb - va; CT += cb - ca;
            };
            if (Q.nuc_on) {
                {                                                           // main part of call R
                    const uint32_t cws = left < (int32_t)Q.w ? (uint32_t)left : Q.w;
                    const bool always_main = ov == 0u || R == 0u;
                    const uint32_t from = always_main ? 0u : (pg.start_index > ov ? pg.start_index : ov);
                    if (from < cws) add_range(rel + (int32_t)from, rel + (int32_t)cws);
                }
                if (carries && R > 0u) {                                    // carry of call R - 1: i >= max(its start index, step)
                    const int32_t left1 = left + (int32_t)Q.s;
                    const uint32_t cws = left1 < (int32_t)Q.w ? (uint32_t)left1 : Q.w;
                    const uint32_t si = R - 1u == 0u ? 0u : pg.start_index;
                    const uint32_t from = si > Q.s ? si : Q.s;
                    if (from < cws) add_range(rel - (int32_t)Q.s + (int32_t)from, rel - (int32_t)Q.s + (int32_t)cws);
                }
            }
            const uint32_t tC = CT & 0xFFFFu, tT = CT >> 16, tG = VG >> 16, tA = (VG & 0xFFFFu) - tC - tT - tG;
            const uint4 cov = *(const uint4 *)&wacc[4u * tid];
            uint32_t *const wr = win_out + (c_seg_win_base[T.seg] + R) * 8ull;
            const int32_t span_hi = rel + (left < (int32_t)Q.w ? left : (int32_t)Q.w);
            if (rel >= 0 && span_hi <= tn) {                                 // the tile holds the whole record: this thread is its only writer
                gen_store((uint4 *)wr, make_uint4(tA, tC, tG, tT));
                gen_store((uint4 *)(wr + 4), cov);
            } else {
                if (tA) gen_add(wr + 0, tA);
                if (tC) gen_add(wr + 1, tC);
                if (tG) gen_add(wr + 2, tG);
                if (tT) gen_add(wr + 3, tT);
                if (cov.x) gen_add(wr + 4, cov.x);
                if (cov.y) gen_add(wr + 5, cov.y);
                if (cov.z) gen_add(wr + 6, cov.z);
                if (cov.w) gen_add(wr + 7, cov.w);
            }
        }
        // 5'. match records
        if (!any_spill && total <= slot_cap && npush != 0u) {
            uint32_t *dst = records + (u64)tile * slot_cap;
#pragma unroll 1
            for (uint32_t e0 = 0; e0 < ncand; e0 += 64u) {
                const uint32_t e = e0 + lane;
                const uint32_t ent = e < ncand ? list[e] : 0u;
                const u64 m = __builtin_amdgcn_ballot_w64((ent & 4u) != 0u);
                if (m == 0ull) continue;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                // the stream's record: tile position << 5 | length index << 2 | canonical << 1 | forward
                if (ent & 4u) gen_store(dst + (base + rank), ((wave * 1024u + (ent >> 6)) << 5) | (((ent >> 3) & 7u) << 2) | (ent & 3u));
                base += (uint32_t)__popcll(m);
            }
        }
        if (!more) break;
        tile = tile_next;
        T = Tn;
        __syncthreads();                                                   // planes, lists and accumulators are rewritten next
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The WIDE form (round 4; rebuilt in round 5): pattern sets beyond the two forms above — more than 8 distinct lengths or a pattern
// longer than 32 bases, up to the 63 lengths of up to 63 bases a ts_pattern[] can hold (the reference's trie has no limit of its
// own, include/teloscope.h:40-57).
//   * 128-bit codes (lo: bases 0..31, hi: 32..62) searched in (lo, hi)-sorted lists (in LDS when they fit), behind a prefix table
//     in LDS: pre6, an entry per 6-mer whose bit li says "these bases begin a pattern of length index li" (16, 32 or 64 bits wide);
//   * per position three masks in LDS — matched lengths, which of them are forward, which canonical (16, 32 or 64 bits);
//   * a halo of 64 bases, and records that carry six bits of length index: position << 8 | index << 2 | canonical << 1 | forward.
// Persistent workgroups (three / two / one per CU by the width of the masks: 47 / 66 / 132 KB of LDS), the tables built once.  Per tile:
//   1. stage: plane dwords by v_perm / v_dot4 and per-dword nucleotide prefix sums, as the list form; the next tile's bases are
//      requested before this one is worked on;
//   2. candidates: a lane looks pre6 up for its sixteen consecutive positions and the wave lists, in position order, the few
//      positions whose next six bases begin a pattern; then a LANE PER LISTED POSITION builds the 128-bit code, finds the reach to
//      the next non-ACGT base and searches the lengths pre6 named (round 4 did all of that for every position);
//   3. window records: a lane per listed position adds the covered bases of its matches to LDS accumulators of the records they
//      belong to (main part of record kw, carry of record kw + 1 — analyzeWindow's index arithmetic turned round), then a lane per
//      record adds the nucleotide counts of its two parts from the prefix sums and writes it; a tile that adds to more than 64
//      records (tiny steps) keeps round 4's wave per record part, over a bitmap of the match positions;
//   4. match records: the push test and the record per listed position, in list order (= position, then length).
// profiles/r05/general_unordered_device_blocks.txt has every step with its time: 41.75 -> 12.78 ms per 3 Gb on a nine-length set.
constexpr uint32_t kWideCodeWords = (kTile + TS_WIDE_HALO) / 16u + 6u;     // 2-bit plane, dwords (five are read per position)
constexpr uint32_t kWideInvalWords = (kTile + TS_WIDE_HALO) / 32u + 4u;    // validity plane (three are read per position)
constexpr uint32_t kWideCumWords = 260;                                     // per-dword nucleotide prefix sums (256 + the end sentinel)
constexpr uint32_t kWideWacc = 64;                                          // window records a tile may add to with a lane per record (more: a wave per record part)

struct WideAcc { uint32_t nuc, can, non, fwd, rev; };

// The part of window call kw that lies in a tile, for the wide form: covered bases of its matches into `a` (per lane, summed by
// the caller), the nucleotide counts of the part as a wave-uniform pair {A | T << 16, C | G << 16} added to nAT / nCG.
// Round 5: the positions that hold a match come from a bitmap of the tile (a lane takes 32 positions and visits the set bits:
// one position in eighty holds a match on random sequence), and the nucleotide counts are two lookups in per-dword prefix sums —
// the round-4 loop visited every position of every window part, three to four times per base, and was half the kernel.
template <typename M, typename PC, bool NUC_ONLY = false>
__device__ __forceinline__ void window_tile_part_wide(const uint32_t *hitmap, const M *hit, const M *fwdm, const M *canm,
                                                      const uint32_t *lens, const TsGenericGeom &Q, u64 n, u64 kw, bool carry,
                                                      u64 P0, uint32_t ntile, uint32_t lane, WideAcc &a, PC prefix_counts,
                                                      uint32_t &nAT, uint32_t &nCG) {
    const u64 wstart = kw * Q.s;
    const uint32_t cws = (uint32_t)((n - wstart) < Q.w ? (n - wstart) : Q.w);
    const uint32_t ov = Q.w - Q.s;
    const bool always_main = (ov == 0 || wstart == 0);
    const uint32_t t1 = Q.s - Q.longest, t2 = ov - Q.longest;           // uint32 on purpose (src/teloscope.cpp:413-415)
    uint32_t start_index = always_main ? 0u : (t1 < t2 ? t1 : t2);
    if (carry && start_index < Q.s) start_index = Q.s;               // the carry only takes i >= step
    if (start_index >= cws) return;
    const u64 lo = wstart + start_index, hi = wstart + cws;
    const uint32_t qlo = lo > P0 ? (uint32_t)(lo - P0 < ntile ? lo - P0 : ntile) : 0u;
    const uint32_t qhi = hi > P0 ? (uint32_t)(hi - P0 < ntile ? hi - P0 : ntile) : 0u;
    if (qlo >= qhi) return;
    const uint32_t ioff = (uint32_t)(P0 - wstart);
    const bool all_nuc = carry || always_main;
    if (Q.nuc_on) {
        // valid bases at window indices i >= ov (all of the part for a carry or a first window): tile positions [nlo, qhi)
        uint32_t nlo = qlo;
        if (!all_nuc) {
            const long long b = (long long)ov + (long long)wstart - (long long)P0;      // i >= ov  <=>  q >= b
            if (b > (long long)qlo) nlo = b >= (long long)qhi ? qhi : (uint32_t)b;
        }
        if (nlo < qhi) {
            uint32_t vg1, ct1, vg0, ct0;
            prefix_counts(qhi, vg1, ct1);
            prefix_counts(nlo, vg0, ct0);
            const uint32_t vg = vg1 - vg0, ct = ct1 - ct0;               // {valid | G << 16}, {C | T << 16}: fields never borrow
            const uint32_t G_ = vg >> 16, C_ = ct & 0xFFFFu, T_ = ct >> 16, A_ = (vg & 0xFFFFu) - G_ - C_ - T_;
            nAT += A_ | (T_ << 16);
            nCG += C_ | (G_ << 16);
        }
    }
    if (NUC_ONLY) return;                            // (a lane per record: the matches' share is added up per match)
    for (uint32_t wd = (qlo >> 5) + lane; wd <= ((qhi - 1u) >> 5); wd += 64u) {
        uint32_t bits = hitmap[wd];
        const uint32_t q0 = wd << 5;
        if (q0 < qlo) bits &= ~0u << (qlo - q0);
        if (q0 + 32u > qhi) bits &= ~0u >> (q0 + 32u - qhi);
        for (; bits; bits &= bits - 1u) {
            const uint32_t q = q0 + (uint32_t)__builtin_ctz(bits);
            const uint32_t i = q + ioff;
            u64 m = hit[q];
            const u64 f = fwdm[q], c = canm[q];
            for (; m; m &= m - 1ull) {
                const uint32_t li = (uint32_t)__builtin_ctzll(m);
                const uint32_t l = lens[li];
                const uint32_t j = i + l - 1u;
                if (j >= cws) continue;                                 // scanLimit: may not cross the window end
                if (!carry && !(always_main || j >= ov)) continue;
                if ((c >> li) & 1ull) a.can += l; else a.non += l;
                if ((f >> li) & 1ull) a.fwd += l; else a.rev += l;
            }
        }
    }
}

// M: the per-position masks' type (uint16_t for up to 16 lengths, uint32_t for up to 32, u64 for up to 63: LDS per workgroup
// 40 / 64 / 130 KB, i.e. three / two / one workgroup per CU); P: the prefix table's entry type (uint32_t or u64)
template <typename M, typename P>
__global__ __launch_bounds__(256)
void ts_general_wide(const unsigned char *in, const TsGeneralTile *tiles, uint32_t ntiles, const u64 *seg_len,
                     const u64 *seg_win_base, const u64 *seg_nwin, const TsWidePatterns W, const TsGenericGeom Q, int tips, uint32_t slot_cap, uint32_t lds_pat,
                     uint32_t *tile_stats, uint32_t *records, uint32_t *win_out, uint32_t *overflow) {
    extern __shared__ __align__(16) unsigned char lds[];
    // layout: hit M[kTile] | fwd M[kTile] | can M[kTile] | pre6 P[4096] | plo u64[lds_pat] | phi u64[lds_pat] | lens u32[64] | first u32[68] |
    //         part u32[8] | wtot u32[8] | wbase u32[16] | hitmap u32[128] | cumVG u32[260] | cumCT u32[260] | codes2 | inval | valid2 | lists u16[4][1024] | wacc u32[64][4] | pfl u8[lds_pat]
    //         (lds_pat: the pattern lists in LDS when they fit — a search step is then an LDS read; out of device memory the searches of the
    //         few lanes that hold a candidate were 80 % of the kernel's time)
    M *hit = (M *)lds;
    M *fwdm = hit + kTile;
    M *canm = fwdm + kTile;
    P *pre6 = (P *)(canm + kTile);
    u64 *plo = (u64 *)(pre6 + 4096u);
    u64 *phi = plo + lds_pat;
    uint32_t *lens = (uint32_t *)(phi + lds_pat);
    uint32_t *first = lens + 64;
    uint32_t *part = first + 68;
    uint32_t *wtot = part + 8;                       // per wave: nucleotide totals of its 64 plane dwords {valid | G << 16, C | T << 16}
    uint32_t *wbase = wtot + 8;                      // the same summed over the waves before (entries 0..4, twice)
    uint32_t *hitmap = wbase + 16;                   // a bit per tile position that holds a match
    uint32_t *cumVG = hitmap + 128;                  // per plane dword: counts of the dwords before it in its wave
    uint32_t *cumCT = cumVG + kWideCumWords;
    uint32_t *codes2 = cumCT + kWideCumWords;
    uint32_t *inval = codes2 + kWideCodeWords;
    uint32_t *valid2 = inval + kWideInvalWords;      // the low bit of each base's pair set: valid (the 2-bit plane's layout)
    unsigned short *lists = (unsigned short *)(valid2 + kWideCodeWords);   // per wave: its candidate positions (of 1024), ascending
    uint32_t *wacc = (uint32_t *)(lists + 4u * 1024u);                     // [kWideWacc][4]: canonical, non-canonical, forward, reverse covered, per window record of the tile
    unsigned char *pfl = (unsigned char *)(wacc + kWideWacc * 4u);
    if (blockIdx.x >= ntiles) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const bool in_lds = lds_pat != 0u;
    for (uint32_t i = tid; i < lds_pat; i += 256u) { plo[i] = W.lo[i]; phi[i] = W.hi[i]; pfl[i] = W.flags[i]; }
    for (uint32_t i = tid; i < 4096u; i += 256u) pre6[i] = (P)0;
    if (tid < 64u) lens[tid] = tid < W.nlen ? W.len[tid] : 0xFFFFFFFFu;
    if (tid <= W.nlen && tid < 68u) first[tid] = W.first[tid];
    if (tid == 0u) { cumVG[256] = 0u; cumCT[256] = 0u; }
    // nucleotide counts of the tile positions [0, q), q <= 4096: {valid | G << 16, C | T << 16}
    auto prefix_counts = [&](uint32_t q, uint32_t &VG, uint32_t &CT) {
        const uint32_t d = q >> 4, m = (1u << (2u * (q & 15u))) - 1u;
        const uint32_t v = valid2[d] & m & 0x55555555u, w = codes2[d], lo = w & v, hi = (w >> 1) & v;
        const uint32_t g = (uint32_t)__popc(lo & hi);
        VG = cumVG[d] + wbase[d >> 6] + ((uint32_t)__popc(v) | (g << 16));
        CT = cumCT[d] + wbase[8u + (d >> 6)] + (((uint32_t)__popc(lo) - g) | (((uint32_t)__popc(hi) - g) << 16));
    };
    bool tables_built = false;
    const uint32_t fold_mask = Q.fold ? 0xDFDFDFDFu : 0xFFFFFFFFu;
    // sixteen bases of a tile (the layout keeps 64 readable bytes behind the last region; regions start on 16-byte boundaries and
    // tiles at multiples of 4096 inside them — a misaligned tile is read byte by byte)
    auto load16 = [&](const TsGeneralTile &X, uint32_t i) -> uint4 {
        if (i >= X.avail) return make_uint4(0u, 0u, 0u, 0u);
        const unsigned char *p = in + X.in_off + i;
        if (((uintptr_t)p & 15u) == 0u) return *(const uint4 *)p;
        uint32_t d[4];
        for (uint32_t q = 0; q < 4u; ++q) {
            uint32_t x = 0;
            for (uint32_t r = 0; r < 4u; ++r) x |= (i + 4u * q + r < X.avail ? (uint32_t)p[4u * q + r] : 0u) << (8u * r);
            d[q] = x;
        }
        return make_uint4(d[0], d[1], d[2], d[3]);
    };
    // plane dword i / 16 from sixteen bases (ASCII & 6 = twice the code: A 0, C 2, T 4, G 6, and the v_perm selector of the letter
    // that code stands for; a byte that is not that letter, up to case folding, is invalid)
    auto stage16 = [&](const uint4 v, uint32_t i, uint32_t avail, bool counts) {
        uint32_t cw = 0, iv = 0xFFFFu;
        if (i < avail) {
            const uint32_t d[4] = {v.x, v.y, v.z, v.w};
            uint32_t t[4], b4[4];
#pragma unroll
            for (uint32_t q = 0; q < 4u; ++q) {
                t[q] = d[q] & 0x06060606u;
                const uint32_t e = __builtin_amdgcn_perm(0xFF47FF54u, 0xFF43FF41u, t[q]);
                const uint32_t dd = (d[q] & fold_mask) ^ e;
                const uint32_t nz = ((((dd & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | dd) & 0x80808080u) >> 7;
                b4[q] = __builtin_amdgcn_udot4(nz, 0x08040201u, 0u, false);
            }
            cw = pack16(t);
            iv = b4[0] | (b4[1] << 4) | (b4[2] << 8) | (b4[3] << 12);
            if (i + 16u > avail) iv = (iv | (~0u << (avail - i))) & 0xFFFFu;       // bases behind the region's end
        }
        codes2[i >> 4] = cw;
        ((unsigned short *)inval)[i >> 4] = (unsigned short)iv;
        uint32_t ok = ~iv & 0xFFFFu;
        ok = (ok | (ok << 8)) & 0x00FF00FFu; ok = (ok | (ok << 4)) & 0x0F0F0F0Fu;
        ok = (ok | (ok << 2)) & 0x33333333u; ok = (ok | (ok << 1)) & 0x55555555u;
        valid2[i >> 4] = ok;                            // (the low bit of each base's pair)
        if (counts) {                                   // (plane dword tid; every thread is here)
            const uint32_t lo1 = cw & ok, hi1 = (cw >> 1) & ok, g = (uint32_t)__popc(lo1 & hi1);
            const uint32_t vg = (uint32_t)__popc(ok) | (g << 16);
            const uint32_t ct = ((uint32_t)__popc(lo1) - g) | (((uint32_t)__popc(hi1) - g) << 16);
            const uint32_t ivg = wave_inclusive_dpp(vg), ict = wave_inclusive_dpp(ct);
            cumVG[tid] = ivg - vg; cumCT[tid] = ict - ct;
            if (lane == 63u) { wtot[wave] = ivg; wtot[4u + wave] = ict; }
        }
    };
    TsGeneralTile T = tiles[blockIdx.x];
    uint4 v_main = load16(T, tid * 16u), v_halo = tid < kWideCodeWords - 256u ? load16(T, 4096u + tid * 16u) : make_uint4(0u, 0u, 0u, 0u);
  TsGeneralTile Tn = T;
  for (uint32_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x, T = Tn) {
    if (tables_built) __syncthreads();               // (the tile before is done with the planes and the masks)
    // 1. stage: bases [0, avail) of the tile, avail <= kTile + 64; positions beyond avail are invalid
    const uint32_t avail = T.avail;
    stage16(v_main, tid * 16u, avail, true);
    if (tid < kWideCodeWords - 256u) stage16(v_halo, 4096u + tid * 16u, avail, false);
    // the next tile's bases are on their way while this one is worked on
    if (tile + gridDim.x < ntiles) {
        Tn = tiles[tile + gridDim.x];
        v_main = load16(Tn, tid * 16u);
        if (tid < kWideCodeWords - 256u) v_halo = load16(Tn, 4096u + tid * 16u);
    }
    __syncthreads();
    if (tid < 5u) {                                  // sums over the waves before (entry 4: the whole tile)
        uint32_t a = 0, b = 0;
        for (uint32_t v = 0; v < tid; ++v) { a += wtot[v]; b += wtot[4u + v]; }
        wbase[tid] = a; wbase[8u + tid] = b;
    }
    // the prefix table: every pattern's first min(l, 6) bases under every extension to six (once per workgroup)
    if (!tables_built)
    for (uint32_t li = 0; li < W.nlen; ++li) {
        const uint32_t l = lens[li], q = l < 6u ? l : 6u, ext_bits = 2u * (6u - q);
        const uint32_t f0 = first[li], cnt = first[li + 1u] - f0;
        for (uint32_t x = tid; x < (cnt << ext_bits); x += 256u) {
            u64 code_lo;
            if (in_lds) code_lo = plo[f0 + (x >> ext_bits)]; else code_lo = W.lo[f0 + (x >> ext_bits)];        // (not a ternary: that is a flat read)
            const uint32_t pre = (uint32_t)code_lo & ((1u << (2u * q)) - 1u);
            const uint32_t idx = pre | ((x & ((1u << ext_bits) - 1u)) << (2u * q));
            if constexpr (sizeof(P) == 2) atomicOr((uint32_t *)pre6 + (idx >> 1), (1u << li) << (16u * (idx & 1u)));   // (LDS atomics are 32 bits wide)
            else atomicOr(&pre6[idx], (P)((P)1 << li));
        }
    }
    tables_built = true;
    __syncthreads();
    // 2. matches.  Round 5: candidates first — a position whose next six bases begin no pattern (pre6: ninety-nine in a hundred on
    // random sequence) is settled by one table lookup; a lane looks up its sixteen consecutive positions and the wave lists the
    // few that remain, in position order.  The 128-bit codes, the reach to the next non-ACGT base and the searches are then done
    // with a LANE PER LISTED POSITION (round 4 did all of it for every position, a lane in sixty-four doing useful work).
    unsigned short *const list = lists + wave * 1024u;
    uint32_t ncand = 0;
    for (uint32_t i = tid; i < kTile * (uint32_t)sizeof(M) / 16u; i += 256u) ((uint4 *)hit)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid < 128u) hitmap[tid] = 0u;
    if (!(Q.abl & 128u)) {
        const uint32_t wd0 = wave * 64u + lane, j0 = wd0 * 16u;
        const uint32_t c0 = codes2[wd0], c1 = codes2[wd0 + 1u];
        uint32_t mask16 = 0;
#pragma unroll
        for (uint32_t i = 0; i < 16u; ++i) {
            const uint32_t idx = (i <= 10u ? (c0 >> (2u * i)) : __builtin_amdgcn_alignbit(c1, c0, 2u * i)) & 0xFFFu;
            mask16 |= (pre6[idx] != (P)0 ? 1u : 0u) << i;
        }
        if (j0 + 16u > T.n) mask16 &= j0 < T.n ? (1u << (T.n - j0)) - 1u : 0u;     // positions behind the tile's last
        const uint32_t cnt = (uint32_t)__popc(mask16);
        const uint32_t incl = wave_inclusive_dpp(cnt);
        ncand = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        uint32_t at = incl - cnt;
        for (uint32_t m = mask16; m; m &= m - 1u) list[at++] = (unsigned short)(lane * 16u + (uint32_t)__builtin_ctz(m));
    }
    __syncthreads();                                   // (the masks are zero; a wave reads its own list only)
    for (uint32_t e0 = 0; e0 < ncand; e0 += 64u) {
        const uint32_t e = e0 + lane;
        if (e < ncand) {
            const uint32_t j = wave * 1024u + list[e];
            u64 h = 0, f = 0, c = 0;
            const uint32_t wd = j >> 4, sh = 2u * (j & 15u);
            const uint32_t c0 = codes2[wd], c1 = codes2[wd + 1u], c2 = codes2[wd + 2u], c3 = codes2[wd + 3u], c4 = codes2[wd + 4u];
            const u64 lo = (u64)__funnelshift_r(c0, c1, sh) | ((u64)__funnelshift_r(c1, c2, sh) << 32);
            const u64 hi = (u64)__funnelshift_r(c2, c3, sh) | ((u64)__funnelshift_r(c3, c4, sh) << 32);
            const uint32_t vd = j >> 5, vs = j & 31u;
            const uint32_t ilo = __funnelshift_r(inval[vd], inval[vd + 1u], vs), ihi = __funnelshift_r(inval[vd + 1u], inval[vd + 2u], vs);
            const uint32_t nvalid = ilo ? (uint32_t)__builtin_ctz(ilo) : (ihi ? 32u + (uint32_t)__builtin_ctz(ihi) : 64u);
            const uint32_t rem = avail > j ? avail - j : 0u;
            const uint32_t maxlen = nvalid < rem ? nvalid : rem;
            // (the same search twice, over the lists in LDS or in device memory: one loop with `in_lds ? lds : global` reads made
            // every read a FLAT access, which for LDS is far slower than a ds_read)
            auto search = [&](const auto *list_lo, const auto *list_hi, const auto *list_fl) {
                for (u64 cand = (u64)pre6[(uint32_t)lo & 0xFFFu]; cand; cand &= cand - 1ull) {
                    const uint32_t li = (uint32_t)__builtin_ctzll(cand);
                    const uint32_t l = lens[li];
                    if (l > maxlen) break;               // lengths ascend: a non-ACGT base or the region's end stops this and every longer pattern
                    const u64 klo = l >= 32u ? lo : (lo & ((1ull << (2u * l)) - 1ull));
                    const u64 khi = l <= 32u ? 0ull : (hi & ((1ull << (2u * (l - 32u))) - 1ull));  // (l <= 63: the shift is below 64)
                    uint32_t a = first[li], b = first[li + 1u];
                    const uint32_t end = b;
                    while (a < b) {                      // binary search in the (lo, hi)-sorted list of this length
                        const uint32_t mid = (a + b) >> 1;
                        const u64 mlo = list_lo[mid], mhi = list_hi[mid];
                        if (mlo < klo || (mlo == klo && mhi < khi)) a = mid + 1u; else b = mid;
                    }
                    if (a < end && list_lo[a] == klo && list_hi[a] == khi) {
                        const uint32_t fl = list_fl[a];
                        h |= 1ull << li;
                        if (fl & 1u) f |= 1ull << li;
                        if (fl & 2u) c |= 1ull << li;
                    }
                }
            };
            if (in_lds) search(plo, phi, pfl); else search(W.lo, W.hi, W.flags);
            if (h) {
                hit[j] = (M)h; fwdm[j] = (M)f; canm[j] = (M)c;
                atomicOr(&hitmap[j >> 5], 1u << (j & 31u));
            }
        }
    }
    __syncthreads();
    const u64 n = seg_len[T.seg];
    const u64 P0 = T.seg_rel;
    // 3. window records (full scans): as in ts_general_fused
    if (!tips && T.n && !(Q.abl & 256u)) {
        // (no 64-bit division: the host hands over P0 = k_p0 s + r_p0 per tile, w = cw s + rw per call and the segments' window counts)
        const u64 nwin = seg_nwin[T.seg];
        const u64 kw_lo = P0 >= Q.w ? T.k_p0 + 1u - Q.cw - (Q.rw > T.r_p0 ? 1u : 0u) : 0u;     // first call whose window reaches the tile: (P0 - w) / s + 1
        u64 kw_hi = T.k_p0 + (T.r_p0 + T.n - 1u) / Q.s;                                     // last call that starts inside it (r_p0 < s, n <= 4096: 32 bits unless s is huge)
        if ((u64)T.r_p0 + T.n - 1u > 0xFFFFFFFFull) kw_hi = T.k_p0 + ((u64)T.r_p0 + T.n - 1u) / Q.s;
        if (kw_hi >= nwin) kw_hi = nwin - 1u;
        const bool carries = Q.w != Q.s;
        u64 rec_hi = kw_hi + (carries ? 1u : 0u);
        if (rec_hi >= nwin) rec_hi = nwin - 1u;
        uint32_t *const wrec = win_out + seg_win_base[T.seg] * 8ull;
        if (rec_hi - kw_lo < kWideWacc) {
            // Round 5: a LANE PER MATCH POSITION adds the covered bases to the accumulators of the records the match belongs to
            // (the windows that hold the position: main part of record kw, carry of record kw + 1 — the conditions are
            // window_tile_part_wide's, turned round), and a LANE PER RECORD adds the nucleotide counts of its two parts from the
            // prefix sums and writes it.  (Before: a wave per record part, twenty of them per tile, each with its own geometry.)
            const uint32_t nrec = (uint32_t)(rec_hi - kw_lo) + 1u;
            if (tid < kWideWacc) *(uint4 *)&wacc[4u * tid] = make_uint4(0u, 0u, 0u, 0u);
            __syncthreads();
            const uint32_t ov = Q.w - Q.s;
            const uint32_t t1 = Q.s - Q.longest, t2 = ov - Q.longest;        // uint32 on purpose (src/teloscope.cpp:413-415)
            const uint32_t si_inner = t1 < t2 ? t1 : t2;
            for (uint32_t e0 = 0; e0 < ncand; e0 += 64u) {
                const uint32_t e = e0 + lane;
                if (e >= ncand) continue;
                const uint32_t j = wave * 1024u + list[e];
                const u64 m = hit[j];
                if (!m) continue;
                const u64 f = fwdm[j], c = canm[j];
                const u64 p = P0 + j;
                const u64 rj = (u64)T.r_p0 + j;                               // p / s with P0 = k_p0 s + r_p0: a 32-bit division
                const u64 kmax = T.k_p0 + (rj <= 0xFFFFFFFFull ? (u64)((uint32_t)rj / Q.s) : rj / Q.s);
                // the first window that holds p: (p - w) / s + 1 = k_p0 - cw + floor((r_p0 + j - rw) / s) + 1
                const long long num = (long long)rj - (long long)Q.rw;
                const u64 kmin = p >= Q.w ? T.k_p0 - Q.cw + (num < 0 ? 0ull : 1ull + (num <= 0xFFFFFFFFll ? (u64)((uint32_t)num / Q.s) : (u64)num / Q.s)) : 0ull;
                for (u64 kw = kmin; kw <= kmax && kw < nwin; ++kw) {
                    const u64 wstart = kw * Q.s;
                    const uint32_t i = (uint32_t)(p - wstart);
                    const uint32_t cws = (uint32_t)((n - wstart) < Q.w ? (n - wstart) : Q.w);
                    const bool always_main = (ov == 0u || kw == 0u);
                    const uint32_t start_index = always_main ? 0u : si_inner;
                    const bool main_i = i >= start_index && i < cws;
                    const bool carry_i = carries && kw + 1u <= rec_hi && i >= (start_index < Q.s ? Q.s : start_index) && i < cws;
                    if (!main_i && !carry_i) continue;
                    uint32_t mc = 0, mn = 0, mf = 0, mr = 0, cc = 0, cn = 0, cf = 0, cr = 0;
                    for (u64 mm = m; mm; mm &= mm - 1ull) {
                        const uint32_t li = (uint32_t)__builtin_ctzll(mm);
                        const uint32_t l = lens[li];
                        const uint32_t jend = i + l - 1u;
                        if (jend >= cws) continue;                              // scanLimit: may not cross the window end
                        const bool isc = (c >> li) & 1ull, isf = (f >> li) & 1ull;
                        if (main_i && (always_main || jend >= ov)) { if (isc) mc += l; else mn += l; if (isf) mf += l; else mr += l; }
                        if (carry_i) { if (isc) cc += l; else cn += l; if (isf) cf += l; else cr += l; }
                    }
                    uint32_t *const am = &wacc[4u * (uint32_t)(kw - kw_lo)];
                    if (mc) atomicAdd(am + 0, mc); if (mn) atomicAdd(am + 1, mn); if (mf) atomicAdd(am + 2, mf); if (mr) atomicAdd(am + 3, mr);
                    if (cc) atomicAdd(am + 4, cc); if (cn) atomicAdd(am + 5, cn); if (cf) atomicAdd(am + 6, cf); if (cr) atomicAdd(am + 7, cr);
                }
            }
            __syncthreads();
            if (tid < nrec) {
                const u64 R = kw_lo + tid;
                WideAcc unused = {0, 0, 0, 0, 0};
                uint32_t tAT = 0, tCG = 0;                                       // A | T << 16, C | G << 16
                window_tile_part_wide<M, decltype(prefix_counts), true>(hitmap, hit, fwdm, canm, lens, Q, n, R, false, P0, T.n, lane, unused, prefix_counts, tAT, tCG);
                if (carries && R > 0u) window_tile_part_wide<M, decltype(prefix_counts), true>(hitmap, hit, fwdm, canm, lens, Q, n, R - 1u, true, P0, T.n, lane, unused, prefix_counts, tAT, tCG);
                const uint4 acc = *(const uint4 *)&wacc[4u * tid];
                const uint4 lo4 = make_uint4(tAT & 0xFFFFu, tCG & 0xFFFFu, tCG >> 16, tAT >> 16);      // A, C, G, T
                const u64 span_lo = R * Q.s;
                const u64 span_hi = span_lo + Q.w < n ? span_lo + Q.w : n;
                const bool sole = span_lo >= P0 && span_hi <= P0 + T.n;
                uint32_t *const out = wrec + R * 8ull;
                if (sole) { *(uint4 *)out = lo4; *(uint4 *)(out + 4) = acc; }
                else {
                    if (lo4.x) atomicAdd(out + 0, lo4.x); if (lo4.y) atomicAdd(out + 1, lo4.y); if (lo4.z) atomicAdd(out + 2, lo4.z); if (lo4.w) atomicAdd(out + 3, lo4.w);
                    if (acc.x) atomicAdd(out + 4, acc.x); if (acc.y) atomicAdd(out + 5, acc.y); if (acc.z) atomicAdd(out + 6, acc.z); if (acc.w) atomicAdd(out + 7, acc.w);
                }
            }
        } else
        for (u64 R = kw_lo + wave; R <= rec_hi; R += 4u) {
            WideAcc a = {0, 0, 0, 0, 0};
            uint32_t tAT = 0, tCG = 0;                                       // A | T << 16, C | G << 16 (wave-uniform)
            window_tile_part_wide(hitmap, hit, fwdm, canm, lens, Q, n, R, false, P0, T.n, lane, a, prefix_counts, tAT, tCG);
            if (carries && R > 0u) window_tile_part_wide(hitmap, hit, fwdm, canm, lens, Q, n, R - 1u, true, P0, T.n, lane, a, prefix_counts, tAT, tCG);
            const uint32_t tcan = wave_total(a.can), tnon = wave_total(a.non), tfwd = wave_total(a.fwd), trev = wave_total(a.rev);
            const uint32_t mine = lane == 0u ? (tAT & 0xFFFFu) : lane == 1u ? (tCG & 0xFFFFu) : lane == 2u ? (tCG >> 16) : lane == 3u ? (tAT >> 16)
                                : lane == 4u ? tcan : lane == 5u ? tnon : lane == 6u ? tfwd : trev;
            const u64 span_lo = R * Q.s;
            const u64 span_hi = span_lo + Q.w < n ? span_lo + Q.w : n;
            const bool sole = span_lo >= P0 && span_hi <= P0 + T.n;
            if (lane < 8u) {
                if (sole) wrec[R * 8ull + lane] = mine;
                else if (mine) atomicAdd(&wrec[R * 8ull + lane], mine);
            }
        }
    }
    __syncthreads();                                   // (the masks are rewritten below)
    // 4. match records: the matches the reference pushes, position then length order; wave v owns positions [1024 v, 1024 v + 1024).
    // A lane per LISTED position (the list is in position order): round 4 walked all 4096 positions twice, and the push test — 64-bit
    // divisions — ran under divergence for every group of 64 positions that held a match.
    PushGeom pg{};                                     // (as the list form derives it: sums and compares of the host's quotients)
    {
        const uint32_t ov = Q.w - Q.s;
        pg.P0 = P0; pg.n = n; pg.s = Q.s; pg.w = Q.w; pg.ov = ov;
        const uint32_t t1 = Q.s - Q.longest, t2 = ov - Q.longest;        // uint32 on purpose (src/teloscope.cpp:413-415)
        pg.start_index = t1 < t2 ? t1 : t2;
        pg.kP0 = T.k_p0; pg.rP0 = T.r_p0;
        if (ov == 0u) {
            pg.k0 = T.k_p0; pg.r0 = T.r_p0;
            pg.N0 = (n - P0) + T.r_p0;                                    // n - k0 s
        } else if (P0 > ov) {
            // P0 - ov = (k_p0 - cw + 1) s + (r_p0 - rw)
            const bool borrow = T.r_p0 < Q.rw;
            pg.k1 = T.k_p0 + 1u - Q.cw - (borrow ? 1u : 0u);
            pg.r1 = T.r_p0 - Q.rw + (borrow ? Q.s : 0u);
            pg.D1 = pg.r1 + ov;                                           // P0 - k1 s
            pg.dsub = 0u;
        } else {
            pg.k1 = 0u; pg.r1 = 0u; pg.D1 = (uint32_t)P0; pg.dsub = (uint32_t)(ov - P0);
        }
    }
    uint32_t wave_cnt = 0;
    for (uint32_t e0 = 0; e0 < ncand; e0 += 64u) {
        const uint32_t e = e0 + lane;
        u64 keep = 0;
        if (e < ncand) {
            const uint32_t j = wave * 1024u + list[e];
            for (u64 m = hit[j]; m; m &= m - 1ull) {
                const uint32_t li = (uint32_t)__builtin_ctzll(m);
                u64 unused_rec;
                if (tips || full_scan_pushes(j, lens[li], pg, &unused_rec)) keep |= 1ull << li;
            }
            hit[j] = (M)keep;
        }
        wave_cnt += (uint32_t)__popcll(keep);
    }
    wave_cnt = wave_total(wave_cnt);
    if (lane == 0u) part[wave] = wave_cnt;
    __syncthreads();
    uint32_t base = 0, total = 0;
    for (uint32_t v = 0; v < 4u; ++v) { if (v < wave) base += part[v]; total += part[v]; }
    if (tid == 0u) {
        *(uint4 *)&tile_stats[4ull * tile] = make_uint4(total, 0u, 0u, 0u);
        if (total > slot_cap) atomicOr(overflow, 1u);
    }
    if (total > slot_cap || wave_cnt == 0u) continue;
    uint32_t *dst = records + (u64)tile * slot_cap;
    for (uint32_t e0 = 0; e0 < ncand; e0 += 64u) {
        const uint32_t e = e0 + lane;
        const uint32_t j = e < ncand ? wave * 1024u + list[e] : 0u;
        const u64 keep = e < ncand ? (u64)hit[j] : 0ull;
        if (__builtin_amdgcn_ballot_w64(keep != 0ull) == 0ull) continue;
        const uint32_t c = (uint32_t)__popcll(keep);
        const uint32_t incl = wave_inclusive(c, lane);
        uint32_t at = base + incl - c;
        const u64 f = fwdm[j], cm = canm[j];
        for (u64 m = keep; m; m &= m - 1ull) {
            const uint32_t li = (uint32_t)__builtin_ctzll(m);
            dst[at++] = (j << 8) | (li << 2) | ((uint32_t)((cm >> li) & 1ull) << 1) | (uint32_t)((f >> li) & 1ull);
        }
        base += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    }
  }
}

// One wave per tile: its records from its slot to their place in the dense, tile-ordered stream.
__global__ __launch_bounds__(256)
void ts_general_compact(const uint32_t *tile_stats, const u64 *tile_off, const uint32_t *records, uint32_t slot_cap,
                        uint32_t ntiles, uint32_t *dense) {
    const uint32_t t = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (t >= ntiles) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = tile_stats[4ull * t];
    const uint32_t *src = records + (u64)t * slot_cap;
    uint32_t *dst = dense + tile_off[t];
    for (uint32_t i = lane; i < n; i += 64u) dst[i] = src[i];
}


// ---- the dense stream in the reference's PUSH order (mixed-length sets under w > s)
//
// scanSegment pushes a match from the first window that holds its END (src/teloscope.cpp:485-509: ascending window, then
// ascending position, then ascending length), so with patterns of different lengths and overlapping windows allMatches is
// not quite in position order: a long match that straddles a window's end is pushed by the NEXT window, behind shorter
// matches that begin after it — and getTerminalBlocks / getInterstitialBlocks walk the stream as it lies
// (:29-256; a position that steps back breaks the chain: the unsigned gap wraps).  The pushing window of a match that ends
// at e is key(e) = e < w - s ? 0 : (e - (w - s)) / s for a segment of at least w bases (one window, key 0, otherwise): monotone in e.
// Push order = the position-ordered stream stably sorted by key, and two records can only be out of order when they start
// fewer than (longest - shortest) bases apart.  So every record finds its place by itself: its index in position order, plus
// the records behind it with a smaller key, minus the records ahead of it with a larger one — a look at the few neighbours
// that start within `spread` bases, and only for records that end within `spread` bases of a window's end.  Neighbours may lie
// in the tile before or behind (the slots are read, nothing of this kernel's output).
//
// The stream stays tile-structured: a record of tile t that has to come before a record of tile t - 1 (its key is smaller
// than the largest key of tile t - 1) MOVES into tile t - 1 — the tile's offset grows by the number that moved, the record's
// tile-relative position by the positions of tile t - 1 — so that every reader of {tile_off, records} (blockcall.hip, the
// host's expansion) walks the reference's sequence by walking the tiles.  The per-tile counts are recomputed from the offsets
// (ts_general_block_inputs).
struct PushOrder {
    uint32_t w, s, spread, shift, li_mask;
    u64 gen_lens;                        // up to eight lengths, six bits each (the list / strided forms' records); 0: wide_len
    const uint32_t *wide_len;            // the wide form's lengths (device)
};

__device__ __forceinline__ uint32_t push_len(const PushOrder &O, uint32_t rec) {
    const uint32_t li = (rec >> 2) & O.li_mask;
    return O.gen_lens ? (uint32_t)(O.gen_lens >> (6u * li)) & 63u : O.wide_len[li];
}

__global__ __launch_bounds__(256)
void ts_general_compact_push(const TsGeneralTile *gtiles, const uint32_t *tile_stats, u64 *tile_off, const uint32_t *records,
                             uint32_t slot_cap, uint32_t ntiles, const u64 *seg_len, const PushOrder O, uint32_t *dense) {
    const uint32_t t = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (t >= ntiles) return;
    const uint32_t lane = threadIdx.x & 63u;
    const TsGeneralTile G = gtiles[t];
    const uint32_t n = tile_stats[4ull * t];
    const u64 off = tile_off[t];
    const uint32_t *const src = records + (u64)t * slot_cap;
    // (neighbours: the tiles before and behind when they continue this one — same segment, adjoining positions)
    const bool has_prev = t > 0u && gtiles[t - 1u].seg == G.seg && gtiles[t - 1u].seg_rel + gtiles[t - 1u].n == G.seg_rel;
    const bool has_next = t + 1u < ntiles && gtiles[t + 1u].seg == G.seg && G.seg_rel + G.n == gtiles[t + 1u].seg_rel;
    const uint32_t n_prev = has_prev ? tile_stats[4ull * (t - 1u)] : 0u, n_next = has_next ? tile_stats[4ull * (t + 1u)] : 0u;
    const u64 prev_rel = has_prev ? gtiles[t - 1u].seg_rel : G.seg_rel, next_rel = has_next ? gtiles[t + 1u].seg_rel : 0ull;
    const uint32_t prev_positions = (uint32_t)(G.seg_rel - prev_rel);
    const uint32_t *const srcp = src - slot_cap, *const srcn = src + slot_cap;
    const u64 ov = O.w - O.s;
    const bool one_window = seg_len[G.seg] < O.w;              // every key is 0: position order is push order
    // keys without a 64-bit division per record: everything this wave looks at ends at or behind X0
    const u64 X0 = prev_rel;
    const bool fast = X0 >= ov && O.s <= 0x7FFFFFFFu;
    const u64 q0 = fast ? (X0 - ov) / O.s : 0ull;
    const uint32_t r0 = fast ? (uint32_t)((X0 - ov) - q0 * O.s) : 0u;
    auto key = [&](u64 e) -> u64 {
        if (one_window) return 0ull;
        if (fast) return q0 + (r0 + (uint32_t)(e - X0)) / O.s;     // (e - X0 < 3 tiles: no overflow beside r0 < s < 2^31)
        return e < ov ? 0ull : (e - ov) / O.s;
    };
    // the largest key of the tile before: that of the record that ends last, one of those that start within `spread`
    // bases of its last record
    u64 kmax_prev = 0;
    if (n_prev && !one_window) {
        const u64 p_last = prev_rel + (srcp[n_prev - 1u] >> O.shift);
        bool more = true;
        for (uint32_t b0 = 0; b0 < n_prev && more; b0 += 64u) {
            u64 k = 0;
            bool in = false;
            if (b0 + lane < n_prev) {
                const uint32_t r = srcp[n_prev - 1u - b0 - lane];
                const u64 p = prev_rel + (r >> O.shift);
                in = p + O.spread >= p_last;
                if (in) k = key(p + push_len(O, r) - 1u);
            }
            for (int sh = 1; sh < 64; sh <<= 1) {                  // wave maximum (rare path: a plain shuffle tree)
                const u64 o = ((u64)(uint32_t)__shfl_xor((int)(uint32_t)(k >> 32), sh) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)k, sh);
                k = o > k ? o : k;
            }
            kmax_prev = k > kmax_prev ? k : kmax_prev;
            more = __ballot(b0 + lane < n_prev && !in) == 0ull;    // all 64 were in reach: the batch before them may be too
        }
    }
    uint32_t moved_total = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += 64u) {
        const uint32_t i = i0 + lane;
        const bool valid = i < n;
        uint32_t rec = valid ? src[i] : 0u;
        const u64 p = G.seg_rel + (rec >> O.shift);
        const u64 e = p + push_len(O, rec) - 1u;
        const u64 k0 = key(e);
        int disp = 0;
        if (valid && !one_window) {
            // records behind this one with a smaller key: only if a window ends in (e - spread, e]
            if (e >= O.spread && key(e - O.spread) != k0) {
                for (uint32_t d = 1;; ++d) {
                    const uint32_t j = i + d;
                    uint32_t rj; u64 base;
                    if (j < n) { rj = src[j]; base = G.seg_rel; }
                    else if (j - n < n_next) { rj = srcn[j - n]; base = next_rel; }
                    else break;
                    const u64 pj = base + (rj >> O.shift);
                    if (pj - p >= O.spread) break;
                    if (key(pj + push_len(O, rj) - 1u) < k0) ++disp;
                }
            }
            // records ahead of it with a larger key: only if a window ends in (e, e + spread]
            if (key(e + O.spread) != k0) {
                for (uint32_t d = 1;; ++d) {
                    uint32_t rj; u64 base;
                    if (d <= i) { rj = src[i - d]; base = G.seg_rel; }
                    else if (d - i <= n_prev) { rj = srcp[n_prev - (d - i)]; base = prev_rel; }
                    else break;
                    const u64 pj = base + (rj >> O.shift);
                    if (p - pj >= O.spread) break;
                    if (key(pj + push_len(O, rj) - 1u) > k0) --disp;
                }
            }
        }
        const bool moved = valid && has_prev && k0 < kmax_prev;
        if (moved) rec += prev_positions << O.shift;
        if (valid) dense[off + i + (long long)disp] = rec;
        moved_total += (uint32_t)__popcll(__ballot(moved));
    }
    if (lane == 0u) tile_off[t] = off + moved_total;
}

}  // namespace

unsigned long long ts_k_general_lds_bytes(const TsGenericPatterns *G, uint32_t *lds_patterns) {
    const uint32_t npat = G->first[G->nlen];
    const uint32_t lp = npat <= kMaxLdsPatterns ? (npat ? npat : 1u) : 0u;
    *lds_patterns = lp;
    return (unsigned long long)kTile * 4u + (unsigned long long)lp * 8u + 8u * 128u * 4u + 32u + kCodeWords * 4u + kInvalWords * 4u +
           ((lp + 15u) & ~15u);
}

// list != 0: the list form of the pass (see ts_general_fused_list: the caller has checked that a tile adds to at most
// ts_k_general_list_max_records() window records; *overflow bit 1 then means "run this group again with list = 0")
uint32_t ts_k_general_list_max_records(void) { return kWaccMax; }

int ts_k_launch_general_fused(const unsigned char *in, const TsGeneralTile *tiles, uint32_t ntiles,
                              const unsigned long long *seg_len, const unsigned long long *seg_win_base, const unsigned long long *seg_nwin,
                              const TsGenericPatterns *G, const TsGenericGeom *Q, int tips, uint32_t slot_cap,
                              uint32_t *tile_stats, uint32_t *records, uint32_t *win_out, uint32_t *overflow, int list, int num_cu, void *stream) {
    if (ntiles == 0) return 0;
    uint32_t lp = 0;
    const unsigned long long lds = ts_k_general_lds_bytes(G, &lp);
    if (list && lp) {
        uint32_t nshort = 0;
        for (uint32_t li = 0; li < G->nlen; ++li) nshort += G->len[li] <= kShortLen ? 1u : 0u;
        const unsigned long long lds2 = 4ull * kListWave * 2u + (((unsigned long long)lp * 8u + 15u) & ~15ull) + 4096u + nshort * 1024u + kWaccMax * 16u + 32u + 48u + 32u + 64u +
                                        kCodeWords * 4u + kInvalWords * 4u + kCodeWords * 4u + 2u * kCumWords * 4u + ((lp + 15u) & ~15u);
        // persistent workgroups: as many as the device holds at once (LDS bound), each strides over the tiles
        static const uint32_t wg_override = [] { const char *e = getenv("TS_GEN_WGS"); return e ? (uint32_t)atoi(e) : 0u; }();
        // (what the runtime says fits: a grid of more workgroups than are resident at once ends in a round of stragglers —
        // 1280 workgroups where 1024 fit measured 7.0 ms against 6.0, profiles/r04/general_occupancy.txt)
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ts_general_fused_list, 256, (size_t)lds2) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            per_cu = (int)std::max<unsigned long long>(1ull, std::min<unsigned long long>(8ull, (128ull << 10) / lds2));
        }
        uint32_t grid = (uint32_t)(num_cu > 0 ? num_cu : 256) * (uint32_t)per_cu;
        if (wg_override) grid = wg_override;
        if (grid > ntiles) grid = ntiles;
        hipLaunchKernelGGL(ts_general_fused_list, dim3(grid), dim3(256), (size_t)lds2, (hipStream_t)stream, in, tiles, ntiles,
                           (const u64 *)seg_len, (const u64 *)seg_win_base, (const u64 *)seg_nwin, *G, *Q, tips, slot_cap, lp, nshort, tile_stats, records, win_out, overflow);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(ts_general_fused, dim3(ntiles), dim3(256), (size_t)lds, (hipStream_t)stream, in, tiles, ntiles,
                       (const u64 *)seg_len, (const u64 *)seg_win_base, *G, *Q, tips, slot_cap, lp, tile_stats, records, win_out, overflow);
    return (int)hipGetLastError();
}

namespace {
// What block calling on the device (blockcall.hip) needs of a group scanned by the kernels above: per tile a TsTile whose
// in_off - (its segment's base) is the tile's segment-relative position, and the canonical / forward counts of its records in
// the tile directory (words 1 and 2; the fused pass leaves only the count).  One wave per tile over the dense stream.
__global__ __launch_bounds__(64)
void ts_general_block_inputs(const TsGeneralTile *gtiles, const u64 *tile_off, const uint32_t *dense, const u64 *seg_base,
                             uint32_t ntiles, TsTile *tiles, uint32_t *tile_stats, int count_from_offsets) {
    const uint32_t t = blockIdx.x;
    if (t >= ntiles) return;
    const uint32_t lane = threadIdx.x;
    const TsGeneralTile G = gtiles[t];
    // (count_from_offsets: the push-ordered stream moved a few records across tile borders; tile_off has ntiles + 1 entries.
    // Otherwise the fused pass's count: the records may lie in the tiles' slots, which are not adjacent)
    const u64 o0 = tile_off[t];
    const uint32_t n = count_from_offsets ? (uint32_t)(tile_off[t + 1u] - o0) : tile_stats[4ull * t];
    const uint32_t *src = dense + o0;
    uint32_t ncan = 0, nfwd = 0;
    for (uint32_t i = lane; i < n; i += 64u) {
        const uint32_t r = src[i];
        nfwd += r & 1u; ncan += (r >> 1) & 1u;                 // (general records: forward is bit 0, canonical bit 1)
    }
    ncan = wave_total(ncan); nfwd = wave_total(nfwd);
    if (lane == 0u) {
        if (count_from_offsets) tile_stats[4ull * t] = n;
        tile_stats[4ull * t + 1u] = ncan;
        tile_stats[4ull * t + 2u] = nfwd;
        TsTile T{};
        T.in_off = seg_base[G.seg] + G.seg_rel;
        T.own_len = G.n; T.nrel = G.n; T.nwin = 0; T.win_out = 0; T.seg = G.seg;
        tiles[t] = T;
    }
}
}  // namespace

int ts_k_launch_general_block_inputs(const TsGeneralTile *gtiles, const unsigned long long *tile_off, const uint32_t *dense,
                                     const unsigned long long *seg_base, uint32_t ntiles, TsTile *tiles, uint32_t *tile_stats,
                                     int count_from_offsets, void *stream) {
    if (ntiles == 0) return 0;
    hipLaunchKernelGGL(ts_general_block_inputs, dim3(ntiles), dim3(64), 0, (hipStream_t)stream, gtiles, (const u64 *)tile_off, dense,
                       (const u64 *)seg_base, ntiles, tiles, tile_stats, count_from_offsets);
    return (int)hipGetLastError();
}

namespace {
// The tile directory of records that stay where the fused pass wrote them: tile t's begin at t * slot_cap.
__global__ __launch_bounds__(256)
void ts_general_slot_offsets(u64 *tile_off, uint32_t ntiles, uint32_t slot_cap) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t <= ntiles) tile_off[t] = (u64)t * slot_cap;
}
}  // namespace

// Blocks-only calls over a stream in position order need no dense stream: block calling (blockcall.hip) addresses records
// through the tile directory, so the directory is pointed at the slots and the compaction — every record read and written
// once more — is not run.
int ts_k_launch_general_slot_offsets(unsigned long long *tile_off, uint32_t ntiles, uint32_t slot_cap, void *stream) {
    hipLaunchKernelGGL(ts_general_slot_offsets, dim3(ntiles / 256u + 1u), dim3(256), 0, (hipStream_t)stream, (u64 *)tile_off, ntiles, slot_cap);
    return (int)hipGetLastError();
}

int ts_k_launch_general_compact(const uint32_t *tile_stats, const unsigned long long *tile_off, const uint32_t *records,
                                uint32_t slot_cap, uint32_t ntiles, uint32_t *dense, void *stream) {
    if (ntiles == 0) return 0;
    hipLaunchKernelGGL(ts_general_compact, dim3((ntiles + 3u) / 4u), dim3(256), 0, (hipStream_t)stream, tile_stats,
                       (const u64 *)tile_off, records, slot_cap, ntiles, dense);
    return (int)hipGetLastError();
}

int ts_k_launch_general_compact_push(const TsGeneralTile *gtiles, const uint32_t *tile_stats, unsigned long long *tile_off,
                                     const uint32_t *records, uint32_t slot_cap, uint32_t ntiles, const unsigned long long *seg_len,
                                     uint32_t w, uint32_t s, uint32_t spread, int wide, unsigned long long gen_lens,
                                     const uint32_t *wide_len, uint32_t *dense, void *stream) {
    if (ntiles == 0) return 0;
    PushOrder O{w, s, spread, wide ? 8u : 5u, wide ? 63u : 7u, wide ? 0ull : gen_lens, wide_len};
    hipLaunchKernelGGL(ts_general_compact_push, dim3((ntiles + 3u) / 4u), dim3(256), 0, (hipStream_t)stream, gtiles, tile_stats,
                       (u64 *)tile_off, records, slot_cap, ntiles, (const u64 *)seg_len, O, dense);
    return (int)hipGetLastError();
}

int ts_k_launch_general_wide(const unsigned char *in, const TsGeneralTile *tiles, uint32_t ntiles,
                             const unsigned long long *seg_len, const unsigned long long *seg_win_base, const unsigned long long *seg_nwin,
                             const TsWidePatterns *W, const TsGenericGeom *Q, int tips, uint32_t slot_cap,
                             uint32_t *tile_stats, uint32_t *records, uint32_t *win_out, uint32_t *overflow, void *stream) {
    if (ntiles == 0) return 0;
    const size_t fixed = 64u * 4u + 68u * 4u + 32u + 32u + 64u + 128u * 4u + 2u * kWideCumWords * 4u + 2u * kWideCodeWords * 4u + kWideInvalWords * 4u + 4u * 1024u * 2u + (kWideWacc + 1u) * 16u;
    static const int cus = [] { int v = 0, dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256; return v; }();
    auto launch = [&](auto kernel, size_t mask_bytes, size_t pre_bytes) -> hipError_t {
        const size_t tables = 3u * (size_t)kTile * mask_bytes + 4096u * pre_bytes + fixed;
        // the pattern lists in LDS (17 bytes per pattern) when they fit beside the tables
        uint32_t lds_pat = W->npat <= 2048u ? W->npat : 0u;
        if (tables + (size_t)lds_pat * 16u + ((lds_pat + 15u) & ~15u) > (160u << 10)) lds_pat = 0u;
        const size_t lds = tables + (size_t)lds_pat * 16u + ((lds_pat + 15u) & ~15u);
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   // (up to 160 KB: above the default limit)
        if (e != hipSuccess) return e;
        // persistent workgroups: as many as fit beside each other (LDS), each takes the tiles blockIdx.x, + gridDim.x, ...
        const size_t per_cu = std::max<size_t>(1, std::min<size_t>(8, (160u << 10) / std::max<size_t>(lds, 1)));
        const uint32_t grid = (uint32_t)std::min<size_t>(ntiles, (size_t)cus * per_cu);
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), lds, (hipStream_t)stream, in, tiles, ntiles,
                           (const u64 *)seg_len, (const u64 *)seg_win_base, (const u64 *)seg_nwin, *W, *Q, tips, slot_cap, lds_pat, tile_stats, records, win_out, overflow);
        return hipSuccess;
    };
    hipError_t e;
    if (W->nlen <= 16u) e = launch(ts_general_wide<uint16_t, uint16_t>, 2, 2);
    else if (W->nlen <= 32u) e = launch(ts_general_wide<uint32_t, uint32_t>, 4, 4);
    else e = launch(ts_general_wide<u64, u64>, 8, 8);
    if (e != hipSuccess) return (int)e;
    return (int)hipGetLastError();
}
