// generic.hip — the GENERAL path of libteloscan (gfx950): every parameter set the tiled kernel in kernels.hip
// does not take.  That kernel covers uniform-length pattern sets (3 <= k <= 8) under the geometry where the
// reference's per-window carry loop has a closed form; what is left — mixed-length pattern sets (several matches per
// position), pattern lengths up to 32, and window/step pairs where the reference's `uint32` start index wraps
// (src/teloscope.cpp:413-415) — is restated here LITERALLY (the main / carry attribution of analyzeWindow, not its
// closed form), for a whole batch of segments at a time and with nothing but ordering work left to the host:
//
//   ts_general_fused    ONE pass, one workgroup per tile of 4096 positions of one scanned region: bases staged into
//                       LDS as 2-bit codes, the pattern lists (per length: ascending 2-bit codes + {forward,
//                       canonical}) and a prefix bitmap per length beside them; every position's matches (3 bits
//                       {match, forward, canonical} per length) stay in LDS; from there the workgroup adds its share to
//                       the window records (a wave per window part, literal main / carry attribution, DPP reduction,
//                       atomics) and writes the match records the reference pushes, in position order, into the
//                       tile's slot.
//   ts_general_compact  the tiles' slots into one dense tile-ordered stream (after a prefix sum over the counts).
//
// Traffic: 1 B/base in, 32 B/window and 2 x 4 B/match out.  (Rounds 1-2 ran three kernels around a 4 B/base match mask
// in HBM: ~13 B/base.)  The host orders mixed-length records by pushing window where they are out of that order,
// expands them and calls blocks.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ts_internal.h"

namespace {

typedef unsigned long long u64;

constexpr uint32_t kTile = TS_GENERAL_TILE;           // positions per tile
constexpr uint32_t kHalo = 32;                        // bases staged beyond it (longest pattern <= 32)
constexpr uint32_t kMaxLdsPatterns = 2048;

__device__ __forceinline__ uint32_t base_code_byte(uint32_t c, uint32_t fold) {
    if (fold) c &= 0xDFu;
    return c == 'A' ? 0u : c == 'C' ? 1u : c == 'T' ? 2u : c == 'G' ? 3u : 0xFFu;
}

// Is the match (segment-relative position p, length l) pushed to the reference's match vectors by a full scan of a
// segment of n bases?  (src/teloscope.cpp:485: by the window whose own scan sees it with j >= overlap, or always in
// window 0 / when windows do not overlap; restated from the window loop's index arithmetic, uint32 wrap included.)
__device__ __forceinline__ bool full_scan_pushes(u64 p, uint32_t l, u64 n, const TsGenericGeom &Q) {
    const uint32_t s = Q.s, w = Q.w, ov = w - s;
    const u64 e = p + l - 1u;
    if (ov == 0u) {
        const u64 k = p / s;
        const u64 left = n - k * s;
        const u64 cws = left < w ? left : w;
        return (p - k * s) + l <= cws;                  // may not cross its only window's end
    }
    if (e < (n < w ? n : (u64)w)) return true;          // window 0 scans everything it holds
    const u64 k = (e - ov) / s;                         // the one window with j >= overlap
    const uint32_t t1 = s - Q.longest, t2 = ov - Q.longest;
    const uint32_t start_index = t1 < t2 ? t1 : t2;
    return p >= k * s && (p - k * s) >= start_index;
}

struct Acc { uint32_t v[8]; };            // A C T G (code order), canonical, non-canonical, forward, reverse covered

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

// What one analyzeWindow() call over window `kw` adds either to its own record (carry == false: bases with
// i >= mainlo, matches with j >= ov, or everything for window 0 / ov == 0) or to the NEXT window's record
// (carry == true: i >= step) — restricted to the positions [P0, P0 + n) the tile holds in LDS; the wave's lanes
// stride over i.  (src/teloscope.cpp:387-534, the index arithmetic in uint32 as there.)
__device__ __forceinline__ void window_tile_part(const unsigned char *codes, const uint32_t *mask, const TsGenericPatterns &G,
                                                 const TsGenericGeom &Q, u64 n, u64 kw, bool carry, u64 P0, uint32_t ntile,
                                                 uint32_t lane, Acc &a) {
    const u64 wstart = kw * Q.s;
    const uint32_t cws = (uint32_t)((n - wstart) < Q.w ? (n - wstart) : Q.w);
    const uint32_t ov = Q.w - Q.s;
    const bool always_main = (ov == 0 || wstart == 0);
    // uint32 arithmetic on purpose (wraps when the longest pattern exceeds step or overlap)
    const uint32_t t1 = Q.s - Q.longest, t2 = ov - Q.longest;
    uint32_t start_index = always_main ? 0u : (t1 < t2 ? t1 : t2);
    if (carry && start_index < Q.s) start_index = Q.s;               // the carry only takes i >= step
    if (start_index >= cws) return;
    u64 lo = wstart + start_index, hi = wstart + cws;                 // segment-relative positions the call visits
    if (lo < P0) lo = P0;
    if (hi > P0 + ntile) hi = P0 + ntile;
    for (u64 p = lo + lane; p < hi; p += 64u) {
        const uint32_t i = (uint32_t)(p - wstart), q = (uint32_t)(p - P0);
        if (Q.nuc_on) {
            const uint32_t c = codes[q];
            if (c > 3u) continue;
            if (carry || always_main || i >= ov) a.v[c]++;
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
            if (b & 4u) a.v[4] += l; else a.v[5] += l;
            if (b & 2u) a.v[6] += l; else a.v[7] += l;
        }
    }
}

// ONE pass per tile of 4096 positions of one scanned region (round 3; rounds 1-2 wrote a 4 B/base match mask to HBM
// and read it back three times).  The workgroup
//   1. stages the tile's bases into LDS as 2-bit codes (coalesced 16-byte loads; a non-ACGT byte becomes 0xFF) beside
//      the pattern lists (per length: ascending 2-bit codes + {forward, canonical}) and one prefix bitmap per length
//      (which min(l, 6)-mers start a pattern: most positions stop there);
//   2. matches: every position extends its l-mer code length by length and looks it up (bitmap, then binary search
//      in LDS); the result — 3 bits {match, forward, canonical} per length — stays in LDS;
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
    // layout: mask u32[kTile] | pcode u64[lds_patterns] | bitmap u32[8][128] | part u32[8] | pflag u8[lds_patterns] | codes u8[...]
    uint32_t *mask = (uint32_t *)lds;
    u64 *pcode = (u64 *)(lds + kTile * 4u);
    uint32_t *bitmap = (uint32_t *)(lds + kTile * 4u + (size_t)lds_patterns * 8u);
    uint32_t *part = bitmap + 8u * 128u;
    unsigned char *pflag = (unsigned char *)(part + 8);
    unsigned char *codes = pflag + ((lds_patterns + 15u) & ~15u);
    if (blockIdx.x >= ntiles) return;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t npat = G.first[G.nlen];
    const bool lds_lists = lds_patterns != 0u;
    for (uint32_t i = tid; i < 8u * 128u; i += 256u) bitmap[i] = 0u;
    if (lds_lists)
        for (uint32_t i = tid; i < npat; i += 256u) { pcode[i] = G.codes[i]; pflag[i] = G.flags[i]; }
    __syncthreads();
    for (uint32_t li = 0; li < G.nlen; ++li) {
        const uint32_t q = G.len[li] < 6u ? G.len[li] : 6u;
        for (uint32_t i = G.first[li] + tid; i < G.first[li + 1]; i += 256u) {
            const uint32_t pre = (uint32_t)(lds_lists ? pcode[i] : G.codes[i]) & ((1u << (2u * q)) - 1u);
            atomicOr(&bitmap[li * 128u + (pre >> 5)], 1u << (pre & 31u));
        }
    }
    const TsGeneralTile T = tiles[blockIdx.x];
    // 1. stage: bases [0, avail) of the tile (avail = what lies between its start and the region end, at most
    // kTile + kHalo); layout offsets are 16-byte aligned per segment, tiles start at multiples of kTile inside it
    const uint32_t avail = T.avail;
    const unsigned char *src = in + T.in_off;
    for (uint32_t i = tid * 16u; i < avail; i += 256u * 16u) {
        unsigned char b[16];
        if (((uintptr_t)(src + i) & 15u) == 0u && i + 16u <= avail) {
            *(uint4 *)b = *(const uint4 *)(src + i);
        } else {
            for (uint32_t q = 0; q < 16u; ++q) b[q] = i + q < avail ? src[i + q] : (unsigned char)0;
        }
        for (uint32_t q = 0; q < 16u; ++q) codes[i + q] = (unsigned char)base_code_byte(b[q], Q.fold);
    }
    __syncthreads();
    // 2. matches
    for (uint32_t j = tid; j < T.n; j += 256u) {
        uint32_t out = 0, have = 0;
        u64 code = 0;
        bool ok = true;
        for (uint32_t li = 0; li < G.nlen; ++li) {
            const uint32_t l = G.len[li];
            if (j + l > avail) break;                // lengths ascend; a match may not cross the region end
            while (ok && have < l) {
                const uint32_t c = codes[j + have];
                if (c > 3u) { ok = false; break; }
                code |= (u64)c << (2u * have);
                ++have;
            }
            if (!ok) break;                          // a non-ACGT base kills this and every longer pattern
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
    if (!tips && T.n) {
        const u64 nwin = (n + Q.s - 1u) / Q.s;
        const u64 kw_lo = P0 >= Q.w ? (P0 - Q.w) / Q.s + 1u : 0u;             // first call whose window reaches the tile
        u64 kw_hi = (P0 + T.n - 1u) / Q.s;                                    // last call that starts inside it
        if (kw_hi >= nwin) kw_hi = nwin - 1u;
        const bool carries = Q.w != Q.s;
        const u64 items = (kw_hi - kw_lo + 1u) * (carries ? 2u : 1u);
        uint32_t *const wrec = win_out + seg_win_base[T.seg] * 8ull;
        for (u64 it = wave; it < items; it += 4u) {
            const u64 kw = kw_lo + (carries ? it >> 1 : it);
            const bool carry = carries && (it & 1u);
            if (carry && kw + 1u >= nwin) continue;
            Acc a = {{0, 0, 0, 0, 0, 0, 0, 0}};
            window_tile_part(codes, mask, G, Q, n, kw, carry, P0, T.n, lane, a);
            uint32_t t[8];
            for (int i = 0; i < 8; ++i) t[i] = wave_total(a.v[i]);
            // A C G T (codes A0 C1 T2 G3), then the four covered counters
            const uint32_t mine = lane == 0u ? t[0] : lane == 1u ? t[1] : lane == 2u ? t[3] : lane == 3u ? t[2]
                                : lane == 4u ? t[4] : lane == 5u ? t[5] : lane == 6u ? t[6] : t[7];
            if (lane < 8u && mine) atomicAdd(&wrec[(kw + (carry ? 1u : 0u)) * 8ull + lane], mine);
        }
    }
    // 4. match records: wave v owns the 1024 consecutive positions [1024 v, 1024 v + 1024)
    uint32_t wave_cnt = 0;
    for (uint32_t r = 0; r < 16u; ++r) {
        const uint32_t j = wave * 1024u + r * 64u + lane;
        uint32_t v = j < T.n ? (mask[j] & 0xFFFFFFu) : 0u;
        uint32_t keep = 0;
        for (uint32_t li = 0; v && li < G.nlen; ++li)
            if (((v >> (3u * li)) & 1u) && (tips || full_scan_pushes(P0 + j, G.len[li], n, Q))) keep |= 1u << li;
        if (j < T.n) mask[j] = v | (keep << 24);
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
    if (total > slot_cap) return;
    uint32_t *dst = records + (u64)blockIdx.x * slot_cap;
    for (uint32_t r = 0; r < 16u; ++r) {
        const uint32_t j = wave * 1024u + r * 64u + lane;
        const uint32_t v = j < T.n ? mask[j] : 0u;
        const uint32_t keep = v >> 24;
        const uint32_t c = (uint32_t)__popc(keep);
        const uint32_t incl = wave_inclusive(c, lane);
        uint32_t at = base + incl - c;
        for (uint32_t li = 0; li < G.nlen; ++li)
            if ((keep >> li) & 1u) dst[at++] = (j << 5) | (li << 2) | ((v >> (3u * li + 1u)) & 3u);
        base += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
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

}  // namespace

unsigned long long ts_k_general_lds_bytes(const TsGenericPatterns *G, uint32_t *lds_patterns) {
    const uint32_t npat = G->first[G->nlen];
    const uint32_t lp = npat <= kMaxLdsPatterns ? (npat ? npat : 1u) : 0u;
    *lds_patterns = lp;
    return (unsigned long long)kTile * 4u + (unsigned long long)lp * 8u + 8u * 128u * 4u + 32u + ((lp + 15u) & ~15u) + kTile + kHalo + 16u;
}

int ts_k_launch_general_fused(const unsigned char *in, const TsGeneralTile *tiles, uint32_t ntiles,
                              const unsigned long long *seg_len, const unsigned long long *seg_win_base,
                              const TsGenericPatterns *G, const TsGenericGeom *Q, int tips, uint32_t slot_cap,
                              uint32_t *tile_stats, uint32_t *records, uint32_t *win_out, uint32_t *overflow, void *stream) {
    if (ntiles == 0) return 0;
    uint32_t lp = 0;
    const unsigned long long lds = ts_k_general_lds_bytes(G, &lp);
    hipLaunchKernelGGL(ts_general_fused, dim3(ntiles), dim3(256), (size_t)lds, (hipStream_t)stream, in, tiles, ntiles,
                       (const u64 *)seg_len, (const u64 *)seg_win_base, *G, *Q, tips, slot_cap, lp, tile_stats, records, win_out, overflow);
    return (int)hipGetLastError();
}

int ts_k_launch_general_compact(const uint32_t *tile_stats, const unsigned long long *tile_off, const uint32_t *records,
                                uint32_t slot_cap, uint32_t ntiles, uint32_t *dense, void *stream) {
    if (ntiles == 0) return 0;
    hipLaunchKernelGGL(ts_general_compact, dim3((ntiles + 3u) / 4u), dim3(256), 0, (hipStream_t)stream, tile_stats,
                       (const u64 *)tile_off, records, slot_cap, ntiles, dense);
    return (int)hipGetLastError();
}
