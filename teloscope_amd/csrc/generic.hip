// generic.hip — the GENERAL path of libteloscan (gfx950): every parameter set the tiled kernel in kernels.hip
// does not take.  That kernel covers uniform-length pattern sets (3 <= k <= 8) under the geometry where the
// reference's per-window carry loop has a closed form; what is left — mixed-length pattern sets (several matches per
// position), pattern lengths up to 32, and window/step pairs where the reference's `uint32` start index wraps
// (src/teloscope.cpp:413-415) — is restated here LITERALLY (the main / carry attribution of analyzeWindow, not its
// closed form), for a whole batch of segments at a time and with nothing but ordering work left to the host:
//
//   ts_general_match    one workgroup per tile of 4096 positions of one scanned region.  The tile's bases are staged
//                       once into LDS as 2-bit codes (coalesced 16-byte loads; a non-ACGT byte becomes 0xFF), the
//                       pattern lists — per length, ascending 2-bit codes + {forward, canonical} — sit in LDS beside
//                       them; every position extends its l-mer code length by length and looks it up by binary
//                       search in LDS.  Result: one dword per position, 3 bits {match, forward, canonical} per length.
//   ts_general_records  (count pass, then emit pass around a prefix sum over the tiles) the matches the reference
//                       pushes to its match vectors, as packed records in position order: a full scan keeps a match
//                       only if some window's own scan pushes it (src/teloscope.cpp:485; the window it belongs to is
//                       recomputed on the host, which orders by it), a tips-only scan keeps everything in the region.
//   ts_general_windows  one WAVEFRONT per window: the lanes stride over the bases analyzeWindow visits for this
//                       window's own record and for the carry from the previous window, a DPP reduction adds the
//                       eight counters up.
//
// Traffic: 1 B/base in, 4 B/base of match mask out and ~2 x 4 B/base back in (windows), 4 B/match out — HBM-bound
// streaming, a few times the tiled kernel's traffic, no LDS tables larger than the pattern lists.
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

__global__ __launch_bounds__(256)
void ts_general_match(const unsigned char *in, const TsGeneralTile *tiles, uint32_t ntiles, const TsGenericPatterns G,
                      uint32_t fold, uint32_t *mask) {
    __shared__ unsigned char codes[kTile + kHalo + 16];
    __shared__ u64 pcode[kMaxLdsPatterns];
    __shared__ unsigned char pflag[kMaxLdsPatterns];
    const uint32_t tid = threadIdx.x;
    const uint32_t npat = G.first[G.nlen];
    const bool lds_lists = npat <= kMaxLdsPatterns;
    if (lds_lists)
        for (uint32_t i = tid; i < npat; i += 256u) { pcode[i] = G.codes[i]; pflag[i] = G.flags[i]; }
    if (blockIdx.x >= ntiles) return;
    const TsGeneralTile T = tiles[blockIdx.x];
    // stage: bases [0, avail) of the tile (avail = what lies between its start and the region end, at most
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
        for (uint32_t q = 0; q < 16u; ++q) codes[i + q] = (unsigned char)base_code_byte(b[q], fold);
    }
    __syncthreads();
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
        mask[T.in_off + j] = out;
    }
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

// EMIT == false: counts the records of every tile into tile_stats[4 t] (the layout ts_k_launch_tile_offsets
// reads); EMIT == true: writes them, in position then length order, from tile_off[t].
// Record: (tile-relative position << 5) | (length index << 2) | canonical << 1 | forward.
template <bool EMIT>
__global__ __launch_bounds__(256)
void ts_general_records(const uint32_t *mask, const TsGeneralTile *tiles, uint32_t ntiles, const u64 *seg_len,
                        const TsGenericPatterns G, const TsGenericGeom Q, int tips, uint32_t *tile_stats,
                        const u64 *tile_off, uint32_t *records) {
    __shared__ uint32_t part[256];
    if (blockIdx.x >= ntiles) return;
    const TsGeneralTile T = tiles[blockIdx.x];
    const uint32_t tid = threadIdx.x;
    const u64 n = seg_len[T.seg];
    // thread t owns the 16 consecutive positions [16 t, 16 t + 16): thread order is position order
    const uint32_t j0 = tid * 16u;
    uint32_t m[16];
    uint32_t cnt = 0;
    for (uint32_t q = 0; q < 16u; ++q) {
        const uint32_t j = j0 + q;
        uint32_t v = j < T.n ? mask[T.in_off + j] : 0u;
        if (v && !tips) {
            for (uint32_t li = 0; li < G.nlen; ++li)
                if (((v >> (3u * li)) & 1u) && !full_scan_pushes(T.seg_rel + j, G.len[li], n, Q)) v &= ~(7u << (3u * li));
        }
        m[q] = v;
        for (uint32_t li = 0; li < G.nlen; ++li) cnt += (v >> (3u * li)) & 1u;
    }
    part[tid] = cnt;
    __syncthreads();
    for (uint32_t o = 1; o < 256u; o <<= 1) {
        const uint32_t add = tid >= o ? part[tid - o] : 0u;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    if (!EMIT) {
        if (tid == 255u) *(uint4 *)&tile_stats[4ull * blockIdx.x] = make_uint4(part[255], 0u, 0u, 0u);
        return;
    }
    uint32_t *dst = records + tile_off[blockIdx.x] + (part[tid] - cnt);
    for (uint32_t q = 0; q < 16u; ++q) {
        const uint32_t v = m[q];
        if (!v) continue;
        for (uint32_t li = 0; li < G.nlen; ++li) {
            const uint32_t b = (v >> (3u * li)) & 7u;
            if (b & 1u) *dst++ = ((j0 + q) << 5) | (li << 2) | (b >> 1);
        }
    }
}

struct Acc { uint32_t v[8]; };            // A C T G (code order), canonical, non-canonical, forward, reverse covered

// What one analyzeWindow() call over window `kw` adds either to its own record (carry == false: bases with
// i >= mainlo, matches with j >= ov, or everything for window 0 / ov == 0) or to the next window's record
// (carry == true: i >= step); the wave's lanes stride over i.
__device__ __forceinline__ void window_scan_part(const unsigned char *seq, const uint32_t *mask, const TsGenericPatterns &G,
                                                 const TsGenericGeom &Q, u64 n, u64 kw, bool carry, uint32_t lane, Acc &a) {
    const u64 wstart = kw * Q.s;
    const uint32_t cws = (uint32_t)((n - wstart) < Q.w ? (n - wstart) : Q.w);
    const uint32_t ov = Q.w - Q.s;
    const bool always_main = (ov == 0 || wstart == 0);
    // uint32 arithmetic on purpose (wraps when the longest pattern exceeds step or overlap)
    const uint32_t t1 = Q.s - Q.longest, t2 = ov - Q.longest;
    uint32_t start_index = always_main ? 0u : (t1 < t2 ? t1 : t2);
    if (carry && start_index < Q.s) start_index = Q.s;               // the carry only takes i >= step
    if (start_index >= cws) return;
    for (uint32_t i = start_index + lane; i < cws; i += 64u) {
        const u64 p = wstart + i;
        if (Q.nuc_on) {
            const uint32_t c = base_code_byte(seq[p], Q.fold);
            if (c > 3u) continue;
            if (carry || always_main || i >= ov) a.v[c]++;
        }
        const uint32_t m = mask[p];
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

__device__ __forceinline__ uint32_t wave_total(uint32_t v) {
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

// one wave per window; seg_win_base[i] = index of segment i's first window record (ascending), [nseg] = all windows
__global__ __launch_bounds__(256)
void ts_general_windows(const unsigned char *in, const uint32_t *mask, const TsGenericPatterns G, const TsGenericGeom Q,
                        const u64 *seg_win_base, const u64 *seg_in_off, const u64 *seg_len, uint32_t nseg, u64 nwin,
                        uint32_t *out) {
    const u64 wi = (u64)blockIdx.x * 4u + (threadIdx.x >> 6);
    if (wi >= nwin) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t lo = 0, hi = nseg;                                       // the segment whose windows contain wi
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (seg_win_base[mid] <= wi) lo = mid; else hi = mid;
    }
    const u64 kw = wi - seg_win_base[lo];
    const unsigned char *seq = in + seg_in_off[lo];
    const uint32_t *msk = mask + seg_in_off[lo];
    const u64 n = seg_len[lo];
    Acc a = {{0, 0, 0, 0, 0, 0, 0, 0}};
    window_scan_part(seq, msk, G, Q, n, kw, false, lane, a);
    if (kw > 0 && Q.w != Q.s) window_scan_part(seq, msk, G, Q, n, kw - 1, true, lane, a);
    uint32_t t[8];
    for (int i = 0; i < 8; ++i) t[i] = wave_total(a.v[i]);
    if (lane == 0) {
        uint32_t *o = out + wi * 8ull;
        o[0] = t[0]; o[1] = t[1]; o[2] = t[3]; o[3] = t[2];          // A C G T (codes A0 C1 T2 G3)
        o[4] = t[4]; o[5] = t[5]; o[6] = t[6]; o[7] = t[7];
    }
}

}  // namespace

int ts_k_launch_general_match(const unsigned char *in, const TsGeneralTile *tiles, uint32_t ntiles,
                              const TsGenericPatterns *G, uint32_t fold, uint32_t *mask, void *stream) {
    if (ntiles == 0) return 0;
    hipLaunchKernelGGL(ts_general_match, dim3(ntiles), dim3(256), 0, (hipStream_t)stream, in, tiles, ntiles, *G, fold, mask);
    return (int)hipGetLastError();
}

int ts_k_launch_general_records(const uint32_t *mask, const TsGeneralTile *tiles, uint32_t ntiles,
                                const unsigned long long *seg_len, const TsGenericPatterns *G, const TsGenericGeom *Q,
                                int tips, uint32_t *tile_stats, const unsigned long long *tile_off, uint32_t *records,
                                int emit, void *stream) {
    if (ntiles == 0) return 0;
    if (emit)
        hipLaunchKernelGGL((ts_general_records<true>), dim3(ntiles), dim3(256), 0, (hipStream_t)stream, mask, tiles, ntiles,
                           (const u64 *)seg_len, *G, *Q, tips, tile_stats, (const u64 *)tile_off, records);
    else
        hipLaunchKernelGGL((ts_general_records<false>), dim3(ntiles), dim3(256), 0, (hipStream_t)stream, mask, tiles, ntiles,
                           (const u64 *)seg_len, *G, *Q, tips, tile_stats, (const u64 *)tile_off, records);
    return (int)hipGetLastError();
}

int ts_k_launch_general_windows(const unsigned char *in, const uint32_t *mask, const TsGenericPatterns *G,
                                const TsGenericGeom *Q, const unsigned long long *seg_win_base,
                                const unsigned long long *seg_in_off, const unsigned long long *seg_len, uint32_t nseg,
                                unsigned long long nwin, uint32_t *out, void *stream) {
    if (nwin == 0) return 0;
    const unsigned long long nb = (nwin + 3ull) / 4ull;
    hipLaunchKernelGGL(ts_general_windows, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, in, mask, *G, *Q,
                       (const u64 *)seg_win_base, (const u64 *)seg_in_off, (const u64 *)seg_len, nseg, nwin, out);
    return (int)hipGetLastError();
}
