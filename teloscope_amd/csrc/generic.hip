// generic.hip — general-case kernels of libteloscan (gfx950).
//
// The tiled kernel in kernels.hip covers uniform-length pattern sets under the geometry where
// the reference's per-window carry loop has a closed form.  Everything else — mixed-length
// pattern sets, pattern lengths up to 32, and window/step pairs where the reference's
// `uint32` start index wraps (src/teloscope.cpp:413-415) — goes through these two kernels,
// which restate the reference's semantics literally instead of in closed form:
//
//   ts_generic_match    one thread per base: for every pattern length, the 2-bit code of the
//                       l-mer starting there is looked up in a sorted code list (binary search);
//                       result = one dword per base, 3 bits {match, forward, canonical} per length.
//   ts_generic_windows  one thread per window: re-derives exactly which bases and matches
//                       analyzeWindow (src/teloscope.cpp:387-534) adds to this window's record
//                       in its own scan ("main") and through the carry from the previous window.
//
// This is the slow, exact path (4 B/base of intermediate traffic); it exists so that no
// parameter set is answered by anything other than the GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ts_internal.h"

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ int base_code_dev(unsigned char c, uint32_t fold) {
    if (fold) c &= 0xDFu;
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'T': return 2;
        case 'G': return 3;
        default:  return -1;
    }
}

// seq[0..n): the segment (or tips region).  mask[p] bit 3*li+0: a pattern of length lens[li]
// starts at p and ends inside [0,n); +1: forward; +2: canonical.
__global__ void ts_generic_match(const unsigned char *seq, u64 n, const TsGenericPatterns G, uint32_t fold,
                                 uint32_t *mask) {
    const u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    uint32_t out = 0;
    u64 code = 0;
    uint32_t have = 0;                       // bases encoded so far
    bool ok = true;
    for (uint32_t li = 0; li < G.nlen; ++li) {
        const uint32_t l = G.len[li];
        if (p + l > n) break;                // lengths ascend
        while (ok && have < l) {
            const int c = base_code_dev(seq[p + have], fold);
            if (c < 0) { ok = false; break; }
            code |= (u64)c << (2u * have);
            ++have;
        }
        if (!ok) break;                      // a non-ACGT base kills this and every longer pattern
        // binary search in the sorted code list of this length
        uint32_t lo = G.first[li], hi = G.first[li + 1];
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            const u64 cm = G.codes[mid];
            if (cm < code) lo = mid + 1; else hi = mid;
        }
        if (lo < G.first[li + 1] && G.codes[lo] == code)
            out |= (1u | ((uint32_t)G.flags[lo] << 1)) << (3u * li);
    }
    mask[p] = out;
}

struct Acc { uint32_t nuc[4]; uint32_t can, noncan, fwd, rev; };

// Adds what one analyzeWindow() call over window `kw` contributes either to its own record
// (carry == false: bases with i >= mainlo, matches with j >= ov or always for window 0 / ov == 0)
// or to the next window's record (carry == true: i >= step).
__device__ void window_scan_part(const unsigned char *seq, const uint32_t *mask, const TsGenericPatterns &G,
                                 const TsGenericGeom &Q, u64 kw, bool carry, Acc &a) {
    const u64 wstart = kw * Q.s;
    const uint32_t cws = (uint32_t)((Q.n - wstart) < Q.w ? (Q.n - wstart) : Q.w);
    const uint32_t ov = Q.w - Q.s;
    const bool always_main = (ov == 0 || wstart == 0);
    // uint32 arithmetic on purpose (wraps when the longest pattern exceeds step or overlap)
    const uint32_t t1 = Q.s - Q.longest, t2 = ov - Q.longest;
    const uint32_t start_index = always_main ? 0u : (t1 < t2 ? t1 : t2);
    for (uint32_t i = start_index; i < cws; ++i) {
        if (carry && i < Q.s) { i = Q.s - 1u; continue; }          // the carry only takes i >= step
        const u64 p = wstart + i;
        if (Q.nuc_on) {
            const int c = base_code_dev(seq[p], Q.fold);
            if (c < 0) continue;
            if (carry || always_main || i >= ov) a.nuc[c]++;
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
            if (b & 4u) a.can += l; else a.noncan += l;
            if (b & 2u) a.fwd += l; else a.rev += l;
        }
    }
}

__global__ void ts_generic_windows(const unsigned char *seq, const uint32_t *mask, const TsGenericPatterns G,
                                   const TsGenericGeom Q, u64 nwin, uint32_t *out) {
    const u64 kw = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (kw >= nwin) return;
    Acc a = {{0, 0, 0, 0}, 0, 0, 0, 0};
    window_scan_part(seq, mask, G, Q, kw, false, a);
    if (kw > 0 && Q.w != Q.s) window_scan_part(seq, mask, G, Q, kw - 1, true, a);
    uint32_t *o = out + kw * 8ull;
    o[0] = a.nuc[0]; o[1] = a.nuc[1]; o[2] = a.nuc[3]; o[3] = a.nuc[2];   // A C G T (codes A0 C1 T2 G3)
    o[4] = a.can; o[5] = a.noncan; o[6] = a.fwd; o[7] = a.rev;
}

}  // namespace

int ts_k_launch_generic_match(const unsigned char *seq, unsigned long long n, const TsGenericPatterns *G,
                              uint32_t fold, uint32_t *mask, void *stream) {
    if (n == 0) return 0;
    const unsigned long long nb = (n + 255ull) / 256ull;
    hipLaunchKernelGGL(ts_generic_match, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, seq, n, *G, fold, mask);
    return (int)hipGetLastError();
}

int ts_k_launch_generic_windows(const unsigned char *seq, const uint32_t *mask, const TsGenericPatterns *G,
                                const TsGenericGeom *Q, unsigned long long nwin, uint32_t *out, void *stream) {
    if (nwin == 0) return 0;
    const unsigned long long nb = (nwin + 63ull) / 64ull;
    hipLaunchKernelGGL(ts_generic_windows, dim3((unsigned)nb), dim3(64), 0, (hipStream_t)stream, seq, mask, *G, *Q,
                       nwin, out);
    return (int)hipGetLastError();
}
