// unpack.hip — the device half of the packed upload (pack.cpp): a staged chunk of 2-bit codes back into the byte layout
// the kernels read, at HBM speed (1 B/base written, 0.25 read: ~0.3 ms per 256 MB group against the 4.5 ms its upload
// takes even packed), then 'N' over the chunk's invalid runs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ts_internal.h"

namespace {

// Sixteen bases per thread: one aligned 16-byte store.  dst_aligned: a 16-byte aligned address at or below the chunk's
// first byte; lead = bytes between the two (0..15).  packed: chunk position x at bits 2 (x & 3) of byte x >> 2, 4-byte
// aligned, padded by 8 readable bytes; the chunk's first byte is position `first` (< 64: chunks are packed from a
// 64-position boundary of the layout).
// (64-thread workgroups: these kernels run beside the scan of the group before, where a 256-thread workgroup cannot be
// placed — blockcall.hip, kSideWg)
__global__ __launch_bounds__(64)
void ts_unpack_bases(const uint32_t *packed, uint32_t first, unsigned char *dst_aligned, uint32_t lead, unsigned long long n) {
    const unsigned long long g = (unsigned long long)blockIdx.x * 64u + threadIdx.x;        // 16-byte group of the destination
    const unsigned long long b0 = g * 16ull;                                                  // its first byte, relative to dst_aligned
    if (b0 >= lead + n) return;
    // chunk position of that byte (negative for the group that holds the lead: handled by the byte path)
    if (b0 >= lead && b0 + 16ull <= lead + n) {
        const unsigned long long x = b0 - lead + first;
        const uint32_t w0 = packed[x >> 4], w1 = packed[(x >> 4) + 1ull];
        const uint32_t c = __funnelshift_r(w0, w1, 2u * (uint32_t)(x & 15ull));               // 16 codes
        uint32_t o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t c4 = (c >> (8 * q)) & 0xFFu;                                       // four codes
            // codes -> v_perm selectors (one per byte), letters 'A' 'C' 'T' 'G' by code
            const uint32_t sel = (c4 & 3u) | ((c4 >> 2) & 3u) << 8 | ((c4 >> 4) & 3u) << 16 | ((c4 >> 6) & 3u) << 24;
            o[q] = __builtin_amdgcn_perm(0u, 0x47544341u, sel);
        }
        *(uint4 *)(dst_aligned + b0) = make_uint4(o[0], o[1], o[2], o[3]);
        return;
    }
    for (uint32_t i = 0; i < 16u; ++i) {                                                      // the two end groups
        const unsigned long long b = b0 + i;
        if (b < lead || b >= lead + n) continue;
        const unsigned long long x = b - lead + first;
        const uint32_t code = (packed[x >> 4] >> (2u * (uint32_t)(x & 15ull))) & 3u;
        dst_aligned[b] = (unsigned char)((0x47544341u >> (8u * code)) & 0xFFu);
    }
}

// One wave per invalid run {start, len} (positions relative to the chunk): 'N'.
__global__ __launch_bounds__(64)
void ts_poke_invalid(const uint2 *runs, uint32_t nruns, unsigned char *dst) {
    const uint32_t r = blockIdx.x;
    if (r >= nruns) return;
    const uint2 run = runs[r];
    for (uint32_t i = threadIdx.x & 63u; i < run.y; i += 64u) dst[(unsigned long long)run.x + i] = (unsigned char)'N';
}

}  // namespace

int ts_k_launch_unpack(const void *packed, uint32_t first, void *dst, unsigned long long n, const void *runs, uint32_t nruns,
                       void *runs_base, void *stream) {
    if (n == 0) return 0;
    const uintptr_t a = (uintptr_t)dst;
    const uint32_t lead = (uint32_t)(a & 15u);
    const unsigned long long groups = (lead + n + 15ull) / 16ull;
    hipLaunchKernelGGL(ts_unpack_bases, dim3((unsigned)((groups + 63ull) / 64ull)), dim3(64), 0, (hipStream_t)stream,
                       (const uint32_t *)packed, first, (unsigned char *)(a - lead), lead, n);
    if (nruns)
        hipLaunchKernelGGL(ts_poke_invalid, dim3(nruns), dim3(64), 0, (hipStream_t)stream, (const uint2 *)runs, nruns,
                           (unsigned char *)runs_base);
    return (int)hipGetLastError();
}
