// exchange.hip — gfx950 kernels behind ts_batch_export / ts_batch_adopt: the device side of the multi-GPU
// gather.  A scan leaves its match records in per-wave regions, addressable through the tile
// directory; what ranks exchange is ONE stream per rank in tile order (= position order inside every segment),
// so that the concatenation over ranks is again a tile-ordered stream and the assembling rank only has to
// prefix-sum the tile counts to get its directory back.  All of it is HBM-bound copying: coalesced 4-byte
// records, one wave per tile.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ts_internal.h"


namespace {

typedef unsigned long long u64;

// One wavefront per workgroup and no LDS: these kernels also run beside the persistent scan kernel of the next batch, whose
// workgroups hold every CU's LDS (a kernel that asks for a single byte of it waits until they are gone).
constexpr uint32_t kScanThreads = 64;
constexpr uint32_t kScanPerThread = 8;
constexpr uint32_t kScanBlock = kScanThreads * kScanPerThread;      // tiles per workgroup

// wave-wide exclusive prefix sum of one u64 per lane; returns the lane's exclusive prefix, *total = the wave's sum
__device__ __forceinline__ u64 block_excl_scan(u64 v, u64 *total) {
    u64 incl = v;
    const uint32_t lane = threadIdx.x & 63u;
    for (uint32_t o = 1; o < 64u; o <<= 1) {
        const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)incl, (int)o), hi = (uint32_t)__shfl_up((int)(uint32_t)(incl >> 32), (int)o);
        if (lane >= o) incl += ((u64)hi << 32) | lo;
    }
    const uint32_t tlo = (uint32_t)__shfl((int)(uint32_t)incl, 63), thi = (uint32_t)__shfl((int)(uint32_t)(incl >> 32), 63);
    *total = ((u64)thi << 32) | tlo;
    return incl - v;
}

// pass 1: records per workgroup of kScanBlock tiles; also raises the "incomplete" flag when a wave's region
// overflowed in the scan (its tile counts then promise records that were never stored)
__global__ __launch_bounds__(kScanThreads)
void ts_tile_count_blocks(const uint32_t *tile_stats, uint32_t ntiles, u64 *block_sums,
                          const uint32_t *wave_fill, uint32_t region_cap, uint32_t nwaves, u64 *total_out) {
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * kScanPerThread;
    u64 v = 0;
    for (uint32_t i = 0; i < kScanPerThread; ++i)
        if (base + i < ntiles) v += tile_stats[4ull * (base + i)];
    u64 total;
    (void)block_excl_scan(v, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
    if (wave_fill) {
        bool over = false;
        for (uint32_t w = blockIdx.x * kScanThreads + threadIdx.x; w < nwaves; w += gridDim.x * kScanThreads)
            over |= wave_fill[w] > region_cap;
        if (__ballot(over) != 0ull && threadIdx.x == 0) atomicOr(total_out + 1, 1ull);
    }
}

// pass 2 (one workgroup): exclusive scan of the block sums in place; the grand total goes to block_sums[nblocks]
__global__ __launch_bounds__(kScanThreads)
void ts_tile_scan_blocks(u64 *block_sums, uint32_t nblocks) {
    u64 carry = 0;
    for (uint32_t b0 = 0; b0 < nblocks; b0 += kScanThreads) {
        const uint32_t i = b0 + threadIdx.x;
        const u64 v = i < nblocks ? block_sums[i] : 0ull;
        u64 total;
        const u64 ex = block_excl_scan(v, &total);
        if (i < nblocks) block_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) block_sums[nblocks] = carry;
}

// pass 3: tile_off[t] = records of the tiles before t; tile_off[ntiles] = all records.  With total_out: the
// grand total and the "stream does not fit" flag.
__global__ __launch_bounds__(kScanThreads)
void ts_tile_offsets(const uint32_t *tile_stats, uint32_t ntiles, const u64 *block_sums, uint32_t nblocks,
                     u64 *tile_off, u64 *total_out, u64 capacity) {
    const uint32_t base = blockIdx.x * kScanBlock + threadIdx.x * kScanPerThread;
    uint32_t c[kScanPerThread];
    u64 v = 0;
    for (uint32_t i = 0; i < kScanPerThread; ++i) {
        c[i] = base + i < ntiles ? tile_stats[4ull * (base + i)] : 0u;
        v += c[i];
    }
    u64 total;
    u64 run = block_sums[blockIdx.x] + block_excl_scan(v, &total);
    for (uint32_t i = 0; i < kScanPerThread; ++i) {
        if (base + i < ntiles) tile_off[base + i] = run;
        run += c[i];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const u64 all = block_sums[nblocks];
        tile_off[ntiles] = all;
        if (total_out) {
            total_out[0] = all;
            if (all > capacity) atomicOr(total_out + 1, 1ull);
        }
    }
}

// One wave per tile: its records, wherever the scan's wave put them, to their place in the tile-ordered stream.
__global__ __launch_bounds__(256)
void ts_tile_order_copy(const uint32_t *tile_stats, const u64 *region_off, const uint32_t *regions,
                        const u64 *dense_off, uint32_t ntiles, uint32_t *dense, const u64 *total_out, int rec16) {
    if (total_out[1] != 0ull) return;                        // incomplete: the caller rescans (ts_batch_sync) and exports again
    const uint32_t t = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (t >= ntiles) return;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = tile_stats[4ull * t];
    uint32_t *dst = dense + dense_off[t];
    // (the tile-ordered stream is 32 bits per record whatever the regions hold)
    if (rec16) { const uint16_t *src = (const uint16_t *)regions + region_off[t]; for (uint32_t i = lane; i < n; i += 64u) dst[i] = src[i]; }
    else { const uint32_t *src = regions + region_off[t]; for (uint32_t i = lane; i < n; i += 64u) dst[i] = src[i]; }
}

// The exchange's wire format: every u32 of a rank's three arrays fits 16 bits (a packed match record is a
// tile-relative position < 2^14 plus two flag bits; window fields are at most k x window; tile counts at most a
// tile's bases), so they travel as u16 — half the bytes over the per-link-bound xGMI gather — and are widened again
// where they land.  Eight values per lane: one 16-byte load, two 16-byte stores.
__global__ __launch_bounds__(256)
void ts_widen_u16(const uint16_t *src, uint32_t *dst, u64 n) {
    const u64 i8 = ((u64)blockIdx.x * 256u + threadIdx.x) * 8ull;
    if (i8 >= n) return;
    if (i8 + 8ull <= n && (((uintptr_t)(src + i8)) & 15u) == 0u && (((uintptr_t)(dst + i8)) & 15u) == 0u) {
        const uint4 v = *(const uint4 *)(src + i8);
        *(uint4 *)(dst + i8) = make_uint4(v.x & 0xFFFFu, v.x >> 16, v.y & 0xFFFFu, v.y >> 16);
        *(uint4 *)(dst + i8 + 4) = make_uint4(v.z & 0xFFFFu, v.z >> 16, v.w & 0xFFFFu, v.w >> 16);
    } else {
        for (u64 i = i8; i < n && i < i8 + 8ull; ++i) dst[i] = src[i];
    }
}

uint32_t scan_blocks(uint32_t ntiles) { return ntiles ? (ntiles + kScanBlock - 1u) / kScanBlock : 1u; }

}  // namespace

unsigned long long ts_k_scan_tmp_bytes(uint32_t ntiles) { return ((unsigned long long)scan_blocks(ntiles) + 2ull) * 8ull; }

int ts_k_launch_tile_offsets(const uint32_t *tile_stats, uint32_t ntiles, unsigned long long *tile_off, void *tmp,
                             void *stream) {
    const uint32_t nb = scan_blocks(ntiles);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(ts_tile_count_blocks, dim3(nb), dim3(kScanThreads), 0, st, tile_stats, ntiles, (u64 *)tmp,
                       (const uint32_t *)nullptr, 0u, 0u, (u64 *)nullptr);
    hipLaunchKernelGGL(ts_tile_scan_blocks, dim3(1), dim3(kScanThreads), 0, st, (u64 *)tmp, nb);
    hipLaunchKernelGGL(ts_tile_offsets, dim3(nb), dim3(kScanThreads), 0, st, tile_stats, ntiles, (const u64 *)tmp, nb,
                       tile_off, (u64 *)nullptr, 0ull);
    return (int)hipGetLastError();
}

int ts_k_launch_tile_order_export(const uint32_t *tile_stats, const unsigned long long *region_off, const uint32_t *regions,
                                  const uint32_t *wave_fill, uint32_t region_cap, uint32_t nwaves, uint32_t ntiles,
                                  unsigned long long *dense_off, void *tmp, uint32_t *dense, unsigned long long capacity,
                                  unsigned long long *total_out, int rec16, void *stream) {
    const uint32_t nb = scan_blocks(ntiles);
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(total_out, 0, 16, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(ts_tile_count_blocks, dim3(nb), dim3(kScanThreads), 0, st, tile_stats, ntiles, (u64 *)tmp,
                       wave_fill, region_cap, nwaves, total_out);
    hipLaunchKernelGGL(ts_tile_scan_blocks, dim3(1), dim3(kScanThreads), 0, st, (u64 *)tmp, nb);
    hipLaunchKernelGGL(ts_tile_offsets, dim3(nb), dim3(kScanThreads), 0, st, tile_stats, ntiles, (const u64 *)tmp, nb,
                       dense_off, total_out, capacity);
    if (ntiles)
        hipLaunchKernelGGL(ts_tile_order_copy, dim3((ntiles + 3u) / 4u), dim3(256), 0, st, tile_stats, region_off, regions,
                           (const u64 *)dense_off, ntiles, dense, (const u64 *)total_out, rec16);
    return (int)hipGetLastError();
}

int ts_k_launch_widen_u16(const uint16_t *src, uint32_t *dst, unsigned long long n, void *stream) {
    if (n == 0) return 0;
    const unsigned long long nb = (n + 2047ull) / 2048ull;
    hipLaunchKernelGGL(ts_widen_u16, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, src, dst, n);
    return (int)hipGetLastError();
}


// ---------------------------------------------------------------------------------------------------------------------
// Box calibration (bench.py: roofline.box): what THIS device issues and streams, measured the same way on every box, so that a
// bench line from a box that runs everything a few per cent slower (power cap, clocks) can be told from a slower kernel.
//   issue probe  every SIMD holds four waves that each execute `iters` x 64 independent v_and_b32 (hand-written, nothing to
//                fuse or reorder): wave-instructions per nanosecond over the whole device;
//   copy probe   a grid-strided 16-byte copy of `bytes`: read + write bytes per nanosecond.
namespace {
#define TS_V8_AND "v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n" \
                  "v_and_b32 %4, %8, %4\n v_and_b32 %5, %8, %5\n v_and_b32 %6, %8, %6\n v_and_b32 %7, %8, %7\n"
__global__ __launch_bounds__(256) void ts_issue_probe(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a0 = seed * threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const uint32_t s = seed | 0xFFFF0000u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g)
            asm volatile(TS_V8_AND : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));
    }
    if ((a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7) == 0x12345u) out[0] = a0;        // (never true: keeps the chain alive)
}
__global__ __launch_bounds__(256) void ts_copy_probe(const uint4 *src, uint4 *dst, unsigned long long n16) {
    for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < n16; i += (unsigned long long)gridDim.x * 256ull) dst[i] = src[i];
}
}  // namespace

// issue_per_ns / copy_bytes_per_ns: best of five launches each (six run, the first — a cold start — is not counted).
// scratch: device memory of 2 x bytes (bytes a multiple of 16).
int ts_k_box_probe(void *scratch, unsigned long long bytes, int num_cu, double *issue_per_ns, double *copy_bytes_per_ns, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    struct Events {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }      // on every return path
    } ev;
    if (hipEventCreate(&ev.e0) != hipSuccess || hipEventCreate(&ev.e1) != hipSuccess) return (int)hipGetLastError();
    hipEvent_t e0 = ev.e0, e1 = ev.e1;
    const int iters = 4000;
    const unsigned grid = (unsigned)(num_cu > 0 ? num_cu : 256) * 4u;             // 4 workgroups of 4 waves per CU: four waves per SIMD
    double best_issue = 0, best_copy = 0;
    for (int r = 0; r < 6; ++r) {
        (void)hipEventRecord(e0, st);
        hipLaunchKernelGGL(ts_issue_probe, dim3(grid), dim3(256), 0, st, (uint32_t *)scratch, iters, 0x9E3779B9u);
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) return (int)hipGetLastError();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)grid * 4.0 * iters * 64.0;                   // waves x iterations x 64 instructions
        if (r && ms > 0 && instr / (ms * 1e6) > best_issue) best_issue = instr / (ms * 1e6);
    }
    const unsigned long long n16 = bytes / 16u;
    for (int r = 0; r < 6 && n16; ++r) {
        (void)hipEventRecord(e0, st);
        hipLaunchKernelGGL(ts_copy_probe, dim3(grid * 4u), dim3(256), 0, st, (const uint4 *)scratch, (uint4 *)((char *)scratch + bytes), n16);
        (void)hipEventRecord(e1, st);
        if (hipEventSynchronize(e1) != hipSuccess) return (int)hipGetLastError();
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (r && ms > 0 && 2.0 * (double)bytes / (ms * 1e6) > best_copy) best_copy = 2.0 * (double)bytes / (ms * 1e6);
    }
    *issue_per_ns = best_issue; *copy_bytes_per_ns = best_copy;
    return (int)hipGetLastError();
}
