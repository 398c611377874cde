// Microbenchmark: sustained issue rate of the integer VALU instructions the scan kernel is made of, at 4, 5 and 8
// waves per SIMD, to price its instruction count.  Every kind runs 8 independent dependency chains per
// lane, so nothing waits for a result; the kernel is compiled for <= 64 VGPRs, the grid and the block size set the
// occupancy (256 CUs: 256 x 1024 threads = 4 waves per SIMD, 512 x 640 = 5, 512 x 1024 = 8).
//   hipcc -O3 --offload-arch=gfx950 profiles/micro/valu_rate.hip -o profiles/micro/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i * 0x9E3779B9u;
    uint32_t s = seed | 1u, t = seed * 7u + 3u;
    asm volatile("" : "+s"(s), "+s"(t));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) a[i] = __builtin_amdgcn_alignbit(a[i], s, 7);
            if (KIND == 1) a[i] = __builtin_amdgcn_ubfe(a[i], 5, 7) + s;
            if (KIND == 2) a[i] = (a[i] << 3) + s;                         // v_lshl_add_u32
            if (KIND == 3) a[i] = (s >> (a[i] & 31u)) ^ a[i];              // v_lshrrev + v_xor
            if (KIND == 4) a[i] = __builtin_amdgcn_udot4(a[i], 0x40100401u, s, false);
            if (KIND == 5) a[i] = __builtin_amdgcn_perm(a[i], s, 0x07020501u);
            if (KIND == 6) a[i] = __popc(a[i]) + s;
            if (KIND == 7) a[i] = a[i] & s;                                // v_and_b32
            if (KIND == 8) a[i] = __builtin_amdgcn_bitop3_b32(a[i], s, t, 0xBE);
            if (KIND == 9) a[i] = (a[i] & s) | t;                          // v_and_or_b32
            if (KIND == 10) a[i] = (a[i] << 8) | s;                        // v_lshl_or_b32
            if (KIND == 11) a[i] = a[i] + s + t;                           // v_add3_u32
            if (KIND == 12) a[i] = __umul24(a[i], s);                      // v_mul_u32_u24
            if (KIND == 13) a[i] = a[i] > s ? a[i] - t : a[i] + t;         // v_cmp + v_cndmask / 2 adds
            if (KIND == 14) a[i] = a[i] + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a[i], 0x111, 0xf, 0xf, false);   // DPP row_shr:1 add
            if (KIND == 15) a[i] = (uint32_t)__builtin_ctz(a[i] | 0x80000000u) + s;   // v_ffbl + or + add
            if (KIND == 16) a[i] = __builtin_amdgcn_mbcnt_lo(a[i], s);     // v_mbcnt_lo
            if (KIND == 17) a[i] = __builtin_amdgcn_sad_u8(a[i], s, t);    // v_sad_u8
            if (KIND == 18) a[i] = (a[i] & 0xAAAAAAAAu) | (s & 0x55555555u);   // v_bfi_b32
        }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int KIND>
void run(const char *name, int ops_per_iter) {
    uint32_t *d;
    hipMalloc(&d, 512 * 1024 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const struct { int grid, block, waves; } occ[3] = {{256, 1024, 4}, {512, 640, 5}, {512, 1024, 8}};
    printf("%-24s", name);
    for (int o = 0; o < 3; ++o) {
        hipLaunchKernelGGL(k<KIND>, dim3(occ[o].grid), dim3(occ[o].block), 0, 0, d, 100, 3u);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(occ[o].grid), dim3(occ[o].block), 0, 0, d, iters, 3u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double winstr = (double)occ[o].waves * iters * ops_per_iter;      // wave-instructions per SIMD
        printf("  %d waves/SIMD: %.3f ms = %.2f cycles/instr @2.4GHz", occ[o].waves, ms, ms * 1e6 / winstr * 2.4);
    }
    printf("\n");
    hipFree(d);
}

int main() {
    run<0>("v_alignbit_b32", 8);
    run<1>("v_bfe_u32 + v_add", 16);
    run<2>("v_lshl_add_u32", 8);
    run<3>("v_lshrrev + v_xor", 16);
    run<4>("v_dot4_u32_u8", 8);
    run<5>("v_perm_b32", 8);
    run<6>("v_bcnt (+add fused)", 8);
    run<7>("v_and_b32", 8);
    run<8>("v_bitop3_b32", 8);
    run<9>("v_and_or_b32", 8);
    run<10>("v_lshl_or_b32", 8);
    run<11>("v_add3_u32", 8);
    run<12>("v_mul_u32_u24", 8);
    run<13>("v_cmp+cndmask+2 add (4)", 32);
    run<14>("DPP v_add row_shr:1", 8);
    run<15>("v_ffbl+v_or+v_add (3)", 24);
    run<16>("v_mbcnt_lo", 8);
    run<17>("v_sad_u8", 8);
    run<18>("v_bfi_b32 (+and)", 16);
    return 0;
}
