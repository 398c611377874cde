// Microbenchmark: sustained issue rate of the integer VALU instructions the scan kernel is made of,
// at 4 waves per SIMD (16 waves per CU), to price its instruction count.  hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int KIND>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed * (threadIdx.x + 1) + i * 0x9E3779B9u;
    uint32_t s = seed | 1u;
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
    hipMalloc(&d, 256 * 1024 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(1024), 0, 0, d, 100, 3u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(1024), 0, 0, d, iters, 3u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: 4 waves x iters x ops
    const double winstr = 4.0 * iters * ops_per_iter;
    printf("%-22s %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, ms,
           ms * 1e6 / winstr, ms * 1e6 / winstr * 2.4);
    hipFree(d);
}

int main() {
    run<0>("v_alignbit_b32", 8);
    run<1>("v_bfe_u32 + add", 16);
    run<2>("v_lshl_add_u32", 8);
    run<3>("v_lshrrev + v_xor", 16);
    run<4>("v_dot4_u32_u8", 8);
    run<5>("v_perm_b32", 8);
    run<6>("v_bcnt + add", 8);
    run<7>("v_and_b32", 8);
    return 0;
}
