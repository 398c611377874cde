// Microbenchmark: what limits the VALU issue rate of a SIMD at the scan kernel's occupancy?  valu_rate.hip found one
// integer wave64 instruction per ~5 cycles at four waves per SIMD, not the 4 the 16-lane datapath allows.  This one
// writes the instruction stream by hand (inline asm, so the compiler neither fuses nor reorders) and varies: waves per
// SIMD (1..8; one workgroup per CU is forced by its LDS allocation), instructions per loop iteration (8 or 64: the share
// of s_add / s_cmp / s_cbranch), the encoding (VOP2 = 4 bytes, VOP3 = 8 bytes) and what shares the issue port
// (scalar ALU instructions, LDS reads) between the vector ones.
//   hipcc -O3 --offload-arch=gfx950 profiles/micro/valu_issue.hip -o profiles/micro/valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define V8_AND  "v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n" \
                "v_and_b32 %4, %8, %4\n v_and_b32 %5, %8, %5\n v_and_b32 %6, %8, %6\n v_and_b32 %7, %8, %7\n"
#define V8_PERM "v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n" \
                "v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9\n"
// 8 vector + 4 scalar ALU instructions
#define V8_SALU "v_and_b32 %0, %8, %0\n s_add_u32 %10, %10, 1\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n s_xor_b32 %10, %10, %9\n v_and_b32 %3, %8, %3\n" \
                "v_and_b32 %4, %8, %4\n s_add_u32 %10, %10, 3\n v_and_b32 %5, %8, %5\n v_and_b32 %6, %8, %6\n s_xor_b32 %10, %10, %9\n v_and_b32 %7, %8, %7\n"
// 8 vector + 2 LDS reads (results waited for once per group)
#define V8_LDS  "v_and_b32 %0, %8, %0\n ds_read_u8 %11, %12\n v_and_b32 %1, %8, %1\n v_and_b32 %2, %8, %2\n v_and_b32 %3, %8, %3\n" \
                "v_and_b32 %4, %8, %4\n ds_read_u8 %13, %12 offset:64\n v_and_b32 %5, %8, %5\n v_and_b32 %6, %8, %6\n v_and_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n"

// one dependency chain: every instruction needs the one before it
#define V8_DEP  "v_and_b32 %0, %8, %0\n v_and_b32 %0, %8, %0\n v_and_b32 %0, %8, %0\n v_and_b32 %0, %8, %0\n" \
                "v_and_b32 %0, %8, %0\n v_and_b32 %0, %8, %0\n v_and_b32 %0, %8, %0\n v_and_b32 %0, %8, %0\n"
#define V8_DEP_ALIGN "v_alignbit_b32 %0, %1, %0, 2\n v_alignbit_b32 %0, %2, %0, 2\n v_alignbit_b32 %0, %3, %0, 2\n v_alignbit_b32 %0, %4, %0, 2\n" \
                "v_alignbit_b32 %0, %5, %0, 2\n v_alignbit_b32 %0, %6, %0, 2\n v_alignbit_b32 %0, %7, %0, 2\n v_alignbit_b32 %0, %1, %0, 2\n"
#define V8_DEP_DPP "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n" \
                "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n" \
                "v_add_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n" \
                "v_add_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n"
// two chains interleaved
#define V8_DEP2 "v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n" \
                "v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n v_and_b32 %0, %8, %0\n v_and_b32 %1, %8, %1\n"

template <int KIND, int GROUPS>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int iters, uint32_t seed) {
    extern __shared__ uint8_t lds[];
    uint32_t a0 = seed * threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    uint32_t s = seed | 0xFFFF0000u, t = 0x07020501u, sc = seed, l0 = 0, l1 = 0;
    uint32_t addr = (threadIdx.x * 97u) & 0x3FFFu;             // random-ish byte addresses, like the pair-table probes
    lds[threadIdx.x] = (uint8_t)seed;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) {
            if (KIND == 0)
                asm volatile(V8_AND : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));
            if (KIND == 1)
                asm volatile(V8_PERM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s), "v"(t));
            if (KIND == 2)
                asm volatile(V8_SALU : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s), "+s"(t), "+s"(sc) : : "scc");
            if (KIND == 4)
                asm volatile(V8_DEP : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));
            if (KIND == 5)
                asm volatile(V8_DEP_ALIGN : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));
            if (KIND == 6)
                asm volatile(V8_DEP_DPP : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));
            if (KIND == 7)
                asm volatile(V8_DEP2 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "s"(s));
            if (KIND == 3)
                asm volatile(V8_LDS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s), "+s"(t), "+s"(sc),
                             "+v"(l0), "+v"(addr), "+v"(l1) : : "memory");
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ sc ^ l0 ^ l1;
}

template <int KIND, int GROUPS>
void run(const char *name) {
    uint32_t *d;
    hipMalloc(&d, 512 * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    // waves per SIMD = blocks per CU x threads / 256; the LDS allocation decides how many blocks share a CU
    const struct { int grid, block, lds, waves; } occ[] = {{256, 256, 100 << 10, 1}, {256, 512, 100 << 10, 2}, {256, 768, 100 << 10, 3},
                                                            {256, 1024, 100 << 10, 4}, {512, 768, 60 << 10, 6}, {512, 1024, 60 << 10, 8}};
    const int iters = 80000 / GROUPS;
    hipFuncSetAttribute((const void *)k<KIND, GROUPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 << 10);
    printf("%-34s", name);
    fflush(stdout);
    for (auto &o : occ) {
        hipLaunchKernelGGL((k<KIND, GROUPS>), dim3(o.grid), dim3(o.block), o.lds, 0, d, 100, 3u);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, GROUPS>), dim3(o.grid), dim3(o.block), o.lds, 0, d, iters, 3u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double winstr = (double)o.waves * iters * GROUPS * 8;              // vector wave-instructions per SIMD
        printf("  %dw %.2f ns", o.waves, ms * 1e6 / winstr);
        fflush(stdout);
    }
    printf("\n");
    hipFree(d);
}

int main() {
    printf("ns per vector wave-instruction per SIMD (4 cycles at 2.4 GHz = 1.67 ns)\n");
    run<0, 1>("v_and (VOP2), 8 per iteration");
    run<0, 8>("v_and (VOP2), 64 per iteration");
    run<1, 1>("v_perm (VOP3), 8 per iteration");
    run<1, 8>("v_perm (VOP3), 64 per iteration");
    run<2, 8>("v_and + 1 scalar per 2, 64/it");
    run<3, 8>("v_and + 1 ds_read_u8 per 4, 64/it");
    run<4, 8>("v_and, ONE dependent chain");
    run<7, 8>("v_and, two dependent chains");
    run<5, 8>("v_alignbit, one dependent chain");
    run<6, 8>("DPP v_add row_shr, one chain");
    return 0;
}
