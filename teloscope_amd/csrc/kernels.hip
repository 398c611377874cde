// kernels.hip — gfx950 (MI355X / CDNA4) kernels of libteloscan.
//
// ts_scan_tiles: the telomeric-motif scan of Teloscope::scanSegment / analyzeWindow
// (reference src/teloscope.cpp:387-658) for uniform-length pattern sets, as one persistent
// kernel.  Integer/byte work bounded by HBM (1 B/base in, 32 B/window + 4 B/match out);
// no MFMA on purpose.
//
// Per tile (see ts_internal.h):
//   phase 1  every wavefront resolves 1008 positions per iteration: one coalesced 16 B/lane
//            load, SWAR ASCII -> 2-bit codes (v_perm / v_sad_u8 / v_dot4), rolling k-mer per
//            position, ONE conflict-free ds_read_b128 from a 16x-replicated bit table in LDS
//            giving {match, forward, canonical}; results are kept as bit planes in LDS.
//   phase 2  per-step-block partial sums by range popcounts over the planes (LDS atomics),
//            windows assembled from ceil(w/s)+1 block partials -> 8 x u32 per window.
//   emit     each tile reserves room for its matches with one atomic add on a global cursor
//            (issued early, consumed late) and every wave compacts its matches to packed
//            32-bit records; the per-tile directory {offset, count} makes the dense stream
//            addressable in position order.  No workgroup ever waits on another one.
// Every per-tile global round trip (tile ticket, tile descriptor, record offset) is issued
// at least one phase before its result is needed: on a CU that is streaming, a dependent
// global access costs microseconds.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ts_internal.h"

namespace {

typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Wave-wide inclusive prefix sum in 6 DPP adds (row_shr 1/2/4/8 inside each row of 16,
// then row_bcast:15 into rows 1,3 and row_bcast:31 into rows 2,3); no LDS traffic.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
    return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}

__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    v = dpp_add<0x111, 0xf>(v);
    v = dpp_add<0x112, 0xf>(v);
    v = dpp_add<0x114, 0xf>(v);
    v = dpp_add<0x118, 0xf>(v);
    v = dpp_add<0x142, 0xa>(v);
    v = dpp_add<0x143, 0xc>(v);
    return v;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_incl(v), 63);
}

// spread the 16 bits of v to the even bit positions of a dword
__device__ __forceinline__ uint32_t spread16(uint32_t x) {
    x = (x | (x << 8)) & 0x00FF00FFu;
    x = (x | (x << 4)) & 0x0F0F0F0Fu;
    x = (x | (x << 2)) & 0x33333333u;
    x = (x | (x << 1)) & 0x55555555u;
    return x;
}

// bits [lo, hi) of a 64-bit word, 0 <= lo < hi <= 64
__device__ __forceinline__ u64 mask64(uint32_t lo, uint32_t hi) {
    u64 m = (hi >= 64u) ? ~0ull : ((1ull << hi) - 1ull);
    return m & (~0ull << lo);
}

// clear bits [lo, hi) of a bit plane held as dwords (hi - lo <= 31)
__device__ __forceinline__ void plane_clear(uint32_t *p, uint32_t lo, uint32_t hi) {
    const uint32_t d0 = lo >> 5, d1 = (hi - 1u) >> 5;
    const uint32_t m0 = ~0u << (lo & 31u);
    const uint32_t m1 = ~0u >> (31u - ((hi - 1u) & 31u));
    if (d0 == d1) {
        atomicAnd(&p[d0], ~(m0 & m1));
    } else {
        atomicAnd(&p[d0], ~m0);
        atomicAnd(&p[d1], ~m1);
    }
}

struct Lds {
    uint32_t *codes;            // 2-bit codes, 16 bases per dword
    uint16_t *pV, *pM, *pF, *pC;   // validity / match / forward / canonical planes (16 positions per halfword)
    uint32_t *blk;              // per step-block accumulators
    uint32_t *misc;
};

__host__ __device__ inline uint32_t align16(uint32_t x) { return (x + 15u) & ~15u; }

__host__ __device__ inline uint32_t lds_layout(const TsScanParams &P, uint32_t off[8]) {
    uint32_t o = 0;
    off[0] = o; o += P.table_rows << P.row_shift;
    off[1] = o; o += align16(P.nch * 63u * 4u + 16u);
    const uint32_t pb = align16(P.nch * 63u * 2u + 16u);
    for (int i = 2; i <= 5; ++i) { off[i] = o; o += pb; }          // V, M, F, C
    off[6] = o; o += align16(P.max_blocks * TS_BLK_STRIDE * 4u);
    off[7] = o; o += 256u;
    return o;
}

// Diagnostic build only (-DTS_PHASE_TIMERS): per-phase cycle shares, summed over tiles, from
// s_memtime stamps taken by thread 0 (never in the product build; stamps perturb timing).
#ifdef TS_PHASE_TIMERS
#define TS_STAMP(i) do { if (tid == 0) { __builtin_amdgcn_sched_barrier(0); stamps[i] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define TS_STAMP(i) do { } while (0)
#endif

// misc[] slots
enum { MI_TILE = 0, MI_INVALID = 1, MI_OFF_LO = 2, MI_OFF_HI = 3, MI_TOT = 4 /* [8] */, MI_CAN = 12,
       MI_FWD = 20, MI_DESC = 32 /* TsTile of the next tile, 8 dwords */ };

// ---------------------------------------------------------------------------------------
// 2 workgroups per CU (LDS-limited) = 4 waves per SIMD: cap the allocation at 128 VGPRs
__global__ __launch_bounds__(TS_WG_THREADS, 4)
void ts_scan_tiles(const TsScanParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint32_t off[8];
    lds_layout(P, off);
    Lds L;
    L.codes = (uint32_t *)(lds_raw + off[1]);
    L.pV = (uint16_t *)(lds_raw + off[2]);
    L.pM = (uint16_t *)(lds_raw + off[3]);
    L.pF = (uint16_t *)(lds_raw + off[4]);
    L.pC = (uint16_t *)(lds_raw + off[5]);
    L.blk = (uint32_t *)(lds_raw + off[6]);
    L.misc = (uint32_t *)(lds_raw + off[7]);

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;

    // the replicated bit table stays in LDS for the lifetime of this (persistent) workgroup
    {
        const uint4 *src = (const uint4 *)P.table;
        uint4 *dst = (uint4 *)(lds_raw + off[0]);
        const uint32_t n16 = (P.table_rows << P.row_shift) >> 4;
        for (uint32_t i = tid; i < n16; i += TS_WG_THREADS) dst[i] = src[i];
    }

    const uint32_t k = P.k;
    const uint32_t rowbits = 2u * k - 5u;
    const uint32_t repoff = (lane & P.rep_mask) * 16u;
    // LDS byte address of the table (the dynamic-LDS base is where the table lives)
    const uint32_t tab_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)(lds_raw + off[0]);

    // Tile tickets and descriptors are fetched one tile ahead; misc[MI_TILE] / misc[MI_DESC]
    // always describe the tile the next iteration scans.
    if (tid == 0) L.misc[MI_TILE] = atomicAdd(P.ticket, 1u);
    __syncthreads();
    if (tid < 8u) {
        const uint32_t t0 = L.misc[MI_TILE];
        if (t0 < P.ntiles) L.misc[MI_DESC + tid] = ((const uint32_t *)&P.tiles[t0])[tid];
    }

    for (;;) {
        __syncthreads();                         // previous iteration fully retired (and table visible)
        const uint32_t tile = L.misc[MI_TILE];
        if (tile >= P.ntiles) break;
        TsTile T;
        {
            const uint32_t *d = &L.misc[MI_DESC];
            T.in_off = ((u64)d[1] << 32) | d[0];
            T.win_out = ((u64)d[3] << 32) | d[2];
            T.nrel = d[4]; T.nwin = d[5]; T.own_len = d[6]; T.seg = d[7];
        }
        if (tid == 0) L.misc[MI_INVALID] = 0;
        __syncthreads();                         // everyone has read the ticket before it is replaced
        uint32_t next_tile = 0;
        if (tid == 0) next_tile = atomicAdd(P.ticket, 1u);     // consumed after phase 1

#ifdef TS_PHASE_TIMERS
        u64 stamps[7];
#endif
        TS_STAMP(0);
        const uint32_t sh = (uint32_t)(T.in_off & 15ull);
        const unsigned char *src = P.in + (T.in_off - sh);
        const uint32_t nblk = T.nwin + P.q;                       // step blocks whose partials are needed
        const uint32_t span = nblk * P.s;
        const uint32_t count_lim = T.nrel < span ? T.nrel : span;  // u-range that is ever counted
        const uint32_t xend = sh + T.nrel;                         // plane coord of the segment end
        uint32_t need = sh + (T.nrel < span + 16u ? T.nrel : span + 16u);
        uint32_t nch = (need + 16u + TS_CHUNK - 1u) / TS_CHUNK;
        if (nch > P.nch) nch = P.nch;

        for (uint32_t i = tid; i < P.max_blocks * TS_BLK_STRIDE; i += TS_WG_THREADS) L.blk[i] = 0;

        // ------------------------------------------------------------------ phase 1
        {
            uint32_t c = wave;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c < nch) v = *(const uint4 *)(src + (size_t)c * TS_CHUNK + lane * 16u);
            while (c < nch) {
                const uint32_t cn = c + TS_WAVES;
                uint4 vn = make_uint4(0, 0, 0, 0);
                if (cn < nch) vn = *(const uint4 *)(src + (size_t)cn * TS_CHUNK + lane * 16u);

                // ASCII -> 2-bit codes (A0 C1 T2 G3) and a validity check, 4 bases per dword
                const uint32_t x[4] = {v.x, v.y, v.z, v.w};
                uint32_t t[4], e[4], sad = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    t[i] = (x[i] >> 1) & 0x07070707u;
                    e[i] = __builtin_amdgcn_perm(0xFFFFFFFFu, 0x47544341u, t[i]);
                    sad = __builtin_amdgcn_sad_u8(x[i] & P.fold_mask, e[i], sad);
                }
                uint32_t w2 = __builtin_amdgcn_udot4(t[3], 0x40100401u, 0u, false);
                w2 = __builtin_amdgcn_udot4(t[2], 0x40100401u, w2 << 8, false);
                w2 = __builtin_amdgcn_udot4(t[1], 0x40100401u, w2 << 8, false);
                w2 = __builtin_amdgcn_udot4(t[0], 0x40100401u, w2 << 8, false);

                const uint32_t pos0 = c * TS_CHUNK + lane * 16u;     // plane coord of this lane's first base
                uint32_t inv16 = 0;
                const bool slow = __any(sad != 0) || (c * TS_CHUNK + 1024u > xend);
                if (slow) {                                           // wave-uniform, rare
                    uint32_t b4[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t d = (x[i] & P.fold_mask) ^ e[i];
                        const uint32_t nz = ((((d & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | d) & 0x80808080u) >> 7;
                        b4[i] = __builtin_amdgcn_udot4(nz, 0x08040201u, 0u, false);
                        t[i] &= 0x03030303u;
                    }
                    inv16 = b4[0] | (b4[1] << 4) | (b4[2] << 8) | (b4[3] << 12);
                    if (pos0 + 16u > xend) {
                        const uint32_t nv = xend > pos0 ? xend - pos0 : 0u;
                        inv16 |= (0xFFFFu << nv) & 0xFFFFu;
                    }
                    w2 = __builtin_amdgcn_udot4(t[3], 0x40100401u, 0u, false);
                    w2 = __builtin_amdgcn_udot4(t[2], 0x40100401u, w2 << 8, false);
                    w2 = __builtin_amdgcn_udot4(t[1], 0x40100401u, w2 << 8, false);
                    w2 = __builtin_amdgcn_udot4(t[0], 0x40100401u, w2 << 8, false);
                    if (lane == 0) L.misc[MI_INVALID] = 1;
                }

                // next lane's 16 bases (DPP wave_shl:1, lane i <- lane i+1; lane 63's value is unused)
                const uint32_t nxt = (uint32_t)__builtin_amdgcn_mov_dpp((int)w2, 0x130, 0xf, 0xf, false);

                // one table probe per position: row = k-mer >> 5, bit = k-mer & 31.  All sixteen
                // ds_read_b128 are issued back to back (inline asm: hipcc would narrow them to b96 and
                // serialise them on a register-reuse wait), then consumed in two halves behind counted
                // lgkmcnt waits, so LDS latency is paid once per chunk instead of once per probe.
                uint32_t tmp[16];
                u32x4 ent[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    tmp[j] = (j == 0) ? w2 : __builtin_amdgcn_alignbit(nxt, w2, 2 * j);
                    const uint32_t row = __builtin_amdgcn_ubfe(tmp[j], 5, rowbits);
                    const uint32_t addr = tab_base + (row << P.row_shift) + repoff;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(ent[j]) : "v"(addr));
                }
                uint32_t aM = 0, aF = 0, aC = 0;
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                // every probe's full 16-byte destination stays allocated until its wait has passed
                // (the unused 4th dword must not be handed to another value while the read is in flight)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(ent[j]));
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    aM = __builtin_amdgcn_alignbit(ent[j].x >> (tmp[j] & 31u), aM, 1);
                    aF = __builtin_amdgcn_alignbit(ent[j].y >> (tmp[j] & 31u), aF, 1);
                    aC = __builtin_amdgcn_alignbit(ent[j].z >> (tmp[j] & 31u), aC, 1);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 8; j < 16; ++j) asm volatile("" ::"v"(ent[j]));
#pragma unroll
                for (int j = 8; j < 16; ++j) {
                    aM = __builtin_amdgcn_alignbit(ent[j].x >> (tmp[j] & 31u), aM, 1);
                    aF = __builtin_amdgcn_alignbit(ent[j].y >> (tmp[j] & 31u), aF, 1);
                    aC = __builtin_amdgcn_alignbit(ent[j].z >> (tmp[j] & 31u), aC, 1);
                }
                uint32_t M16 = aM >> 16, F16 = aF >> 16, C16 = aC >> 16;

                if (slow) {                                           // k-mers touching an invalid base
                    const uint32_t inv32 = inv16 | ((uint32_t)__builtin_amdgcn_mov_dpp((int)inv16, 0x130, 0xf, 0xf, false) << 16);
                    uint32_t kb = 0;
                    for (uint32_t i = 0; i < k; ++i) kb |= inv32 >> i;
                    M16 &= ~kb; F16 &= ~kb; C16 &= ~kb;
                }

                if (lane < 63u) {
                    const uint32_t h = c * 63u + lane;
                    L.codes[h] = w2;
                    L.pM[h] = (uint16_t)M16;
                    L.pF[h] = (uint16_t)F16;
                    L.pC[h] = (uint16_t)C16;
                    L.pV[h] = (uint16_t)(~inv16);
                }
                c = cn;
                v = vn;
            }
        }
        if (tid == 0) L.misc[MI_TILE] = next_tile;
        __syncthreads();
        TS_STAMP(1);
        // descriptor of the next tile: loaded now, parked in LDS after phase 2a
        uint32_t next_desc = 0;
        const uint32_t nt_id = L.misc[MI_TILE];
        if (tid < 8u && nt_id < P.ntiles) next_desc = ((const uint32_t *)&P.tiles[nt_id])[tid];

        // -------------------------------------------- w == s: matches may not straddle a window end
        if (P.straddle_fix && k > 1u) {
            for (uint32_t b = tid; b < nblk; b += TS_WG_THREADS) {
                const uint32_t hi = sh + (b + 1u) * P.s;
                if (hi <= nch * TS_CHUNK) {
                    const uint32_t lo = hi - (k - 1u);
                    plane_clear((uint32_t *)L.pM, lo, hi);
                    plane_clear((uint32_t *)L.pF, lo, hi);
                    plane_clear((uint32_t *)L.pC, lo, hi);
                }
            }
            __syncthreads();
        }
        TS_STAMP(2);

        // ------------------------------------------------- emit step A: owned match totals per wave
        const uint32_t own_end = sh + T.own_len;                  // plane coord
        const uint32_t h_hi = (own_end + 15u) >> 4;
        const uint32_t per = (((h_hi + TS_WAVES - 1u) / TS_WAVES) + 63u) & ~63u;
        const uint32_t hw0 = wave * per;
        const uint32_t hw1 = (hw0 + per < h_hi) ? hw0 + per : h_hi;
        {
            uint32_t cm = 0, cc = 0, cf = 0;
            for (uint32_t h = hw0 + lane; h < hw1; h += 64u) {
                const uint32_t hb = h << 4;
                const uint32_t lo = hb > sh ? 0u : sh - hb;
                const uint32_t hi = (hb + 16u <= own_end) ? 16u : own_end - hb;
                const uint32_t m = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
                const uint32_t M = L.pM[h] & m;
                cm += __popc(M);
                cc += __popc(M & L.pC[h]);
                cf += __popc(M & L.pF[h]);
            }
            cm = wave_sum(cm); cc = wave_sum(cc); cf = wave_sum(cf);
            if (lane == 0) { L.misc[MI_TOT + wave] = cm; L.misc[MI_CAN + wave] = cc; L.misc[MI_FWD + wave] = cf; }
        }
        __syncthreads();

        // ------------------- reserve room for this tile's records: one atomic add, consumed later
        uint32_t agg = 0;
#pragma unroll
        for (int i = 0; i < TS_WAVES; ++i) agg += L.misc[MI_TOT + i];
        u64 rec_off = 0;
        if (tid == 0) {
            rec_off = atomicAdd(P.cursor, (u64)agg);
            uint32_t tc = 0, tf = 0;
            for (int i = 0; i < TS_WAVES; ++i) { tc += L.misc[MI_CAN + i]; tf += L.misc[MI_FWD + i]; }
            *(uint4 *)&P.tile_stats[4u * tile] = make_uint4(agg, tc, tf, 0u);
        }
        TS_STAMP(3);

        // ------------------------------------------------------------------ phase 2a
        const bool has_invalid = L.misc[MI_INVALID] != 0;
        if (P.windows_on) {
            const uint32_t ngran = (sh + count_lim + 63u) >> 6;
            const u64 *gM = (const u64 *)L.pM, *gF = (const u64 *)L.pF, *gC = (const u64 *)L.pC;
            for (uint32_t g = tid; g < ngran; g += TS_WG_THREADS) {
                const uint32_t gx = g << 6;
                const uint32_t x_lo = gx > sh ? gx : sh;
                const uint32_t x_hi = (gx + 64u < sh + count_lim) ? gx + 64u : sh + count_lim;
                if (x_lo >= x_hi) continue;
                const uint32_t u0 = x_lo - sh, u1 = x_hi - sh;
                // block index / offset of u0 (u0 < 2^22: multiply-high by ceil(2^32/s), then fix up)
                uint32_t b0 = __umulhi(u0, P.s_inv);
                if (b0 * P.s > u0) --b0;
                uint32_t o0 = u0 - b0 * P.s;
                if (o0 >= P.s) { o0 -= P.s; ++b0; }

                {   // matches, split at offsets {0, hh+1} of every step block
                    const u64 M = gM[g], F = gF[g], C = gC[g];
                    if (M) {
                        uint32_t u = u0, b = b0, o = o0;
                        while (u < u1) {
                            const bool head = o <= P.hh;
                            uint32_t eu = head ? (u - o + P.hh + 1u) : (u - o + P.s);
                            if (eu > u1) eu = u1;
                            const u64 m = mask64(u + sh - gx, eu + sh - gx);
                            const uint32_t nm = __popcll(M & m);
                            if (nm) {
                                uint32_t *a = &L.blk[b * TS_BLK_STRIDE + (head ? 8u : 11u)];
                                const uint32_t nf = __popcll(F & m), nc = __popcll(C & m);
                                atomicAdd(&a[0], nm);
                                if (nc) atomicAdd(&a[1], nc);
                                if (nf) atomicAdd(&a[2], nf);
                            }
                            o += eu - u; u = eu;
                            if (o >= P.s) { o = 0; ++b; }
                        }
                    }
                }
                if (P.nuc_on) {   // nucleotides, split at offsets {0, r}
                    uint32_t u = u0, b = b0, o = o0;
                    while (u < u1) {
                        const bool head = o < P.r;
                        uint32_t eu = head ? (u - o + P.r) : (u - o + P.s);
                        if (eu > u1) eu = u1;
                        const uint32_t xa = u + sh, xb = eu + sh;
                        uint32_t n1 = 0, n2 = 0, n3 = 0, nt = 0;
                        if (xb - xa == 64u && !has_invalid) {           // whole granule, all bases valid
                            const uint4 cd4 = *(const uint4 *)&L.codes[g << 2];
                            const uint32_t cd[4] = {cd4.x, cd4.y, cd4.z, cd4.w};
#pragma unroll
                            for (int tq = 0; tq < 4; ++tq) {
                                const uint32_t l0 = cd[tq] & 0x55555555u, h0 = (cd[tq] >> 1) & 0x55555555u;
                                n3 += __popc(l0 & h0);
                                n1 += __popc(l0 & ~h0);
                                n2 += __popc(h0 & ~l0);
                            }
                            nt = 64u;
                        } else {
#pragma unroll
                            for (uint32_t tq = 0; tq < 4; ++tq) {
                                const uint32_t d = (g << 2) + tq, base = d << 4;
                                const uint32_t lo = xa > base ? xa : base;
                                const uint32_t hi = xb < base + 16u ? xb : base + 16u;
                                if (lo < hi) {
                                    const uint32_t nb2 = (hi - lo) * 2u;
                                    uint32_t sel = (nb2 >= 32u ? ~0u : ((1u << nb2) - 1u)) << ((lo - base) * 2u);
                                    sel &= 0x55555555u;
                                    if (has_invalid) sel &= spread16(L.pV[d]);
                                    const uint32_t cd = L.codes[d];
                                    const uint32_t l0 = cd & sel, h0 = (cd >> 1) & sel;
                                    n3 += __popc(l0 & h0);
                                    n1 += __popc(l0 & ~h0);
                                    n2 += __popc(h0 & ~l0);
                                    nt += __popc(sel);
                                }
                            }
                        }
                        uint32_t *a = &L.blk[b * TS_BLK_STRIDE + (head ? 0u : 4u)];
                        const uint32_t n0 = nt - n1 - n2 - n3;
                        if (n0) atomicAdd(&a[0], n0);
                        if (n1) atomicAdd(&a[1], n1);
                        if (n2) atomicAdd(&a[2], n2);
                        if (n3) atomicAdd(&a[3], n3);
                        o += eu - u; u = eu;
                        if (o >= P.s) { o = 0; ++b; }
                    }
                }
            }
        }
        TS_STAMP(4);
        if (tid < 8u) L.misc[MI_DESC + tid] = next_desc;
        if (tid == 0) {
            L.misc[MI_OFF_LO] = (uint32_t)rec_off;
            L.misc[MI_OFF_HI] = (uint32_t)(rec_off >> 32);
            P.tile_off[tile] = rec_off;
        }
        __syncthreads();

        // --------------------------------------------------------------- phase 2b: window records
        if (P.windows_on) {
            for (uint32_t i = tid; i < T.nwin; i += TS_WG_THREADS) {
                uint32_t n[4] = {0, 0, 0, 0}, m[3] = {0, 0, 0};
                for (uint32_t j = 0; j < P.q; ++j) {
                    const uint32_t *a = &L.blk[(i + j) * TS_BLK_STRIDE];
#pragma unroll
                    for (int cidx = 0; cidx < 4; ++cidx) n[cidx] += a[cidx] + a[4 + cidx];
                }
                if (P.r) {
                    const uint32_t *a = &L.blk[(i + P.q) * TS_BLK_STRIDE];
#pragma unroll
                    for (int cidx = 0; cidx < 4; ++cidx) n[cidx] += a[cidx];
                }
                for (uint32_t j = 0; j < P.qq; ++j) {
                    const uint32_t *a = &L.blk[(i + j) * TS_BLK_STRIDE];
#pragma unroll
                    for (int cidx = 0; cidx < 3; ++cidx) m[cidx] += a[8 + cidx] + a[11 + cidx];
                }
                {
                    const uint32_t *a = &L.blk[(i + P.qq) * TS_BLK_STRIDE];
#pragma unroll
                    for (int cidx = 0; cidx < 3; ++cidx) m[cidx] += a[8 + cidx];
                }
                uint4 *dst = (uint4 *)(P.windows_out + (T.win_out + i) * 8ull);
                dst[0] = make_uint4(n[0], n[1], n[3], n[2]);                // A C G T
                dst[1] = make_uint4(m[1] * k, (m[0] - m[1]) * k, m[2] * k, (m[0] - m[2]) * k);
            }
        }
        TS_STAMP(5);

        // ------------------------------------------------------- emit step B: packed records
        {
            u64 obase = ((u64)L.misc[MI_OFF_HI] << 32) | (u64)L.misc[MI_OFF_LO];
            for (uint32_t i = 0; i < wave; ++i) obase += L.misc[MI_TOT + i];
            for (uint32_t h0 = hw0; h0 < hw1; h0 += 64u) {
                const uint32_t h = h0 + lane;
                uint32_t M = 0, F = 0, C = 0;
                if (h < hw1) {
                    const uint32_t hb = h << 4;
                    const uint32_t lo = hb > sh ? 0u : sh - hb;
                    const uint32_t hi = (hb + 16u <= own_end) ? 16u : own_end - hb;
                    const uint32_t m = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
                    M = L.pM[h] & m; F = L.pF[h]; C = L.pC[h];
                }
                const uint32_t cnt = __popc(M);
                const uint32_t incl = wave_scan_incl(cnt);
                u64 o = obase + (u64)(incl - cnt);
                const uint32_t ubase = (h << 4) - sh;                 // tile-relative position of bit 0
                while (M) {
                    const uint32_t j = (uint32_t)__builtin_ctz(M);
                    M &= M - 1u;
                    const uint32_t rec = ((ubase + j) << 2) | (((F >> j) & 1u) << 1) | ((C >> j) & 1u);
                    if (o < P.match_cap) P.matches_out[o] = rec;
                    ++o;
                }
                obase += (u64)(uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
            }
        }
#ifdef TS_PHASE_TIMERS
        __syncthreads();
        TS_STAMP(6);
        if (tid == 0 && P.phase_cycles)
            for (int i = 0; i < 6; ++i) atomicAdd(&P.phase_cycles[i], stamps[i + 1] - stamps[i]);
#endif
    }
}

// Per-segment hit summary {windows, matches, canonical, forward}: the buffer ranks gather.
__global__ void ts_segment_summary(const uint32_t *tile_stats, const uint32_t *seg_first_tile,
                                   const uint64_t *seg_nwin, uint32_t nseg, u64 *out) {
    const uint32_t sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= nseg) return;
    const uint32_t t0 = seg_first_tile[sidx], t1 = seg_first_tile[sidx + 1];
    u64 nm = 0, nc = 0, nf = 0;
    for (uint32_t t = t0; t < t1; ++t) {
        nm += tile_stats[4u * t]; nc += tile_stats[4u * t + 1u]; nf += tile_stats[4u * t + 2u];
    }
    out[4ull * sidx + 0] = seg_nwin[sidx];
    out[4ull * sidx + 1] = nm;
    out[4ull * sidx + 2] = nc;
    out[4ull * sidx + 3] = nf;
}

}  // namespace

int ts_k_lds_bytes(const TsScanParams *p) {
    uint32_t off[8];
    return (int)lds_layout(*p, off);
}

int ts_k_prepare(uint32_t lds_bytes) {
    return (int)hipFuncSetAttribute((const void *)ts_scan_tiles,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
}

int ts_k_occupancy(uint32_t lds_bytes) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ts_scan_tiles, TS_WG_THREADS, lds_bytes) != hipSuccess)
        return 1;
    return nb < 1 ? 1 : nb;
}

int ts_k_launch_scan(const TsScanParams *p, uint32_t grid, uint32_t lds_bytes, void *stream) {
    hipLaunchKernelGGL(ts_scan_tiles, dim3(grid), dim3(TS_WG_THREADS), lds_bytes, (hipStream_t)stream, *p);
    return (int)hipGetLastError();
}

int ts_k_launch_summary(const uint32_t *tile_stats, const uint32_t *seg_first_tile, const uint64_t *seg_nwin,
                        uint32_t nseg, unsigned long long *out, void *stream) {
    if (nseg == 0) return 0;
    hipLaunchKernelGGL(ts_segment_summary, dim3((nseg + 255u) / 256u), dim3(256), 0, (hipStream_t)stream,
                       tile_stats, seg_first_tile, seg_nwin, nseg, out);
    return (int)hipGetLastError();
}
