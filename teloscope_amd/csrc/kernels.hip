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
//   emit     tile match totals are chained through a decoupled look-back (one 8-byte
//            status|value word per tile, agent-scope relaxed atomics), then every wave
//            compacts its matches to packed 32-bit records at their exact global offsets,
//            so the match stream is dense and position-ordered without a second pass.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ts_internal.h"

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ __forceinline__ u64 wave_sum64(u64 v) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// inclusive prefix sum across the 64 lanes
__device__ __forceinline__ uint32_t wave_scan_incl(uint32_t v) {
    const uint32_t l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(v, o);
        if (l >= (uint32_t)o) v += t;
    }
    return v;
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
    uint32_t *table;     // rows * 64 dwords
    uint32_t *codes;     // 2-bit codes, 16 bases per dword
    uint16_t *pM, *pF, *pC, *pV;   // bit planes, 16 positions per halfword
    uint32_t *blk;       // per step-block accumulators
    uint32_t *misc;
};

__host__ __device__ inline uint32_t align16(uint32_t x) { return (x + 15u) & ~15u; }

__host__ __device__ inline uint32_t lds_layout(const TsScanParams &P, uint32_t off[8]) {
    uint32_t o = 0;
    off[0] = o; o += P.table_rows * P.row_stride;
    off[1] = o; o += align16(P.nch * 63u * 4u + 16u);
    const uint32_t pb = align16(P.nch * 63u * 2u + 16u);
    off[2] = o; o += pb;
    off[3] = o; o += pb;
    off[4] = o; o += pb;
    off[5] = o; o += pb;
    off[6] = o; o += align16(P.max_blocks * TS_BLK_STRIDE * 4u);
    off[7] = o; o += 256u;
    return o;
}

// misc[] slots
enum { MI_TILE = 0, MI_INVALID = 1, MI_EXCL_LO = 2, MI_EXCL_HI = 3, MI_TOT = 4, MI_CAN = 12, MI_FWD = 20 };

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(TS_WG_THREADS)
void ts_scan_tiles(const TsScanParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint32_t off[8];
    lds_layout(P, off);
    Lds L;
    L.table = (uint32_t *)(lds_raw + off[0]);
    L.codes = (uint32_t *)(lds_raw + off[1]);
    L.pM = (uint16_t *)(lds_raw + off[2]);
    L.pF = (uint16_t *)(lds_raw + off[3]);
    L.pC = (uint16_t *)(lds_raw + off[4]);
    L.pV = (uint16_t *)(lds_raw + off[5]);
    L.blk = (uint32_t *)(lds_raw + off[6]);
    L.misc = (uint32_t *)(lds_raw + off[7]);

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t wave = tid >> 6;

    // the replicated bit table stays in LDS for the lifetime of this (persistent) workgroup
    {
        const uint4 *src = (const uint4 *)P.table;
        uint4 *dst = (uint4 *)L.table;
        const uint32_t n16 = P.table_rows * (P.row_stride >> 4);
        for (uint32_t i = tid; i < n16; i += TS_WG_THREADS) dst[i] = src[i];
    }

    const uint32_t k = P.k;
    const uint32_t rowbits = 2u * k - 5u;
    const uint32_t repoff = (lane & P.rep_mask) * 16u;
    const unsigned char *tab_bytes = (const unsigned char *)L.table;

    for (;;) {
        __syncthreads();                         // previous tile fully retired (and table visible)
        if (tid == 0) L.misc[MI_TILE] = atomicAdd(P.ticket, 1u);
        __syncthreads();
        const uint32_t tile = L.misc[MI_TILE];
        if (tile >= P.ntiles) break;

        const TsTile T = P.tiles[tile];
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
        if (tid == 0) L.misc[MI_INVALID] = 0;

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

                const uint32_t nxt = __shfl_down(w2, 1);             // next 16 bases (lane 63: unused)

                // one table probe per position: row = k-mer >> 5, bit = k-mer & 31
                uint32_t aM = 0, aF = 0, aC = 0;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const uint32_t tmp = (j == 0) ? w2 : __builtin_amdgcn_alignbit(nxt, w2, 2 * j);
                    const uint32_t row = __builtin_amdgcn_ubfe(tmp, 5, rowbits);
                    const uint4 ent = *(const uint4 *)(tab_bytes + row * P.row_stride + repoff);
                    asm volatile("" ::"v"(ent.w));   // keep all 16 bytes live: ds_read_b128 (4 LDS cycles), not b96 (8)
                    aM = __builtin_amdgcn_alignbit(ent.x >> (tmp & 31u), aM, 1);
                    aF = __builtin_amdgcn_alignbit(ent.y >> (tmp & 31u), aF, 1);
                    aC = __builtin_amdgcn_alignbit(ent.z >> (tmp & 31u), aC, 1);
                }
                uint32_t M16 = aM >> 16, F16 = aF >> 16, C16 = aC >> 16;

                if (slow) {                                           // k-mers touching an invalid base
                    const uint32_t inv32 = inv16 | (__shfl_down(inv16, 1) << 16);
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
        __syncthreads();

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

                {   // matches, split at offsets {0, hh+1} of every step block
                    const u64 M = gM[g], F = gF[g], C = gC[g];
                    uint32_t u = u0, b = u / P.s, o = u - b * P.s;
                    while (u < u1) {
                        const bool head = o <= P.hh;
                        uint32_t eu = head ? (u - o + P.hh + 1u) : (u - o + P.s);
                        if (eu > u1) eu = u1;
                        const u64 m = mask64(u + sh - gx, eu + sh - gx);
                        const uint32_t nm = __popcll(M & m), nf = __popcll(F & m), nc = __popcll(C & m);
                        uint32_t *a = &L.blk[b * TS_BLK_STRIDE + (head ? 8u : 11u)];
                        if (nm) atomicAdd(&a[0], nm);
                        if (nc) atomicAdd(&a[1], nc);
                        if (nf) atomicAdd(&a[2], nf);
                        o += eu - u; u = eu;
                        if (o >= P.s) { o = 0; ++b; }
                    }
                }
                if (P.nuc_on) {   // nucleotides, split at offsets {0, r}
                    uint32_t u = u0, b = u / P.s, o = u - b * P.s;
                    while (u < u1) {
                        const bool head = o < P.r;
                        uint32_t eu = head ? (u - o + P.r) : (u - o + P.s);
                        if (eu > u1) eu = u1;
                        const uint32_t xa = u + sh, xb = eu + sh;
                        uint32_t n1 = 0, n2 = 0, n3 = 0, nt = 0;
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

        // --------------------------------------------------- publish aggregate, assemble windows
        uint32_t agg = 0;
#pragma unroll
        for (int i = 0; i < TS_WAVES; ++i) agg += L.misc[MI_TOT + i];
        if (tid == 0) {
            const u64 word = ((tile == 0 ? TS_ST_INCL : TS_ST_AGG) << 62) | (u64)agg;
            __hip_atomic_store(&P.state[tile], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t tc = 0, tf = 0;
            for (int i = 0; i < TS_WAVES; ++i) { tc += L.misc[MI_CAN + i]; tf += L.misc[MI_FWD + i]; }
            P.tile_stats[2u * tile] = tc;
            P.tile_stats[2u * tile + 1u] = tf;
        }

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

        // ------------------------------------------------------- decoupled look-back (wave 0)
        if (wave == 0) {
            u64 excl = 0;
            if (tile > 0) {
                int64_t base = (int64_t)tile - 1;
                uint32_t spins = 0;
                for (;;) {
                    const int64_t idx = base - (int64_t)lane;
                    u64 wv = (TS_ST_INCL << 62);
                    if (idx >= 0)
                        wv = __hip_atomic_load(&P.state[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t st = (uint32_t)(wv >> 62);
                    const u64 inval = __ballot(st == TS_ST_INVALID);
                    const u64 incl = __ballot(st == TS_ST_INCL);
                    const uint32_t first_incl = incl ? (uint32_t)__builtin_ctzll(incl) : 64u;
                    const uint32_t first_inv = inval ? (uint32_t)__builtin_ctzll(inval) : 64u;
                    const uint32_t last = first_incl < 63u ? first_incl : 63u;   // lanes [0,last] are summed
                    if (first_inv <= last) {                                      // a needed predecessor is not ready
                        if (++spins > (1u << 24)) { if (lane == 0) atomicExch(P.error_flag, 1u); break; }
                        __builtin_amdgcn_s_sleep(2);
                        continue;
                    }
                    const u64 val = (lane <= last) ? (wv & ((1ull << 62) - 1ull)) : 0ull;
                    excl += wave_sum64(val);
                    if (first_incl < 64u) break;
                    base -= 64;
                }
                if (lane == 0) {
                    const u64 word = (TS_ST_INCL << 62) | (excl + (u64)agg);
                    __hip_atomic_store(&P.state[tile], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (lane == 0) {
                L.misc[MI_EXCL_LO] = (uint32_t)excl;
                L.misc[MI_EXCL_HI] = (uint32_t)(excl >> 32);
                P.tile_prefix[tile] = excl;
            }
        }
        __syncthreads();

        // ------------------------------------------------------- emit step B: packed records
        {
            u64 obase = ((u64)L.misc[MI_EXCL_HI] << 32) | (u64)L.misc[MI_EXCL_LO];
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
                obase += (u64)__shfl(incl, 63);
            }
        }
    }
}

// Per-segment hit summary {windows, matches, canonical, forward}: the buffer ranks gather.
__global__ void ts_segment_summary(const TsTile *tiles, const u64 *tile_prefix, const u64 *state,
                                   const uint32_t *tile_stats, const uint32_t *seg_first_tile,
                                   const uint64_t *seg_nwin, uint32_t nseg, uint32_t ntiles, u64 *out) {
    const uint32_t sidx = blockIdx.x * blockDim.x + threadIdx.x;
    if (sidx >= nseg) return;
    const uint32_t t0 = seg_first_tile[sidx], t1 = seg_first_tile[sidx + 1];
    u64 nm = 0, nc = 0, nf = 0;
    if (t1 > t0) {
        const u64 endp = (t1 < ntiles) ? tile_prefix[t1] : (state[ntiles - 1] & ((1ull << 62) - 1ull));
        nm = endp - tile_prefix[t0];
        for (uint32_t t = t0; t < t1; ++t) { nc += tile_stats[2u * t]; nf += tile_stats[2u * t + 1u]; }
    }
    out[4ull * sidx + 0] = seg_nwin[sidx];
    out[4ull * sidx + 1] = nm;
    out[4ull * sidx + 2] = nc;
    out[4ull * sidx + 3] = nf;
    (void)tiles;
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

int ts_k_launch_summary(const TsTile *tiles, const unsigned long long *tile_prefix,
                        const unsigned long long *state, const uint32_t *tile_stats,
                        const uint32_t *seg_first_tile, const uint64_t *seg_nwin,
                        uint32_t nseg, uint32_t ntiles, unsigned long long *out, void *stream) {
    if (nseg == 0) return 0;
    hipLaunchKernelGGL(ts_segment_summary, dim3((nseg + 255u) / 256u), dim3(256), 0, (hipStream_t)stream,
                       tiles, tile_prefix, state, tile_stats, seg_first_tile, seg_nwin, nseg, ntiles, out);
    return (int)hipGetLastError();
}
