// pack.cpp — the host half of the packed upload (round 3): bases leave the host as 2-bit codes, a quarter of the
// bytes of the ASCII they arrive as, plus a short list of the positions that are not A/C/G/T.
//
// The host entry points are bound by the PCIe link at 1 B/base (46 Gbases/s).  The staging threads
// that copy — and, for FASTA text, strip — the caller's bases into the pinned ring pack them on the way: four bases per
// byte (base i of a group at bits 2i..2i+1, codes A 0, C 1, T 2, G 3: ASCII & 6, halved — the scan kernel's own code),
// every byte that is not one of the four letters (after case folding, when the context folds case) recorded as part
// of an invalid RUN and packed as code 0.  unpack.hip turns a chunk back into the byte layout the kernels read
// (letters, 'N' over the invalid runs) at HBM speed, so no kernel changes and the result cannot depend on the route.
#include <immintrin.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "host.hpp"

namespace ts {

namespace {

inline void push_invalid(PackRuns &r, uint32_t pos) {
    if (r.open_len && r.open_start + r.open_len == pos) { ++r.open_len; return; }
    if (r.open_len) r.runs.push_back({r.open_start, r.open_len});
    r.open_start = pos;
    r.open_len = 1;
}

// four bases -> one byte; invalid ones noted
inline unsigned char pack4_scalar(const unsigned char *s, size_t n, uint32_t fold_and, uint32_t pos, PackRuns &r) {
    uint32_t out = 0;
    for (size_t i = 0; i < n; ++i) {
        const uint32_t c = s[i] & fold_and;
        const uint32_t code = (c >> 1) & 3u;
        if (c != ((0x47544341u >> (8u * code)) & 0xFFu)) push_invalid(r, pos + (uint32_t)i);      // 'A' 'C' 'T' 'G' by code
        else out |= code << (2u * i);
    }
    return (unsigned char)out;
}

__attribute__((target("avx2")))
size_t pack_avx2(const unsigned char *src, size_t n, unsigned char *dst, bool fold, uint32_t pos0, PackRuns &r) {
    const __m256i fold_mask = _mm256_set1_epi8(fold ? (char)0xDF : (char)0xFF);
    const __m256i three = _mm256_set1_epi8(3);
    const __m256i lut = _mm256_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                         'A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i w14 = _mm256_set1_epi16(0x0401);                 // bytes {1, 4}: c0 + 4 c1
    const __m256i w116 = _mm256_set1_epi32(0x00100001);            // words {1, 16}: (c0 + 4 c1) + 16 (c2 + 4 c3)
    const __m256i gather = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                            0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    size_t i = 0;
    // 128 bases per round: four independent chains, ONE 32-byte store (the four blocks' bytes, one per dword, narrowed by
    // two saturating packs and put back in order by a dword permute)
    const __m256i order = _mm256_setr_epi32(0, 4, 1, 5, 2, 6, 3, 7);
    // TS_PACK_MODE (experiment knob): bit 0 = prefetch the source 1 KB ahead, bit 1 = non-temporal stores when dst is aligned
    static const int mode = [] { const char *e = getenv("TS_PACK_MODE"); return e ? atoi(e) : 0; }();
    const bool nt_store = (mode & 2) && (((uintptr_t)dst & 31u) == 0u);
    for (; i + 128 <= n; i += 128) {
        if (mode & 1) { _mm_prefetch((const char *)src + i + 1024, _MM_HINT_NTA); _mm_prefetch((const char *)src + i + 1088, _MM_HINT_NTA); }
        __m256i t32[4];
        uint32_t bad[4];
#pragma GCC unroll 4
        for (int q = 0; q < 4; ++q) {
            const __m256i v = _mm256_and_si256(_mm256_loadu_si256((const __m256i *)(src + i + 32 * q)), fold_mask);
            __m256i code = _mm256_and_si256(_mm256_srli_epi16(v, 1), three);
            const __m256i ok = _mm256_cmpeq_epi8(v, _mm256_shuffle_epi8(lut, code));
            bad[q] = ~(uint32_t)_mm256_movemask_epi8(ok);
            code = _mm256_and_si256(code, ok);
            t32[q] = _mm256_madd_epi16(_mm256_maddubs_epi16(code, w14), w116);
        }
        const __m256i ab = _mm256_packus_epi32(t32[0], t32[1]), cd = _mm256_packus_epi32(t32[2], t32[3]);
        const __m256i out = _mm256_permutevar8x32_epi32(_mm256_packus_epi16(ab, cd), order);
        if (nt_store) _mm256_stream_si256((__m256i *)(dst + (i >> 2)), out);
        else _mm256_storeu_si256((__m256i *)(dst + (i >> 2)), out);
        if (bad[0] | bad[1] | bad[2] | bad[3]) {
            for (int q = 0; q < 4; ++q) {
                uint32_t b = bad[q];
                while (b) { push_invalid(r, pos0 + (uint32_t)i + 32u * (uint32_t)q + (uint32_t)__builtin_ctz(b)); b &= b - 1u; }
            }
        }
    }
    if (nt_store) _mm_sfence();
    for (; i + 32 <= n; i += 32) {
        const __m256i v = _mm256_and_si256(_mm256_loadu_si256((const __m256i *)(src + i)), fold_mask);
        __m256i code = _mm256_and_si256(_mm256_srli_epi16(v, 1), three);
        const __m256i ok = _mm256_cmpeq_epi8(v, _mm256_shuffle_epi8(lut, code));
        const uint32_t bad = ~(uint32_t)_mm256_movemask_epi8(ok);
        code = _mm256_and_si256(code, ok);
        const __m256i t32 = _mm256_madd_epi16(_mm256_maddubs_epi16(code, w14), w116);
        const __m256i p = _mm256_shuffle_epi8(t32, gather);
        const uint32_t lo = (uint32_t)_mm_cvtsi128_si32(_mm256_castsi256_si128(p));
        const uint32_t hi = (uint32_t)_mm_cvtsi128_si32(_mm256_extracti128_si256(p, 1));
        std::memcpy(dst + (i >> 2), &lo, 4);
        std::memcpy(dst + (i >> 2) + 4, &hi, 4);
        if (bad) {
            uint32_t b = bad;
            while (b) { push_invalid(r, pos0 + (uint32_t)i + (uint32_t)__builtin_ctz(b)); b &= b - 1u; }
        }
    }
    return i;
}

}  // namespace

// Packs the n bases at src (ASCII) into dst, four per byte, the first base at bits 0..1 of dst[0]; the last byte is
// padded with code 0 when n is not a multiple of 4.  pos0: the position (relative to the chunk) of src[0], for the runs.
void pack_bases(const unsigned char *src, size_t n, unsigned char *dst, bool fold, uint32_t pos0, PackRuns &r) {
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    size_t i = have_avx2 ? pack_avx2(src, n, dst, fold, pos0, r) : 0;
    const uint32_t fold_and = fold ? 0xDFu : 0xFFu;
    for (; i < n; i += 4) dst[i >> 2] = pack4_scalar(src + i, n - i < 4 ? n - i : 4, fold_and, pos0 + (uint32_t)i, r);
}

void PackRuns::finish() {
    if (open_len) runs.push_back({open_start, open_len});
    open_len = 0;
}

}  // namespace ts
