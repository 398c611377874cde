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

namespace {

// appends `nbits` (<= 64) low bits of `bits` to a little-endian bit stream: acc holds `fill` (< 64) pending bits
struct BitOut {
    unsigned char *dst;
    uint64_t acc = 0;
    uint32_t fill = 0;
    inline void put(uint64_t bits, uint32_t nbits) {
        if (!nbits) return;
        acc |= bits << fill;
        if (fill + nbits >= 64u) {
            std::memcpy(dst, &acc, 8);
            dst += 8;
            acc = fill ? bits >> (64u - fill) : 0ull;
            fill = fill + nbits - 64u;
        } else fill += nbits;
    }
    inline void flush() {                                             // the pending bits, whole bytes, zero padded
        for (uint32_t b = 0; b < fill; b += 8) *dst++ = (unsigned char)(acc >> b);
        acc = 0; fill = 0;
    }
};

__attribute__((target("avx2,bmi2")))
size_t pack_text_avx2(const char **cursor, const char *end, size_t n, BitOut &out, bool fold, uint32_t pos0, PackRuns &r) {
    const __m256i fold_mask = _mm256_set1_epi8(fold ? (char)0xDF : (char)0xFF);
    const __m256i three = _mm256_set1_epi8(3);
    const __m256i lut = _mm256_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                         'A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m256i w14 = _mm256_set1_epi16(0x0401), w116 = _mm256_set1_epi32(0x00100001);
    const __m256i gather = _mm256_setr_epi8(0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1,
                                            0, 4, 8, 12, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1);
    const __m256i lf = _mm256_set1_epi8('\n'), crv = _mm256_set1_epi8('\r');
    const char *p = *cursor;
    size_t done = 0;
    // 32 text bytes per round while 32 more bases are wanted and byte p[32] is readable (it decides whether a carriage return in
    // the chunk's last byte belongs to a line end)
    while (n - done >= 32 && end - p >= 33) {
        const __m256i raw = _mm256_loadu_si256((const __m256i *)p);
        const uint32_t mlf = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(raw, lf));
        const uint32_t mcr = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(raw, crv));
        const uint32_t ends = mlf | (mcr & ((mlf >> 1) | (p[32] == '\n' ? 0x80000000u : 0u)));      // bytes that are (part of) a line end
        const __m256i v = _mm256_and_si256(raw, fold_mask);
        __m256i code = _mm256_and_si256(_mm256_srli_epi16(v, 1), three);
        const __m256i ok = _mm256_cmpeq_epi8(v, _mm256_shuffle_epi8(lut, code));
        uint32_t bad = ~(uint32_t)_mm256_movemask_epi8(ok);
        code = _mm256_and_si256(code, ok);
        const __m256i t32 = _mm256_madd_epi16(_mm256_maddubs_epi16(code, w14), w116);
        const __m256i q = _mm256_shuffle_epi8(t32, gather);
        uint64_t bits = (uint64_t)(uint32_t)_mm_cvtsi128_si32(_mm256_castsi256_si128(q)) |
                        ((uint64_t)(uint32_t)_mm_cvtsi128_si32(_mm256_extracti128_si256(q, 1)) << 32);
        uint32_t nb = 32u;
        if (ends) {                                                  // take the line ends' slots out of the codes and of the invalid mask
            const uint32_t keep = ~ends;
            const uint64_t keep2 = _pdep_u64((uint64_t)keep, 0x5555555555555555ull) * 3ull;
            bits = _pext_u64(bits, keep2);
            bad = (uint32_t)_pext_u32(bad, keep);
            nb = (uint32_t)__builtin_popcount(keep);
        }
        out.put(bits, 2u * nb);
        while (bad) { push_invalid(r, pos0 + (uint32_t)done + (uint32_t)__builtin_ctz(bad)); bad &= bad - 1u; }
        done += nb;
        p += 32;
    }
    *cursor = p;
    return done;
}

}  // namespace

size_t pack_text(const char **cursor, const char *end, size_t n, unsigned char *dst, bool fold, uint32_t pos0, PackRuns &r) {
    static const bool fast = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("bmi2");
    BitOut out;
    out.dst = dst;
    size_t done = fast ? pack_text_avx2(cursor, end, n, out, fold, pos0, r) : 0;
    // the last bytes (and everything, without AVX2 / BMI2): a byte at a time
    const uint32_t fold_and = fold ? 0xDFu : 0xFFu;
    const char *p = *cursor;
    while (done < n && p < end) {
        const unsigned char raw = (unsigned char)*p;
        if (raw == '\n') { ++p; continue; }
        if (raw == '\r' && (p + 1 == end || p[1] == '\n')) { ++p; continue; }
        const uint32_t c = raw & fold_and, code = (c >> 1) & 3u;
        if (c != ((0x47544341u >> (8u * code)) & 0xFFu)) { push_invalid(r, pos0 + (uint32_t)done); out.put(0ull, 2u); }
        else out.put((uint64_t)code, 2u);
        ++done; ++p;
    }
    // (line ends right behind the last base stay for the next call: it skips them)
    out.flush();
    *cursor = p;
    return done;
}

void PackRuns::finish() {
    if (open_len) runs.push_back({open_start, open_len});
    open_len = 0;
}

}  // namespace ts
