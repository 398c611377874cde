// shard.hip — gfx950 kernels that pack a shard's results into its message (ts_batch_pack_shard, shard.cpp).
//
// Several devices share one scan by owning consecutive tile ranges of its plan (the reference merges
// its thread-pool jobs' PathData in seqPos order, src/input.cpp:719-733, include/teloscope.h:262-266).  What a
// device hands over is what the reference's writers consume (src/teloscope.cpp:700-868), not the raw match stream:
//
//   window records, bit-packed     A, C, G, T and the canonical / non-canonical / forward match COUNTS of every owned
//                                  window, bit_width(window) bits each (covered bases = count x k; reverse = canonical
//                                  + non-canonical - forward): 9 bytes per window at w = 1000 instead of 32
//   visible match records          canonicalMatches and the nonCanonicalMatches of the terminal zone
//                                  (src/teloscope.cpp:486-496): ~3 % of the stream; 16 bits each (tile-relative
//                                  position < 2^14, forward, canonical) + a 16-bit count per tile
//   blocks                         terminal and interstitial blocks, called on this device (blockcall.hip)
//
// All of it is HBM-bound streaming over data the scan left resident; nothing here synchronises with the host.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ts_internal.h"

constexpr unsigned kSideWg = 64;        // one-wave workgroups: what can be placed beside the resident scan (see blockcall.hip)

namespace {

typedef unsigned long long u64;

// A record is visible iff it is canonical or lies in its segment's terminal zone (isTerminal, src/teloscope.cpp:451-459:
// absI <= terminalLimit || absI >= segmentSize - terminalLimit, the latter only evaluated as written for segments longer
// than the limit; shorter ones are terminal as a whole).
struct Zone { u64 lo_end, hi_begin; };        // terminal iff rel <= lo_end || rel >= hi_begin
__device__ __forceinline__ Zone zone_of(u64 seg_len, uint32_t terminal_limit) {
    Zone z;
    z.lo_end = terminal_limit;
    z.hi_begin = seg_len > terminal_limit ? seg_len - terminal_limit : 0ull;
    return z;
}

// segment of owned tile t: the shard's segment table is short (a few tens of entries), tiles of a segment are
// consecutive
__device__ __forceinline__ uint32_t seg_of_tile(const TsShardPackParams &P, uint32_t t) {
    const uint32_t base = P.segs[0].seg;
    return P.tiles[t].seg - base;
}

// Visible records per owned tile -> vis_stats[4 * i] (the stride the tile-offset scan of exchange.hip reads) and the
// message's u16 per-tile section.  A tile outside the terminal zone holds exactly its canonical records (the scan
// counted them: tile_stats[1]); one inside it holds all its records; only the few tiles the zone's edge cuts are read.
// One thread per tile.  (The records themselves are written by the interstitial pass, blockcall.hip, which reads the
// whole stream anyway.)
__global__ __launch_bounds__(kSideWg)
void ts_shard_visible_count(const TsShardPackParams P, uint32_t *vis_stats) {
    const uint32_t i = blockIdx.x * kSideWg + threadIdx.x;
    const uint32_t nown = P.own1 - P.own0;
    const uint32_t lane = threadIdx.x & 63u;
    bool edge = false;
    uint32_t vis = 0;
    if (i < nown) {
        const uint32_t t = P.own0 + i;
        const TsTile T = P.tiles[t];
        const uint4 st = *(const uint4 *)&P.tile_stats[4ull * t];
        const TsShardSegIn S = P.segs[seg_of_tile(P, t)];
        const Zone z = zone_of(S.len, P.terminal_limit);
        const u64 rel0 = T.in_off - S.in_off, rel1 = rel0 + T.own_len;    // the tile's records lie in [rel0, rel1)
        if (rel1 - 1 <= z.lo_end || rel0 >= z.hi_begin) vis = st.x;                           // wholly terminal
        else if (rel0 > z.lo_end && rel1 <= z.hi_begin) vis = st.y;                          // wholly interstitial
        else edge = st.x != 0u;
    }
    // the tiles the zone's edge cuts (two per long segment): the whole wave counts each of them
    for (u64 todo = __ballot(edge); todo; todo &= todo - 1ull) {
        const uint32_t l = (uint32_t)__builtin_ctzll(todo);
        const uint32_t t = P.own0 + (i - lane + l);
        const TsTile T = P.tiles[t];
        const TsShardSegIn S = P.segs[seg_of_tile(P, t)];
        const Zone z = zone_of(S.len, P.terminal_limit);
        const u64 rel0 = T.in_off - S.in_off;
        const uint32_t cnt = P.tile_stats[4ull * t];
        const u64 src = P.tile_off[t];
        uint32_t n = 0;
        for (uint32_t j = lane; j < cnt; j += 64u) {
            const uint32_t r = P.rec16 ? (uint32_t)((const uint16_t *)P.matches)[src + j] : P.matches[src + j];
            const u64 rel = rel0 + (r >> 2);
            n += ((r & 1u) || rel <= z.lo_end || rel >= z.hi_begin) ? 1u : 0u;
        }
        for (int o = 32; o >= 1; o >>= 1) n += (uint32_t)__shfl_xor((int)n, o);
        if (lane == l) vis = n;
    }
    if (i < nown) {
        vis_stats[4ull * i] = vis;
        ((uint16_t *)(P.msg + P.off_tilevis))[i] = (uint16_t)vis;
    }
}

// The visible records of the owned tiles as the scan left them (per-wave regions, TsTileChain words 2-3 say where) into the
// message's section in tile order, and the u16 count per tile.  A thread per tile — its directory entries are read
// coalesced, and a tile outside the terminal zones holds a dozen canonical records, which the thread copies itself, four
// loads in flight.  The few tiles with more (a telomere: every k-th base; a terminal zone: every match) lie next to each
// other — copied where they are found they were one wave's serial loop, 75 us whatever the batch size — and go onto a list
// {tile, source, destination, count} that ts_shard_copy_visible_big spreads over whole workgroups.
constexpr uint32_t kVisOwn = 32;                              // records a thread copies itself
struct BigCopy { u64 src, dst; uint32_t n, pad; };            // 24 bytes

__device__ __forceinline__ uint32_t vis_get(const TsShardPackParams &P, u64 at) {
    return P.vis_src_wide ? ((const uint32_t *)P.vis_src)[at] : ((const uint16_t *)P.vis_src)[at];
}
__device__ __forceinline__ void vis_put(const TsShardPackParams &P, uint32_t dst_bytes, u64 at, uint32_t r) {
    if (dst_bytes == 2u) ((uint16_t *)(P.msg + P.off_visible))[at] = (uint16_t)r;
    else ((uint32_t *)(P.msg + P.off_visible))[at] = r;
}

__global__ __launch_bounds__(kSideWg)
void ts_shard_copy_visible(const TsShardPackParams P, const u64 *vis_off, u64 capacity, uint32_t dst_bytes, BigCopy *big, uint32_t *n_big) {
    const uint32_t i = blockIdx.x * kSideWg + threadIdx.x;
    const uint32_t nown = P.own1 - P.own0;
    uint32_t n = 0;
    u64 src0 = 0, dst0 = 0;
    if (i < nown) {
        const uint32_t t = P.own0 + i;
        n = P.tile_stats[4ull * t + 3u];
        ((uint16_t *)(P.msg + P.off_tilevis))[i] = (uint16_t)n;
        const uint2 where = *(const uint2 *)&P.chain[4ull * t + 2u];
        src0 = ((u64)where.y << 32) | where.x;
        dst0 = vis_off[i];
        // A wave whose visible region overflowed in the scan (wave_fill[nwaves + w] > vis_cap; ts_overflow_flag reports it and the
        // caller rescans) did not store the records it promises here: src0 = wave * vis_cap + the wave's cursor, and a cursor past
        // the region's end would be read out of the next waves' regions — for the last waves, past the end of vis_src.
        if (P.vis_cap == 0u || (src0 % P.vis_cap) + n > P.vis_cap) n = 0;
    }
    if (vis_off[nown] > capacity) return;                     // does not fit: nothing is written, the header reports it
    if (n <= kVisOwn) {
        uint32_t j = 0;
        for (; j + 4u <= n; j += 4u) {
            const uint32_t a = vis_get(P, src0 + j), b = vis_get(P, src0 + j + 1u), c = vis_get(P, src0 + j + 2u), d = vis_get(P, src0 + j + 3u);
            vis_put(P, dst_bytes, dst0 + j, a); vis_put(P, dst_bytes, dst0 + j + 1u, b);
            vis_put(P, dst_bytes, dst0 + j + 2u, c); vis_put(P, dst_bytes, dst0 + j + 3u, d);
        }
        for (; j < n; ++j) vis_put(P, dst_bytes, dst0 + j, vis_get(P, src0 + j));
    } else {
        const uint32_t slot = atomicAdd(n_big, 1u);           // (the list holds an entry per owned tile: it cannot overflow)
        big[slot] = BigCopy{src0, dst0, n, 0u};
    }
}

__global__ __launch_bounds__(kSideWg)
void ts_shard_copy_visible_big(const TsShardPackParams P, uint32_t dst_bytes, const BigCopy *big, const uint32_t *n_big) {
    const uint32_t n = *n_big;
    for (uint32_t e = blockIdx.x; e < n; e += gridDim.x) {
        const BigCopy B = big[e];
        for (uint32_t j = threadIdx.x; j < B.n; j += kSideWg) vis_put(P, dst_bytes, B.dst + j, vis_get(P, B.src + j));
    }
}

// Window records of the owned windows, bit-packed: fields [A C G T] (when nucleotide counts are on), canonical,
// non-canonical and forward match counts, field_bits each, least significant first, in window_bytes bytes.
__global__ __launch_bounds__(kSideWg)
void ts_shard_pack_windows(const TsShardPackParams P, uint32_t window_bytes) {
    const u64 i = (u64)blockIdx.x * kSideWg + threadIdx.x;
    const u64 n = P.own_win1 - P.own_win0;
    if (i >= n) return;
    const uint32_t *r = P.windows + (P.own_win0 - P.win_lo + i) * 8ull;
    const uint4 nuc = *(const uint4 *)r, cov = *(const uint4 *)(r + 4);
    const uint32_t B = P.field_bits;
    u64 lo = 0, hi = 0;
    uint32_t at = 0;
    auto put = [&](uint32_t v) {
        lo |= at < 64u ? (u64)v << at : 0ull;
        if (at + B > 64u) hi |= at >= 64u ? (u64)v << (at - 64u) : (u64)v >> (64u - at);
        at += B;
    };
    if (P.nuc_on) { put(nuc.x); put(nuc.y); put(nuc.z); put(nuc.w); }
    put(cov.x / P.k); put(cov.y / P.k); put(cov.z / P.k);
    unsigned char *dst = P.msg + P.off_windows + i * window_bytes;
    for (uint32_t b = 0; b < window_bytes; ++b)
        dst[b] = (unsigned char)(b < 8u ? lo >> (8u * b) : hi >> (8u * (b - 8u)));
}

// A scan that overflowed a wave's record region (or, when it emits, its visible-record region) promises records it never
// stored: TS_SHARD_F_SCAN_OVERFLOW into the (zeroed) header's flags.  A thread per wave.
__global__ __launch_bounds__(kSideWg)
void ts_overflow_flag(const TsShardPackParams P) {
    const uint32_t w = blockIdx.x * kSideWg + threadIdx.x;
    bool o = false;
    if (w < P.nwaves) {
        o = P.wave_fill[w] > P.region_cap;
        if (P.chain) o |= P.wave_fill[P.nwaves + w] > P.vis_cap;
    }
    if (__ballot(o) != 0ull && (threadIdx.x & 63u) == 0u) atomicOr(&((TsShardHeader *)P.msg)->flags, TS_SHARD_F_SCAN_OVERFLOW);
}

// The header: what the host knows arrives by value, what the device found out is filled in here (n_blocks was
// counted in place by the block-calling kernels).
__global__ __launch_bounds__(64)
void ts_shard_header(const TsShardPackParams P, TsShardHeader H, const u64 *vis_off) {
    // (whether the scan overflowed a wave's region: checked over the waves by ts_overflow_flag, launched with the terminal walks,
    // which leaves TS_SHARD_F_SCAN_OVERFLOW in the zeroed header's flags — 5120 waves are an eighty-step loop for one wave here)
    const bool o = (((const TsShardHeader *)P.msg)->flags & TS_SHARD_F_SCAN_OVERFLOW) != 0u;
    const TsShardSeg *segs = (const TsShardSeg *)(P.msg + P.off_segs);
    bool ctx = false;
    for (uint32_t s = threadIdx.x; s < P.n_segs; s += 64u) ctx |= (segs[s].flags & TS_SEG_F_CONTEXT) != 0u;
    const bool any_o = o, any_ctx = __ballot(ctx) != 0ull;       // (one wavefront, no LDS: see exchange.hip)
    if (threadIdx.x != 0) return;
    TsShardHeader *dst = (TsShardHeader *)P.msg;
    uint32_t flags = (any_o ? TS_SHARD_F_SCAN_OVERFLOW : 0u) | (any_ctx ? TS_SHARD_F_CONTEXT : 0u);
    H.n_blocks = dst->n_blocks;
    H.n_visible = vis_off ? vis_off[P.own1 - P.own0] : 0ull;
    if (H.n_visible > H.visible_capacity) flags |= TS_SHARD_F_VISIBLE_OVERFLOW;
    if (H.n_blocks > H.block_capacity) flags |= TS_SHARD_F_BLOCK_OVERFLOW;
    H.flags = flags;
    *dst = H;
}

}  // namespace

// exchange.hip
int ts_k_launch_tile_offsets(const uint32_t *tile_stats, uint32_t ntiles, unsigned long long *tile_off, void *tmp, void *stream);
unsigned long long ts_k_scan_tmp_bytes(uint32_t ntiles);

// scratch of a pack: [vis_stats: 4 x u32 per owned tile][vis_off: u64 per owned tile + 1][the prefix sum's own scratch]
static unsigned long long shard_tmp_off(uint32_t own_tiles, int which) {
    const unsigned long long a = ((unsigned long long)own_tiles + 1ull) * 16ull;
    const unsigned long long b = a + ((unsigned long long)own_tiles + 2ull) * 8ull;
    return which == 0 ? 0ull : which == 1 ? a : ((b + 15ull) & ~15ull);
}
// ... [the list of tiles whose visible records a whole workgroup copies: a counter, then 24 bytes per owned tile]
static unsigned long long shard_big_off(uint32_t own_tiles) { return (shard_tmp_off(own_tiles, 2) + ts_k_scan_tmp_bytes(own_tiles) + 31ull) & ~15ull; }
unsigned long long ts_k_shard_tmp_bytes(uint32_t own_tiles) { return shard_big_off(own_tiles) + 16ull + 24ull * ((unsigned long long)own_tiles + 1ull); }

// Phase 1 (before block calling): visible records per owned tile and where each tile's go.  Fills `vis` for the
// interstitial pass (off == nullptr when the shard has no owned tile or no visible records: a tips-only batch).
int ts_k_launch_shard_count(const TsShardPackParams *P, const TsShardHeader *H, void *tmp, int with_visible, TsVisibleOut *vis, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const uint32_t nown = P->own1 - P->own0;
    *vis = TsVisibleOut{};
    if (!with_visible || !nown) return 0;
    uint32_t *vis_stats = (uint32_t *)((char *)tmp + shard_tmp_off(nown, 0));
    u64 *vis_off = (u64 *)((char *)tmp + shard_tmp_off(nown, 1));
    void *scan_tmp = (char *)tmp + shard_tmp_off(nown, 2);
    hipLaunchKernelGGL(ts_shard_visible_count, dim3((nown + kSideWg - 1u) / kSideWg), dim3(kSideWg), 0, st, *P, vis_stats);
    int e = ts_k_launch_tile_offsets(vis_stats, nown, vis_off, scan_tmp, stream);
    if (e) return e;
    vis->off = vis_off;
    vis->dst = P->msg + P->off_visible;
    vis->capacity = H->visible_capacity;
    vis->own0 = P->own0; vis->own1 = P->own1;
    vis->rec_bytes = H->visible_bytes;
    vis->terminal_limit = P->terminal_limit;
    return (int)hipGetLastError();
}

void *ts_k_shard_big_counter(void *tmp, uint32_t own_tiles) { return (char *)tmp + shard_big_off(own_tiles); }

int ts_k_launch_shard_visible(const TsShardPackParams *P, const TsShardHeader *H, void *tmp, int prezeroed, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const uint32_t nown = P->own1 - P->own0;
    if (!nown || !P->chain) return 0;
    u64 *vis_off = (u64 *)((char *)tmp + shard_tmp_off(nown, 1));
    void *scan_tmp = (char *)tmp + shard_tmp_off(nown, 2);
    // per-tile counts: word 3 of the owned tiles' tile_stats rows (the prefix sum reads every fourth dword)
    int e = ts_k_launch_tile_offsets(P->tile_stats + 4ull * P->own0 + 3u, nown, vis_off, scan_tmp, stream);
    if (e) return e;
    uint32_t *n_big = (uint32_t *)((char *)tmp + shard_big_off(nown));
    BigCopy *big = (BigCopy *)((char *)tmp + shard_big_off(nown) + 16);
    if (!prezeroed) {
        hipError_t he = hipMemsetAsync(n_big, 0, 4, st);
        if (he != hipSuccess) return (int)he;
    }
    hipLaunchKernelGGL(ts_shard_copy_visible, dim3((nown + kSideWg - 1u) / kSideWg), dim3(kSideWg), 0, st, *P, (const u64 *)vis_off,
                       (u64)H->visible_capacity, H->visible_bytes, big, n_big);
    hipLaunchKernelGGL(ts_shard_copy_visible_big, dim3(1024), dim3(kSideWg), 0, st, *P, H->visible_bytes, (const BigCopy *)big, (const uint32_t *)n_big);
    return (int)hipGetLastError();
}

// ---- do two streams run kernels at the same time?  HIP maps a process's streams onto a few hardware queues (four by default), and
// kernels of two streams that share one run one after the other whatever their events say.  A wave that waits ~1 ms on the constant
// 100 MHz clock goes to stream a, an empty kernel behind it to stream b: b's kernel is done while a's still waits exactly when the
// two streams sit on different queues.
__global__ __launch_bounds__(64)
void ts_probe_wait(unsigned long long ticks, unsigned long long *sink) {
    const unsigned long long t0 = wall_clock64();
    unsigned long long t = t0;
    while (t - t0 < ticks) { __builtin_amdgcn_s_sleep(64); t = wall_clock64(); }
    if (sink && threadIdx.x == 0) *sink = t - t0;
}
__global__ __launch_bounds__(64)
void ts_probe_empty() {}

int ts_k_streams_concurrent(void *a, void *b, int *concurrent) {
    hipEvent_t ea = nullptr, eb = nullptr;
    int rc = 0;
    *concurrent = 0;
    if (hipEventCreateWithFlags(&ea, hipEventDisableTiming) != hipSuccess) return 1;
    if (hipEventCreateWithFlags(&eb, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(ea); return 1; }
    if (hipStreamSynchronize((hipStream_t)a) != hipSuccess || hipStreamSynchronize((hipStream_t)b) != hipSuccess) rc = 1;
    if (!rc) {
        hipLaunchKernelGGL(ts_probe_wait, dim3(1), dim3(64), 0, (hipStream_t)a, 100000ull, (unsigned long long *)nullptr);   // 1 ms
        if (hipEventRecord(ea, (hipStream_t)a) != hipSuccess) rc = 1;
        hipLaunchKernelGGL(ts_probe_empty, dim3(1), dim3(64), 0, (hipStream_t)b);
        if (hipEventRecord(eb, (hipStream_t)b) != hipSuccess) rc = 1;
        if (hipGetLastError() != hipSuccess) rc = 1;
    }
    if (!rc) {
        if (hipEventSynchronize(eb) != hipSuccess) rc = 1;
        else {
            const hipError_t q = hipEventQuery(ea);
            if (q == hipErrorNotReady) *concurrent = 1;
            else if (q != hipSuccess) rc = 1;
        }
        if (hipEventSynchronize(ea) != hipSuccess) rc = 1;
    }
    (void)hipGetLastError();
    (void)hipEventDestroy(ea);
    (void)hipEventDestroy(eb);
    return rc;
}

// After the header was zeroed, any time before ts_k_launch_shard_pack: the scan-overflow flag.
int ts_k_launch_shard_overflow(const TsShardPackParams *P, void *stream) {
    if (P->nwaves)
        hipLaunchKernelGGL(ts_overflow_flag, dim3((P->nwaves + kSideWg - 1u) / kSideWg), dim3(kSideWg), 0, (hipStream_t)stream, *P);
    return (int)hipGetLastError();
}

// The packed window records (independent of block calling).
int ts_k_launch_shard_windows(const TsShardPackParams *P, const TsShardHeader *H, void *stream) {
    const u64 nwin = P->own_win1 - P->own_win0;
    if (nwin)
        hipLaunchKernelGGL(ts_shard_pack_windows, dim3((unsigned)((nwin + kSideWg - 1ull) / kSideWg)), dim3(kSideWg), 0, (hipStream_t)stream, *P, H->window_bytes);
    return (int)hipGetLastError();
}

// Last (after block calling, which counted the blocks in the zeroed header and wrote the visible records): the header.
int ts_k_launch_shard_pack(const TsShardPackParams *P, const TsShardHeader *H, void *tmp, int with_visible, void *stream) {
    const uint32_t nown = P->own1 - P->own0;
    u64 *vis_off = (u64 *)((char *)tmp + shard_tmp_off(nown, 1));
    hipLaunchKernelGGL(ts_shard_header, dim3(1), dim3(64), 0, (hipStream_t)stream, *P, *H, (const u64 *)((with_visible && nown) ? vis_off : nullptr));
    return (int)hipGetLastError();
}
