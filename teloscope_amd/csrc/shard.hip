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
__global__ __launch_bounds__(256)
void ts_shard_visible_count(const TsShardPackParams P, uint32_t *vis_stats) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
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
        const uint32_t *src = P.matches + P.tile_off[t];
        uint32_t n = 0;
        for (uint32_t j = lane; j < cnt; j += 64u) {
            const uint32_t r = src[j];
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

// Window records of the owned windows, bit-packed: fields [A C G T] (when nucleotide counts are on), canonical,
// non-canonical and forward match counts, field_bits each, least significant first, in window_bytes bytes.
__global__ __launch_bounds__(256)
void ts_shard_pack_windows(const TsShardPackParams P, uint32_t window_bytes) {
    const u64 i = (u64)blockIdx.x * 256u + threadIdx.x;
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

// The header: what the host knows arrives by value, what the device found out is filled in here (n_blocks was
// counted in place by the block-calling kernels).
__global__ __launch_bounds__(64)
void ts_shard_header(const TsShardPackParams P, TsShardHeader H, const u64 *vis_off) {
    bool o = false;
    for (uint32_t w = threadIdx.x; w < P.nwaves; w += 64u) o |= P.wave_fill[w] > P.region_cap;
    const TsShardSeg *segs = (const TsShardSeg *)(P.msg + P.off_segs);
    bool ctx = false;
    for (uint32_t s = threadIdx.x; s < P.n_segs; s += 64u) ctx |= (segs[s].flags & TS_SEG_F_CONTEXT) != 0u;
    const bool any_o = __ballot(o) != 0ull, any_ctx = __ballot(ctx) != 0ull;       // (one wavefront, no LDS: see exchange.hip)
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
unsigned long long ts_k_shard_tmp_bytes(uint32_t own_tiles) { return shard_tmp_off(own_tiles, 2) + ts_k_scan_tmp_bytes(own_tiles) + 16ull; }

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
    hipLaunchKernelGGL(ts_shard_visible_count, dim3((nown + 255u) / 256u), dim3(256), 0, st, *P, vis_stats);
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

// The packed window records (independent of block calling).
int ts_k_launch_shard_windows(const TsShardPackParams *P, const TsShardHeader *H, void *stream) {
    const u64 nwin = P->own_win1 - P->own_win0;
    if (nwin)
        hipLaunchKernelGGL(ts_shard_pack_windows, dim3((unsigned)((nwin + 255ull) / 256ull)), dim3(256), 0, (hipStream_t)stream, *P, H->window_bytes);
    return (int)hipGetLastError();
}

// Last (after block calling, which counted the blocks in the zeroed header and wrote the visible records): the header.
int ts_k_launch_shard_pack(const TsShardPackParams *P, const TsShardHeader *H, void *tmp, int with_visible, void *stream) {
    const uint32_t nown = P->own1 - P->own0;
    u64 *vis_off = (u64 *)((char *)tmp + shard_tmp_off(nown, 1));
    hipLaunchKernelGGL(ts_shard_header, dim3(1), dim3(64), 0, (hipStream_t)stream, *P, *H, (const u64 *)((with_visible && nown) ? vis_off : nullptr));
    return (int)hipGetLastError();
}
