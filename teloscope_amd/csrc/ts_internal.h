// ts_internal.h — structures shared by the host side of libteloscan and its HIP kernels.
#ifndef TS_INTERNAL_H
#define TS_INTERNAL_H

#include <stdint.h>

// Geometry of the tiled uniform-k scan kernel (kernels.hip: ts_scan_tiles).
//
// A segment (or tips-only region) is cut into tiles; ONE WAVEFRONT scans one tile.  A tile OWNS
// `nwin` consecutive windows (window k covers [k*s, k*s+w)) = the `nwin*s` bases where those
// windows start, and additionally reads a halo so that every owned window and every k-mer
// that starts in an owned base is complete.  Positions inside a tile are "plane coordinates":
// byte offset from the 16-byte-aligned address at or below the tile's first owned base.

#define TS_MAX_WG_THREADS 1024         // up to 16 wavefronts per workgroup, one workgroup per CU
#ifndef TS_LIST
#define TS_LIST         512            // entries of a wave's match queue (power of two >= 64; a denser chunk is appended in lane groups)
#endif
#define TS_CHUNK        2016           // positions a wave resolves per iteration (63 lanes x 32)
#define TS_BLK_COUNTERS 10             // 4 NB dwords of window nucleotide counts, then per step block: match head[3] rest[3]
#define TS_TICKET_STRIDE 64            // dwords between two ticket counters (each in its own 256 bytes)
#define TS_MAX_TICKET_GROUPS 64
#define TS_IN_PAD       4096           // bytes of zero padding after the last segment in the input buffer

struct TsTile {                 // 32 bytes
    uint64_t in_off;            // byte offset (input buffer) of the tile's first owned base
    uint64_t win_out;           // index of the first owned window record
    uint32_t nrel;              // bases from the first owned base to the end of the segment/region (clamped)
    uint32_t nwin;              // owned windows (tips mode: 1 pseudo block, no record written)
    uint32_t own_len;           // owned bases = min(nrel, nwin*s)
    uint32_t seg;               // segment index (host bookkeeping / summary kernel)
};

struct TsScanParams {
    const uint8_t  *in;
    const TsTile   *tiles;
    const uint32_t *table;      // pair table (table_rows dwords) followed by the flag table (fc_bytes)
    uint32_t       *windows_out;    // 8 x u32 per window
    uint32_t       *matches_out;    // packed records, one region of region_cap records per wave
    unsigned long long *tile_off;   // tile directory: first record of each tile (index into matches_out)
    uint32_t       *tile_stats;     // tile directory: {matches, canonical, forward, visible (with emit, else 0)} per tile
    uint32_t       *wave_fill;      // per wave: records it needed (> region_cap means overflow); with emit, behind them per wave: visible records it needed (> vis_cap)
    uint32_t       *tile_tickets;   // 2 x ticket_groups ticket counters, TS_TICKET_STRIDE dwords apart; set [ticket_slot] is zero at launch
    uint32_t        region_cap;     // records per wave region
    uint32_t        ntiles;
    uint32_t        waves_per_wg;
    uint32_t        table_rows;     // dwords of the pair table: 4^(k+1) / 16, or 4^(k+1) / 4 as a byte table
    uint32_t        pair_byte_table;// 1: one byte per (k+1)-mer (k <= 6); 0: 2 bits per (k+1)-mer
    uint32_t        fc_bytes;       // flag table size in bytes (16-byte multiple)
    uint32_t        fc_byte_table;  // 1: one byte {forward, canonical} per k-mer; 0: 2 bits per k-mer
    uint32_t        k;              // pattern length
    uint32_t        s, w;           // step and window (tips mode: s = w = tile size)
    uint32_t        s_inv;          // ceil(2^16 / s): 24-bit multiply division by s (tile-relative positions < 2^16)
    uint32_t        s_magic22;      // ceil(2^22 / s) and
    uint32_t        div_exact;      // 1: u x s_magic22 >> 22 == u / s for every tile position u (no correction step)
    uint32_t        halo_blocks;    // step blocks read beyond the owned ones: ceil(w / s) - 1
    uint32_t        nch;            // chunks of TS_CHUNK positions per tile
    uint32_t        max_windows;    // windows per tile (rows of the LDS record buffer)
    uint32_t        acc_copies;     // lane-interleaved copies of the window match accumulators (power of two)
    uint32_t        stage_cap;      // match records a wave can stage in LDS before it flushes them
    uint32_t        stage_u16;      // 1: the stage holds 16-bit entries (position << 2 | flags < 2^16: tiles of at most 16 k positions)
    uint32_t        fold_mask;      // 0xDFDFDFDF (fold case) or 0xFFFFFFFF
    uint32_t        straddle_fix;   // 1: w == s, drop matches that straddle a window end
    uint32_t        windows_on;     // 0 in tips-only mode
    uint32_t        nuc_on;         // nucleotide counts wanted (-g / -e)
    uint32_t        block_sums;     // 1: w is a multiple of s, nucleotide counts are summed per step block
    uint32_t        acc_blocks;     // 1: (w a multiple of s) the match fields of the window records accumulate per step block too
    uint32_t        dynamic_tiles;  // 1: a wave takes the next free tile (ticket counter); 0: tiles dealt round-robin
    uint32_t        ticket_slot;    // which of the two counter sets this launch counts on (it zeroes the other)
    uint32_t        ticket_groups;  // groups of workgroups with a counter each (<= 64; group g owns the tiles t = g mod groups)
    uint32_t        wgs_per_cu;     // host side: workgroups that share a CU (1, or 2 of 10 waves: selects the 80-VGPR build)
    // ---- what the scan hands to block calling and to a shard's message (window scans only; emit == 0: none of it)
    uint32_t        emit;           // 1: visible records + per-tile chain summaries are produced; 2 (tips-only read batches): the indices of the
                                    // canonical records among every tile's records instead, no summaries (vis_out, tile_chain words 2-3, tile_stats word 3 as for 1)
    uint32_t        kdist;          // -k (maxMatchDistance): matches farther apart than this start a new chain
    uint32_t        vis_wide;       // 0: visible records are u16 (tile positions < 2^14), 1: u32
    uint32_t        vis_cap;        // visible records per wave region
    void           *vis_out;        // visible records, one region of vis_cap records per wave (u16 or u32 each)
    const uint32_t *tile_zone;      // per tile: {zlo, zhi} 16 bits each: a non-canonical record at tile position u is visible
                                    // (lies in its segment's terminal zone) iff u < zlo || u >= zhi
    uint32_t       *tile_chain;     // per tile: TsTileChain (4 dwords)
    // ---- a shard's scan that packs its window records itself (ts_batch_bind_shard_message; emit only).  nullptr: 8 x u32 per
    // window to windows_out.  Else the record of plan window i, win_pack_lo <= i < win_pack_hi (the owned windows), goes to
    // win_packed + i * win_pack_bytes in the message's bit-packed form (shard.hip: ts_shard_pack_windows) and windows_out is
    // not written at all.
    uint8_t        *win_packed;
    unsigned long long win_pack_lo, win_pack_hi;
    uint32_t        win_pack_bytes, win_field_bits;
    // ---- a read filter's batch (tips-only, 16-bit stage entries, every read terminal zone as a whole, no emit): the records leave as
    // 16 bits each — the stage's own entries: position (14 bits) << 2 | forward << 1 | canonical — into matches_out read as uint16_t
    // (regions of region_cap records as before).  Their only reader is the read predicate (predicate.hip), and the records are a
    // tenth of what the read step moves through HBM, a step that runs at what HBM gives (profiles/r05/reads_u16_records.txt).
    uint32_t        rec16;
};

// What ts_scan_tiles leaves per tile beside {matches, canonical, forward, visible} in tile_stats: the summary the
// interstitial screening needs instead of re-reading the tile's records, and where the tile's visible records are.
// A chain (getInterstitialBlocks, src/teloscope.cpp:235-253) links consecutive matches of any kind at most -k apart; an
// "internal head" is a record of the tile more than -k behind its predecessor IN the tile.
//   word 0   position of the tile's first record | position of its last record << 16      (tile-relative, < 2^16)
//   word 1   bits 0-14  canonical records ahead of the first internal head (all of them when there is none), saturating
//            bit  15    the tile holds an internal head
//            bits 16-30 canonical records from the last internal head on (all of them when there is none), saturating
//            bit  31    a chain between two internal heads of the tile holds four or more canonical records
//   word 2,3 index (u64) of the tile's first visible record in vis_out
#define TS_CHAIN_HEADS 0x8000u
#define TS_CHAIN_INNER 0x80000000u
#define TS_ZONE_NONE   0xFFFF0000u      // tile_zone word of a tile no position of which is terminal

// parameters of getTerminalBlocks for the device-side predicate (kernels.hip: ts_terminal_predicate)
struct TsPredParams {
    uint32_t terminal_limit;
    uint32_t max_match_dist, min_block_len, max_block_dist, min_block_counts;
    float    min_block_density;
    uint32_t k;                         // match length (uniform)
    uint32_t long_list;                 // least number of match records that makes a read "long" (walked by a whole wave, predicate.hip)
};

// ---- device block calling (blockcall.hip) ----
struct TsDevBlock {                     // ts_block (include/teloscan.h) + bookkeeping, 64 bytes
    unsigned long long start;
    uint32_t block_len, block_counts, forward_count, reverse_count, canonical_count, non_canonical_count;
    uint32_t total_covered, fwd_covered, can_covered;
    uint8_t  has_valid_or, is_longest;
    char     block_label;
    uint8_t  reserved;
    uint32_t seg;                       // segment index
    uint32_t kind;                      // 0/1 terminal (forward / reverse walk), 2 interstitial
    uint32_t seq;                       // push order among the segment's terminal blocks
    uint32_t pad;
};

struct TsBlockCallParams {
    const TsTile *tiles;
    const unsigned long long *tile_off;
    const uint32_t *tile_stats;
    const uint32_t *matches;
    TsDevBlock *blocks;
    uint32_t *n_blocks;                 // atomic counter; > block_cap means overflow
    uint32_t block_cap;
    uint32_t *cand;                     // chains the interstitial screening lists for exact evaluation: {tile, record index} pairs
    uint32_t *n_cand;                   // atomic counter (zero at launch); > cand_cap: reported as a block overflow
    uint32_t cand_cap;
    uint32_t terminal_limit, max_match_dist, min_block_len, max_block_dist, min_block_counts;
    float    min_block_density;
    uint32_t k;                         // match length (uniform)
    uint32_t its_min_len;               // 2 * patterns.front().size()
    uint32_t rec16;                     // the tiled kernel's records at 16 bits each (TsScanParams.rec16): `matches` is then read as uint16_t
    unsigned long long gen_lens;        // 0: the tiled kernel's records (position << 2 | forward << 1 | canonical, length k);
                                        // else the general kernels' (position << 5 | length index << 2 | canonical << 1 | forward)
                                        // and the up to eight pattern lengths, six bits each, length index i at bits 6i..
    // the general path's other cases (blockcall.hip, MODE 1)
    uint32_t wide;                      // records of the wide form: position << 8 | length index (six bits) << 2 | canonical << 1 | forward
    uint32_t unordered;                 // the stream lies in the reference's PUSH order (generic.hip: ts_general_compact_push), not in position order
    const uint32_t *wide_len;           // wide: the lengths by index (device)
    unsigned long long *its_range;      // unordered: per segment {first, end} stream indices of the interstitial search (ts_its_range writes, the search reads)
};

// ---- shard results (shard.hip, shard.cpp): what one device contributes to a scan that several devices share ----
// A shard OWNS a run of consecutive tiles of the batch's plan and additionally scans a few CONTEXT tiles either side
// where a segment continues on a neighbour (blocks are chains of matches: a chain that begins in an owned tile is
// followed into the context).  After its scan it packs ONE message:
//
//   TsShardHeader | TsShardSeg x n_segs | packed window records | u16 visible-record count per owned tile |
//   visible records (u16 or u32 each, capacity) | TsDevBlock x capacity
//
// "visible" = the match records a writer reads: canonicalMatches, and nonCanonicalMatches of the terminal zone
// (src/teloscope.cpp:486-496, writers :700-868); everything else the match stream holds is consumed by block calling,
// which has happened on the device.  Section offsets are a pure function of the plan, the split and the capacity scale
// (shard_layout, shard.cpp), so sender and receiver agree on them without talking.
#define TS_SHARD_MAGIC   0x44485354u    // "TSHD"
#define TS_SHARD_VERSION 1u
#define TS_SHARD_F_VISIBLE_OVERFLOW 0x1u    // more visible records than the section holds (n_visible says how many)
#define TS_SHARD_F_BLOCK_OVERFLOW   0x2u    // more blocks than the section holds (n_blocks says how many)
#define TS_SHARD_F_SCAN_OVERFLOW    0x4u    // a wave's record region overflowed in the scan: ts_batch_sync, then pack again
#define TS_SHARD_F_CONTEXT          0x8u    // a chain or a terminal walk ran out of the context tiles: the segment needs the full path

struct TsShardHeader {                  // 128 bytes
    uint32_t magic, version;
    uint32_t part, n_parts;
    unsigned long long own_begin, own_end;      // tiles (plan indices)
    unsigned long long ext_begin, ext_end;
    unsigned long long seg_begin;               // first segment with an owned tile
    uint32_t n_segs;
    uint32_t flags;                             // TS_SHARD_F_*
    unsigned long long n_visible;               // visible records of the owned tiles (may exceed the capacity: overflow)
    uint32_t n_blocks;                          // blocks emitted (may exceed the capacity: overflow)
    uint32_t visible_bytes;                     // 2 or 4 per visible record
    unsigned long long visible_capacity;
    uint32_t block_capacity;
    uint32_t window_bytes;                      // bytes per packed window record
    unsigned long long n_windows;               // owned window records
    unsigned long long msg_bytes;
    unsigned long long reserved[2];             // [0]: scratch of the pack (candidate counter), 0 on the wire; [1]: the capacity scale
};
static_assert(sizeof(TsShardHeader) == 128, "TsShardHeader is 128 bytes on the wire");

#define TS_SEG_F_HAS_START   0x1u       // the shard owns the segment's first tile: it walked the forward list from the start
#define TS_SEG_F_HAS_END     0x2u       // ... its last tile: it walked the reverse list from the end
#define TS_SEG_F_FWD_WALKED  0x4u       // the forward walk ran (>= 2 forward matches among the tiles the shard sees)
#define TS_SEG_F_REV_WALKED  0x8u
#define TS_SEG_F_CONTEXT     0x10u      // a walk or a chain of this segment ran out of context
struct TsShardSeg {                     // 64 bytes, one per segment with an owned tile
    unsigned long long fwd_boundary, rev_boundary;  // segment-relative; what this shard used for its interstitial search
    unsigned long long n_matches, n_canonical, n_forward;   // over the OWNED tiles of the segment
    unsigned long long seen_matches, seen_forward;  // over all its tiles the shard scanned (owned + context)
    uint32_t flags;
    uint32_t reserved;
};
static_assert(sizeof(TsShardSeg) == 64, "TsShardSeg is 64 bytes on the wire");

struct TsShardSegIn {                   // per-segment table of the shard kernels, 64 bytes
    unsigned long long in_off, len, abs_pos;    // input layout offset, length, absolute position of the segment
    unsigned long long lo_rel, hi_rel;          // segment-relative position of the first base of tile t0 / behind the last base of
                                                // tile t1 - 1 (0 and len when the shard holds none of its tiles)
    uint32_t t0, t1;                            // its tiles the shard scanned, as indices into the batch's (range-local) arrays
    uint32_t o0, o1;                            // its OWNED tiles, same indexing (o0 == o1: none)
    uint32_t flags;                             // TS_SEG_F_HAS_START / HAS_END
    uint32_t seg;                               // plan index of the segment
};

struct TsVisibleOut {                   // the visible records of a shard's owned tiles, written by the interstitial pass
    const unsigned long long *off;      // per owned tile (+1): index of its first visible record; nullptr = not wanted
    void *dst;                          // the message's visible section
    unsigned long long capacity;        // records it holds (more in all: nothing is written, the header reports it)
    uint32_t own0, own1;                // owned tiles, range-local
    uint32_t rec_bytes;                 // 2 or 4
    uint32_t terminal_limit;
};

struct TsShardPackParams {
    const TsTile *tiles;                // range-local arrays of the batch (the range = owned + context tiles)
    const unsigned long long *tile_off;
    const uint32_t *tile_stats;
    const uint32_t *matches;      uint32_t rec16;      // (read as uint16_t when rec16: TsScanParams.rec16)
    const uint32_t *windows;            // 8 x u32 per window of the range, from win_lo
    const uint32_t *wave_fill;
    uint32_t region_cap, nwaves;
    uint32_t own0, own1;                // owned tiles, range-local
    unsigned long long win_lo;          // plan index of the range's first window record
    unsigned long long own_win0, own_win1;      // owned window records (plan indices)
    const TsShardSegIn *segs;
    uint32_t n_segs;
    uint32_t terminal_limit, k, nuc_on, field_bits;
    unsigned char *msg;
    unsigned long long off_segs, off_windows, off_tilevis, off_visible, off_blocks;   // byte offsets of the sections
    // what the scan left for the message (kp.emit; chain == nullptr: the visible records come out of the match stream)
    const uint32_t *chain;              // TsTileChain per tile (words 2, 3: where the tile's visible records are in vis_src)
    const void *vis_src;                // the scan's visible records, per-wave regions
    uint32_t vis_src_wide;              // 1: u32 each, 0: u16
    uint32_t vis_cap;                   // records per wave region (wave_fill[nwaves + w] > vis_cap: that wave's region overflowed)
};

// ---- general kernels (generic.hip) ----
struct TsGenericPatterns {
    const unsigned long long *codes;    // per length: ascending 2-bit codes (base i at bits 2i..2i+1)
    const uint8_t *flags;               // bit0 forward, bit1 canonical (parallel to codes)
    uint32_t nlen;                      // distinct pattern lengths, ascending
    uint32_t len[8];
    uint32_t first[9];                  // codes[first[i] .. first[i+1]) have length len[i]
};

// The WIDE form's tables (generic.hip: ts_general_wide): pattern sets beyond the table forms above — up to 63 distinct lengths of
// up to 63 bases, what a ts_pattern can hold.  Codes are 128 bits (lo: bases 0..31, hi: bases 32..62), per length ascending by
// (lo, hi); everything lives in device memory.
struct TsWidePatterns {
    const unsigned long long *lo, *hi;
    const uint8_t *flags;               // bit0 forward, bit1 canonical (parallel to lo / hi)
    const uint32_t *len;                // nlen lengths, ascending
    const uint32_t *first;              // nlen + 1: patterns [first[i], first[i + 1]) have length len[i]
    uint32_t nlen, npat;
};
#define TS_WIDE_HALO 64                 // bases staged beyond a tile for the wide form (the other forms: 32)

struct TsGenericGeom {
    uint32_t s, w, longest;
    uint32_t nuc_on, fold;
    uint32_t abl;                       // TS_GEN_ABL (profiling): strided form: 1 no matching, 2 no window records, 4 no match records, 8 nothing
                                        // after the table load; list form: 16 no candidates, 32 no per-candidate pass, 64 no window records
    uint32_t s_magic;                   // floor(2^32 / s) + 1 (s >= 2): x / s == umulhi(x, s_magic) for x < 2^32 / s (the list form, s <= 8192)
    uint32_t cw, rw;                    // w / s and w - cw s
};

#define TS_GENERAL_TILE 4096            // positions per tile of the general kernels (generic.hip)
struct TsGeneralTile {                  // 40 bytes
    unsigned long long in_off;          // byte offset (input layout) of the tile's first base; the match mask is indexed alike
    unsigned long long seg_rel;         // segment-relative position of that base (P0)
    unsigned long long k_p0;            // P0 / step and
    uint32_t n;                         // positions of the tile (<= TS_GENERAL_TILE)
    uint32_t avail;                     // bases from the tile's first base to the end of its region (clamped to n + 32)
    uint32_t seg;                       // segment index within the group
    uint32_t r_p0;                      // P0 - k_p0 step: the host walks them from tile to tile, so that the list kernel never
                                        // divides 64-bit numbers (five such divisions per tile and wave were a third of its
                                        // vector instructions)
};

struct TsLaunchInfo {
    uint32_t grid;
    uint32_t lds_bytes;
};

#ifdef __cplusplus
extern "C++" {
// Implemented in kernels.hip (compiled by hipcc).  All return a hipError_t as int.
int  ts_k_lds_bytes(const TsScanParams *p);
int  ts_k_prepare(uint32_t lds_bytes);                       // raises the dynamic-LDS limit
int  ts_k_launch_scan(const TsScanParams *p, uint32_t grid, uint32_t lds_bytes, void *stream);
int  ts_k_launch_summary(const uint32_t *tile_stats, const uint32_t *seg_first_tile,
                         const uint64_t *seg_nwin, uint32_t nseg, unsigned long long *out, void *stream);
unsigned long long ts_k_general_lds_bytes(const TsGenericPatterns *G, uint32_t *lds_patterns);
int  ts_k_launch_general_fused(const unsigned char *in, const TsGeneralTile *tiles, uint32_t ntiles,
                               const unsigned long long *seg_len, const unsigned long long *seg_win_base, const unsigned long long *seg_nwin,
                               const TsGenericPatterns *G, const TsGenericGeom *Q, int tips, uint32_t slot_cap,
                               uint32_t *tile_stats, uint32_t *records, uint32_t *win_out, uint32_t *overflow, int list, int num_cu, void *stream);
uint32_t ts_k_general_list_max_records(void);
// the wide form of the same pass: any set a ts_pattern[] can express; records are (tile position << 8) | (length index << 2) |
// canonical << 1 | forward; tiles' avail is clamped to n + TS_WIDE_HALO; *overflow bit 0 as above
int  ts_k_launch_general_wide(const unsigned char *in, const TsGeneralTile *tiles, uint32_t ntiles,
                              const unsigned long long *seg_len, const unsigned long long *seg_win_base, const unsigned long long *seg_nwin,
                              const TsWidePatterns *W, const TsGenericGeom *Q, int tips, uint32_t slot_cap,
                              uint32_t *tile_stats, uint32_t *records, uint32_t *win_out, uint32_t *overflow, void *stream);
                               // (records: ntiles x slot_cap entries; win_out zeroed by the caller; *overflow raised when a tile
                               //  holds more than slot_cap records — its count is still written; list != 0: the list form of
                               //  the pass — for parameter sets whose tiles add to at most ts_k_general_list_max_records()
                               //  window records and whose pattern lists fit LDS; *overflow bit 1: run again with list = 0)
// after the compaction: what device block calling needs of the group (TsTile per tile with in_off = seg_base[seg] + the tile's
// segment-relative position; canonical / forward counts into words 1, 2 of the tile directory)
int  ts_k_launch_general_block_inputs(const TsGeneralTile *gtiles, const unsigned long long *tile_off, const uint32_t *dense,
                                      const unsigned long long *seg_base, uint32_t ntiles, TsTile *tiles, uint32_t *tile_stats,
                                      int count_from_offsets, void *stream);
int  ts_k_launch_general_slot_offsets(unsigned long long *tile_off, uint32_t ntiles, uint32_t slot_cap, void *stream);
int  ts_k_launch_general_compact_push(const TsGeneralTile *gtiles, const uint32_t *tile_stats, unsigned long long *tile_off,
                                      const uint32_t *records, uint32_t slot_cap, uint32_t ntiles, const unsigned long long *seg_len,
                                      uint32_t w, uint32_t s, uint32_t spread, int wide, unsigned long long gen_lens,
                                      const uint32_t *wide_len, uint32_t *dense, void *stream);
int  ts_k_launch_general_compact(const uint32_t *tile_stats, const unsigned long long *tile_off, const uint32_t *records,
                                 uint32_t slot_cap, uint32_t ntiles, uint32_t *dense, void *stream);
int  ts_k_launch_predicate(const TsTile *tiles, const unsigned long long *tile_off, const uint32_t *tile_stats,
                           const uint32_t *matches, unsigned long long nrec_limit, const uint32_t *seg_first_tile,
                           const unsigned long long *seg_in_off, const unsigned long long *seg_len,
                           uint32_t nseg, const TsPredParams *Q, unsigned char *pass, uint32_t *long_list,
                           uint32_t *long_count, int all_terminal, const uint32_t *wave_fill, uint32_t region_cap,
                           uint32_t nwaves, uint32_t *overflow, const uint32_t *chain, const void *canon_idx, uint32_t vis_cap, int rec16, void *stream);
                           // (chain + canon_idx, both or neither: a read batch's scan left the indices of the canonical records — u16 each,
                           //  per-wave regions of vis_cap, TsTileChain words 2-3 say where a tile's are, tile_stats word 3 how many — and
                           //  the predicate visits only the chains that hold one)
                           // (nrec_limit: records that may be READ behind `matches` — the predicate fetches aligned 16-byte blocks;
                           //  long_list: nseg entries of scratch + long_count: one counter, for the reads a whole wave walks;
                           //  all_terminal: no segment is longer than the terminal limit — every read batch — the lean kernel;
                           //  wave_fill / region_cap / nwaves -> *overflow: raised, and nothing judged, when the scan overflowed a wave's region)
// blockcall.hip: terminal walks per segment of `segs` (nseg entries; bounds: 2 x u64 per segment), then the interstitial
// search over the batch's ntiles (range-local) tiles; seg_base = plan index of segs[0]'s segment; seg_out (nullable):
// what a shard reports per segment
// vis (nullable): where the interstitial pass leaves the visible records of the owned tiles (a shard's message)
// sums (nullable): 5 x u64 per segment, the per-segment totals for a caller that wants them
// chain + work (both or neither): the scan's per-tile chain summaries (TsTileChain) and scratch of (ntiles + 1) dwords — the
// interstitial search then screens the tiles by their summaries and walks only the listed ones (ignored when vis asks for
// the visible records out of the match stream)
int  ts_k_launch_block_call(const TsBlockCallParams *Q, const TsShardSegIn *segs, uint32_t nseg, uint32_t seg_base,
                            uint32_t ntiles, unsigned long long *bounds, TsShardSeg *seg_out, int with_its,
                            const TsVisibleOut *vis, unsigned long long *sums, const uint32_t *chain, uint32_t *work, void *stream);
// the terminal walks alone (they answer scanSegment's ">= 2 matches" gates from the tile directory themselves)
int  ts_k_launch_terminal(const TsBlockCallParams *Q, const TsShardSegIn *segs, uint32_t nseg, uint32_t seg_base, uint32_t ntiles,
                          unsigned long long *bounds, TsShardSeg *seg_out, void *stream);
// per-segment totals into sums (5 x u64 per segment) or into the segments' entries of a shard's message (seg_out)
int  ts_k_launch_segment_sums(const TsBlockCallParams *Q, const TsShardSegIn *segs, uint32_t nseg, uint32_t seg_base, uint32_t ntiles,
                              unsigned long long *sums, TsShardSeg *seg_out, int prezeroed, void *stream);
int  ts_k_launch_zero(void *p0, unsigned long long n0, void *p1, unsigned long long n1, void *p2, unsigned long long n2,
                      void *p3, unsigned long long n3, void *stream);
int  ts_k_launch_interstitial(const TsBlockCallParams *Q, const TsShardSegIn *segs, uint32_t nseg, uint32_t seg_base, uint32_t ntiles,
                              const unsigned long long *bounds, TsShardSeg *seg_out, const TsVisibleOut *vis,
                              const uint32_t *chain, uint32_t *work, int prezeroed, void *stream);
// shard.hip: packed window records, visible records + per-tile counts, and the header of a shard's message
int  ts_k_launch_shard_count(const TsShardPackParams *P, const TsShardHeader *H, void *tmp, int with_visible, TsVisibleOut *vis, void *stream);
// the same from the scan's own visible records (P->chain != nullptr): per-tile counts out of tile_stats, prefix sum, then the
// records copied into the message in tile order — nothing is left for the interstitial pass to write
int  ts_k_launch_shard_visible(const TsShardPackParams *P, const TsShardHeader *H, void *tmp, int prezeroed, void *stream);
void *ts_k_shard_big_counter(void *tmp, uint32_t own_tiles);      // the 4 bytes ts_k_launch_shard_visible wants zero
int  ts_k_launch_shard_windows(const TsShardPackParams *P, const TsShardHeader *H, void *stream);
// 1 in *concurrent when a kernel on stream b runs while one on stream a is still running (the streams sit on different hardware
// queues); both streams are synchronised first; ~1 ms.
int  ts_k_streams_concurrent(void *a, void *b, int *concurrent);
int  ts_k_launch_shard_overflow(const TsShardPackParams *P, void *stream);     // after the header's memset, before ts_k_launch_shard_pack
int  ts_k_launch_shard_pack(const TsShardPackParams *P, const TsShardHeader *H, void *tmp, int with_visible, void *stream);
unsigned long long ts_k_shard_tmp_bytes(uint32_t own_tiles);
// exchange.hip: tile directory of a dense tile-ordered stream, and the export of a scan's records into one
unsigned long long ts_k_scan_tmp_bytes(uint32_t ntiles);
int  ts_k_launch_tile_offsets(const uint32_t *tile_stats, uint32_t ntiles, unsigned long long *tile_off, void *tmp,
                              void *stream);
int  ts_k_launch_tile_order_export(const uint32_t *tile_stats, const unsigned long long *region_off,
                                   const uint32_t *regions, const uint32_t *wave_fill, uint32_t region_cap,
                                   uint32_t nwaves, uint32_t ntiles, unsigned long long *dense_off, void *tmp,
                                   uint32_t *dense, unsigned long long capacity, unsigned long long *total_out,
                                   int rec16, void *stream);
// unpack.hip: a staged chunk of 2-bit codes (pack.cpp) -> the byte layout: chunk positions [first, first + n) to dst, then
// 'N' over the chunk's invalid runs ({start, len} pairs, device memory; run position x lies at runs_base + x).
// packed: 4-byte aligned, 8 readable bytes behind the last code.
int  ts_k_launch_unpack(const void *packed, uint32_t first, void *dst, unsigned long long n, const void *runs, uint32_t nruns,
                        void *runs_base, void *stream);
// exchange.hip: box calibration (see there)
int  ts_k_read_index_built(void);      // kernels.hip: built with -DTS_READ_INDEX_BUILD=1 (the read filter's canonical-index experiment)
int  ts_k_box_probe(void *scratch, unsigned long long bytes, int num_cu, double *issue_per_ns, double *copy_bytes_per_ns, void *stream);
int  ts_k_launch_widen_u16(const uint16_t *src, uint32_t *dst, unsigned long long n, void *stream);
int  ts_k_launch_compact(const uint32_t *regions, const uint32_t *wave_fill,
                         const unsigned long long *wave_dense_base, uint32_t region_cap,
                         uint32_t nwaves, uint32_t *dense, int rec16, void *stream);
}
#endif

#endif
