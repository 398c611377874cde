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
    uint32_t       *tile_stats;     // tile directory: {matches, canonical, forward, 0} per tile
    uint32_t       *wave_fill;      // per wave: records it needed (> region_cap means overflow)
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
    uint32_t        halo_blocks;    // step blocks read beyond the owned ones: ceil(w / s) - 1
    uint32_t        nch;            // chunks of TS_CHUNK positions per tile
    uint32_t        max_windows;    // windows per tile (rows of the LDS record buffer)
    uint32_t        acc_copies;     // lane-interleaved copies of the window match accumulators (power of two)
    uint32_t        stage_cap;      // match records a wave can stage in LDS before it flushes them
    uint32_t        fold_mask;      // 0xDFDFDFDF (fold case) or 0xFFFFFFFF
    uint32_t        straddle_fix;   // 1: w == s, drop matches that straddle a window end
    uint32_t        windows_on;     // 0 in tips-only mode
    uint32_t        nuc_on;         // nucleotide counts wanted (-g / -e)
    uint32_t        block_sums;     // 1: w is a multiple of s, nucleotide counts are summed per step block
    uint32_t        dynamic_tiles;  // 1: a wave takes the next free tile (ticket counter); 0: tiles dealt round-robin
    uint32_t        ticket_slot;    // which of the two counter sets this launch counts on (it zeroes the other)
    uint32_t        ticket_groups;  // groups of workgroups with a counter each (<= 64; group g owns the tiles t = g mod groups)
    uint32_t        wgs_per_cu;     // host side: workgroups that share a CU (1, or 2 of 10 waves: selects the 80-VGPR build)
};

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
    uint32_t terminal_limit, max_match_dist, min_block_len, max_block_dist, min_block_counts;
    float    min_block_density;
    uint32_t k;                         // match length (uniform)
    uint32_t its_min_len;               // 2 * patterns.front().size()
};

// ---- general kernels (generic.hip) ----
struct TsGenericPatterns {
    const unsigned long long *codes;    // per length: ascending 2-bit codes (base i at bits 2i..2i+1)
    const uint8_t *flags;               // bit0 forward, bit1 canonical (parallel to codes)
    uint32_t nlen;                      // distinct pattern lengths, ascending
    uint32_t len[8];
    uint32_t first[9];                  // codes[first[i] .. first[i+1]) have length len[i]
};

struct TsGenericGeom {
    uint32_t s, w, longest;
    uint32_t nuc_on, fold;
};

#define TS_GENERAL_TILE 4096            // positions per tile of the general kernels (generic.hip)
struct TsGeneralTile {                  // 32 bytes
    unsigned long long in_off;          // byte offset (input layout) of the tile's first base; the match mask is indexed alike
    unsigned long long seg_rel;         // segment-relative position of that base
    uint32_t n;                         // positions of the tile (<= TS_GENERAL_TILE)
    uint32_t avail;                     // bases from the tile's first base to the end of its region (clamped to n + 32)
    uint32_t seg;                       // segment index within the group
    uint32_t pad;
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
int  ts_k_launch_general_match(const unsigned char *in, const TsGeneralTile *tiles, uint32_t ntiles,
                               const TsGenericPatterns *G, uint32_t fold, uint32_t *mask, void *stream);
int  ts_k_launch_general_records(const uint32_t *mask, const TsGeneralTile *tiles, uint32_t ntiles,
                                 const unsigned long long *seg_len, const TsGenericPatterns *G, const TsGenericGeom *Q,
                                 int tips, uint32_t *tile_stats, const unsigned long long *tile_off, uint32_t *records,
                                 int emit, void *stream);
int  ts_k_launch_general_windows(const unsigned char *in, const uint32_t *mask, const TsGenericPatterns *G,
                                 const TsGenericGeom *Q, const unsigned long long *seg_win_base,
                                 const unsigned long long *seg_in_off, const unsigned long long *seg_len, uint32_t nseg,
                                 unsigned long long nwin, uint32_t *out, void *stream);
int  ts_k_launch_predicate(const TsTile *tiles, const unsigned long long *tile_off, const uint32_t *tile_stats,
                           const uint32_t *matches, unsigned long long nrec_limit, const uint32_t *seg_first_tile,
                           const unsigned long long *seg_in_off, const unsigned long long *seg_len,
                           uint32_t nseg, const TsPredParams *Q, unsigned char *pass, uint32_t *long_list,
                           uint32_t *long_count, int all_terminal, void *stream);
                           // (nrec_limit: records that may be READ behind `matches` — the predicate fetches aligned 16-byte blocks;
                           //  long_list: nseg entries of scratch + long_count: one counter, for the reads a whole wave walks;
                           //  all_terminal: no segment is longer than the terminal limit — every read batch — the lean kernel)
int  ts_k_launch_block_call(const TsBlockCallParams *Q, const uint32_t *seg_first_tile,
                            const unsigned long long *seg_in_off, const unsigned long long *seg_len,
                            const unsigned long long *seg_abs, uint32_t nseg, uint32_t ntiles,
                            unsigned long long *bounds, int with_its, void *stream);
// exchange.hip: tile directory of a dense tile-ordered stream, and the export of a scan's records into one
unsigned long long ts_k_scan_tmp_bytes(uint32_t ntiles);
int  ts_k_launch_tile_offsets(const uint32_t *tile_stats, uint32_t ntiles, unsigned long long *tile_off, void *tmp,
                              void *stream);
int  ts_k_launch_tile_order_export(const uint32_t *tile_stats, const unsigned long long *region_off,
                                   const uint32_t *regions, const uint32_t *wave_fill, uint32_t region_cap,
                                   uint32_t nwaves, uint32_t ntiles, unsigned long long *dense_off, void *tmp,
                                   uint32_t *dense, unsigned long long capacity, unsigned long long *total_out,
                                   void *stream);
int  ts_k_launch_widen_u16(const uint16_t *src, uint32_t *dst, unsigned long long n, void *stream);
int  ts_k_launch_compact(const uint32_t *regions, const uint32_t *wave_fill,
                         const unsigned long long *wave_dense_base, uint32_t region_cap,
                         uint32_t nwaves, uint32_t *dense, void *stream);
}
#endif

#endif
