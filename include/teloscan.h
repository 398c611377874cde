/*
 * teloscan.h — C-ABI of libteloscan.so: the MI355X (gfx950) implementation of
 * Teloscope's telomeric-motif scan path.
 *
 * This is the drop-in boundary.  The reference has no plugin registry; the seam
 * is two C++ member functions,
 *
 *     SegmentData Teloscope::scanSegment(std::string&, uint64_t absPos, bool tipsOnly);
 *                                                   (include/teloscope.h:260, src/teloscope.cpp:537)
 *     bool        ReadTelomereFilter::matches(std::string);
 *                                                   (include/read-filter.h:16, src/read-filter.cpp:37)
 *
 * and every entry point below names the reference interface it replaces.
 * Plain pointers and sizes only; no C++ or torch types.  All functions that
 * return int return TS_OK (0) or a negative ts_status; ts_last_error() gives
 * the text.  The library never calls exit().
 *
 * There is no CPU fallback: if no HIP device is usable, ts_create() fails
 * with TS_ERR_NO_DEVICE.
 */
#ifndef TELOSCAN_H
#define TELOSCAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TELOSCAN_ABI_VERSION 4

/* ts_params.device value of a PLANNING-ONLY context: no HIP call is ever made behind it.  It plans batches
 * (ts_batch_create, ts_batch_get_info, ts_batch_get_tiles, ts_batch_partition, ts_batch_range_info) so that a
 * host without a GPU — the rank that only merges, a test — sees the same tiling as the ranks that scan; every
 * entry point that needs the device fails on it with TS_ERR_NO_DEVICE.  It is not a CPU scan path. */
#define TS_DEVICE_NONE (-2)

typedef enum ts_status {
    TS_OK               =  0,
    TS_ERR_INVALID_ARG  = -1,
    TS_ERR_NO_DEVICE    = -2,   /* no usable HIP device (no CPU fallback exists) */
    TS_ERR_HIP          = -3,   /* a HIP runtime call failed */
    TS_ERR_ALLOC        = -4,
    TS_ERR_UNSUPPORTED  = -5,   /* parameter set no device path implements (pattern > 32 bases, > 8 lengths, ...) */
    TS_ERR_STATE        = -6
} ts_status;

/* ScaffoldType, include/tools.h:13-19 (same enumerator order). */
typedef enum ts_scaffold_type {
    TS_T2T = 0, TS_GAPPED_T2T, TS_MISASSEMBLY, TS_GAPPED_MISASSEMBLY,
    TS_INCOMPLETE, TS_GAPPED_INCOMPLETE, TS_NONE, TS_GAPPED_NONE,
    TS_DISCORDANT, TS_GAPPED_DISCORDANT
} ts_scaffold_type;

/* One expanded search pattern = one element of UserInputTeloscope::patternInfo
 * plus the isCanonical flag the Teloscope ctor derives (include/teloscope.h:241-247). */
typedef struct ts_pattern {
    char    seq[64];        /* NUL-terminated, A/C/G/T only */
    uint8_t len;
    uint8_t is_forward;
    uint8_t is_canonical;
    uint8_t reserved;
} ts_pattern;

/* The fields of UserInputTeloscope (include/input.h:15-64) the scan path reads. */
typedef struct ts_params {
    uint32_t struct_size;        /* = sizeof(ts_params), for ABI growth */
    uint32_t window_size;        /* -w, default 1000 */
    uint32_t step;               /* -s, default 1000, must be <= window_size */
    uint32_t terminal_limit;     /* -t, default 50000 */
    uint16_t max_match_dist;     /* -k, default 50 */
    uint16_t min_block_len;      /* -l, default 300 */
    uint16_t max_block_dist;     /* -d, default 500 */
    uint16_t min_block_counts;   /* default 2 */
    float    min_block_density;  /* -y, default 0.5 */
    uint16_t canonical_size;     /* strlen(canonical) */
    uint8_t  out_gc;             /* -g : nucleotide counts + gc_content are produced */
    uint8_t  out_entropy;        /* -e : nucleotide counts + shannon_entropy */
    uint8_t  out_matches;        /* -m */
    uint8_t  out_its;            /* -i */
    uint8_t  fold_case;          /* 1: a/c/g/t match like A/C/G/T (what every reference
                                       caller gets by calling unmaskSequence first);
                                    0: strict scanSegment semantics (lower case = non-ACGT) */
    uint8_t  reserved0;
    int32_t  device;             /* HIP device ordinal, -1 = current device, TS_DEVICE_NONE = planning only */
    uint32_t reserved1;
} ts_params;

/* MatchInfo, include/teloscope.h:89-95 (matchSeq is seq.substr(position-absPos, match_size)). */
#define TS_MATCH_FORWARD   0x1u
#define TS_MATCH_CANONICAL 0x2u
#define TS_MATCH_TERMINAL  0x4u   /* isTerminal of src/teloscope.cpp:451-459 */
typedef struct ts_match {
    uint64_t position;
    uint16_t match_size;
    uint8_t  flags;
    uint8_t  reserved[5];
} ts_match;

/* WindowData, include/teloscope.h:119-137, without the fields no writer reads. */
typedef struct ts_window {
    uint64_t window_start;           /* absolute (absPos + window offset) */
    uint32_t current_window_size;
    uint32_t nucleotide_counts[4];   /* A C G T; zero unless out_gc || out_entropy */
    float    gc_content;             /* getGCContent, include/teloscope.h:211 (if out_gc) */
    float    shannon_entropy;        /* getShannonEntropy, include/teloscope.h:199 (if out_entropy) */
    uint32_t canonical_covered;
    uint32_t non_canonical_covered;
    uint32_t fwd_covered;
    uint32_t rev_covered;
    uint32_t reserved;
} ts_window;

/* TelomereBlock, include/teloscope.h:103-117. */
typedef struct ts_block {
    uint64_t start;
    uint32_t block_len;
    uint32_t block_counts;
    uint32_t forward_count;
    uint32_t reverse_count;
    uint32_t canonical_count;
    uint32_t non_canonical_count;
    uint32_t total_covered;
    uint32_t fwd_covered;
    uint32_t can_covered;
    uint8_t  has_valid_or;
    uint8_t  is_longest;
    char     block_label;
    uint8_t  reserved;
} ts_block;

/* One scanSegment() call: (sequence, absPos, tipsOnly).
 *
 * input_format TS_INPUT_BASES (0): `seq` points to the segment's `len` bases.
 * input_format TS_INPUT_TEXT_PIECES: `seq` points to a ts_text_piece array instead — FASTA body text as it lies in the
 * file, line ends included; the pieces, in order, hold the segment's `len` bases.  The library skips the line ends while
 * it stages the bases for upload, so a front end never has to join the lines of a record into one buffer (what gfalibs
 * does before walkPath, and the largest host cost once the scan itself takes milliseconds).  The entry points read
 * pieces until `len` bases are covered (at most `n_pieces` of them); of the last piece only what is needed. */
#define TS_INPUT_BASES       0
#define TS_INPUT_TEXT_PIECES 1
#define TS_INPUT_PACKED2     2
typedef struct ts_text_piece {
    const char *text;        /* text_len bytes: n_bases bases and the line ends between / behind them ('\n', and a '\r'
                                right before one or at the very end of the piece) */
    uint64_t    text_len;    /* at most 16 MiB */
    uint64_t    n_bases;
} ts_text_piece;
/* input_format TS_INPUT_PACKED2: `seq` points to ONE ts_packed_seq — the segment's bases already as 2-bit codes (four per
 * byte, base i at bits 2 (i & 3) of byte i >> 2; A 0, C 1, T 2, G 3: what ts_pack_bases writes) plus the runs of positions
 * that are not A/C/G/T (after case folding, if the context folds case), ascending and non-overlapping, segment-relative.
 * A front end that parses FASTA touches every base once anyway: packing there (ts_pack_bases, or its own loop) means the
 * bases are read from host memory ONCE between the file and the PCIe link — the library's staging threads then copy a
 * quarter of the bytes instead of reading the ASCII a second time, which is what bounds the host entry points once the
 * link carries packed bases.  Results cannot depend on the input format: the device restores the same
 * byte layout ('N' over the runs) that TS_INPUT_BASES uploads.  Tiled kernel's parameter sets only, like the text pieces. */
typedef struct ts_packed_run { uint64_t start, len; } ts_packed_run;
typedef struct ts_packed_seq {
    const uint8_t       *codes;     /* (len + 3) / 4 bytes */
    const ts_packed_run *runs;      /* may be NULL when n_runs == 0 */
    uint64_t             n_runs;
} ts_packed_seq;
typedef struct ts_segment_in {
    const char *seq;        /* borrowed for the duration of the call; need not be NUL-terminated */
    uint64_t    len;
    uint64_t    abs_pos;
    uint8_t     tips_only;
    uint8_t     input_format;   /* TS_INPUT_BASES / TS_INPUT_TEXT_PIECES / TS_INPUT_PACKED2 (any parameter set the library scans) */
    uint8_t     reserved[2];
    uint32_t    n_pieces;       /* TS_INPUT_TEXT_PIECES: entries of the ts_text_piece array (the walk never reads past it;
                                   pieces that hold fewer than `len` bases are TS_ERR_INVALID_ARG) */
} ts_segment_in;

/* SegmentData, include/teloscope.h:139-148.  `matches` holds what the
 * reference pushes to allMatches (full scan) or to fwdMatches+revMatches
 * (tips-only), in the reference's push order; the other reference vectors are
 * subsequences selected by flags:
 *   canonicalMatches     = flags & CANONICAL                  (full scan only)
 *   nonCanonicalMatches  = !(flags & CANONICAL) && TERMINAL   (full scan only)
 *   fwdMatches / revMatches = flags & FORWARD / not
 * Arrays are owned by the library until ts_free_segments(). */
typedef struct ts_segment_out {
    ts_window *windows;             uint64_t n_windows;
    ts_match  *matches;             uint64_t n_matches;
    ts_block  *terminal_blocks;     uint64_t n_terminal_blocks;
    ts_block  *interstitial_blocks; uint64_t n_interstitial_blocks;
} ts_segment_out;

typedef struct ts_ctx   ts_ctx;
typedef struct ts_batch ts_batch;

int         ts_abi_version(void);
const char *ts_last_error(const ts_ctx *ctx);      /* ctx may be NULL: last create error */
int         ts_device_count(void);                 /* usable HIP devices, 0 if none */

/* ---- pattern preparation: expandPatternsWithOrientation, src/tools.cpp:201-283,
 *      plus the canonical-orientation rule of src/main.cpp:287-296. ---------------- */
/* canonical_in: the -c argument (upper-cased by the callee). Writes the
 * lexicographically smaller of (canonical, revcomp) to fwd_out, the other to rev_out
 * (each >= 64 bytes). */
int ts_canonical_orientation(const char *canonical_in, char *fwd_out, char *rev_out);
/* raw_csv: comma-separated -p list (IUPAC allowed). On success *out is a malloc'd
 * array of *n_out patterns sorted like the reference's patternInfo; free with
 * ts_free_patterns(). */
int  ts_expand_patterns(const char *raw_csv, int edit_distance, const char *canonical_fwd,
                        ts_pattern **out, size_t *n_out);
void ts_free_patterns(ts_pattern *p);

/* ---- context: the Teloscope object (ctor include/teloscope.h:241-247). Builds the
 *      k-mer match tables on the device.  Thread-safe for concurrent scan/filter calls, and made for them: the reference
 *      calls scanSegment / matches from all its pool workers at once (src/input.cpp:977, :786, src/bam.cpp:217-220), and
 *      calls that arrive while the device is busy are COALESCED — the next run takes every waiting call of one kind
 *      (ts_scan_segments, ts_scan_segments_blocks, ts_filter_reads; full scans and tips-only apart) as one batch and hands
 *      each caller its own results.  Sixty-four threads with one segment each cost about what one call with sixty-four
 *      segments costs; results never depend on who was merged with whom.
 *      LIMITS of the pattern set (the reference's trie has none, include/teloscope.h:40-57): what a ts_pattern[] can express — a
 *      pattern of up to 63 bases, hence up to 63 distinct lengths — is scanned; a longer pattern is refused by ts_create and by
 *      ts_expand_patterns (never truncated), a non-ACGT pattern makes every scan fail with TS_ERR_UNSUPPORTED (loudly: there
 *      is no CPU path to fall back to).  Uniform-length sets of 3..8 bases under w == s or k <= min(s, w - s) take the tiled
 *      kernel; sets of up to 8 distinct lengths of up to 32 bases the general kernels' table forms; everything else the
 *      general kernels' wide form (128-bit codes; a correct path for rare sets, not a fast one). */
ts_ctx *ts_create(const ts_params *params, const ts_pattern *patterns, size_t n_patterns);
void    ts_destroy(ts_ctx *ctx);
/* 1 if full scans with this (window, step, patterns) run on the tiled uniform-k kernel,
 * 0 if on the general kernels (mixed-length sets, k > 9, or a longest pattern exceeding
 * min(step, window-step), where the reference's start-index arithmetic wraps). */
int     ts_uses_fast_path(const ts_ctx *ctx);
/* Measurement aid (no counterpart in the reference): what THIS device issues — wave-instructions per ns of hand-written
 * independent integer instructions at four waves per SIMD — and streams — read + write bytes per ns of a 1 GiB copy —, so that
 * a benchmark line from a box that runs everything a few per cent slower can be told from a slower kernel (bench.py: roofline.box). */
int     ts_box_probe(ts_ctx *ctx, double *valu_wave_instr_per_ns, double *copy_bytes_per_ns);
/* 1 if segments of this kind (full scan / tips-only) may come as TS_INPUT_TEXT_PIECES or TS_INPUT_PACKED2: every parameter
 * set the library scans (until ABI 3 the general kernels wanted the bases joined). */
int     ts_takes_text_input(const ts_ctx *ctx, int tips_only);
/* The host entry points read a handful of measurement / test knobs from the environment (TS_TIMING, TS_PACKED_UPLOAD,
 * TS_PACKED_MIN_BYTES, TS_STAGE_THREADS, TS_GEN_HOST_BLOCKS, TS_GEN_PREFETCH, TS_GEN_LIST, TS_GEN_ABL) ONCE, when the context
 * is made — never per call.  This reads them again (tests and A/B scripts that flip one between two calls on one context).
 * No counterpart in the reference (its options are fixed by main, /root/reference/src/main.cpp:149-184). */
int     ts_refresh_env(ts_ctx *ctx);
/* HIP puts the streams of a process on a few hardware queues (four unless GPU_MAX_HW_QUEUES says otherwise) and does not say which;
 * kernels of two streams that share a queue run one after the other, whatever events allow.  Returns 1 when a kernel on stream_b
 * runs while one on stream_a is still running, 0 when the two streams share a queue (a caller that wants a pack beside the next
 * scan — ts_batch_pack_shard — makes streams until it holds a scan stream and pack streams that do not), negative on error.
 * Waits for the work both streams hold, then takes ~1 ms.  The library tries its own side stream (the terminal walks of
 * ts_batch_pack_shard) against every scan / pack stream it meets in the same way.  No counterpart in the reference (one thread
 * per path, /root/reference/src/input.cpp:719-733). */
int     ts_streams_concurrent(ts_ctx *ctx, void *stream_a, void *stream_b);
/* Restricts the CALLING thread (and the threads it starts from then on) to the CPUs of the NUMA node the context's
 * device is attached to; returns 1 if it did, 0 if the topology is unknown, the thread's mask holds none of those CPUs,
 * or TS_NO_NUMA_BIND is set.  The library's own pipeline threads do this by themselves; a front end calls it on the threads
 * that read, parse and format around the scan (the reference's ThreadPool workers, src/input.cpp:719-733): on a two-socket
 * host a buffer or a staging thread on the other socket costs 15-20 % of the PCIe-inclusive rate. */
int     ts_bind_thread_to_device(const ts_ctx *ctx);

/* ---- The packed upload's host half, as a utility (host only, no device needed): `n` bases at `src` -> 2-bit codes at
 *      `dst` ((n + 3) / 4 bytes; base i at bits 2 (i & 3) .. +1 of byte i >> 2; A 0, C 1, T 2, G 3 — the scan kernel's own
 *      code — and 0 for every byte that is not one of the four letters), and the runs of such bytes {start, length} in
 *      `runs` (up to run_cap; *n_runs = how many there are — more than run_cap: TS_ERR_INVALID_ARG, nothing usable).
 *      fold_case as in ts_params.  This is what the host entry points do to every chunk they upload (unless
 *      TS_PACKED_UPLOAD=0): a quarter of the bytes cross PCIe, and a kernel restores the byte layout in HBM. */
typedef struct ts_invalid_run { uint32_t start, len; } ts_invalid_run;
int     ts_pack_bases(const char *src, uint64_t n, int fold_case, uint8_t *dst, ts_invalid_run *runs, uint64_t run_cap,
                      uint64_t *n_runs);

/* ---- Teloscope::scanSegment, batched (src/teloscope.cpp:537-658).  out[i] receives the
 *      SegmentData of segs[i]; a one-element call equals one scanSegment() call.  Host
 *      buffers in, host results out (H2D, kernels, D2H and host block calling inside). */
int  ts_scan_segments(ts_ctx *ctx, const ts_segment_in *segs, size_t n_segs, ts_segment_out *out);
void ts_free_segments(ts_segment_out *out, size_t n_segs);

/* ---- ReadTelomereFilter (src/read-filter.cpp:10-45).  ts_create_read_filter applies
 *      makeReadFilterInput's overrides (min_block_len 42 unless min_block_len_set,
 *      terminal_limit UINT32_MAX/2, all genome-wide outputs off); ts_filter_reads is
 *      matches() over a batch: trailing '\r' stripped, case folded, pass[i] = 1 iff the
 *      read has a terminal block. */
ts_ctx *ts_create_read_filter(const ts_params *params, int min_block_len_set,
                              const ts_pattern *patterns, size_t n_patterns);
int     ts_filter_reads(ts_ctx *ctx, const char *const *seqs, const uint64_t *lens,
                        size_t n_reads, uint8_t *pass);

/* ---- Teloscope::labelTerminalBlocks (src/teloscope.cpp:259-383): the step walkPath runs
 *      on the concatenated terminal blocks of a path (src/input.cpp:1031).  Sorts blocks in
 *      place, sets is_longest, writes the granular label (needs 2*n+1 bytes). */
int ts_label_terminal_blocks(ts_block *blocks, size_t n, uint16_t gaps, uint64_t path_size,
                             uint32_t terminal_limit, char *label_out, int *scaffold_type_out);

/* ---- window float metrics (include/teloscope.h:199-214), evaluated on the host from
 *      the integer counts exactly as the reference does. */
float ts_gc_content(const uint32_t counts[4], uint32_t window_size);
float ts_shannon_entropy(const uint32_t counts[4], uint32_t window_size);

/* ts_scan_segments for callers that do not read the match vectors (every reference run without -m:
 * writeBEDFile reads windows, blocks and canonicalMatches.size() only, src/teloscope.cpp:700-868).
 * Scan, terminal/interstitial block calling and the counts below all happen on the device; out[i] is
 * what ts_scan_segments returns with matches == NULL / n_matches == 0, and counts[i] (optional, may be
 * NULL) carries the sizes the match vectors would have had.  Free out with ts_free_segments(). */
typedef struct ts_segment_counts {
    uint64_t n_windows;         /* = windows.size() (0 for a tips-only segment) */
    uint64_t n_matches;         /* full scan: allMatches.size(); tips-only: fwdMatches.size() + revMatches.size() */
    uint64_t n_canonical;       /* canonicalMatches.size() of a full scan */
    uint64_t n_forward;         /* fwdMatches.size() */
} ts_segment_counts;
int ts_scan_segments_blocks(ts_ctx *ctx, const ts_segment_in *segs, size_t n_segs, ts_segment_out *out,
                            ts_segment_counts *counts);

/* ---- device-resident batches: the same scan with inputs and outputs kept in HBM.
 *      Used by bench.py and the multi-GPU driver; ts_scan_segments is built on it. ----- */
typedef struct ts_batch_info {
    uint64_t n_segments;
    uint64_t total_bases;
    uint64_t input_bytes;       /* size of the device input buffer the batch expects */
    uint64_t n_windows;         /* window records produced */
    uint64_t n_tiles;
    uint64_t match_capacity;    /* match records the device buffer can hold */
    uint64_t n_matches;         /* valid after ts_batch_sync() */
    uint64_t algorithmic_bytes; /* window scan: 1 B/base + 32 B/window + 4 B/match (after sync); tips-only / read batch:
                                   1 B per scanned base + 1 bit per segment (its match stream is an intermediate) */
    double   last_kernel_ms;    /* HIP-event time of the last scan (after sync) */
    double   avg_kernel_ms;     /* mean HIP-event time of the scans enqueued since the previous sync
                                   (the latest 64 at most) */
    uint64_t kernel_launches;   /* how many scans avg_kernel_ms averages over */
} ts_batch_info;

/* Plans a batch of n_segs segments of the given lengths (all tips_only or all full scan).
 * match_capacity 0 = default (bases/4). */
ts_batch *ts_batch_create(ts_ctx *ctx, const uint64_t *seg_lens, const uint64_t *abs_pos,
                          size_t n_segs, int tips_only, uint64_t match_capacity);
void      ts_batch_destroy(ts_batch *b);
/* Byte offset of segment i inside the device input buffer (16-byte aligned). */
uint64_t  ts_batch_segment_offset(const ts_batch *b, size_t i);
/* Device pointer of the batch's own input buffer (input_bytes long). */
void     *ts_batch_input_ptr(ts_batch *b);
/* Copies one segment's bases host -> device input buffer. */
int       ts_batch_upload(ts_batch *b, size_t i, const char *seq);
/* Enqueues the scan on `stream` (a hipStream_t, NULL = default stream), reading bases from
 * d_input (NULL = the batch's own input buffer). Asynchronous. */
int       ts_batch_scan(ts_batch *b, const void *d_input, void *stream);
/* Makes `stream` wait for the batch's last scan (no-op when it is the scan's own stream): a stream wait on the event the library
 * records behind every scan anyway — a caller that records an event of its own behind the scan puts a second packet on the scan's
 * queue, 0.011-0.013 ms per scan that the next scan starts later (profiles/r05/shard_step_queues.txt).  What runs beside the next
 * scan (ts_batch_pack_shard, ts_batch_read_pass on a stream of their own) is ordered this way.  The reference orders the same two
 * steps by program order inside one job (scanSegment: scan, then block calling, /root/reference/src/teloscope.cpp:600-657). */
int       ts_batch_wait_scan(ts_batch *b, void *stream);
/* Width of the match records the batch's scans leave in their per-wave regions: 16 bits each — the kernel's own staged entries,
 * (tile-relative position << 2) | forward << 1 | canonical with positions below 2^14 — wherever the batch's tiles are that short
 * (every geometry the planner picks for windows up to 8 192 bases, and read batches), else 32.  Every reader inside the library
 * knows both (block calling, the read predicate, pack; export, downloads and the dense stream widen to 32 bits, so nothing that
 * leaves the device changes); the records are a tenth of the bytes a scan moves through HBM.  ts_batch_set_record_bits(b, 32)
 * before the first scan keeps 32-bit regions — what a caller that reads them raw through ts_batch_matches_ptr needs (that
 * pointer is NULL for 16-bit regions); 16 asks for 16 (TS_ERR_UNSUPPORTED when the tiles are too long).  TS_REC32=1 (read when the
 * context is made) makes 32 the default.  The reference has no such stream at all (it walks SegmentData in place,
 * /root/reference/src/teloscope.cpp:600-657, src/read-filter.cpp:10-45). */
int       ts_batch_set_record_bits(ts_batch *b, int bits);
/* Which of the batch's scans are timed: every `every`-th (1, the default: all; 0: none).  A timed scan has an event recorded in
 * front of it as well as the one behind every scan; ts_batch_info's kernel times are means over the timed scans since the last
 * ts_batch_sync.  An event is a packet on the scan's queue: a caller that enqueues scans back to back and does not need every
 * one's time (a rank's steps) saves 0.005 ms per untimed scan. */
int       ts_batch_set_timing(ts_batch *b, uint32_t every);
/* Waits for the last scan, reads back counters; grows the match buffer and rescans if it
 * overflowed. */
int       ts_batch_sync(ts_batch *b);
int       ts_batch_get_info(const ts_batch *b, ts_batch_info *info);
/* Device pointers to the raw result buffers (valid after sync): window records
 * (8 x uint32 each: A,C,G,T, canonical/nonCanonical/fwd/rev covered) and the stream of
 * packed 32-bit match records ((tile-relative position << 2) | fwd << 1 | canonical).  Records
 * of one tile are contiguous and position-ordered; tiles are placed in completion order and
 * located through the batch's tile directory (ts_batch_download does that).  ts_batch_matches_ptr is NULL
 * while the regions hold 16-bit records (ts_batch_set_record_bits). */
const void *ts_batch_windows_ptr(const ts_batch *b);
const void *ts_batch_matches_ptr(const ts_batch *b);
/* D2H + host post-processing (absolute positions, terminal flags, block calling):
 * fills out[0..n_segments). Free with ts_free_segments(). */
int       ts_batch_download(ts_batch *b, const char *const *host_seqs, ts_segment_out *out);
/* Block calling ON THE DEVICE (getTerminalBlocks / getInterstitialBlocks, src/teloscope.cpp:29-256)
 * from the resident match stream, for a synced batch: fills out[i].terminal_blocks /
 * interstitial_blocks (and windows for a full scan) exactly as ts_batch_download would, but
 * leaves out[i].matches empty — only the blocks (a few per segment) cross PCIe.  Free with
 * ts_free_segments(). */
int       ts_batch_download_blocks(ts_batch *b, ts_segment_out *out);
/* Per-segment summary (n_windows, n_matches, n_canonical, n_forward as 4 x uint64 per
 * segment) written to a device buffer of 32*n_segments bytes: the "hit buffer" ranks
 * gather over RCCL. */
int       ts_batch_segment_summary(ts_batch *b, void *d_out, void *stream);


/* ---- the tile directory and tile-range shards: multi-GPU form of the scan.
 *      The reference runs one thread-pool job per path (src/input.cpp:719-724) and merges the jobs' PathData
 *      in seqPos order (sortBySeqPos, include/teloscope.h:262-266).  Here the unit of work is a TILE of the
 *      batch's plan (a run of consecutive windows of one segment, plus its w-s halo); ranks scan disjoint tile
 *      ranges of the SAME plan and one rank adopts all their results, after which it holds exactly what a
 *      single-GPU scan of the whole batch holds.  Nothing here communicates: the exchange itself
 *      (RCCL through torch.distributed) belongs to the caller, teloscope_amd/distributed.py. ---------- */
typedef struct ts_tile_info {
    uint64_t seg_index;        /* segment the tile belongs to */
    uint64_t seg_offset;       /* segment-relative position of the tile's first owned base; the position field
                                  of a packed match record is relative to it */
    uint64_t first_window;     /* index of the tile's first window record in the batch's window array */
    uint32_t n_windows;        /* window records the tile owns (0 in a tips-only batch) */
    uint32_t owned_bases;
} ts_tile_info;
/* Fills out[0..n) with tiles first .. first+n-1 of the plan (host data; works on a planning-only context). */
int ts_batch_get_tiles(const ts_batch *b, uint64_t first, uint64_t n, ts_tile_info *out);

typedef struct ts_range_info {
    uint64_t tile_begin, tile_end;
    uint64_t window_begin, window_end;   /* window records the range owns */
    uint64_t input_begin, input_end;     /* bytes of the batch's input layout the range reads (halo and the kernel's
                                            over-read slack included; bytes past a segment's data may hold anything) */
    uint64_t bases;                      /* owned bases */
} ts_range_info;
int ts_batch_range_info(const ts_batch *b, uint64_t tile_begin, uint64_t tile_end, ts_range_info *out);
/* Deterministic split of the plan into n_parts consecutive tile ranges of equal owned bases (+-1 tile);
 * part p owns [*tile_begin, *tile_end).  Consecutive ranges make the gather a concatenation: window records,
 * the tile directory and the tile-ordered record stream of part p follow those of part p-1. */
int ts_batch_partition(const ts_batch *b, uint32_t n_parts, uint32_t part, uint64_t *tile_begin, uint64_t *tile_end);
/* Before the first scan: the batch will execute only tiles [tile_begin, tile_end) — device buffers are sized for
 * the range, ts_batch_scan expects d_input to point at byte input_begin of the input layout
 * (ts_batch_range_info), and ts_batch_upload copies only the part of a segment the range reads. */
int ts_batch_restrict(ts_batch *b, uint64_t tile_begin, uint64_t tile_end);
/* on != 0: the batch's scans (window scans only; a tips-only batch ignores it) also leave, in HBM, what the calls that
 * follow a scan would otherwise re-read the whole match stream for: the records a writer reads — canonicalMatches and
 * the terminal nonCanonicalMatches, /root/reference/src/teloscope.cpp:486-496 — as a second, sparse output, and per tile
 * a summary of its chains of matches (getInterstitialBlocks' grouping, src/teloscope.cpp:235-253).  Device block
 * calling (ts_batch_download*, the host entry points) then looks only at the tiles that can hold an interstitial block,
 * and ts_batch_pack_shard copies the visible records instead of filtering 4 bytes per match.  It costs the scan kernel
 * about a tenth of its time, so it is OFF for a batch that is only scanned (results left in HBM) and switched ON by
 * ts_batch_restrict_shard and by the host entry points, whose downloads always call blocks; results are identical
 * either way.  May be changed between scans; the fourth word of a tile's directory entry is its visible-record count
 * while on. */
int ts_batch_set_emit(ts_batch *b, int on);
/* Caller-owned result buffers (device): the scan writes the range's window records (32 B each, from
 * window_begin) to d_windows and its tile directory entries {matches, canonical, forward, 0 or visible} x uint32 (16 B per
 * tile, from tile_begin) to d_tile_stats.  Either may be NULL = the batch's own buffer.  Before the first scan. */
int ts_batch_bind_results(ts_batch *b, void *d_windows, void *d_tile_stats);
/* After a scan (asynchronous on `stream`): packs the range's match records into ONE stream in tile order
 * (= position order within each segment) at d_dense (capacity in records) and writes to d_total (2 x uint64 on
 * the device) {records the range produced, 1 if the stream is incomplete: a wave's region overflowed — call
 * ts_batch_sync, which grows it and rescans — or dense_capacity was too small}. */
int ts_batch_export(ts_batch *b, void *d_dense, uint64_t dense_capacity, void *d_total, void *stream);
/* On a whole (unrestricted) batch: take results produced elsewhere — window records, tile directory entries and
 * the tile-ordered record stream of ALL tiles, i.e. the concatenation over the parts of what ts_batch_bind_results
 * / ts_batch_export delivered — as this batch's results.  The buffers stay owned by the caller and must outlive
 * the batch's use.  Afterwards ts_batch_download, ts_batch_download_blocks and ts_batch_segment_summary work as
 * after a local scan + sync. */
int ts_batch_adopt(ts_batch *b, void *d_windows, void *d_tile_stats, const void *d_dense, uint64_t n_matches,
                   void *stream);
/* The exchange's wire format.  Every u32 of the three result arrays fits 16 bits when ts_batch_wire16_ok() says so
 * (packed match records always do: position < 2^14 plus two flag bits; tile counts too; window fields when
 * pattern length x window <= 65535), so ranks may send them as u16 — half the bytes over the link-bound gather — and
 * the destination widens them where they land: dst[i] = src[i] for n values, asynchronously on `stream`
 * (device pointers; ctx names the device). */
int ts_batch_wire16_ok(const ts_batch *b);
int ts_wire_widen_u16(ts_ctx *ctx, const void *d_src_u16, void *d_dst_u32, uint64_t n, void *stream);
/* Device pointer of the batch's tile directory entries (16 B per tile of the range). */
const void *ts_batch_tile_stats_ptr(const ts_batch *b);

/* ---- shard results: several devices share ONE scan and hand over what the reference's writers read ----------------
 *      The reference fans paths out to its thread-pool workers and merges their PathData in seqPos order
 *      (src/input.cpp:719-733, sortBySeqPos include/teloscope.h:262-266).  Of a path's results its writers read the
 *      windows, the blocks, canonicalMatches and the terminal nonCanonicalMatches (src/teloscope.cpp:486-496, :700-868;
 *      walkPath moves only those two match vectors into PathData, src/input.cpp:1000-1009) — about 3 % of the match
 *      records; allMatches / fwdMatches / revMatches feed block calling (:646-655) and nothing else.  A SHARD therefore
 *      calls its blocks on its own device and packs ONE message — bit-packed window records, the visible match records
 *      (16 bits each), its blocks — which is all that crosses xGMI (rank form, teloscope_amd/distributed.py) or PCIe
 *      (ts_scan_segments_multi): ~60 MB per 3 Gb scan at 8 devices instead of the 245 MB of the full exchange above.
 *
 *      A shard OWNS the p-th of n consecutive tile ranges (ts_batch_partition: equal bases; a boundary inside a segment
 *      keeps at least terminal zone + context tiles from both its ends) and also scans CONTEXT tiles either side where
 *      a segment continues on a neighbour, so that a chain of matches that starts in an owned tile can be followed.
 *      Terminal blocks belong to the shard that owns that end of the segment, an interstitial block to the shard that
 *      owns the tile it starts in.  What the shards assume about each other (the bounds of the interstitial search) is
 *      checked by ts_shards_finalize; when it does not hold — a telomere that reaches beyond the context tiles —
 *      ts_shards_finalize returns TS_SHARD_NEED_FULL and the caller takes the full path for that batch. ------------- */
typedef struct ts_shard_info {
    uint32_t n_parts, part;
    uint64_t own_begin, own_end;         /* tiles the shard owns */
    uint64_t ext_begin, ext_end;         /* tiles it scans: owned + context */
    uint64_t window_begin, window_end;   /* window records it owns */
    uint64_t input_begin, input_end;     /* bytes of the batch's input layout its scan reads (ts_batch_scan expects d_input
                                            to point at byte input_begin) */
    uint64_t bases;                      /* owned bases */
    uint64_t seg_begin, seg_end;         /* segments with an owned tile */
    uint64_t msg_bytes;                  /* size of its result message at this capacity scale */
    uint64_t visible_capacity;           /* visible match records / blocks the message has room for */
    uint32_t block_capacity;
    uint32_t window_bytes;               /* bytes per packed window record */
    uint32_t visible_bytes;              /* bytes per visible match record (2, or 4 when a tile-relative position needs more) */
    uint32_t context_tiles;
} ts_shard_info;
/* Host only (works on a planning-only context): sender and receiver compute the same numbers.  `scale` >= 1 multiplies
 * the capacities of the message's variable sections (visible records, blocks). */
int ts_batch_shard_info(const ts_batch *b, uint32_t n_parts, uint32_t part, uint32_t scale, ts_shard_info *out);
/* Before the first scan: the batch becomes shard `part` of `n_parts` (ts_batch_restrict to its scanned range). */
int ts_batch_restrict_shard(ts_batch *b, uint32_t n_parts, uint32_t part, uint32_t scale);
int ts_batch_set_shard_scale(ts_batch *b, uint32_t scale);
/* After ts_batch_scan, asynchronous on `stream`, no host synchronisation: block calling on the device and the packed
 * message at d_msg (device memory, msg_bytes >= ts_shard_info.msg_bytes). */
int ts_batch_pack_shard(ts_batch *b, void *d_msg, uint64_t msg_bytes, void *stream);
/* Optional, after ts_batch_restrict_shard: the message buffer (device memory, msg_bytes >= ts_shard_info.msg_bytes) the
 * shard's scans will be packed into.  Knowing it, an emitting scan writes the records of the windows it owns straight into the
 * message's window section in their bit-packed form (7 fields x bit_width(window) bits: what a writer reads of a WindowData,
 * /root/reference/src/teloscope.cpp:785-812, /root/reference/include/teloscope.h:119-137) instead of 8 x uint32 per window
 * that ts_batch_pack_shard then packs in a pass of its own: the pack has one kernel fewer and the scan stores a quarter of
 * the bytes.  The 8 x uint32 records of such a scan are NOT produced (ts_batch_windows_ptr / ts_batch_download see none), and
 * ts_batch_pack_shard must be given the same buffer (TS_ERR_STATE otherwise).  The message's bytes do not depend on it.
 * d_msg == NULL unbinds; ts_batch_set_shard_scale unbinds (the message's size changes with the scale). */
int ts_batch_bind_shard_message(ts_batch *b, void *d_msg, uint64_t msg_bytes);
#define TS_SHARD_OVERFLOW_VISIBLE 0x1u   /* ts_shard_status.flags */
#define TS_SHARD_OVERFLOW_BLOCKS  0x2u
#define TS_SHARD_OVERFLOW_SCAN    0x4u   /* a wave's record region overflowed in the scan: ts_batch_sync, then pack again */
#define TS_SHARD_OUT_OF_CONTEXT   0x8u   /* a chain or a terminal walk ran out of the context tiles */
typedef struct ts_shard_status {
    uint32_t part, n_parts, flags, n_blocks;
    uint64_t n_visible, visible_capacity;
    uint32_t block_capacity;
    uint32_t scale_factor_needed;        /* 1 when everything fitted, else the factor by which to raise the scale */
    uint64_t msg_bytes;
} ts_shard_status;
/* Reads a message's header (host memory). */
int ts_shard_peek(const void *msg, uint64_t msg_bytes, ts_shard_status *out);
/* positive returns of ts_shards_finalize: the messages could not be turned into results as they are */
#define TS_SHARD_RETRY_SYNC 1            /* a shard's scan overflowed: ts_batch_sync on it, pack again */
#define TS_SHARD_RETRY_GROW 2            /* a message overflowed: pack again with a larger scale (ts_shard_peek says which) */
#define TS_SHARD_NEED_FULL  3            /* the shards' assumptions about each other do not hold for this input */
/* Host: the messages of all n_parts shards of `plan` (a whole batch of the same plan; may sit on a planning-only
 * context) -> out[i] = SegmentData of segment i with windows, blocks and, as `matches`, the VISIBLE records only
 * (canonicalMatches + terminal nonCanonicalMatches, position order); counts[i] (optional) = the sizes the match
 * vectors had on the devices.  Free out with ts_free_segments(). */
int ts_shards_finalize(const ts_batch *plan, const void *const *msgs, const uint64_t *msg_bytes, uint32_t n_parts,
                       ts_segment_out *out, ts_segment_counts *counts);

/* ---- Teloscope::scanSegment over several devices (one host thread per context, as ts_filter_reads_multi): the batch's
 *      plan is split into one shard per context; every device uploads the bases its shard reads and downloads its
 *      shard's message over its OWN PCIe link; the host merges (ts_shards_finalize).  Replaces the reference's one job
 *      per path + merge under a mutex (src/input.cpp:719-733, :1036-1037).  out[i].matches holds the VISIBLE records
 *      only (see above); counts may be NULL.  Contexts must have been created with the same parameters and patterns and
 *      may share a device.  Parameter sets outside the tiled kernel have no shard results: their segments are dealt WHOLE to the
 *      contexts, in consecutive runs of equal bases (the reference's one job per path); inputs for which the shards' assumptions
 *      fail run on ctxs[0] alone (same results). */
int ts_scan_segments_multi(ts_ctx *const *ctxs, size_t n_ctx, const ts_segment_in *segs, size_t n_segs,
                           ts_segment_out *out, ts_segment_counts *counts);

/* ---- ReadTelomereFilter over several devices (the read shard of --fastq-subset / --bam-subset:
 *      the reference deals a batch's reads to its thread-pool workers in chunks and writes the chunk outputs in
 *      chunk order, src/input.cpp:753-812).  ctxs[0..n_ctx) are read-filter contexts (ts_create_read_filter),
 *      normally one per GPU; the batch is cut into n_ctx consecutive shards of equal bases, each filtered by one
 *      host thread on its context, and pass[] comes back in input order.  Contexts may share a device. */
/* ReadTelomereFilter::matches for reads that already are in HBM: on a tips-only batch made from a read-filter
 * context and scanned on `stream` (ts_batch_scan), writes pass[i] (one byte per read, device memory) asynchronously
 * on the same stream; the match stream never leaves the device. */
int ts_batch_read_pass(ts_batch *b, void *d_pass, void *stream);
/* The scan may have overflowed a wave's record region (tiles are taken on demand, so the per-wave fill differs from launch
 * to launch: a launch that fitted, even one ts_batch_sync vouched for, says nothing about the next).  ts_batch_read_pass
 * then judges NOTHING — d_pass keeps what it held — and raises a flag on the device that stays up until it is read here:
 * *overflowed = 1 if any pass since the last call was skipped for that reason (then: ts_batch_sync, which regrows and
 * rescans, and ts_batch_read_pass again).  Waits for the device. */
int ts_batch_read_pass_status(ts_batch *b, int *overflowed);
int ts_filter_reads_multi(ts_ctx *const *ctxs, size_t n_ctx, const char *const *seqs, const uint64_t *lens,
                          size_t n_reads, uint8_t *pass);

/* ---- the rank form's ONE exchange for a C++ host (one process per GPU): every rank's shard message
 *      (ts_batch_pack_shard) to rank `dst` in one grouped send / recv over RCCL — xGMI between the GPUs of a node.  The
 *      reference merges its per-path results in process, under a mutex (src/input.cpp:719-733, sortBySeqPos
 *      include/teloscope.h:262-266); this is that merge's transport when the paths were scanned by other processes.
 *      librccl is opened at run time: on a host without it these four calls fail with TS_ERR_UNSUPPORTED and nothing else
 *      changes.  Message sizes come from ts_batch_shard_info on both sides, so a step needs no size exchange and no host
 *      synchronisation.  (teloscope_amd/distributed.py's ShardExchange is the same exchange through torch.distributed.) */
typedef struct ts_exchange ts_exchange;
#define TS_EXCHANGE_ID_BYTES 128
/* One rank makes the id (ncclGetUniqueId) and hands its 128 bytes to the others by whatever the host has: MPI_Bcast, a
 * socket, a file.  On failure ts_exchange_last_error() says why. */
int          ts_exchange_unique_id(void *id_out);
const char  *ts_exchange_last_error(void);
/* Collective over the n_ranks processes (ncclCommInitRank on the context's device); NULL on failure (ts_last_error(ctx)). */
ts_exchange *ts_exchange_create(ts_ctx *ctx, const void *id, int rank, int n_ranks);
void         ts_exchange_destroy(ts_exchange *x);
/* Asynchronous on `stream` (the one the message was packed on, or one that waits for it).  A rank other than dst sends
 * d_msg[0, my_bytes); dst receives rank p's message into d_recv[p] (device memory, msg_bytes[p] bytes) for every p != dst —
 * its own message stays where it is (d_recv[dst] may be NULL; when it is given and differs from d_msg the message also goes
 * through RCCL to itself: the loop-back a one-GPU box rehearses the pattern with).  d_recv / msg_bytes are read on dst only. */
int          ts_exchange_gather(ts_exchange *x, int dst, const void *d_msg, uint64_t my_bytes, void *const *d_recv,
                                const uint64_t *msg_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TELOSCAN_H */
