/*
 * teloscope_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * A plain-C restatement of the reference's telomeric-motif scan path, written
 * from a reading of the reference sources (cited per function in the .c file).
 * It exists only so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg can check / time-compare the HIP product path against it.
 * Nothing under teloscope_amd/ may include, link, load or call this code.
 *
 * Pinning: the real reference cannot be built in this pipeline (its gfalibs
 * submodule is absent and writing stand-in headers is not allowed), so this
 * oracle is pinned by the reference's own fixtures: the validateFiles .tst manifests
 * (stdout of 140+ FASTA runs), testFiles/expected .fq files (read filter) and the
 * known-answer cases in scripts/test_bam_subset.py — see tests/test_oracle_*.py.
 */
#ifndef TELOSCOPE_ORACLE_H
#define TELOSCOPE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Mirrors the fields of UserInputTeloscope (include/input.h:15-64) that the
 * scan path reads. */
typedef struct tso_params {
    uint32_t window_size;       /* -w, default 1000 */
    uint32_t step;              /* -s, default 1000 */
    uint32_t terminal_limit;    /* -t, default 50000 */
    uint16_t max_match_dist;    /* -k, default 50 */
    uint16_t min_block_len;     /* -l, default 300 (42 for reads) */
    uint16_t max_block_dist;    /* -d, default 500 */
    uint16_t min_block_counts;  /* default 2 */
    float    min_block_density; /* -y, default 0.5 */
    uint16_t canonical_size;    /* strlen(-c) */
    uint8_t  out_gc;            /* -g */
    uint8_t  out_entropy;       /* -e */
    uint8_t  out_matches;       /* -m (matchSeq is seq.substr; not stored here) */
    uint8_t  reserved;
} tso_params;

typedef struct tso_pattern {
    char    seq[64];
    uint8_t len;
    uint8_t is_forward;
    uint8_t is_canonical;
    uint8_t ambiguous;   /* same string produced with both orientations (the
                            reference's result then depends on std::sort) */
} tso_pattern;

/* MatchInfo (include/teloscope.h:89-95) without matchSeq. */
typedef struct tso_match {
    uint64_t position;
    uint16_t match_size;
    uint8_t  is_forward;
    uint8_t  is_canonical;
    uint32_t pad;
} tso_match;

/* WindowData (include/teloscope.h:119-137). */
typedef struct tso_window {
    uint64_t window_start;
    uint32_t current_window_size;
    uint32_t nucleotide_counts[4];
    float    gc_content;
    float    shannon_entropy;
    uint16_t canonical_counts, non_canonical_counts, fwd_counts, rev_counts;
    uint32_t canonical_covered, non_canonical_covered, fwd_covered, rev_covered;
    uint8_t  has_can_dimer;
    uint8_t  pad[3];
} tso_window;

/* TelomereBlock (include/teloscope.h:103-117). */
typedef struct tso_block {
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
    uint8_t  pad;
} tso_block;

/* SegmentData (include/teloscope.h:139-148). */
typedef struct tso_segment {
    tso_window *windows;            size_t n_windows;
    tso_block  *terminal_blocks;    size_t n_terminal_blocks;
    tso_block  *interstitial_blocks;size_t n_interstitial_blocks;
    tso_match  *canonical_matches;     size_t n_canonical_matches;
    tso_match  *non_canonical_matches; size_t n_non_canonical_matches;
    tso_match  *fwd_matches;           size_t n_fwd_matches;
    tso_match  *rev_matches;           size_t n_rev_matches;
    tso_match  *all_matches;           size_t n_all_matches;
} tso_segment;

typedef struct tso_ctx tso_ctx;

/* revCom (gfalibs; behaviour pinned by every .tst p/q label). */
void tso_revcom(const char *in, char *out);

/* expandPatternsWithOrientation (src/tools.cpp:201-283). raw_csv is the
 * comma-separated, already upper-cased -p list. Returns malloc'd array. */
tso_pattern *tso_expand_patterns(const char *raw_csv, int edit_distance,
                                 const char *canonical_fwd, size_t *n_out);

/* Teloscope ctor (include/teloscope.h:241-247): builds the trie. */
tso_ctx *tso_create(const tso_params *p, const tso_pattern *pats, size_t npat);
void     tso_destroy(tso_ctx *c);
uint16_t tso_longest_pattern(const tso_ctx *c);

/* Teloscope::scanSegment (src/teloscope.cpp:537-658). */
int  tso_scan_segment(const tso_ctx *c, const char *seq, uint64_t n,
                      uint64_t abs_pos, int tips_only, tso_segment *out);
void tso_free_segment(tso_segment *s);

/* Teloscope::labelTerminalBlocks (src/teloscope.cpp:259-383). blocks is
 * sorted in place; label_out needs 2*n+1 bytes; returns ScaffoldType value
 * (include/tools.h:13-19). */
int  tso_label_terminal_blocks(tso_block *blocks, size_t n, uint16_t gaps,
                               uint64_t path_size, uint32_t terminal_limit,
                               char *label_out);

/* ReadTelomereFilter::matches (src/read-filter.cpp:10-45). The ctx must have
 * been created from tso_read_filter_params(). */
void tso_read_filter_params(const tso_params *in, int min_block_len_set, tso_params *out);
int  tso_read_filter_matches(const tso_ctx *c, const char *seq, uint64_t n);

/* getGCContent / getShannonEntropy (include/teloscope.h:199-214). */
float tso_gc_content(const uint32_t counts[4], uint32_t window_size);
float tso_shannon_entropy(const uint32_t counts[4], uint32_t window_size);

/* cpu_baseline helper for bench.py: scans [seq, seq+n) once as one segment in
 * full-window mode and returns checksums so the work cannot be elided. */
uint64_t tso_bench_scan(const tso_ctx *c, const char *seq, uint64_t n,
                        uint64_t *n_windows, uint64_t *n_matches);

#ifdef __cplusplus
}
#endif
#endif
