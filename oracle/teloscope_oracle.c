/*
 * teloscope_oracle.c — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 *
 * Plain-C restatement of the reference's scan path.  Every function names the
 * reference file:line whose behaviour it follows.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the
 * product (teloscope_amd/) never links or calls anything in this directory.
 *
 * It deliberately keeps the reference's mechanics (4-ary trie walk per base,
 * per-window carry of the overlap region, uint32 wrap-around in startIndex,
 * libstdc++-style lower_bound) rather than the closed forms the HIP kernels
 * use, so that agreement between the two is evidence, not tautology.
 *
 * Parity pinning: see the header and tests/test_oracle_*.py.
 */
#include "teloscope_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ utils */

typedef struct { void *p; size_t n, cap, esz; } vec;

static void vec_init(vec *v, size_t esz) { v->p = NULL; v->n = 0; v->cap = 0; v->esz = esz; }

static void *vec_push(vec *v) {
    if (v->n == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 64;
        void *np = realloc(v->p, nc * v->esz);
        if (!np) abort();
        v->p = np; v->cap = nc;
    }
    return (char *)v->p + (v->n++) * v->esz;
}

/* Trie::charToIndex (include/teloscope.h:27-35): A C G T -> 0..3, else -1 */
static int base_index(char c) {
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default:  return -1;
    }
}

/* gfalibs revCom: reverse + A<->T, C<->G (either case); other symbols kept. */
void tso_revcom(const char *in, char *out) {
    size_t n = strlen(in);
    for (size_t i = 0; i < n; ++i) {
        char c = in[n - 1 - i], r;
        switch (c) {
            case 'A': r = 'T'; break; case 'T': r = 'A'; break;
            case 'C': r = 'G'; break; case 'G': r = 'C'; break;
            case 'a': r = 't'; break; case 't': r = 'a'; break;
            case 'c': r = 'g'; break; case 'g': r = 'c'; break;
            default:  r = c;  break;
        }
        out[i] = r;
    }
    out[n] = '\0';
}

/* --------------------------------------------------------- pattern expansion */

/* IUPAC table, src/tools.cpp:54-70 (same member order). */
static const char *iupac_members(char c) {
    switch (c) {
        case 'A': return "A";   case 'C': return "C";
        case 'G': return "G";   case 'T': return "T";
        case 'R': return "AG";  case 'Y': return "CT";
        case 'M': return "AC";  case 'K': return "GT";
        case 'S': return "CG";  case 'W': return "AT";
        case 'H': return "ACT"; case 'B': return "CGT";
        case 'V': return "ACG"; case 'D': return "AGT";
        case 'N': return "ACGT";
        default:  return "";    /* unknown symbol: no combination survives */
    }
}

typedef struct { char s[64]; uint8_t fwd; } pat_tmp;

/* getCombinations, src/tools.cpp:73-85 */
static void combos_rec(const char *pattern, size_t len, char *cur, size_t idx, vec *out) {
    if (idx == len) {
        char *dst = (char *)vec_push(out);
        memcpy(dst, cur, len); dst[len] = '\0';
        return;
    }
    for (const char *m = iupac_members(pattern[idx]); *m; ++m) {
        cur[idx] = *m;
        combos_rec(pattern, len, cur, idx + 1, out);
    }
}

/* getEditVariants, src/tools.cpp:88-128: every single substitution; for
 * distance 2 the distance-1 list is re-expanded once more (duplicates kept). */
static void edit_variants(const char *pattern, int max_dist, vec *out /* char[64] */) {
    static const char nts[4] = {'A', 'C', 'G', 'T'};
    size_t len = strlen(pattern);
    if (max_dist == 0) return;
    size_t first = out->n;
    for (size_t i = 0; i < len; ++i) {
        char orig = pattern[i];
        if (orig != 'A' && orig != 'C' && orig != 'G' && orig != 'T') continue;
        for (int k = 0; k < 4; ++k) {
            if (nts[k] == orig) continue;
            char *dst = (char *)vec_push(out);
            memcpy(dst, pattern, len + 1);
            dst[i] = nts[k];
        }
    }
    if (max_dist >= 2) {
        size_t d1_end = out->n;
        for (size_t v = first; v < d1_end; ++v) {
            char tmp[64];
            memcpy(tmp, (char *)out->p + v * 64, 64);
            edit_variants(tmp, 1, out);
        }
    }
}

/* isCloserToFwd lambda, src/tools.cpp:206-247 */
static int closer_to_fwd(const char *pattern, const char *can_fwd) {
    char can_rev[64], pat_rev[64];
    tso_revcom(can_fwd, can_rev);
    size_t pl = strlen(pattern), cl = strlen(can_fwd);
    if (pl == cl) {
        unsigned df = 0, dr = 0;
        for (size_t i = 0; i < pl; ++i) {
            if (pattern[i] != can_fwd[i]) ++df;
            if (pattern[i] != can_rev[i]) ++dr;
        }
        return (uint8_t)df <= (uint8_t)dr;
    }
    tso_revcom(pattern, pat_rev);
    const char *shorter = (pl < cl) ? pattern : can_fwd;
    const char *longer  = (pl < cl) ? can_fwd : pattern;
    const char *longer_rev = (pl < cl) ? can_rev : pat_rev;
    size_t sl = strlen(shorter), ll = strlen(longer);
    uint8_t best_f = 255, best_r = 255;
    for (size_t off = 0; off + sl <= ll; ++off) {
        uint8_t d = 0;
        for (size_t i = 0; i < sl; ++i) if (shorter[i] != longer[off + i]) ++d;
        if (d < best_f) best_f = d;
    }
    for (size_t off = 0; off + sl <= ll; ++off) {
        uint8_t d = 0;
        for (size_t i = 0; i < sl; ++i) if (shorter[i] != longer_rev[off + i]) ++d;
        if (d < best_r) best_r = d;
    }
    return best_f <= best_r;
}

/* stable merge sort on the string key (the reference uses std::sort, whose
 * order among EQUAL strings is unspecified; see `ambiguous`). */
static void msort(pat_tmp *a, pat_tmp *tmp, size_t n) {
    if (n < 2) return;
    size_t h = n / 2;
    msort(a, tmp, h); msort(a + h, tmp, n - h);
    size_t i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = (strcmp(a[j].s, a[i].s) < 0) ? a[j++] : a[i++];
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, n * sizeof(pat_tmp));
}

/* expandPatternsWithOrientation, src/tools.cpp:201-283 */
tso_pattern *tso_expand_patterns(const char *raw_csv, int edit_distance,
                                 const char *canonical_fwd, size_t *n_out) {
    vec all; vec_init(&all, sizeof(pat_tmp));
    char can_rev[64];
    tso_revcom(canonical_fwd, can_rev);

    const char *p = raw_csv;
    while (*p) {
        const char *q = strchr(p, ',');
        size_t len = q ? (size_t)(q - p) : strlen(p);
        if (len > 0 && len < 64) {
            char seed[64], cur[64];
            memcpy(seed, p, len); seed[len] = '\0';
            memcpy(cur, seed, len + 1);
            vec combos; vec_init(&combos, 64);
            combos_rec(seed, len, cur, 0, &combos);
            for (size_t ci = 0; ci < combos.n; ++ci) {
                const char *combo = (char *)combos.p + ci * 64;
                int seed_fwd = closer_to_fwd(combo, canonical_fwd);
                vec vars; vec_init(&vars, 64);
                memcpy(vec_push(&vars), combo, 64);
                if (edit_distance > 0) edit_variants(combo, edit_distance, &vars);
                for (size_t vi = 0; vi < vars.n; ++vi) {
                    const char *v = (char *)vars.p + vi * 64;
                    pat_tmp *a = (pat_tmp *)vec_push(&all);
                    memset(a, 0, sizeof *a); strcpy(a->s, v); a->fwd = (uint8_t)seed_fwd;
                    pat_tmp *b = (pat_tmp *)vec_push(&all);
                    memset(b, 0, sizeof *b); tso_revcom(v, b->s); b->fwd = (uint8_t)!seed_fwd;
                }
                free(vars.p);
            }
            free(combos.p);
        }
        if (!q) break;
        p = q + 1;
    }

    pat_tmp *arr = (pat_tmp *)all.p;
    size_t n = all.n;
    if (n > 1) {
        pat_tmp *tmp = (pat_tmp *)malloc(n * sizeof(pat_tmp));
        msort(arr, tmp, n);
        free(tmp);
    }
    tso_pattern *out = (tso_pattern *)calloc(n ? n : 1, sizeof(tso_pattern));
    size_t m = 0;
    for (size_t i = 0; i < n; ++i) {
        if (m > 0 && strcmp(out[m - 1].seq, arr[i].s) == 0) {
            if (out[m - 1].is_forward != arr[i].fwd) out[m - 1].ambiguous = 1;
            continue;
        }
        strcpy(out[m].seq, arr[i].s);
        out[m].len = (uint8_t)strlen(arr[i].s);
        out[m].is_forward = arr[i].fwd;
        /* Teloscope ctor, include/teloscope.h:243-244 */
        out[m].is_canonical = (strcmp(arr[i].s, canonical_fwd) == 0 ||
                               strcmp(arr[i].s, can_rev) == 0);
        ++m;
    }
    free(arr);
    *n_out = m;
    return out;
}

/* -------------------------------------------------------------------- trie */

typedef struct { int32_t child[4]; uint8_t end, fwd, canon; } node;

struct tso_ctx {
    tso_params p;
    node *nodes; size_t n_nodes, cap_nodes;
    uint16_t longest;          /* Trie::longestPatternSize */
    uint16_t first_pattern_len;/* userInput.patterns.front().size() */
};

static int32_t new_node(tso_ctx *c) {
    if (c->n_nodes == c->cap_nodes) {
        c->cap_nodes = c->cap_nodes ? c->cap_nodes * 2 : 256;
        c->nodes = (node *)realloc(c->nodes, c->cap_nodes * sizeof(node));
        if (!c->nodes) abort();
    }
    node *nd = &c->nodes[c->n_nodes];
    nd->child[0] = nd->child[1] = nd->child[2] = nd->child[3] = -1;
    nd->end = nd->fwd = nd->canon = 0;
    return (int32_t)c->n_nodes++;
}

/* Trie::insertPattern, include/teloscope.h:40-57 */
static void trie_insert(tso_ctx *c, const tso_pattern *pt) {
    int32_t cur = 0;
    for (size_t i = 0; i < pt->len; ++i) {
        int idx = base_index(pt->seq[i]);
        if (idx < 0) continue;
        if (c->nodes[cur].child[idx] < 0) {
            int32_t nn = new_node(c);
            c->nodes[cur].child[idx] = nn;
        }
        cur = c->nodes[cur].child[idx];
    }
    c->nodes[cur].end = 1;
    c->nodes[cur].fwd = pt->is_forward;
    c->nodes[cur].canon = pt->is_canonical;
    if (pt->len > c->longest) c->longest = pt->len;
}

tso_ctx *tso_create(const tso_params *p, const tso_pattern *pats, size_t npat) {
    tso_ctx *c = (tso_ctx *)calloc(1, sizeof *c);
    c->p = *p;
    new_node(c);
    for (size_t i = 0; i < npat; ++i) trie_insert(c, &pats[i]);
    c->first_pattern_len = npat ? pats[0].len : 0;
    return c;
}

void tso_destroy(tso_ctx *c) { if (c) { free(c->nodes); free(c); } }
uint16_t tso_longest_pattern(const tso_ctx *c) { return c->longest; }

/* Trie::getChild */
static inline int32_t trie_child(const tso_ctx *c, int32_t nd, char ch) {
    int idx = base_index(ch);
    return idx >= 0 ? c->nodes[nd].child[idx] : -1;
}

/* --------------------------------------------------------- float metrics */

/* getGCContent, include/teloscope.h:211-214: float / uint32 in float, then
 * a double multiply by 100.0, narrowed on return. */
float tso_gc_content(const uint32_t counts[4], uint32_t window_size) {
    uint32_t gc = counts[1] + counts[2];
    float frac = (float)gc / (float)window_size;
    return (float)((double)frac * 100.0);
}

/* getShannonEntropy, include/teloscope.h:199-208: all float32, log2f, rounded
 * to three decimals with roundf. */
float tso_shannon_entropy(const uint32_t counts[4], uint32_t window_size) {
    float entropy = 0.0f;
    for (int i = 0; i < 4; ++i) {
        if (counts[i] > 0) {
            float prob = (float)counts[i] / (float)window_size;
            float term = prob * log2f(prob);
            entropy = entropy - term;
        }
    }
    float scaled = entropy * 1000.0f;
    return roundf(scaled) / 1000.0f;
}

/* ----------------------------------------------------------- block calling */

typedef struct {
    int in_block;
    uint64_t start, end, prev;
    uint32_t counts, fwd, canon, total_cov, fwd_cov, can_cov;
} chain;

static void chain_start(chain *b, const tso_match *m) {
    b->start = m->position;
    b->end = m->position + m->match_size;
    b->prev = m->position;
    b->counts = 1;
    b->fwd = m->is_forward;
    b->canon = m->is_canonical;
    b->total_cov = m->match_size;
    b->fwd_cov = (uint32_t)m->is_forward * m->match_size;
    b->can_cov = (uint32_t)m->is_canonical * m->match_size;
    b->in_block = 1;
}

static void chain_add(chain *b, const tso_match *m) {
    b->counts++;
    b->fwd += m->is_forward;
    b->canon += m->is_canonical;
    b->total_cov += m->match_size;
    b->fwd_cov += (uint32_t)m->is_forward * m->match_size;
    b->can_cov += (uint32_t)m->is_canonical * m->match_size;
    b->prev = m->position;
}

static void chain_to_block(const chain *b, tso_block *o) {
    memset(o, 0, sizeof *o);
    o->start = b->start;
    o->block_len = (uint32_t)(b->end - b->start);
    o->block_counts = b->counts;
    o->forward_count = b->fwd;
    o->reverse_count = b->counts - b->fwd;
    o->canonical_count = b->canon;
    o->non_canonical_count = b->counts - b->canon;
    o->total_covered = b->total_cov;
    o->fwd_covered = b->fwd_cov;
    o->can_covered = b->can_cov;
    o->has_valid_or = 1;
    o->is_longest = 0;
    o->block_label = '\0';
}

/* Teloscope::getTerminalBlocks, src/teloscope.cpp:29-176 */
static uint64_t terminal_blocks(const tso_ctx *c, const tso_match *ms, size_t nm,
                                vec *out, uint64_t seg_size, uint64_t abs_pos, int from_start) {
    const tso_params *P = &c->p;
    uint64_t boundary = from_start ? abs_pos : abs_pos + seg_size;
    int64_t n = (int64_t)nm;
    int64_t idx = from_start ? 0 : n - 1;
    int64_t end_idx = from_start ? n : -1;
    int64_t inc = from_start ? 1 : -1;

    vec subs; vec_init(&subs, sizeof(tso_block));
    chain b; memset(&b, 0, sizeof b);

#define IN_ZONE(pos) ( seg_size <= P->terminal_limit ? 1 :                         \
        ( from_start ? (((pos) - abs_pos) < P->terminal_limit)                     \
                     : (((pos) - abs_pos) >= seg_size - P->terminal_limit) ) )
#define FINALIZE_SUB() do {                                                        \
        if (b.counts >= P->min_block_counts && b.canon > 0 &&                      \
            (float)b.can_cov >= P->min_block_density * (float)(b.end - b.start)) { \
            chain_to_block(&b, (tso_block *)vec_push(&subs));                      \
        }                                                                          \
        b.in_block = 0;                                                            \
    } while (0)

    for (; idx != end_idx; idx += inc) {
        const tso_match *m = &ms[idx];
        if (!b.in_block) {
            if (IN_ZONE(m->position)) chain_start(&b, m);
            else break;
        } else {
            uint64_t gap = from_start ? (m->position - b.prev) : (b.prev - m->position);
            if (gap <= P->max_match_dist) {
                if (from_start) b.end = m->position + m->match_size;
                else            b.start = m->position;
                chain_add(&b, m);
            } else {
                FINALIZE_SUB();
                if (IN_ZONE(m->position)) chain_start(&b, m);
                else break;
            }
        }
    }
    if (b.in_block) FINALIZE_SUB();
#undef IN_ZONE
#undef FINALIZE_SUB

    if (subs.n == 0) { free(subs.p); return boundary; }

    /* phase 2: merge sub-blocks closer than -d */
    tso_block *sb = (tso_block *)subs.p;
    tso_block cur = sb[0];

#define FINALIZE_EXT() do {                                                        \
        if (cur.block_len >= P->min_block_len) {                                   \
            cur.block_label = from_start ? 'p' : 'q';                              \
            uint64_t rel_start = cur.start - abs_pos;                              \
            uint64_t rel_end = rel_start + cur.block_len;                          \
            uint64_t left = rel_start;                                             \
            uint64_t right = (rel_end <= seg_size) ? (seg_size - rel_end) : 0;     \
            cur.has_valid_or = from_start ? (left <= right) : (left >= right);     \
            *(tso_block *)vec_push(out) = cur;                                     \
            boundary = from_start ? (cur.start + cur.block_len) : cur.start;       \
        }                                                                          \
    } while (0)

    for (size_t i = 1; i < subs.n; ++i) {
        tso_block *nx = &sb[i];
        uint64_t gap = from_start ? (nx->start - (cur.start + cur.block_len))
                                  : (cur.start - (nx->start + nx->block_len));
        if (gap <= P->max_block_dist) {
            if (from_start) {
                cur.block_len = (uint32_t)((nx->start + nx->block_len) - cur.start);
            } else {
                cur.block_len = (uint32_t)((cur.start + cur.block_len) - nx->start);
                cur.start = nx->start;
            }
            cur.block_counts += nx->block_counts;
            cur.forward_count += nx->forward_count;
            cur.reverse_count += nx->reverse_count;
            cur.canonical_count += nx->canonical_count;
            cur.non_canonical_count += nx->non_canonical_count;
            cur.total_covered += nx->total_covered;
            cur.fwd_covered += nx->fwd_covered;
            cur.can_covered += nx->can_covered;
        } else {
            FINALIZE_EXT();
            cur = *nx;
        }
    }
    FINALIZE_EXT();
#undef FINALIZE_EXT
    free(subs.p);
    return boundary;
}

/* computeBlockLabel, include/teloscope.h:217-222 */
static char block_label(uint32_t fwd_count, uint32_t counts) {
    float ratio = ((float)fwd_count * 100.0f) / (float)counts;
    if (ratio > 66.6f) return 'p';
    if (ratio < 33.3f) return 'q';
    return 'b';
}

/* Teloscope::getInterstitialBlocks, src/teloscope.cpp:179-256 */
static void interstitial_blocks(const tso_ctx *c, const tso_match *ms, size_t nm, vec *out,
                                uint64_t fwd_boundary, uint64_t rev_boundary) {
    uint16_t merge_dist = c->p.max_match_dist;
    uint16_t min_len = (uint16_t)(2 * c->first_pattern_len);
    const uint16_t min_canon = 4;

    /* std::lower_bound as libstdc++ walks it (the input may be slightly
     * unsorted for mixed-length pattern sets; keep the same probe order) */
    size_t first = 0, len = nm;
    while (len > 0) {
        size_t half = len >> 1, mid = first + half;
        if (ms[mid].position < fwd_boundary) { first = mid + 1; len = len - half - 1; }
        else len = half;
    }
    if (first == nm || ms[first].position >= rev_boundary) return;

    chain b; memset(&b, 0, sizeof b);
#define FINALIZE_ITS() do {                                                        \
        uint32_t blen = (uint32_t)(b.end - b.start);                               \
        char lab = block_label(b.fwd, b.counts);                                   \
        if (blen >= min_len && b.canon >= min_canon &&                             \
            !(lab == 'b' && b.fwd < 2 && (b.counts - b.fwd) < 2)) {                \
            tso_block *o = (tso_block *)vec_push(out);                             \
            chain_to_block(&b, o);                                                 \
            o->block_label = lab;                                                  \
        }                                                                          \
        b.in_block = 0;                                                            \
    } while (0)

    for (size_t i = first; i < nm && ms[i].position < rev_boundary; ++i) {
        const tso_match *m = &ms[i];
        if (!b.in_block) {
            chain_start(&b, m);
        } else if (m->position - b.prev <= merge_dist) {
            b.end = m->position + m->match_size;
            chain_add(&b, m);
        } else {
            FINALIZE_ITS();
            chain_start(&b, m);
        }
    }
    if (b.in_block) FINALIZE_ITS();
#undef FINALIZE_ITS
}

static int cmp_block_start(const void *a, const void *b) {
    const tso_block *x = (const tso_block *)a, *y = (const tso_block *)b;
    return (x->start > y->start) - (x->start < y->start);
}

/* Teloscope::labelTerminalBlocks, src/teloscope.cpp:259-383.
 * Return value = ScaffoldType enumerator (include/tools.h:13-19):
 * 0 T2T 1 GAPPED_T2T 2 MISASSEMBLY 3 GAPPED_MISASSEMBLY 4 INCOMPLETE
 * 5 GAPPED_INCOMPLETE 6 NONE 7 GAPPED_NONE 8 DISCORDANT 9 GAPPED_DISCORDANT */
int tso_label_terminal_blocks(tso_block *blocks, size_t n, uint16_t gaps,
                              uint64_t path_size, uint32_t terminal_limit, char *label_out) {
    const int g = gaps > 0 ? 1 : 0;
    size_t L = 0;
    label_out[0] = '\0';
    for (size_t i = 0; i < n; ++i) blocks[i].is_longest = 0;
    if (n == 0) return 6 + g;

    /* insertion sort keeps equal starts in input order; starts are distinct in
     * practice (one p and one q chain per segment) */
    for (size_t i = 1; i < n; ++i) {
        tso_block t = blocks[i]; size_t j = i;
        while (j > 0 && cmp_block_start(&blocks[j - 1], &t) > 0) { blocks[j] = blocks[j - 1]; --j; }
        blocks[j] = t;
    }

    tso_block **scaf = (tso_block **)malloc(n * sizeof *scaf);
    size_t ns = 0;
    for (size_t i = 0; i < n; ++i) {
        uint64_t bend = blocks[i].start + blocks[i].block_len;
        /* pathSize - terminalLimit is evaluated in uint64 and may wrap */
        if (blocks[i].start < terminal_limit || bend > path_size - (uint64_t)terminal_limit)
            scaf[ns++] = &blocks[i];
    }
    for (size_t i = 0; i < n; ++i) {
        label_out[L++] = blocks[i].block_label;
        if (!blocks[i].has_valid_or) label_out[L++] = '*';
    }
    label_out[L] = '\0';

    tso_block *lp = NULL, *lq = NULL;
    uint64_t best_p = 0, best_q = 0;
    for (size_t i = 0; i < ns; ++i) {
        tso_block *b = scaf[i];
        if (b->block_label == 'p' && b->can_covered > best_p) { lp = b; best_p = b->can_covered; }
        else if (b->block_label == 'q' && b->can_covered > best_q) { lq = b; best_q = b->can_covered; }
    }
    if (lp) lp->is_longest = 1;
    if (lq) lq->is_longest = 1;

    for (size_t i = 0, j = 0; i < n; ++i) {
        if (blocks[i].is_longest && label_out[j] >= 'a' && label_out[j] <= 'z')
            label_out[j] = (char)(label_out[j] - 'a' + 'A');
        ++j;
        if (j < L && label_out[j] == '*') ++j;
    }

    int type;
    if ((lp && !lp->has_valid_or) || (lq && !lq->has_valid_or)) {
        type = 8 + g;
    } else if (lp && lq) {
        type = (lp->start < lq->start) ? 0 + g : 2 + g;
    } else {
        type = -1;
        if (lp) for (size_t i = 0; i < ns && type < 0; ++i)
            if (scaf[i]->block_label == 'p' && scaf[i] != lp && scaf[i]->has_valid_or) type = 2 + g;
        if (type < 0 && lq) for (size_t i = 0; i < ns && type < 0; ++i)
            if (scaf[i]->block_label == 'q' && scaf[i] != lq && scaf[i]->has_valid_or) type = 2 + g;
        if (type < 0) type = (!lp && !lq) ? 6 + g : 4 + g;
    }
    free(scaf);
    return type;
}

/* ------------------------------------------------------------ window scan */

typedef struct {
    vec windows, term, its, canon, noncanon, fwd, rev, all;
} seg_acc;

static void push_match(vec *v, uint64_t pos, uint16_t len, int fwd, int canon) {
    tso_match *m = (tso_match *)vec_push(v);
    m->position = pos; m->match_size = len;
    m->is_forward = (uint8_t)fwd; m->is_canonical = (uint8_t)canon; m->pad = 0;
}

/* Teloscope::analyzeWindow, src/teloscope.cpp:387-534.  `win`/`next` are the
 * record being completed and the carry for the following window. */
static void analyze_window(const tso_ctx *c, const char *window, uint32_t wsize,
                           uint64_t window_start, tso_window *win, tso_window *next,
                           seg_acc *acc, uint64_t seg_size, uint64_t abs_pos) {
    const tso_params *P = &c->p;
    const uint16_t longest = c->longest;
    const uint32_t step = P->step;
    const uint32_t overlap = P->window_size - step;
    const uint32_t tlimit = P->terminal_limit;
    const int count_bases = P->out_gc || P->out_entropy;
    int has_last_canon = 0;
    uint64_t last_canon_pos = 0;

    win->window_start = window_start;

    uint64_t window_end = window_start + wsize;
    uint64_t terminal_end = seg_size > tlimit ? seg_size - tlimit : 0;
    int fully_terminal = (window_end <= tlimit) || (window_start >= terminal_end);
    int fully_interstitial = (window_start > tlimit) && (window_end < terminal_end);

    int always_main = (overlap == 0 || window_start == 0);
    int has_overlap = overlap != 0;

    /* uint32 arithmetic on purpose: wraps when the longest pattern exceeds
     * step or overlap (src/teloscope.cpp:413-415) */
    uint32_t a = step - (uint32_t)longest, b = overlap - (uint32_t)longest;
    uint32_t start_index = always_main ? 0u : (a < b ? a : b);

    for (uint32_t i = start_index; i < wsize; ++i) {
        if (count_bases) {
            int bi = base_index(window[i]);
            if (bi < 0) continue;                       /* also skips the trie walk */
            if (always_main || i >= overlap) win->nucleotide_counts[bi]++;
            if (has_overlap && i >= step) next->nucleotide_counts[bi]++;
        }

        int32_t cur = 0;
        uint32_t lim = i + (uint32_t)longest;
        if (lim > wsize) lim = wsize;

        for (uint32_t j = i; j < lim; ++j) {
            cur = trie_child(c, cur, window[j]);
            if (cur < 0) break;
            if (!c->nodes[cur].end) continue;

            uint16_t mlen = (uint16_t)(j - i + 1);
            int is_fwd = c->nodes[cur].fwd, is_canon = c->nodes[cur].canon;
            uint64_t mpos = abs_pos + window_start + i;

            int is_terminal;
            if (fully_terminal) is_terminal = 1;
            else if (fully_interstitial) is_terminal = 0;
            else {
                uint64_t rel = window_start + i;
                is_terminal = (rel <= tlimit || rel >= terminal_end);
            }

            if (is_canon) {
                if (has_last_canon && (mpos - last_canon_pos) <= P->canonical_size) {
                    if (!win->has_can_dimer && (always_main || j >= overlap)) win->has_can_dimer = 1;
                    if (!next->has_can_dimer && has_overlap && i >= step) next->has_can_dimer = 1;
                }
                last_canon_pos = mpos;
                has_last_canon = 1;
            }

            if (always_main || j >= overlap) {
                if (is_canon) {
                    win->canonical_counts++;
                    win->canonical_covered += mlen;
                    push_match(&acc->canon, mpos, mlen, is_fwd, is_canon);
                } else {
                    win->non_canonical_counts++;
                    win->non_canonical_covered += mlen;
                    if (is_terminal) push_match(&acc->noncanon, mpos, mlen, is_fwd, is_canon);
                }
                if (is_fwd) {
                    win->fwd_counts++;
                    win->fwd_covered += mlen;
                    push_match(&acc->fwd, mpos, mlen, is_fwd, is_canon);
                } else {
                    win->rev_counts++;
                    win->rev_covered += mlen;
                    push_match(&acc->rev, mpos, mlen, is_fwd, is_canon);
                }
                push_match(&acc->all, mpos, mlen, is_fwd, is_canon);
            }

            if (has_overlap && i >= step) {
                if (is_canon) { next->canonical_counts++; next->canonical_covered += mlen; }
                else          { next->non_canonical_counts++; next->non_canonical_covered += mlen; }
                if (is_fwd)   { next->fwd_counts++; next->fwd_covered += mlen; }
                else          { next->rev_counts++; next->rev_covered += mlen; }
            }
        }
    }
}

/* tips-only processRegion lambda, src/teloscope.cpp:546-574 */
static void tips_region(const tso_ctx *c, const char *seq, uint64_t start, uint64_t end,
                        uint64_t abs_pos, seg_acc *acc) {
    for (uint64_t i = start; i < end; ++i) {
        int32_t nd = 0;
        uint64_t lim = i + (uint64_t)c->longest;
        if (lim > end) lim = end;
        for (uint64_t j = i; j < lim; ++j) {
            nd = trie_child(c, nd, seq[j]);
            if (nd < 0) break;
            if (c->nodes[nd].end) {
                uint16_t len = (uint16_t)(j - i + 1);
                if (c->nodes[nd].fwd) push_match(&acc->fwd, abs_pos + i, len, 1, c->nodes[nd].canon);
                else                  push_match(&acc->rev, abs_pos + i, len, 0, c->nodes[nd].canon);
            }
        }
    }
}

/* Teloscope::scanSegment, src/teloscope.cpp:537-658 */
int tso_scan_segment(const tso_ctx *c, const char *seq, uint64_t n, uint64_t abs_pos,
                     int tips_only, tso_segment *out) {
    const tso_params *P = &c->p;
    seg_acc acc;
    vec_init(&acc.windows, sizeof(tso_window));
    vec_init(&acc.term, sizeof(tso_block));
    vec_init(&acc.its, sizeof(tso_block));
    vec_init(&acc.canon, sizeof(tso_match));
    vec_init(&acc.noncanon, sizeof(tso_match));
    vec_init(&acc.fwd, sizeof(tso_match));
    vec_init(&acc.rev, sizeof(tso_match));
    vec_init(&acc.all, sizeof(tso_match));

    if (tips_only) {
        uint32_t twice = 2u * P->terminal_limit;        /* uint32 product, as in the reference */
        if (n > twice) {
            tips_region(c, seq, 0, P->terminal_limit, abs_pos, &acc);
            tips_region(c, seq, n - P->terminal_limit, n, abs_pos, &acc);
        } else {
            tips_region(c, seq, 0, n, abs_pos, &acc);
        }
    } else {
        tso_window prev, next;
        memset(&prev, 0, sizeof prev);
        memset(&next, 0, sizeof next);
        uint64_t wstart = 0;
        uint64_t cur_size = P->window_size < n ? P->window_size : n;
        const char *view = seq;
        while (wstart < n) {
            tso_window w = prev;
            analyze_window(c, view, (uint32_t)cur_size, wstart, &w, &next, &acc, n, abs_pos);
            if (P->out_gc) w.gc_content = tso_gc_content(w.nucleotide_counts, (uint32_t)cur_size);
            if (P->out_entropy) w.shannon_entropy = tso_shannon_entropy(w.nucleotide_counts, (uint32_t)cur_size);
            w.window_start = wstart + abs_pos;
            w.current_window_size = (uint32_t)cur_size;
            *(tso_window *)vec_push(&acc.windows) = w;
            prev = next;
            memset(&next, 0, sizeof next);
            wstart += P->step;
            if (wstart >= n) break;
            cur_size = (n - wstart) < P->window_size ? (n - wstart) : P->window_size;
            view = seq + wstart;
        }
    }

    uint64_t fwd_boundary = abs_pos, rev_boundary = abs_pos + n;
    if (acc.fwd.n >= 2)
        fwd_boundary = terminal_blocks(c, (tso_match *)acc.fwd.p, acc.fwd.n, &acc.term, n, abs_pos, 1);
    if (acc.rev.n >= 2)
        rev_boundary = terminal_blocks(c, (tso_match *)acc.rev.p, acc.rev.n, &acc.term, n, abs_pos, 0);
    if (!tips_only && fwd_boundary < rev_boundary && acc.all.n >= 2)
        interstitial_blocks(c, (tso_match *)acc.all.p, acc.all.n, &acc.its, fwd_boundary, rev_boundary);

    out->windows = (tso_window *)acc.windows.p;             out->n_windows = acc.windows.n;
    out->terminal_blocks = (tso_block *)acc.term.p;         out->n_terminal_blocks = acc.term.n;
    out->interstitial_blocks = (tso_block *)acc.its.p;      out->n_interstitial_blocks = acc.its.n;
    out->canonical_matches = (tso_match *)acc.canon.p;      out->n_canonical_matches = acc.canon.n;
    out->non_canonical_matches = (tso_match *)acc.noncanon.p; out->n_non_canonical_matches = acc.noncanon.n;
    out->fwd_matches = (tso_match *)acc.fwd.p;              out->n_fwd_matches = acc.fwd.n;
    out->rev_matches = (tso_match *)acc.rev.p;              out->n_rev_matches = acc.rev.n;
    out->all_matches = (tso_match *)acc.all.p;              out->n_all_matches = acc.all.n;
    return 0;
}

void tso_free_segment(tso_segment *s) {
    free(s->windows); free(s->terminal_blocks); free(s->interstitial_blocks);
    free(s->canonical_matches); free(s->non_canonical_matches);
    free(s->fwd_matches); free(s->rev_matches); free(s->all_matches);
    memset(s, 0, sizeof *s);
}

/* ------------------------------------------------------------ read filter */

/* makeReadFilterInput, src/read-filter.cpp:10-30 */
void tso_read_filter_params(const tso_params *in, int min_block_len_set, tso_params *out) {
    *out = *in;
    if (!min_block_len_set) out->min_block_len = 42;
    out->terminal_limit = UINT32_MAX / 2;
    out->out_gc = 0; out->out_entropy = 0; out->out_matches = 0;
}

/* ReadTelomereFilter::matches, src/read-filter.cpp:37-45 */
int tso_read_filter_matches(const tso_ctx *c, const char *seq, uint64_t n) {
    if (n > 0 && seq[n - 1] == '\r') --n;
    char *up = (char *)malloc(n ? n : 1);
    for (uint64_t i = 0; i < n; ++i) {          /* unmaskSequence = upper-case */
        char ch = seq[i];
        up[i] = (ch >= 'a' && ch <= 'z') ? (char)(ch - 'a' + 'A') : ch;
    }
    tso_segment s;
    tso_scan_segment(c, up, n, 0, 1, &s);
    int pass = s.n_terminal_blocks != 0;
    tso_free_segment(&s);
    free(up);
    return pass;
}

/* ------------------------------------------------------------------ bench */

uint64_t tso_bench_scan(const tso_ctx *c, const char *seq, uint64_t n,
                        uint64_t *n_windows, uint64_t *n_matches) {
    tso_segment s;
    tso_scan_segment(c, seq, n, 0, 0, &s);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < s.n_windows; ++i) {
        const tso_window *w = &s.windows[i];
        h = (h ^ w->canonical_covered) * 1099511628211ull;
        h = (h ^ w->fwd_covered) * 1099511628211ull;
        h = (h ^ w->nucleotide_counts[2]) * 1099511628211ull;
    }
    for (size_t i = 0; i < s.n_all_matches; ++i)
        h = (h ^ s.all_matches[i].position) * 1099511628211ull;
    *n_windows = s.n_windows;
    *n_matches = s.n_all_matches;
    tso_free_segment(&s);
    return h;
}
