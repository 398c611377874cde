// blocks.cpp — telomere block calling on the compact match stream (host, C++17).
//
// Product restatement of Teloscope::getTerminalBlocks (src/teloscope.cpp:29-176),
// getInterstitialBlocks (:179-256), labelTerminalBlocks (:259-383) and the float window
// metrics (include/teloscope.h:199-214).  O(matches) integer work downstream of the scan
// kernels; the device version of it is the next row of the scope table.
#include "host.hpp"

#include <algorithm>
#include <cmath>

namespace ts {

namespace {

// Running chain of matches closer than -k to each other.
struct Chain {
    bool open = false;
    uint64_t start = 0, end = 0, prev = 0;
    uint32_t counts = 0, fwd = 0, canon = 0, covered = 0, fwd_cov = 0, can_cov = 0;

    void begin(const ts_match &m) {
        const bool f = m.flags & TS_MATCH_FORWARD, c = m.flags & TS_MATCH_CANONICAL;
        start = m.position;
        end = m.position + m.match_size;
        prev = m.position;
        counts = 1;
        fwd = f; canon = c;
        covered = m.match_size;
        fwd_cov = f ? m.match_size : 0;
        can_cov = c ? m.match_size : 0;
        open = true;
    }
    void extend(const ts_match &m) {
        const bool f = m.flags & TS_MATCH_FORWARD, c = m.flags & TS_MATCH_CANONICAL;
        ++counts;
        fwd += f; canon += c;
        covered += m.match_size;
        if (f) fwd_cov += m.match_size;
        if (c) can_cov += m.match_size;
        prev = m.position;
    }
    ts_block to_block() const {
        ts_block b{};
        b.start = start;
        b.block_len = static_cast<uint32_t>(end - start);
        b.block_counts = counts;
        b.forward_count = fwd;
        b.reverse_count = counts - fwd;
        b.canonical_count = canon;
        b.non_canonical_count = counts - canon;
        b.total_covered = covered;
        b.fwd_covered = fwd_cov;
        b.can_covered = can_cov;
        b.has_valid_or = 1;
        b.is_longest = 0;
        b.block_label = '\0';
        return b;
    }
};

char its_label(uint32_t fwd_count, uint32_t counts) {       // computeBlockLabel, teloscope.h:217-222
    const float ratio = (static_cast<float>(fwd_count) * 100.0f) / static_cast<float>(counts);
    if (ratio > 66.6f) return 'p';
    if (ratio < 33.3f) return 'q';
    return 'b';
}

}  // namespace

uint64_t terminal_blocks(const BlockParams &bp, const ts_match *m, const uint32_t *idx, size_t n,
                         std::vector<ts_block> &out, uint64_t seg_size, uint64_t abs_pos, bool from_start,
                         int only_forward) {
    // only_forward >= 0 (and idx == nullptr): m[0..n) holds both orientations in position order and the walk
    // takes the records whose forward flag equals it — the walk leaves the terminal zone after a few thousand
    // records, so the per-orientation index lists of a whole segment need not be built
    uint64_t boundary = from_start ? abs_pos : abs_pos + seg_size;
    auto at = [&](size_t i) -> const ts_match & { return idx ? m[idx[i]] : m[i]; };
    auto in_zone = [&](uint64_t pos) {
        if (seg_size <= bp.terminal_limit) return true;
        const uint64_t rel = pos - abs_pos;
        return from_start ? rel < bp.terminal_limit : rel >= seg_size - bp.terminal_limit;
    };

    // phase 1: walk inwards from the segment end, chaining matches <= -k apart
    std::vector<ts_block> subs;
    Chain ch;
    auto close_chain = [&]() {
        const float need = bp.min_block_density * static_cast<float>(ch.end - ch.start);
        if (ch.counts >= bp.min_block_counts && ch.canon > 0 && static_cast<float>(ch.can_cov) >= need)
            subs.push_back(ch.to_block());
        ch.open = false;
    };
    for (size_t step = 0; step < n; ++step) {
        const ts_match &cur = at(from_start ? step : n - 1 - step);
        if (only_forward >= 0 && ((cur.flags & TS_MATCH_FORWARD) != 0) != (only_forward != 0)) continue;
        if (ch.open) {
            const uint64_t gap = from_start ? cur.position - ch.prev : ch.prev - cur.position;
            if (gap <= bp.max_match_dist) {
                if (from_start) ch.end = cur.position + cur.match_size;
                else            ch.start = cur.position;
                ch.extend(cur);
                continue;
            }
            close_chain();
        }
        if (!in_zone(cur.position)) break;
        ch.begin(cur);
    }
    if (ch.open) close_chain();
    if (subs.empty()) return boundary;

    // phase 2: merge sub-blocks <= -d apart, keep those >= -l
    ts_block cur = subs[0];
    auto close_block = [&]() {
        if (cur.block_len < bp.min_block_len) return;
        cur.block_label = from_start ? 'p' : 'q';
        const uint64_t rel_start = cur.start - abs_pos;
        const uint64_t rel_end = rel_start + cur.block_len;
        const uint64_t left = rel_start;
        const uint64_t right = rel_end <= seg_size ? seg_size - rel_end : 0;
        cur.has_valid_or = from_start ? (left <= right) : (left >= right);
        out.push_back(cur);
        boundary = from_start ? cur.start + cur.block_len : cur.start;
    };
    for (size_t i = 1; i < subs.size(); ++i) {
        const ts_block &nx = subs[i];
        const uint64_t gap = from_start ? nx.start - (cur.start + cur.block_len)
                                        : cur.start - (nx.start + nx.block_len);
        if (gap > bp.max_block_dist) {
            close_block();
            cur = nx;
            continue;
        }
        if (from_start) {
            cur.block_len = static_cast<uint32_t>(nx.start + nx.block_len - cur.start);
        } else {
            cur.block_len = static_cast<uint32_t>(cur.start + cur.block_len - nx.start);
            cur.start = nx.start;
        }
        cur.block_counts += nx.block_counts;
        cur.forward_count += nx.forward_count;
        cur.reverse_count += nx.reverse_count;
        cur.canonical_count += nx.canonical_count;
        cur.non_canonical_count += nx.non_canonical_count;
        cur.total_covered += nx.total_covered;
        cur.fwd_covered += nx.fwd_covered;
        cur.can_covered += nx.can_covered;
    }
    close_block();
    return boundary;
}

void interstitial_blocks(const BlockParams &bp, const ts_match *m, size_t n, std::vector<ts_block> &out,
                         uint64_t fwd_boundary, uint64_t rev_boundary) {
    const uint16_t min_len = static_cast<uint16_t>(2 * bp.first_pattern_len);
    const ts_match *it = std::lower_bound(m, m + n, fwd_boundary,
                                          [](const ts_match &a, uint64_t v) { return a.position < v; });
    if (it == m + n || it->position >= rev_boundary) return;

    Chain ch;
    auto close_chain = [&]() {
        const uint32_t len = static_cast<uint32_t>(ch.end - ch.start);
        const char lab = its_label(ch.fwd, ch.counts);
        const bool weak_balanced = lab == 'b' && ch.fwd < 2 && (ch.counts - ch.fwd) < 2;
        if (len >= min_len && ch.canon >= 4 && !weak_balanced) {
            ts_block b = ch.to_block();
            b.block_label = lab;
            out.push_back(b);
        }
        ch.open = false;
    };
    for (; it != m + n && it->position < rev_boundary; ++it) {
        if (ch.open && it->position - ch.prev <= bp.max_match_dist) {
            ch.end = it->position + it->match_size;
            ch.extend(*it);
            continue;
        }
        if (ch.open) close_chain();
        ch.begin(*it);
    }
    if (ch.open) close_chain();
}

int label_terminal_blocks(ts_block *blocks, size_t n, uint16_t gaps, uint64_t path_size,
                          uint32_t terminal_limit, std::string &label) {
    const int g = gaps > 0 ? 1 : 0;
    label.clear();
    for (size_t i = 0; i < n; ++i) blocks[i].is_longest = 0;
    if (n == 0) return TS_NONE + g;

    std::sort(blocks, blocks + n, [](const ts_block &a, const ts_block &b) { return a.start < b.start; });

    std::vector<ts_block *> ends;                    // scaffold-terminal blocks
    for (size_t i = 0; i < n; ++i) {
        const uint64_t bend = blocks[i].start + blocks[i].block_len;
        if (blocks[i].start < terminal_limit || bend > path_size - static_cast<uint64_t>(terminal_limit))
            ends.push_back(&blocks[i]);
    }
    std::vector<size_t> label_pos(n);
    for (size_t i = 0; i < n; ++i) {
        label_pos[i] = label.size();
        label.push_back(blocks[i].block_label);
        if (!blocks[i].has_valid_or) label.push_back('*');
    }

    ts_block *best_p = nullptr, *best_q = nullptr;
    uint64_t cov_p = 0, cov_q = 0;
    for (ts_block *b : ends) {
        if (b->block_label == 'p' && b->can_covered > cov_p) { best_p = b; cov_p = b->can_covered; }
        else if (b->block_label == 'q' && b->can_covered > cov_q) { best_q = b; cov_q = b->can_covered; }
    }
    if (best_p) best_p->is_longest = 1;
    if (best_q) best_q->is_longest = 1;
    for (size_t i = 0; i < n; ++i)
        if (blocks[i].is_longest)
            label[label_pos[i]] = static_cast<char>(std::toupper(static_cast<unsigned char>(label[label_pos[i]])));

    if ((best_p && !best_p->has_valid_or) || (best_q && !best_q->has_valid_or)) return TS_DISCORDANT + g;
    if (best_p && best_q) return (best_p->start < best_q->start ? TS_T2T : TS_MISASSEMBLY) + g;
    if (best_p)
        for (const ts_block *b : ends)
            if (b->block_label == 'p' && b != best_p && b->has_valid_or) return TS_MISASSEMBLY + g;
    if (best_q)
        for (const ts_block *b : ends)
            if (b->block_label == 'q' && b != best_q && b->has_valid_or) return TS_MISASSEMBLY + g;
    if (!best_p && !best_q) return TS_NONE + g;
    return TS_INCOMPLETE + g;
}

// getGCContent: float division, double multiply, narrowed (include/teloscope.h:211-214)
float gc_content(const uint32_t counts[4], uint32_t window_size) {
    const uint32_t gc = counts[1] + counts[2];
    return static_cast<float>(gc) / window_size * 100.0;
}

// getShannonEntropy: float32 throughout, rounded to 3 decimals (include/teloscope.h:199-208)
float shannon_entropy(const uint32_t counts[4], uint32_t window_size) {
    float entropy = 0.0;
    for (int i = 0; i < 4; ++i) {
        if (counts[i] > 0) {
            const float p = static_cast<float>(counts[i]) / window_size;
            entropy -= p * std::log2(p);
        }
    }
    return std::round(entropy * 1000.0f) / 1000.0f;
}

// The same with the four terms p log2 p looked up: a count c of a window of `w` bases always gives the same term, and a
// scan converts millions of windows of one size (log2f four times per window was most of the host's share of a scan
// without match vectors).  term[c] is computed by the expression above, so the sum is the same float.
void entropy_terms(uint32_t w, std::vector<float> &term) {
    term.assign((size_t)w + 1, 0.0f);
    for (uint32_t c = 1; c <= w; ++c) {
        const float p = static_cast<float>(c) / w;
        term[c] = p * std::log2(p);
    }
}

float shannon_entropy_memo(const uint32_t counts[4], uint32_t window_size, const std::vector<float> &term) {
    if (term.size() != (size_t)window_size + 1) return shannon_entropy(counts, window_size);
    float entropy = 0.0;
    for (int i = 0; i < 4; ++i) {
        if (counts[i] > 0) {
            if (counts[i] > window_size) return shannon_entropy(counts, window_size);
            entropy -= term[counts[i]];
        }
    }
    return std::round(entropy * 1000.0f) / 1000.0f;
}

}  // namespace ts
