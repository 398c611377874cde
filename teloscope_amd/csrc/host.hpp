// host.hpp — internal C++17 interfaces of libteloscan's host side.
#ifndef TS_HOST_HPP
#define TS_HOST_HPP

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/teloscan.h"

namespace ts {

struct Pattern {
    std::string seq;
    bool is_forward = false;
    bool is_canonical = false;
};

// patterns.cpp
std::string rev_com(const std::string &s);
void unmask(std::string &s);
void canonical_orientation(const std::string &canonical_in, std::string &fwd, std::string &rev);
std::vector<Pattern> expand_patterns(const std::string &raw_csv, int edit_distance,
                                     const std::string &canonical_fwd);
int  base_code(char c);
bool build_match_table(const std::vector<Pattern> &pats, uint32_t k, uint32_t max_byte_k, std::vector<uint32_t> &table,
                       uint32_t &rows, uint32_t &fc_bytes, bool &fc_byte_table, bool &pair_byte_table);

// blocks.cpp — block calling on the match stream (src/teloscope.cpp:29-383)
struct BlockParams {
    uint32_t terminal_limit;
    uint16_t max_match_dist, min_block_len, max_block_dist, min_block_counts;
    float    min_block_density;
    uint16_t first_pattern_len;     // userInput.patterns.front().size()
};

// Matches are passed as parallel views over ts_match records; `idx` selects a subsequence
// (fwdMatches / revMatches) without copying, nullptr = all.
uint64_t terminal_blocks(const BlockParams &bp, const ts_match *m, const uint32_t *idx, size_t n,
                         std::vector<ts_block> &out, uint64_t seg_size, uint64_t abs_pos, bool from_start,
                         int only_forward = -1);
void interstitial_blocks(const BlockParams &bp, const ts_match *m, size_t n, std::vector<ts_block> &out,
                         uint64_t fwd_boundary, uint64_t rev_boundary);
int  label_terminal_blocks(ts_block *blocks, size_t n, uint16_t gaps, uint64_t path_size,
                           uint32_t terminal_limit, std::string &label);

// pack.cpp — the packed upload's host half: bases -> 2-bit codes + invalid runs
struct InvalidRun { uint32_t start, len; };          // positions relative to the staged chunk
struct PackRuns {
    std::vector<InvalidRun> runs;
    uint32_t open_start = 0, open_len = 0;
    void finish();
};
void pack_bases(const unsigned char *src, size_t n, unsigned char *dst, bool fold, uint32_t pos0, PackRuns &r);
// The same from FASTA body text: up to n bases from *cursor on (line ends skipped: '\n', and a '\r' right before one or at the very end
// of the text), packed straight from the text — no stripped copy in between.  dst gets (taken + 3) / 4 bytes (the last one padded with
// code 0) and may be written up to 8 bytes beyond that; returns the bases taken (fewer than n only when the text ends) and leaves the
// cursor behind the last byte it consumed (never between a carriage return and its line feed).
size_t pack_text(const char **cursor, const char *end, size_t n, unsigned char *dst, bool fold, uint32_t pos0, PackRuns &r);

float gc_content(const uint32_t counts[4], uint32_t window_size);
float shannon_entropy(const uint32_t counts[4], uint32_t window_size);
void  entropy_terms(uint32_t w, std::vector<float> &term);                       // term[c] = (c / w) log2 (c / w), as shannon_entropy computes it
float shannon_entropy_memo(const uint32_t counts[4], uint32_t window_size, const std::vector<float> &term);   // == shannon_entropy

}  // namespace ts

#endif
