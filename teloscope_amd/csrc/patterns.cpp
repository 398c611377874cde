// patterns.cpp — host-side pattern preparation of libteloscan (C++17).
//
// Product restatement of the reference's pattern expansion
// (expandPatternsWithOrientation, src/tools.cpp:201-283; getCombinations :73-85;
// getEditVariants :88-128; the canonical orientation rule src/main.cpp:287-296) and of
// the gfalibs helpers it leans on (revCom, unmaskSequence).  It defines WHICH k-mers the
// kernels match, so it is part of the path even though it runs once per context.
#include "host.hpp"

#include <algorithm>
#include <cstring>

namespace ts {

std::string rev_com(const std::string &s) {
    std::string r(s.rbegin(), s.rend());
    for (char &c : r) {
        switch (c) {
            case 'A': c = 'T'; break; case 'T': c = 'A'; break;
            case 'C': c = 'G'; break; case 'G': c = 'C'; break;
            case 'a': c = 't'; break; case 't': c = 'a'; break;
            case 'c': c = 'g'; break; case 'g': c = 'c'; break;
            default: break;
        }
    }
    return r;
}

void unmask(std::string &s) {
    for (char &c : s)
        if (c >= 'a' && c <= 'z') c = static_cast<char>(c - 'a' + 'A');
}

namespace {

const char *iupac(char c) {
    switch (c) {
        case 'A': return "A";   case 'C': return "C";   case 'G': return "G";   case 'T': return "T";
        case 'R': return "AG";  case 'Y': return "CT";  case 'M': return "AC";  case 'K': return "GT";
        case 'S': return "CG";  case 'W': return "AT";  case 'H': return "ACT"; case 'B': return "CGT";
        case 'V': return "ACG"; case 'D': return "AGT"; case 'N': return "ACGT";
        default:  return "";
    }
}

void combos(const std::string &pat, std::string &cur, size_t idx, std::vector<std::string> &out) {
    if (idx == pat.size()) { out.push_back(cur); return; }
    for (const char *m = iupac(pat[idx]); *m; ++m) {
        cur[idx] = *m;
        combos(pat, cur, idx + 1, out);
    }
}

void edits(const std::string &pat, int max_dist, std::vector<std::string> &out) {
    if (max_dist <= 0) return;
    static const char nts[4] = {'A', 'C', 'G', 'T'};
    const size_t first = out.size();
    for (size_t i = 0; i < pat.size(); ++i) {
        const char o = pat[i];
        if (o != 'A' && o != 'C' && o != 'G' && o != 'T') continue;
        for (char n : nts) {
            if (n == o) continue;
            out.push_back(pat);
            out.back()[i] = n;
        }
    }
    if (max_dist >= 2) {
        const size_t d1 = out.size();
        for (size_t v = first; v < d1; ++v) {
            const std::string base = out[v];
            edits(base, 1, out);
        }
    }
}

unsigned best_offset_distance(const std::string &shorter, const std::string &longer) {
    unsigned best = 255;
    for (size_t off = 0; off + shorter.size() <= longer.size(); ++off) {
        unsigned d = 0;
        for (size_t i = 0; i < shorter.size(); ++i) d += shorter[i] != longer[off + i];
        best = std::min<unsigned>(best, d & 0xFFu);
    }
    return best;
}

bool closer_to_fwd(const std::string &pat, const std::string &can_fwd, const std::string &can_rev) {
    if (pat.size() == can_fwd.size()) {
        unsigned df = 0, dr = 0;
        for (size_t i = 0; i < pat.size(); ++i) {
            df += pat[i] != can_fwd[i];
            dr += pat[i] != can_rev[i];
        }
        return (df & 0xFFu) <= (dr & 0xFFu);
    }
    const bool pat_short = pat.size() < can_fwd.size();
    const std::string &shorter = pat_short ? pat : can_fwd;
    const std::string longer = pat_short ? can_fwd : pat;
    const std::string longer_rev = pat_short ? can_rev : rev_com(pat);
    return best_offset_distance(shorter, longer) <= best_offset_distance(shorter, longer_rev);
}

}  // namespace

void canonical_orientation(const std::string &canonical_in, std::string &fwd, std::string &rev) {
    std::string c = canonical_in;
    unmask(c);
    const std::string rc = rev_com(c);
    if (c <= rc) { fwd = c; rev = rc; } else { fwd = rc; rev = c; }
}

std::vector<Pattern> expand_patterns(const std::string &raw_csv, int edit_distance,
                                     const std::string &canonical_fwd) {
    const std::string can_rev = rev_com(canonical_fwd);
    std::vector<std::pair<std::string, bool>> all;

    size_t pos = 0;
    while (pos <= raw_csv.size()) {
        size_t comma = raw_csv.find(',', pos);
        if (comma == std::string::npos) comma = raw_csv.size();
        std::string seed = raw_csv.substr(pos, comma - pos);
        pos = comma + 1;
        if (seed.empty() || seed.size() > 63) continue;           // (ts_expand_patterns refuses such a list before it gets here)
        unmask(seed);

        std::vector<std::string> cs;
        std::string cur = seed;
        combos(seed, cur, 0, cs);
        for (const std::string &combo : cs) {
            const bool seed_fwd = closer_to_fwd(combo, canonical_fwd, can_rev);
            std::vector<std::string> vars{combo};
            edits(combo, edit_distance, vars);
            for (const std::string &v : vars) {
                all.emplace_back(v, seed_fwd);
                all.emplace_back(rev_com(v), !seed_fwd);
            }
        }
    }

    // Same call sequence as the reference on the same element order, so that equal strings
    // with different orientation resolve the way its libstdc++ build resolves them.
    std::sort(all.begin(), all.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    all.erase(std::unique(all.begin(), all.end(),
                          [](const auto &a, const auto &b) { return a.first == b.first; }),
              all.end());

    std::vector<Pattern> out;
    out.reserve(all.size());
    for (const auto &pr : all) {
        Pattern p;
        p.seq = pr.first;
        p.is_forward = pr.second;
        p.is_canonical = (pr.first == canonical_fwd || pr.first == can_rev);
        out.push_back(std::move(p));
    }
    return out;
}

// 2-bit code used by the kernels' SWAR decode: (ascii & 6) >> 1  ->  A0 C1 T2 G3
int base_code(char c) {
    switch (c) {
        case 'A': return 0; case 'C': return 1; case 'T': return 2; case 'G': return 3;
        default: return -1;
    }
}

// Match tables for uniform-length pattern sets (k-mer index x: base i at bits 2i..2i+1).
//  * pair table: indexed by the (k+1)-mer y = bases p..p+k; entry = 2 bits {x(p) is a pattern,
//    x(p+1) is a pattern}, one byte per entry for k <= max_byte_k (6, or 7 when the caller has 64 KB of
//    LDS for it), else 16 entries per dword (row = y >> 4).
//    Not replicated: the kernel is
//    bound by VALU issue, and spreading rows over the LDS banks (8 copies) bought 0.4 % when it
//    was measured, while the 28 KB it cost is what the per-wave count planes now live in;
//  * flags {canonical, forward} (bit 0, bit 1: the low bits of a match record) per k-mer: bits 2..3 of the
//    byte pair table's entries (no separate table then), else a flag table, looked up only at matched positions: one byte
//    per k-mer for k <= 7 (cheapest lookup), 2 bits per k-mer at k = 8 (LDS capacity).
// Layout in `table`: [rows dwords][flag table].
bool build_match_table(const std::vector<Pattern> &pats, uint32_t k, uint32_t max_byte_k, std::vector<uint32_t> &table,
                       uint32_t &rows, uint32_t &fc_bytes, bool &fc_byte_table, bool &pair_byte_table) {
    if (k < 3 || k > 8) return false;
    const uint64_t nk = 1ull << (2 * k);
    std::vector<uint8_t> m(nk, 0), fl(nk, 0);
    for (const Pattern &p : pats) {
        if (p.seq.size() != k) return false;
        uint32_t x = 0;
        for (uint32_t i = 0; i < k; ++i) {
            const int c = base_code(p.seq[i]);
            if (c < 0) return false;
            x |= static_cast<uint32_t>(c) << (2 * i);
        }
        m[x] = 1;
        fl[x] = static_cast<uint8_t>((p.is_canonical ? 1 : 0) | (p.is_forward ? 2 : 0));   // a match record's low bits
    }
    const uint64_t npairs = nk * 4;                                   // (k+1)-mers
    // k <= max_byte_k (6, or 7 when the caller has the LDS for it): one BYTE per (k+1)-mer (16 KB at k = 6,
    // 64 KB at k = 7): a probe is then v_bfe / ds_read_u8 / alignbit, no row-and-shift arithmetic; larger k
    // keeps 2 bits per (k+1)-mer
    pair_byte_table = k <= max_byte_k;
    rows = static_cast<uint32_t>(pair_byte_table ? npairs / 4 : npairs / 16);
    fc_byte_table = k <= 7;
    // with a byte pair table the flags of the k-mer at p ride in bits 2..3 of its entries: no flag table
    const size_t fc_words = pair_byte_table ? 0 : static_cast<size_t>(std::max<uint64_t>(fc_byte_table ? nk / 4 : nk / 16, 4));
    fc_bytes = static_cast<uint32_t>(fc_words * 4);
    table.assign(static_cast<size_t>(rows) + fc_words, 0u);
    const uint32_t kmask = static_cast<uint32_t>(nk - 1);
    for (uint64_t y = 0; y < npairs; ++y) {
        const uint32_t bits = (m[y & kmask] ? 1u : 0u) | (m[(y >> 2) & kmask] ? 2u : 0u);
        if (pair_byte_table) {
            const uint32_t entry = bits | (static_cast<uint32_t>(fl[y & kmask]) << 2);
            table[y >> 2] |= entry << (8 * (y & 3));
            continue;
        }
        if (!bits) continue;
        table[y >> 4] |= bits << (2 * (y & 15));
    }
    uint32_t *fc = fc_words ? &table[rows] : nullptr;
    for (uint64_t x = 0; fc && x < nk; ++x) {
        if (fc_byte_table) fc[x >> 2] |= static_cast<uint32_t>(fl[x]) << (8 * (x & 3));
        else fc[x >> 4] |= static_cast<uint32_t>(fl[x]) << (2 * (x & 15));
    }
    return true;
}

}  // namespace ts
