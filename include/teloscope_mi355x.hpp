// teloscope_mi355x.hpp — C++17 host-side mirror of the reference's scan-path interface, above
// the C-ABI of libteloscan.so (include/teloscan.h).
//
// Same class names, member names and argument meaning as the reference, so reference call
// sites compile against it unchanged:
//
//     UserInputTeloscope userInput;                       // include/input.h:15-64
//     userInput.patternInfo = expandPatternsWithOrientation(userInput.rawPatterns,
//                                 userInput.editDistance, userInput.canonicalFwd);   // src/tools.cpp:201
//     Teloscope teloscope(userInput);                      // include/teloscope.h:241
//     SegmentData sd = teloscope.scanSegment(sequence, absPos, tipsOnly);            // :260
//     ReadTelomereFilter filter(userInput);  bool keep = filter.matches(read);       // read-filter.h:16
//
// plus the batched forms a GPU needs (scanSegments, matchesBatch).  Header-only; link with
// -lteloscan.  Errors the reference would answer with exit(EXIT_FAILURE) are thrown as
// std::runtime_error carrying ts_last_error(); nothing falls back to a CPU scan.
#ifndef TELOSCOPE_MI355X_HPP
#define TELOSCOPE_MI355X_HPP

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "teloscan.h"

namespace teloscope_mi355x {

enum class ScaffoldType : uint8_t {            // include/tools.h:13-19
    T2T, GAPPED_T2T, MISASSEMBLY, GAPPED_MISASSEMBLY, INCOMPLETE, GAPPED_INCOMPLETE,
    NONE, GAPPED_NONE, DISCORDANT, GAPPED_DISCORDANT
};

struct UserInputTeloscope {                    // include/input.h:15-64 (fields the scan path reads)
    std::string canonicalFwd = "CCCTAA";
    std::string canonicalRev = "TTAGGG";
    unsigned short int canonicalSize = 6;
    std::vector<std::string> rawPatterns = {"TTAGGG", "CCCTAA"};
    std::vector<std::string> patterns = {"TTAGGG", "CCCTAA"};
    std::vector<std::pair<std::string, bool>> patternInfo;      // (pattern, isForward)
    uint32_t windowSize = 1000;
    uint32_t step = 1000;
    uint32_t terminalLimit = 50000;
    uint8_t editDistance = 1;
    unsigned short int maxMatchDist = 50;
    unsigned short int minBlockLen = 300;
    bool minBlockLenSet = false;
    unsigned short int maxBlockDist = 500;
    unsigned short int minBlockCounts = 2;
    float minBlockDensity = 0.5f;
    bool outWinRepeats = false, outGC = false, outEntropy = false, outMatches = false, outITS = false;
    bool ultraFastMode = true;
    int device = -1;                           // HIP device ordinal (-1 = current); not in the reference
};

struct MatchInfo {                             // include/teloscope.h:89-95
    bool isCanonical = false;
    bool isForward = false;
    uint64_t position = 0;
    uint16_t matchSize = 0;
    std::string matchSeq;
};

struct TelomereBlock {                         // include/teloscope.h:103-117
    uint64_t start = 0;
    uint32_t blockLen = 0, blockCounts = 0, forwardCount = 0, reverseCount = 0, canonicalCount = 0,
             nonCanonicalCount = 0, totalCovered = 0, fwdCovered = 0, canCovered = 0;
    bool hasValidOr = true;
    bool isLongest = false;
    char blockLabel = '\0';
};

struct WindowData {                            // include/teloscope.h:119-137 (fields any writer reads)
    uint64_t windowStart = 0;
    uint32_t currentWindowSize = 0;
    uint32_t nucleotideCounts[4] = {0, 0, 0, 0};
    float gcContent = 0.0f;
    float shannonEntropy = 0.0f;
    uint32_t canonicalCovered = 0, nonCanonicalCovered = 0, fwdCovered = 0, revCovered = 0;
};

struct SegmentData {                           // include/teloscope.h:139-148
    std::vector<WindowData> windows;
    std::vector<TelomereBlock> terminalBlocks;
    std::vector<TelomereBlock> interstitialBlocks;
    std::vector<MatchInfo> canonicalMatches;
    std::vector<MatchInfo> nonCanonicalMatches;
    std::vector<MatchInfo> fwdMatches;
    std::vector<MatchInfo> revMatches;
    std::vector<MatchInfo> allMatches;
};

// src/main.cpp:287-296: the lexicographically smaller of (pattern, reverse complement) is "forward"
inline void setCanonical(UserInputTeloscope &ui, const std::string &canonical) {
    char f[64], r[64];
    if (ts_canonical_orientation(canonical.c_str(), f, r) != TS_OK) throw std::runtime_error("bad canonical pattern");
    ui.canonicalFwd = f;
    ui.canonicalRev = r;
    ui.canonicalSize = static_cast<unsigned short>(ui.canonicalFwd.size());
}

// src/tools.cpp:201-283
inline std::vector<std::pair<std::string, bool>> expandPatternsWithOrientation(
    const std::vector<std::string> &rawPatterns, uint8_t editDistance, const std::string &canonicalFwd) {
    std::string csv;
    for (size_t i = 0; i < rawPatterns.size(); ++i) { if (i) csv += ','; csv += rawPatterns[i]; }
    ts_pattern *arr = nullptr;
    size_t n = 0;
    if (ts_expand_patterns(csv.c_str(), editDistance, canonicalFwd.c_str(), &arr, &n) != TS_OK)
        throw std::runtime_error("expandPatternsWithOrientation: invalid arguments");
    std::vector<std::pair<std::string, bool>> out;
    out.reserve(n);
    for (size_t i = 0; i < n; ++i) out.emplace_back(std::string(arr[i].seq, arr[i].len), arr[i].is_forward != 0);
    ts_free_patterns(arr);
    return out;
}

// How the lines of a ts_text_piece lie, when they are regular (what a FASTA writer produces): `first` bases on the first
// line (a piece may be entered mid-line), then lines of `width` bases (the last one may be shorter), every line end `eol`
// bytes long.  width == 0: irregular — positions are found by walking the lines.
struct TextLines { uint32_t first = 0, width = 0, eol = 0; };

namespace detail {

struct CtxDeleter { void operator()(ts_ctx *c) const { ts_destroy(c); } };
using CtxPtr = std::unique_ptr<ts_ctx, CtxDeleter>;

inline ts_params makeParams(const UserInputTeloscope &ui) {
    ts_params p{};
    p.struct_size = sizeof p;
    p.window_size = ui.windowSize; p.step = ui.step; p.terminal_limit = ui.terminalLimit;
    p.max_match_dist = ui.maxMatchDist; p.min_block_len = ui.minBlockLen;
    p.max_block_dist = ui.maxBlockDist; p.min_block_counts = ui.minBlockCounts;
    p.min_block_density = ui.minBlockDensity; p.canonical_size = ui.canonicalSize;
    p.out_gc = ui.outGC; p.out_entropy = ui.outEntropy; p.out_matches = ui.outMatches; p.out_its = ui.outITS;
    p.fold_case = 1;                                   // every reference caller runs unmaskSequence first
    p.device = ui.device;
    return p;
}

inline std::vector<ts_pattern> makePatterns(UserInputTeloscope &ui) {
    if (ui.patternInfo.empty())
        ui.patternInfo = expandPatternsWithOrientation(ui.rawPatterns, ui.editDistance, ui.canonicalFwd);
    ui.patterns.clear();
    std::vector<ts_pattern> pats;
    for (const auto &pi : ui.patternInfo) {
        ui.patterns.push_back(pi.first);
        if (pi.first.size() > 63) throw std::runtime_error("pattern longer than 63 bases: more than the library's ts_pattern holds");
        ts_pattern t{};
        std::strncpy(t.seq, pi.first.c_str(), 63);
        t.len = static_cast<uint8_t>(pi.first.size());
        t.is_forward = pi.second;
        t.is_canonical = (pi.first == ui.canonicalFwd || pi.first == ui.canonicalRev);   // teloscope.h:243-244
        pats.push_back(t);
    }
    return pats;
}

inline TelomereBlock toBlock(const ts_block &b) {
    TelomereBlock t;
    t.start = b.start; t.blockLen = b.block_len; t.blockCounts = b.block_counts;
    t.forwardCount = b.forward_count; t.reverseCount = b.reverse_count; t.canonicalCount = b.canonical_count;
    t.nonCanonicalCount = b.non_canonical_count; t.totalCovered = b.total_covered;
    t.fwdCovered = b.fwd_covered; t.canCovered = b.can_covered;
    t.hasValidOr = b.has_valid_or != 0; t.isLongest = b.is_longest != 0; t.blockLabel = b.block_label;
    return t;
}

inline ts_block fromBlock(const TelomereBlock &t) {
    ts_block b{};
    b.start = t.start; b.block_len = t.blockLen; b.block_counts = t.blockCounts;
    b.forward_count = t.forwardCount; b.reverse_count = t.reverseCount; b.canonical_count = t.canonicalCount;
    b.non_canonical_count = t.nonCanonicalCount; b.total_covered = t.totalCovered;
    b.fwd_covered = t.fwdCovered; b.can_covered = t.canCovered;
    b.has_valid_or = t.hasValidOr; b.is_longest = t.isLongest; b.block_label = t.blockLabel;
    return b;
}

}  // namespace detail

class Teloscope {                              // include/teloscope.h:166-300 (scan path only)
public:
    struct Segment;
private:
    UserInputTeloscope userInput;
    detail::CtxPtr ctx;                         // the first device's context: every single-device entry point
    // one context per further device: the reference runs one thread-pool job per path (src/input.cpp:719-733); here a
    // batch of segments is cut into one shard per device (ts_scan_segments_multi) and each device uploads, scans and
    // downloads its shard over its own PCIe link
    std::vector<detail::CtxPtr> more;
    std::vector<ts_ctx *> all;

    // writerViewOnly: fill only what writeBEDFile reads (src/teloscope.cpp:700-868) — windows, blocks,
    // canonicalMatches and nonCanonicalMatches; the other three match vectors (all / fwd / rev, which only block
    // calling reads, and that has happened on the device) stay empty.  Of ~90 M matches of a 3 Gb assembly ~3 M are
    // written, and a MatchInfo is 56 bytes with a std::string.
    SegmentData convert(const ts_segment_out &o, const Segment &seg, bool writerViewOnly = false) const {
        const uint64_t absPos = seg.absPos;
        const bool tipsOnly = seg.tipsOnly;
        SegmentData sd;
        // WindowData mirrors ts_window field for field (the trailing `reserved` word is WindowData's padding): the
        // 6 M windows of a 3 Gb assembly are copied as bytes, not converted one by one
        static_assert(sizeof(WindowData) == sizeof(ts_window) && offsetof(WindowData, currentWindowSize) == offsetof(ts_window, current_window_size) &&
                      offsetof(WindowData, nucleotideCounts) == offsetof(ts_window, nucleotide_counts) && offsetof(WindowData, gcContent) == offsetof(ts_window, gc_content) &&
                      offsetof(WindowData, shannonEntropy) == offsetof(ts_window, shannon_entropy) && offsetof(WindowData, canonicalCovered) == offsetof(ts_window, canonical_covered) &&
                      offsetof(WindowData, nonCanonicalCovered) == offsetof(ts_window, non_canonical_covered) && offsetof(WindowData, fwdCovered) == offsetof(ts_window, fwd_covered) &&
                      offsetof(WindowData, revCovered) == offsetof(ts_window, rev_covered), "WindowData and ts_window must share their layout");
        sd.windows.resize(o.n_windows);
        if (o.n_windows) std::memcpy(static_cast<void *>(sd.windows.data()), o.windows, o.n_windows * sizeof(ts_window));
        if (!writerViewOnly) {
            if (!tipsOnly) sd.allMatches.reserve(o.n_matches);
            sd.fwdMatches.reserve(o.n_matches / 2 + 1);
            sd.revMatches.reserve(o.n_matches / 2 + 1);
        }
        for (uint64_t i = 0; i < o.n_matches; ++i) {
            const ts_match &g = o.matches[i];
            if (writerViewOnly && (tipsOnly || !(g.flags & (TS_MATCH_CANONICAL | TS_MATCH_TERMINAL)))) continue;
            MatchInfo m;
            m.position = g.position; m.matchSize = g.match_size;
            m.isForward = (g.flags & TS_MATCH_FORWARD) != 0;
            m.isCanonical = (g.flags & TS_MATCH_CANONICAL) != 0;
            if (userInput.outMatches && !tipsOnly)                      // src/teloscope.cpp:466-468 (of the unmasked sequence)
                m.matchSeq = seg.bases(g.position - absPos, g.match_size);
            if (!writerViewOnly) (m.isForward ? sd.fwdMatches : sd.revMatches).push_back(m);
            if (!tipsOnly) {                                            // routing of src/teloscope.cpp:485-509
                if (!writerViewOnly) sd.allMatches.push_back(m);
                if (m.isCanonical) sd.canonicalMatches.push_back(m);
                else if (g.flags & TS_MATCH_TERMINAL) sd.nonCanonicalMatches.push_back(m);
            }
        }
        for (uint64_t i = 0; i < o.n_terminal_blocks; ++i) sd.terminalBlocks.push_back(detail::toBlock(o.terminal_blocks[i]));
        for (uint64_t i = 0; i < o.n_interstitial_blocks; ++i) sd.interstitialBlocks.push_back(detail::toBlock(o.interstitial_blocks[i]));
        return sd;
    }

public:
    // devices: HIP ordinals, one context each (an ordinal may repeat: several contexts share that GPU);
    // empty = one context on userInput.device
    explicit Teloscope(UserInputTeloscope ui, const std::vector<int> &devices = {}) : userInput(std::move(ui)) {
        std::vector<ts_pattern> pats = detail::makePatterns(userInput);
        const std::vector<int> devs = devices.empty() ? std::vector<int>{userInput.device} : devices;
        for (size_t i = 0; i < devs.size(); ++i) {
            UserInputTeloscope u = userInput;
            u.device = devs[i];
            ts_params p = detail::makeParams(u);
            detail::CtxPtr c(ts_create(&p, pats.data(), pats.size()));
            if (!c) throw std::runtime_error(ts_last_error(nullptr));
            all.push_back(c.get());
            if (i == 0) ctx = std::move(c); else more.push_back(std::move(c));
        }
    }

    size_t deviceCount() const { return all.size(); }

    const UserInputTeloscope &input() const { return userInput; }
    // does the library take TS_INPUT_TEXT_PIECES segments for this parameter set (every set it scans)?
    bool takesTextPieces() const { return ts_takes_text_input(ctx.get(), userInput.ultraFastMode ? 1 : 0) != 0; }
    // keeps the calling thread (and the threads it starts) on the CPUs of the device's NUMA node: ts_bind_thread_to_device
    bool bindThreadToDevice() const { return ts_bind_thread_to_device(ctx.get()) != 0; }

    // one scanSegment call of a batch; the bases are borrowed for the duration of the call (any case:
    // the library folds case itself, as unmaskSequence would have)
    struct Segment {
        const char *data;                      // the bases — or nullptr when `pieces` is set
        size_t size;
        uint64_t absPos;
        bool tipsOnly;
        const ts_text_piece *pieces = nullptr; // FASTA body text as it lies in the file (TS_INPUT_TEXT_PIECES): the
                                               // library skips the line ends on the way to the device
        const TextLines *lines = nullptr;      // per piece, optional: lets bases() jump to a position instead of walking lines
        mutable size_t cursorPiece = 0;        // bases() is asked for ascending positions: where the last answer lay
        mutable uint64_t cursorCum = 0;
        Segment(const char *d, size_t n, uint64_t a, bool t) : data(d), size(n), absPos(a), tipsOnly(t) {}
        Segment(const std::string *s, uint64_t a, bool t) : data(s->data()), size(s->size()), absPos(a), tipsOnly(t) {}
        size_t nPieces = 0;                    // entries of `pieces`
        Segment(const ts_text_piece *p, size_t np, size_t nBases, uint64_t a, bool t, const TextLines *l = nullptr)
            : data(nullptr), size(nBases), absPos(a), tipsOnly(t), pieces(p), lines(l), nPieces(np) {}
        ts_segment_in in() const {
            ts_segment_in x{};
            x.seq = pieces ? reinterpret_cast<const char *>(pieces) : data;
            x.len = size; x.abs_pos = absPos; x.tips_only = static_cast<uint8_t>(tipsOnly);
            x.input_format = pieces ? TS_INPUT_TEXT_PIECES : TS_INPUT_BASES;
            x.n_pieces = static_cast<uint32_t>(nPieces);
            return x;
        }
        // bases [pos, pos + n) of the segment, upper-cased (matchSeq, src/teloscope.cpp:466-468)
        std::string bases(uint64_t pos, size_t n) const {
            std::string out;
            out.reserve(n);
            if (!pieces) out.assign(data + pos, n);
            else {
                if (pos < cursorCum) { cursorPiece = 0; cursorCum = 0; }
                size_t k = cursorPiece;
                uint64_t cum = cursorCum;
                while (pos >= cum + pieces[k].n_bases) cum += pieces[k++].n_bases;
                cursorPiece = k; cursorCum = cum;
                uint64_t skip = pos - cum;
                for (; out.size() < n; ++k, skip = 0) {
                    const ts_text_piece &p = pieces[k];
                    const char *q = p.text, *end = p.text + p.text_len;
                    if (skip && lines && lines[k].width) {                    // regular lines: jump to the position
                        const TextLines &L = lines[k];
                        if (skip >= L.first) {
                            const uint64_t r = skip - L.first;
                            q += L.first + L.eol + (r / L.width) * (uint64_t(L.width) + L.eol) + r % L.width;
                            skip = 0;
                            // (q may sit at a line's end when r % width == 0 and the line is the last, shorter one: the walk below moves on)
                        }
                    }
                    while (q < end && out.size() < n) {                       // line by line
                        const char *nl = static_cast<const char *>(std::memchr(q, '\n', static_cast<size_t>(end - q)));
                        const char *stop = nl ? nl : end;
                        uint64_t line = static_cast<uint64_t>(stop - q);
                        if (line && stop[-1] == '\r') --line;
                        if (skip < line) { out.append(q + skip, static_cast<size_t>(std::min<uint64_t>(line - skip, n - out.size()))); skip = 0; }
                        else skip -= line;
                        q = nl ? nl + 1 : end;
                    }
                }
            }
            for (char &ch : out)
                if (ch >= 'a' && ch <= 'z') ch = static_cast<char>(ch - 32);
            return out;
        }
    };

    // batched scanSegment: result[i] is what scanSegment(*segs[i].sequence, absPos, tipsOnly) returns
    // (writerViewOnly: see convert())
    std::vector<SegmentData> scanSegments(const std::vector<Segment> &segs, bool writerViewOnly = false) {
        std::vector<ts_segment_in> in(segs.size());
        for (size_t i = 0; i < segs.size(); ++i) in[i] = segs[i].in();
        std::vector<ts_segment_out> out(segs.size());
        if (ts_scan_segments(ctx.get(), in.data(), in.size(), out.data()) != TS_OK)
            throw std::runtime_error(ts_last_error(ctx.get()));
        // SegmentData's five MatchInfo vectors (56-byte records holding a std::string) are the costly part
        // of the mirror: segments are converted on up to 16 host threads, largest first
        std::vector<SegmentData> res(segs.size());
        std::vector<size_t> order(segs.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return out[a].n_matches > out[b].n_matches; });
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t k; (k = next.fetch_add(1)) < order.size();) {
                const size_t i = order[k];
                res[i] = convert(out[i], segs[i], writerViewOnly);
            }
        };
        const unsigned nt = static_cast<unsigned>(std::min<size_t>({size_t(16), segs.size(), size_t(std::max(1u, std::thread::hardware_concurrency()))}));
        if (nt <= 1) {
            worker();
        } else {
            std::vector<std::thread> pool;
            for (unsigned i = 0; i < nt; ++i) pool.emplace_back(worker);
            for (std::thread &th : pool) th.join();
        }
        ts_free_segments(out.data(), out.size());
        return res;
    }

    // scanSegments for callers that do not read the match vectors (every run without -m): scan, block
    // calling and counting on the device (ts_scan_segments_blocks).  result[i] has windows and blocks;
    // counts[i] = the sizes the match vectors would have had.
    std::vector<SegmentData> scanSegmentsNoMatches(const std::vector<Segment> &segs, std::vector<ts_segment_counts> &counts) {
        std::vector<ts_segment_in> in(segs.size());
        for (size_t i = 0; i < segs.size(); ++i) in[i] = segs[i].in();
        std::vector<ts_segment_out> out(segs.size());
        counts.assign(segs.size(), ts_segment_counts{0, 0, 0, 0});
        if (ts_scan_segments_blocks(ctx.get(), in.data(), in.size(), out.data(), counts.data()) != TS_OK)
            throw std::runtime_error(ts_last_error(ctx.get()));
        // (window records of a long contig are millions: segments are converted on up to 16 host threads)
        std::vector<SegmentData> res(segs.size());
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t i; (i = next.fetch_add(1)) < segs.size();) res[i] = convert(out[i], segs[i]);
        };
        const unsigned nt = static_cast<unsigned>(std::min<size_t>({size_t(16), segs.size(), size_t(std::max(1u, std::thread::hardware_concurrency()))}));
        if (nt <= 1) {
            worker();
        } else {
            std::vector<std::thread> pool;
            for (unsigned i = 0; i < nt; ++i) pool.emplace_back(worker);
            for (std::thread &th : pool) th.join();
        }
        ts_free_segments(out.data(), out.size());
        return res;
    }

    // scanSegments for a writer (src/teloscope.cpp:700-868 reads a path's windows, blocks, canonicalMatches and terminal
    // nonCanonicalMatches, nothing else), over ALL the object's devices: ts_scan_segments_multi cuts the batch into one
    // shard per device; block calling happens on the devices and only that view comes back.  result[i] has windows,
    // blocks and those two match vectors (the other three stay empty); counts[i] = the sizes the match vectors had.
    std::vector<SegmentData> scanSegmentsWriterView(const std::vector<Segment> &segs, std::vector<ts_segment_counts> &counts) {
        std::vector<ts_segment_in> in(segs.size());
        for (size_t i = 0; i < segs.size(); ++i) in[i] = segs[i].in();
        std::vector<ts_segment_out> out(segs.size());
        counts.assign(segs.size(), ts_segment_counts{0, 0, 0, 0});
        if (ts_scan_segments_multi(all.data(), all.size(), in.data(), in.size(), out.data(), counts.data()) != TS_OK)
            throw std::runtime_error(ts_last_error(ctx.get()));
        std::vector<SegmentData> res(segs.size());
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t i; (i = next.fetch_add(1)) < segs.size();) res[i] = convert(out[i], segs[i], true);
        };
        const unsigned nt = static_cast<unsigned>(std::min<size_t>({size_t(16), segs.size(), size_t(std::max(1u, std::thread::hardware_concurrency()))}));
        if (nt <= 1) {
            worker();
        } else {
            std::vector<std::thread> pool;
            for (unsigned i = 0; i < nt; ++i) pool.emplace_back(worker);
            for (std::thread &th : pool) th.join();
        }
        ts_free_segments(out.data(), out.size());
        return res;
    }

    // SegmentData Teloscope::scanSegment(std::string &sequence, uint64_t absPos, bool tipsOnly)
    SegmentData scanSegment(std::string &sequence, uint64_t absPos, bool tipsOnly) {
        return std::move(scanSegments({Segment{&sequence, absPos, tipsOnly}})[0]);
    }

    // void Teloscope::labelTerminalBlocks(blocks, gaps, terminalLabel, scaffoldType, pathSize, terminalLimit)
    void labelTerminalBlocks(std::vector<TelomereBlock> &blocks, uint16_t gaps, std::string &terminalLabel,
                             ScaffoldType &scaffoldType, uint64_t pathSize, uint32_t terminalLimit) {
        std::vector<ts_block> raw;
        for (const TelomereBlock &b : blocks) raw.push_back(detail::fromBlock(b));
        std::string label(2 * raw.size() + 2, '\0');
        int st = 0;
        if (ts_label_terminal_blocks(raw.data(), raw.size(), gaps, pathSize, terminalLimit, &label[0], &st) != TS_OK)
            throw std::runtime_error("labelTerminalBlocks failed");
        terminalLabel = label.c_str();
        scaffoldType = static_cast<ScaffoldType>(st);
        for (size_t i = 0; i < raw.size(); ++i) blocks[i] = detail::toBlock(raw[i]);
    }
};

class ReadTelomereFilter {                     // include/read-filter.h:10-18
    // one context per device: the reference makes one ReadTelomereFilter per thread-pool job and deals a batch's
    // records to the jobs in chunks (src/input.cpp:780-793); here a batch is dealt to the devices in consecutive
    // shards of equal bases (ts_filter_reads_multi) and the pass bits come back in input order
    std::vector<detail::CtxPtr> ctxs;
    std::vector<ts_ctx *> raw;

public:
    // devices: HIP ordinals, one context each (an ordinal may repeat: several contexts share that GPU);
    // empty = one context on the current device
    explicit ReadTelomereFilter(const UserInputTeloscope &input, const std::vector<int> &devices = {}) {
        UserInputTeloscope ui = input;
        std::vector<ts_pattern> pats = detail::makePatterns(ui);
        const std::vector<int> devs = devices.empty() ? std::vector<int>{ui.device} : devices;
        for (int d : devs) {
            ui.device = d;
            ts_params p = detail::makeParams(ui);
            detail::CtxPtr c(ts_create_read_filter(&p, ui.minBlockLenSet ? 1 : 0, pats.data(), pats.size()));
            if (!c) throw std::runtime_error(ts_last_error(nullptr));
            raw.push_back(c.get());
            ctxs.push_back(std::move(c));
        }
    }

    size_t deviceCount() const { return raw.size(); }

    std::vector<bool> matchesBatch(const std::vector<std::string> &sequences) {
        std::vector<const char *> ptr(sequences.size());
        std::vector<uint64_t> len(sequences.size());
        for (size_t i = 0; i < sequences.size(); ++i) { ptr[i] = sequences[i].data(); len[i] = sequences[i].size(); }
        std::vector<uint8_t> pass(sequences.size());
        matchesPointers(ptr.data(), len.data(), sequences.size(), pass.data());
        return std::vector<bool>(pass.begin(), pass.end());
    }

    // the same on borrowed buffers (no copies): pass[i] = matches(std::string(seqs[i], lens[i]))
    void matchesPointers(const char *const *seqs, const uint64_t *lens, size_t n, uint8_t *pass) {
        if (ts_filter_reads_multi(raw.data(), raw.size(), seqs, lens, n, pass) != TS_OK)
            throw std::runtime_error(ts_last_error(raw[0]));
    }

    // bool ReadTelomereFilter::matches(std::string sequence)
    bool matches(std::string sequence) { return matchesBatch({std::move(sequence)})[0]; }
    // keeps the calling thread (and the threads it starts) on the CPUs of the first device's NUMA node
    bool bindThreadToDevice() const { return !raw.empty() && ts_bind_thread_to_device(raw[0]) != 0; }
};

}  // namespace teloscope_mi355x

#endif
