// teloscope_mi355x_io.hpp — the callers and writers either side of the scan path (SURVEY §8 row f2,
// and the part of f3 the writers need), host-side C++17 above teloscope_mi355x.hpp:
//
//   splitPath / walkPaths   what Teloscope::walkPath does per path (src/input.cpp:942-1041): cut a
//                           record into '+' segments and N-gaps, scan, concatenate, label — but with
//                           ONE batched scanSegments call for every segment of every path, which is
//                           what a GPU back end needs;
//   writeBEDFiles           the eleven output files of handleBEDFile / writeBEDFile
//                           (src/teloscope.cpp:661-957; formats in docs/outputs.md) and the console
//                           path report;
//   printSummary            the assembly summary (src/teloscope.cpp:959-1055), to console and report.
//
// Once the scan takes milliseconds, formatting 6-15 M lines x 5 files is the end-to-end bottleneck, so
// lines are formatted by a pool of host threads into per-task buffers (std::to_chars: integers, and
// floats exactly as operator<<(float) prints them — "%g" with 6 significant digits) and written in
// path order.  Output is byte-identical for any thread count.
#ifndef TELOSCOPE_MI355X_IO_HPP
#define TELOSCOPE_MI355X_IO_HPP

#include <algorithm>
#include <array>
#include <atomic>
#include <charconv>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <ostream>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "teloscope_mi355x.hpp"

namespace teloscope_mi355x {

struct GapInfo {                               // include/teloscope.h:97-100
    uint64_t start = 0;
    uint32_t length = 0;
};

struct PathData {                              // include/teloscope.h:151-163
    unsigned int seqPos = 0;
    std::string header;
    std::vector<GapInfo> gapInfos;
    uint64_t pathSize = 0;
    std::vector<WindowData> windows;
    std::vector<TelomereBlock> terminalBlocks;
    std::vector<TelomereBlock> interstitialBlocks;
    std::vector<MatchInfo> canonicalMatches;
    std::vector<MatchInfo> nonCanonicalMatches;
    std::string terminalLabel;
    ScaffoldType scaffoldType = ScaffoldType::NONE;
    // not in the reference: canonicalMatches.size() as the report prints it, also when the match
    // vectors were never brought to the host (walkPaths without -m counts on the device)
    uint64_t canonicalMatchCount = 0;
};

struct FastaRecord {
    std::string header;                        // first word after '>'
    std::string sequence;
};

// FASTA, plain or gzip-compressed (zlib reads both through the same calls; link with -lz).  The header
// is the first word after '>', line ends and '\r' are dropped, as gfalibs does.  The FASTQ/BAM front
// ends are rows f3/f4.
namespace detail {
// the body lines of one record, [p, end), joined into `seq` without their line ends
inline void joinFastaLines(const char *p, const char *end, std::string &seq) {
    seq.reserve(static_cast<size_t>(end - p));
    while (p < end) {
        const char *nl = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
        const char *stop = nl ? nl : end;
        if (stop > p) seq.append(p, stop[-1] == '\r' ? stop - 1 : stop);
        p = nl ? nl + 1 : end;
    }
    seq.shrink_to_fit();
}
inline std::string fastaHeaderWord(const char *b, const char *e) {
    if (e > b && e[-1] == '\r') --e;
    const char *w = b;
    while (w < e && *w != ' ' && *w != '\t') ++w;
    return std::string(b, w);
}
}  // namespace detail

inline std::vector<FastaRecord> readFasta(const std::string &file) {
    // A regular uncompressed file is mapped: one pass finds the records ('>' at a line start), then the lines of
    // every record are joined by a pool of threads, largest record first.  Anything else goes through zlib below.
    {
        const int fd = ::open(file.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open " + file);
        struct Close { int fd; ~Close() { ::close(fd); } } closer{fd};
        struct stat sb;
        unsigned char magic[2] = {0, 0};
        const bool gz = ::pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        if (!gz && ::fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
            const size_t size = static_cast<size_t>(sb.st_size);
            void *map = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (map != MAP_FAILED) {
                struct Unmap { void *p; size_t n; ~Unmap() { ::munmap(p, n); } } unmap{map, size};
                const char *data = static_cast<const char *>(map), *end = data + size;
                struct Span { const char *head, *body, *stop; };
                std::vector<Span> spans;
                for (const char *p = data; p < end;) {
                    const char *gt = static_cast<const char *>(std::memchr(p, '>', static_cast<size_t>(end - p)));
                    if (!gt) break;
                    p = gt + 1;
                    if (gt != data && gt[-1] != '\n') continue;            // a '>' inside a line
                    if (!spans.empty()) spans.back().stop = gt;
                    const char *nl = static_cast<const char *>(std::memchr(gt, '\n', static_cast<size_t>(end - gt)));
                    spans.push_back(Span{gt + 1, nl ? nl + 1 : end, end});
                    if (!nl) break;
                    p = nl + 1;
                }
                std::vector<FastaRecord> recs(spans.size());
                std::vector<size_t> order(spans.size());
                for (size_t i = 0; i < order.size(); ++i) order[i] = i;
                std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return spans[a].stop - spans[a].body > spans[b].stop - spans[b].body; });
                std::atomic<size_t> next{0};
                auto worker = [&]() {
                    for (size_t k; (k = next.fetch_add(1)) < order.size();) {
                        const Span &sp = spans[order[k]];
                        const char *he = sp.body > sp.head && sp.body[-1] == '\n' ? sp.body - 1 : sp.body;
                        recs[order[k]].header = detail::fastaHeaderWord(sp.head, he);
                        detail::joinFastaLines(sp.body, sp.stop, recs[order[k]].sequence);
                    }
                };
                const unsigned nthr = static_cast<unsigned>(std::min<size_t>(std::min<size_t>(16, spans.size()), std::max(1u, std::thread::hardware_concurrency())));
                if (nthr <= 1) worker();
                else {
                    std::vector<std::thread> pool;
                    for (unsigned t = 0; t < nthr; ++t) pool.emplace_back(worker);
                    for (std::thread &th : pool) th.join();
                }
                return recs;
            }
        }
    }
    gzFile in = gzopen(file.c_str(), "rb");
    if (!in) throw std::runtime_error("cannot open " + file);
    gzbuffer(in, 1u << 20);
    std::vector<FastaRecord> recs;
    std::vector<char> buf(1u << 22);
    bool in_header = false, at_line_start = true;
    std::string header_line;
    for (;;) {
        const int n = gzread(in, buf.data(), static_cast<unsigned>(buf.size()));
        if (n < 0) { gzclose(in); throw std::runtime_error("read error in " + file); }
        if (n == 0) break;
        const char *p = buf.data(), *end = p + n;
        while (p < end) {
            if (in_header) {                                    // collect the header line
                const char *nl = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
                header_line.append(p, nl ? nl : end);
                if (!nl) break;
                if (!header_line.empty() && header_line.back() == '\r') header_line.pop_back();
                FastaRecord r;
                const size_t e = header_line.find_first_of(" \t");
                r.header = header_line.substr(0, e);
                recs.push_back(std::move(r));
                header_line.clear();
                in_header = false; at_line_start = true;
                p = nl + 1;
            } else if (at_line_start && *p == '>') {
                in_header = true;
                ++p;
            } else {                                            // sequence bytes up to the end of the line
                const char *nl = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
                const char *stop = nl ? nl : end;
                if (!recs.empty() && stop > p) {
                    std::string &seq = recs.back().sequence;
                    const size_t before = seq.size();
                    seq.append(p, stop);
                    if (seq.size() > before && seq.back() == '\r') seq.pop_back();
                }
                at_line_start = nl != nullptr;
                p = nl ? nl + 1 : end;
            }
        }
    }
    gzclose(in);
    return recs;
}

// A record as the path model sees it: every run of N/n (X/x) is one gap, what lies between is a '+'
// segment.  Segments are views into the record (no copy; the library folds case itself, which is what
// unmaskSequence does before every reference scanSegment call).  Pinned by testFiles/expected/*_gaps.bed.
struct PathComponents {
    std::vector<std::pair<uint64_t, uint64_t>> segments;       // (absPos = offset in the record, length)
    std::vector<GapInfo> gaps;
};

namespace detail {
// first position in [i, n) whose "is a gap letter" (N n X x) state equals `want`; n if none.  Eight bytes at
// a time: with the case bit set, a gap letter is 'n' or 'x', and (v ^ pattern) has a zero byte exactly there.
inline size_t scanGapState(const char *s, size_t i, size_t n, bool want) {
    auto isGap = [](char c) { return c == 'N' || c == 'n' || c == 'X' || c == 'x'; };
    const uint64_t ones = 0x0101010101010101ull, highs = 0x8080808080808080ull;
    const uint64_t low7 = 0x7F7F7F7F7F7F7F7Full;
    auto zeroBytes = [&](uint64_t v) { return ~(((v & low7) + low7) | v | low7); };            // 0x80 in every zero byte, exactly
    while (i < n && (reinterpret_cast<uintptr_t>(s + i) & 7u)) { if (isGap(s[i]) == want) return i; ++i; }
    for (; i + 8 <= n; i += 8) {
        uint64_t v;
        std::memcpy(&v, s + i, 8);
        v |= 0x2020202020202020ull;
        const uint64_t hit = zeroBytes(v ^ (ones * 'n')) | zeroBytes(v ^ (ones * 'x'));
        // want = true: stop at a word that holds a gap letter; want = false: at one that holds anything else
        if (want ? hit != 0 : hit != highs) break;
    }
    while (i < n && isGap(s[i]) != want) ++i;
    return i;
}
}  // namespace detail

inline PathComponents splitPath(const char *s, size_t n) {
    PathComponents pc;
    size_t i = 0;
    while (i < n) {
        const bool gap = s[i] == 'N' || s[i] == 'n' || s[i] == 'X' || s[i] == 'x';
        const size_t j = detail::scanGapState(s, i + 1, n, !gap);   // the run ends where the state flips
        if (gap) pc.gaps.push_back(GapInfo{i, static_cast<uint32_t>(j - i)});
        else pc.segments.emplace_back(i, j - i);
        i = j;
    }
    return pc;
}
inline PathComponents splitPath(const std::string &seq) { return splitPath(seq.data(), seq.size()); }

// A record as walkPaths sees it: header + bases, wherever they live (a FastaRecord's std::string, or a buffer the
// streaming reader filled).
struct RecordView {
    const std::string *header;
    const char *data;                          // the bases, or nullptr when `pieces` is set
    size_t size;                               // bases
    const ts_text_piece *pieces = nullptr;     // FASTA body text in the file (line ends included), in order
    size_t nPieces = 0;
    const TextLines *lines = nullptr;          // per piece, optional
};

namespace detail {
// text position of base `skip` of a run of FASTA body text (line ends are not bases)
inline const char *textLocate(const char *text, uint64_t textLen, uint64_t skip) {
    const char *p = text, *end = text + textLen;
    while (p < end) {
        const char *nl = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
        const char *stop = nl ? nl : end;
        uint64_t line = static_cast<uint64_t>(stop - p);
        if (line && stop[-1] == '\r') --line;
        if (skip < line) return p + skip;
        skip -= line;
        p = nl ? nl + 1 : end;
    }
    return end;
}

// splitPath over FASTA body text [p, end) that starts at a line start: runs in BASE coordinates (line ends are
// skipped, a run goes on across them); *nBases = the bases of the text; firstIsGap / lastIsGap as splitPaths needs them;
// *lines (optional) = how the lines lie, if they are regular (TextLines)
inline PathComponents splitPathText(const char *p, const char *end, uint64_t *nBases, bool *firstIsGap, bool *lastIsGap,
                                    TextLines *lines = nullptr) {
    PathComponents pc;
    auto isGap = [](char c) { return c == 'N' || c == 'n' || c == 'X' || c == 'x'; };
    uint64_t base = 0, runStart = 0;
    bool have = false, runGap = false;
    *firstIsGap = *lastIsGap = false;
    auto close = [&](uint64_t at) {
        if (!have || at == runStart) return;
        if (runGap) pc.gaps.push_back(GapInfo{runStart, static_cast<uint32_t>(at - runStart)});
        else pc.segments.emplace_back(runStart, at - runStart);
    };
    // line structure: `first` bases on the first line, `width` on every line between the first and the last, the last
    // at most `width`, every line end `eol` bytes
    uint64_t nLines = 0, first = 0, width = 0, eol = 0, prevLen = 0, prevEol = 0;
    bool regular = true;
    while (p < end) {
        const char *nl = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
        const char *stop = nl ? nl : end;
        size_t n = static_cast<size_t>(stop - p);
        const bool cr = n && stop[-1] == '\r';
        if (cr) --n;
        const uint64_t thisEol = (nl ? 1u : 0u) + (cr ? 1u : 0u);
        if (nLines == 0) { first = n; eol = thisEol; }
        else {
            if (prevEol != eol) regular = false;
            if (nLines == 1) width = n;
            else if (prevLen != width) regular = false;                     // the previous line turned out to be a middle one
        }
        prevLen = n; prevEol = thisEol; ++nLines;
        size_t i = 0;
        while (i < n) {
            const bool gap = isGap(p[i]);
            if (!have) { have = true; runGap = gap; runStart = base; *firstIsGap = gap; }
            else if (gap != runGap) { close(base + i); runGap = gap; runStart = base + i; }
            i = scanGapState(p, i + 1, n, !gap);                            // the run goes on to where the state flips (or the line ends)
        }
        if (n) *lastIsGap = isGap(p[n - 1]);
        base += n;
        p = nl ? nl + 1 : end;
    }
    close(base);
    *nBases = base;
    if (lines) {
        *lines = TextLines{};
        if (nLines >= 3 && prevLen > width) regular = false;
        if (nLines == 1) width = first;
        if (width == 0) width = 1;
        if (eol == 0) eol = 1;
        if (regular && nLines >= 1 && first < (1u << 31) && width < (1u << 31))
            *lines = TextLines{static_cast<uint32_t>(first), static_cast<uint32_t>(width), static_cast<uint32_t>(eol)};
    }
    return pc;
}

// appends the runs of a piece (already in record coordinates) to the record's components; when the piece continues
// a record and its first run is of the kind the previous piece ended with, the two are one run
inline void appendPieceRuns(PathComponents &out, const PathComponents &pc, bool continues, bool prevLastIsGap, bool firstIsGap) {
    size_t si = 0, gi = 0;
    if (continues && prevLastIsGap == firstIsGap) {
        if (firstIsGap) { out.gaps.back().length += pc.gaps[0].length; gi = 1; }
        else { out.segments.back().second += pc.segments[0].second; si = 1; }
    }
    out.segments.insert(out.segments.end(), pc.segments.begin() + static_cast<long>(si), pc.segments.end());
    out.gaps.insert(out.gaps.end(), pc.gaps.begin() + static_cast<long>(gi), pc.gaps.end());
}
}  // namespace detail

// splitPath for every record of a group, on the host threads: records are cut into pieces (16 MB) that are scanned
// independently — one 250 Mb chromosome keeps all threads busy — and a run that crosses a cut is put together again.
inline std::vector<PathComponents> splitPaths(const std::vector<RecordView> &records, size_t pieceBytes = size_t(16) << 20) {
    std::vector<PathComponents> comps(records.size());
    struct Piece { size_t rec, a, z; PathComponents pc; bool firstIsGap, lastIsGap; };
    std::vector<Piece> pieces;
    pieceBytes = std::max<size_t>(pieceBytes, 1);
    for (size_t r = 0; r < records.size(); ++r)
        for (size_t a = 0; a < records[r].size; a += pieceBytes) pieces.push_back(Piece{r, a, std::min(records[r].size, a + pieceBytes), {}, false, false});
    std::atomic<size_t> next{0};
    auto worker = [&]() {
        for (size_t i; (i = next.fetch_add(1)) < pieces.size();) {
            Piece &p = pieces[i];
            p.pc = splitPath(records[p.rec].data + p.a, p.z - p.a);
            for (auto &sg : p.pc.segments) sg.first += p.a;
            for (GapInfo &g : p.pc.gaps) g.start += p.a;
            auto isGap = [](char c) { return c == 'N' || c == 'n' || c == 'X' || c == 'x'; };
            p.firstIsGap = isGap(records[p.rec].data[p.a]);
            p.lastIsGap = isGap(records[p.rec].data[p.z - 1]);
        }
    };
    const unsigned nt = static_cast<unsigned>(std::min<size_t>({size_t(16), pieces.size(), size_t(std::max(1u, std::thread::hardware_concurrency()))}));
    if (nt <= 1) worker();
    else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; ++t) pool.emplace_back(worker);
        for (std::thread &th : pool) th.join();
    }
    for (size_t i = 0; i < pieces.size(); ++i) {
        Piece &p = pieces[i];
        detail::appendPieceRuns(comps[p.rec], p.pc, p.a != 0, p.a != 0 && pieces[i - 1].lastIsGap, p.firstIsGap);
    }
    return comps;
}

// walkPath for every record, with one batched scan (result order = record order; seqPos = seqPosBase + index).
inline std::vector<PathData> walkRecordViews(Teloscope &teloscope, const std::vector<RecordView> &records, size_t seqPosBase = 0,
                                             const std::vector<PathComponents> *precomputed = nullptr) {
    const UserInputTeloscope &ui = teloscope.input();
    // (the streaming reader finds the N-runs while the freshly joined bases are still in cache)
    const std::vector<PathComponents> split = precomputed ? std::vector<PathComponents>() : splitPaths(records);
    const std::vector<PathComponents> &comps = precomputed ? *precomputed : split;
    std::vector<Teloscope::Segment> batch;
    // text records that N-runs cut into several segments: every segment gets its own piece list (reserved up front, so
    // the lists do not move while the batch points into them)
    auto oneSegment = [&](size_t pi) {
        return comps[pi].segments.size() == 1 && comps[pi].segments[0].first == 0 && comps[pi].segments[0].second == records[pi].size;
    };
    size_t nSub = 0;
    for (size_t pi = 0; pi < records.size(); ++pi)
        if (records[pi].pieces && !oneSegment(pi)) nSub += comps[pi].segments.size();
    std::vector<std::vector<ts_text_piece>> subPieces;
    std::vector<std::vector<TextLines>> subLines;
    subPieces.reserve(nSub);
    subLines.reserve(nSub);
    for (size_t pi = 0; pi < records.size(); ++pi) {
        const RecordView &rv = records[pi];
        if (!rv.pieces) {
            for (const auto &sg : comps[pi].segments)
                batch.emplace_back(rv.data + sg.first, static_cast<size_t>(sg.second), sg.first, ui.ultraFastMode);
            continue;
        }
        if (oneSegment(pi)) {
            batch.emplace_back(rv.pieces, rv.nPieces, rv.size, 0, ui.ultraFastMode, rv.lines);   // its pieces as they are
            continue;
        }
        // a segment = bases [a, a + n) of the record: the text pieces that hold them, the first one entered at base a
        uint64_t cum = 0;
        size_t k = 0;
        for (const auto &sg : comps[pi].segments) {
            while (k < rv.nPieces && cum + rv.pieces[k].n_bases <= sg.first) cum += rv.pieces[k++].n_bases;
            subPieces.emplace_back();
            subLines.emplace_back();
            std::vector<ts_text_piece> &sp = subPieces.back();
            std::vector<TextLines> &sl = subLines.back();
            uint64_t at = sg.first, left = sg.second, c2 = cum;
            for (size_t q = k; left && q < rv.nPieces; ++q) {
                ts_text_piece t = rv.pieces[q];
                TextLines L = rv.lines ? rv.lines[q] : TextLines{};
                const uint64_t skip = at - c2;
                if (skip) {
                    const char *from = detail::textLocate(t.text, t.text_len, skip);
                    t.text_len -= static_cast<uint64_t>(from - t.text); t.text = from; t.n_bases -= skip;
                    if (L.width) {                                                 // entered mid-line: what is left of that line comes first
                        const char *nl = static_cast<const char *>(std::memchr(from, '\n', static_cast<size_t>(t.text_len)));
                        uint64_t rest = static_cast<uint64_t>((nl ? nl : from + t.text_len) - from);
                        if (rest && from[rest - 1] == '\r') --rest;
                        L.first = static_cast<uint32_t>(rest);
                    }
                }
                const uint64_t n = std::min<uint64_t>(t.n_bases, left);
                t.n_bases = n;                                                     // (of the last piece only what is needed)
                sp.push_back(t);
                sl.push_back(L);
                at += n; left -= n; c2 += rv.pieces[q].n_bases;
            }
            batch.emplace_back(sp.data(), sp.size(), static_cast<size_t>(sg.second), sg.first, ui.ultraFastMode, sl.data());
        }
    }
    // without -m nothing downstream reads a match record: blocks and counts come from the device
    std::vector<ts_segment_counts> counts;
    // (with -m: only the two match vectors the writers read are materialised; block calling has happened on the device)
    // (several devices: the batch is cut into one shard per device, and what comes back is the writers' view either way)
    std::vector<SegmentData> scanned = teloscope.deviceCount() > 1 ? teloscope.scanSegmentsWriterView(batch, counts)
                                     : ui.outMatches ? teloscope.scanSegments(batch, true)
                                                     : teloscope.scanSegmentsNoMatches(batch, counts);
    const bool haveCounts = teloscope.deviceCount() > 1 || !ui.outMatches;

    std::vector<PathData> paths(records.size());
    size_t si = 0;
    for (size_t pi = 0; pi < records.size(); ++pi) {
        PathData &pd = paths[pi];
        pd.seqPos = static_cast<unsigned int>(seqPosBase + pi);
        pd.header = *records[pi].header;
        pd.pathSize = records[pi].size;
        pd.gapInfos = comps[pi].gaps;
        const size_t nseg = comps[pi].segments.size();
        if (nseg == 1) {                                        // the common case: the record is one segment — no copies
            SegmentData &sd = scanned[si];
            pd.canonicalMatchCount = haveCounts ? counts[si].n_canonical : sd.canonicalMatches.size();
            pd.windows = std::move(sd.windows);
            pd.terminalBlocks = std::move(sd.terminalBlocks);
            pd.interstitialBlocks = std::move(sd.interstitialBlocks);
            pd.canonicalMatches = std::move(sd.canonicalMatches);
            pd.nonCanonicalMatches = std::move(sd.nonCanonicalMatches);
            ++si;
        } else {
            for (size_t k = 0; k < nseg; ++k, ++si) {
                SegmentData &sd = scanned[si];
                auto append = [](auto &dst, auto &src) {
                    dst.insert(dst.end(), std::make_move_iterator(src.begin()), std::make_move_iterator(src.end()));
                };
                append(pd.windows, sd.windows);
                append(pd.terminalBlocks, sd.terminalBlocks);
                append(pd.interstitialBlocks, sd.interstitialBlocks);
                pd.canonicalMatchCount += haveCounts ? counts[si].n_canonical : sd.canonicalMatches.size();
                append(pd.canonicalMatches, sd.canonicalMatches);
                append(pd.nonCanonicalMatches, sd.nonCanonicalMatches);
            }
        }
        teloscope.labelTerminalBlocks(pd.terminalBlocks, static_cast<uint16_t>(pd.gapInfos.size()), pd.terminalLabel,
                                      pd.scaffoldType, pd.pathSize, ui.terminalLimit);
    }
    return paths;
}

inline std::vector<PathData> walkPaths(Teloscope &teloscope, const std::vector<FastaRecord> &records) {
    std::vector<RecordView> views(records.size());
    for (size_t i = 0; i < records.size(); ++i) views[i] = RecordView{&records[i].header, records[i].sequence.data(), records[i].sequence.size(), nullptr, 0, nullptr};
    return walkRecordViews(teloscope, views, 0);
}

// ---------------------------------------------------------------------------------------------
// --fastq-subset (Input::readFastqSubset, src/input.cpp:737-832): echo the reads that carry a terminal
// telomere block, in input order, byte for byte.  Records are read by the 4-line rules of
// readFastqRecord (src/input.cpp:113-138: blank lines before a header are skipped, line contents are
// kept verbatim, '\r' included) from a plain or gzip file ("-" = stdin) and filtered in batches of up
// to readsPerBatch reads / basesPerBatch bases — a GPU batch, not the reference's 2048-record one.
// Malformed input throws std::runtime_error with the reference's message.
struct FastqSubsetResult { uint64_t kept = 0, total = 0; };

namespace detail {
inline size_t logicalLineLength(const char *b, const char *e) { return (e > b && e[-1] == '\r') ? static_cast<size_t>(e - b) - 1 : static_cast<size_t>(e - b); }

struct FastqRec { const char *begin, *seq, *end; uint64_t seqLen; bool needNewline; };

// Parses the records that START in [p, limit) of a complete FASTQ text ending at `end` (the last line may lack its
// '\n'), by readFastqRecord's rules (src/input.cpp:113-138).  Returns where it stopped — the start of the first record
// at or beyond `limit`, or `end` — or nullptr at the first thing that is not a well-formed record (the caller then
// re-parses sequentially to report it the way the reference does).
inline const char *parseFastqPiece(const char *p, const char *limit, const char *end, std::vector<FastqRec> &out) {
    auto nextLine = [&](const char *from, const char *&lb, const char *&le, const char *&next) -> bool {
        if (from >= end) return false;
        const char *nl = static_cast<const char *>(std::memchr(from, '\n', static_cast<size_t>(end - from)));
        lb = from; le = nl ? nl : end; next = nl ? nl + 1 : end;
        return true;
    };
    while (p < limit) {
        const char *hb, *he, *nx;
        if (!nextLine(p, hb, he, nx)) break;
        if (logicalLineLength(hb, he) == 0) { p = nx; continue; }             // blank line before a header
        const char *sb, *se, *pb, *pe, *qb, *qe, *n2, *n3, *n4;
        if (!(nextLine(nx, sb, se, n2) && nextLine(n2, pb, pe, n3) && nextLine(n3, qb, qe, n4))) return nullptr;
        if (he == hb || *hb != '@' || pe == pb || *pb != '+' || logicalLineLength(sb, se) != logicalLineLength(qb, qe)) return nullptr;
        out.push_back(FastqRec{hb, sb, qe, static_cast<uint64_t>(se - sb), true});
        p = n4;
    }
    return p;
}

// All records of a mapped FASTQ text, found by the host threads: the text is cut into pieces; a piece starts at the first
// line that looks like a header ('@' first, the line after next begins with '+', the next line does not begin with '@' — a
// quality line may begin with '@', but then the line after it is a header), every piece is parsed on its own, and the
// pieces are accepted only if each parse lands exactly on the next piece's start.  false = not provably well-formed this
// way: parse sequentially instead.
inline bool splitFastqRecords(const char *data, size_t size, std::vector<FastqRec> &records, size_t pieceBytes = size_t(32) << 20) {
    const char *end = data + size;
    const size_t np = std::max<size_t>(1, std::min<size_t>(256, size / std::max<size_t>(pieceBytes, 1)));
    if (np < 2) return false;
    std::vector<const char *> start(np + 1, nullptr);
    start[0] = data; start[np] = end;
    std::atomic<bool> bad{false};
    auto lineAfter = [&](const char *q) -> const char * {                    // start of the line after the one holding q
        const char *nl = static_cast<const char *>(std::memchr(q, '\n', static_cast<size_t>(end - q)));
        return nl ? nl + 1 : end;
    };
    {
        std::atomic<size_t> next{1};
        auto worker = [&]() {
            for (size_t i; (i = next.fetch_add(1)) < np;) {
                const char *q = lineAfter(data + size / np * i - 1);          // a line start at or after the cut
                int tries = 0;
                for (; q < end && tries < 64; ++tries) {
                    const char *l1 = lineAfter(q), *l2 = l1 < end ? lineAfter(l1) : end;
                    if (*q == '@' && l1 < end && *l1 != '@' && l2 < end && *l2 == '+') break;
                    q = l1;
                }
                if (q < end && tries == 64) { bad.store(true); return; }       // 64 lines without a header: not FASTQ-shaped
                start[i] = q < end ? q : end;                                     // (no record starts behind this cut)
            }
        };
        const unsigned nt = static_cast<unsigned>(std::min<size_t>({size_t(16), np, size_t(std::max(1u, std::thread::hardware_concurrency()))}));
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; ++t) pool.emplace_back(worker);
        for (std::thread &th : pool) th.join();
    }
    if (bad.load()) return false;
    for (size_t i = 1; i <= np; ++i) if (start[i] < start[i - 1]) return false;
    std::vector<std::vector<FastqRec>> parts(np);
    {
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t i; (i = next.fetch_add(1)) < np;) {
                if (start[i] == start[i + 1]) continue;
                parts[i].reserve(static_cast<size_t>(start[i + 1] - start[i]) / 20000 + 16);
                const char *stop = parseFastqPiece(start[i], start[i + 1], end, parts[i]);
                if (stop != start[i + 1]) { bad.store(true); return; }
            }
        };
        const unsigned nt = static_cast<unsigned>(std::min<size_t>({size_t(16), np, size_t(std::max(1u, std::thread::hardware_concurrency()))}));
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; ++t) pool.emplace_back(worker);
        for (std::thread &th : pool) th.join();
    }
    if (bad.load()) return false;
    size_t total = 0;
    for (const auto &v : parts) total += v.size();
    records.clear();
    records.reserve(total);
    for (const auto &v : parts) records.insert(records.end(), v.begin(), v.end());
    return true;
}
}  // namespace detail

namespace detail {
// The library's first call in a process pays for its streams, its kernels' code, the device pool and the pinned staging (about
// 75 ms, whatever the size of the call).  The subset tools make it here, on one dummy read and a thread of its own, while their
// first block of input is still being read and parsed; joined before the first real batch.  TS_MIRROR_WARMUP=0 leaves it out.
class FilterWarmUp {
    std::thread th;
public:
    explicit FilterWarmUp(ReadTelomereFilter &filter) {
        const char *wu = std::getenv("TS_MIRROR_WARMUP");
        if (wu && wu[0] == '0') return;
        th = std::thread([&filter] {
            try {
                const std::string seq(4096, 'A');
                const char *p = seq.data();
                uint64_t l = seq.size();
                uint8_t pass = 0;
                filter.matchesPointers(&p, &l, 1, &pass);
            } catch (...) {}
        });
    }
    void join() { if (th.joinable()) th.join(); }
    ~FilterWarmUp() { join(); }
    FilterWarmUp(const FilterWarmUp &) = delete;
    FilterWarmUp &operator=(const FilterWarmUp &) = delete;
};
}  // namespace detail

inline FastqSubsetResult fastqSubset(const std::string &inFile, std::ostream &out, ReadTelomereFilter &filter,
                                     size_t readsPerBatch = 1u << 20, size_t bytesPerBatch = 512u << 20) {
    detail::FilterWarmUp warmUp(filter);
    // The input is read in large blocks into one arena and parsed in place: a batch's sequences are
    // handed to the filter as pointers into the arena and a kept record is echoed as the byte range of
    // its four lines — no per-line copies.
    // (a plain file is read with pread(2) straight into the block; gzip input and stdin go through zlib)
    struct Source {
        gzFile gz = nullptr;
        int fd = -1;
        ~Source() { if (gz) gzclose(gz); if (fd >= 0) ::close(fd); }
        long get(char *dst, size_t n) {
            if (gz) return gzread(gz, dst, static_cast<unsigned>(std::min<size_t>(n, 1u << 30)));
            return static_cast<long>(::read(fd, dst, std::min<size_t>(n, 1u << 30)));
        }
    } src;
    if (inFile != "-") {
        src.fd = ::open(inFile.c_str(), O_RDONLY);
        if (src.fd < 0) throw std::runtime_error("Stream not successful: " + inFile);
        unsigned char magic[2] = {0, 0};
        const bool gz = ::pread(src.fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        if (gz) { src.gz = gzdopen(src.fd, "rb"); src.fd = -1; }
    } else {
        src.gz = gzdopen(0, "rb");
    }
    if (!src.gz && src.fd < 0) throw std::runtime_error("Stream not successful: " + inFile);
    if (src.gz) gzbuffer(src.gz, 1u << 20);
    const bool seekable = !src.gz && ::lseek(src.fd, 0, SEEK_CUR) != static_cast<off_t>(-1);   // a FIFO is read in order
    // Two blocks in turn: while one is parsed and filtered on the GPU, a reader thread fills the other (a plain
    // file by up to eight concurrent pread(2) streams, gzip / stdin through zlib).  A block's data starts at `head`
    // bytes into its buffer, so that the partial record left over from the previous block can be put in front
    // of it without moving the block.
    const size_t blockBytes = std::max<size_t>(bytesPerBatch, 64);
    struct Block {
        std::vector<char> buf;
        size_t begin = 0, end = 0;         // valid bytes [begin, end)
        bool last = false;                 // the input ended inside (or right before) this block
    } blk[2];
    size_t head = std::min<size_t>(blockBytes, 1u << 20);
    uint64_t fileOff = 0;                  // next byte of a plain file
    auto fill = [&](Block &b) {            // reads up to blockBytes bytes to b.buf[head ..)
        if (b.buf.size() < head + blockBytes) b.buf.resize(head + blockBytes);
        char *dst = b.buf.data() + head;
        size_t got = 0;
        b.last = false;
        if (seekable) {
            // slices of the block are read concurrently; a short slice means the file ends there
            const unsigned nth = blockBytes >= (64u << 20) ? std::min(8u, std::max(2u, std::thread::hardware_concurrency() / 2u)) : 1u;
            const size_t share = (blockBytes + nth - 1) / nth;
            std::vector<size_t> done(nth, 0);
            std::vector<int> bad(nth, 0);
            auto readSlice = [&](unsigned t) {
                const size_t lo = std::min(blockBytes, t * share), hi = std::min(blockBytes, lo + share);
                size_t n = 0;
                while (lo + n < hi) {
                    const ssize_t r = ::pread(src.fd, dst + lo + n, std::min<size_t>(hi - lo - n, 1u << 30), static_cast<off_t>(fileOff + lo + n));
                    if (r < 0) { bad[t] = 1; break; }
                    if (r == 0) break;
                    n += static_cast<size_t>(r);
                }
                done[t] = n;
            };
            if (nth == 1) readSlice(0);
            else {
                std::vector<std::thread> pool;
                for (unsigned t = 0; t < nth; ++t) pool.emplace_back(readSlice, t);
                for (std::thread &th : pool) th.join();
            }
            for (unsigned t = 0; t < nth; ++t) {
                if (bad[t]) throw std::runtime_error("read error in FASTQ input");
                const size_t lo = std::min(blockBytes, t * share), hi = std::min(blockBytes, lo + share);
                got += done[t];
                if (done[t] < hi - lo) { b.last = true; break; }      // end of file inside this slice
            }
            fileOff += got;
        } else {
            while (got < blockBytes) {
                const long n = src.get(dst + got, blockBytes - got);
                if (n < 0) throw std::runtime_error("read error in FASTQ input");
                if (n == 0) { b.last = true; break; }
                got += static_cast<size_t>(n);
            }
        }
        b.begin = head;
        b.end = head + got;
    };
    bool first = true;
    FastqSubsetResult res;
    uint64_t recordNumber = 0;
    auto fail = [&](const char *msg) { throw std::runtime_error("FASTQ record " + std::to_string(recordNumber + 1) + ": " + msg); };
    using Rec = detail::FastqRec;
    std::vector<Rec> batch;
    std::vector<const char *> ptr;
    std::vector<uint64_t> len;
    std::vector<uint8_t> pass;
    std::string text;
    // filters `batch` on the GPU and echoes the kept records
    auto runBatch = [&]() {
        if (batch.empty()) return;
        ptr.resize(batch.size()); len.resize(batch.size()); pass.resize(batch.size());
        for (size_t i = 0; i < batch.size(); ++i) { ptr[i] = batch[i].seq; len[i] = batch[i].seqLen; }
        warmUp.join();
        filter.matchesPointers(ptr.data(), len.data(), batch.size(), pass.data());
        text.clear();
        for (size_t i = 0; i < batch.size(); ++i) {
            if (!pass[i]) continue;
            text.append(batch[i].begin, batch[i].end);
            text.push_back('\n');
            ++res.kept;
        }
        res.total += batch.size();
        out.write(text.data(), static_cast<std::streamsize>(text.size()));
        if (!out.good()) throw std::runtime_error("failed while writing FASTQ subset");
    };
    // Parses whole records out of [p, end) — several GPU batches if the range holds more than readsPerBatch reads or
    // bytesPerBatch record bytes — filters them and echoes the kept ones; returns the end of the last whole record.
    auto processRange = [&](const char *p, const char *const end, const bool eof) -> const char * {
        const char *consumed = p;
        // parse whole records out of the block; several GPU batches if it holds more than readsPerBatch reads
        bool more = true;
        while (more) {
            batch.clear();
            more = false;
            size_t batchBytes = 0;
            while (p < end) {
                if (batch.size() >= readsPerBatch || batchBytes >= bytesPerBatch) { more = true; break; }
                // a line = [p, nl); at end of input the last line may lack its '\n'
                auto nextLine = [&](const char *from, const char *&lb, const char *&le, const char *&next) -> bool {
                    if (from >= end) return false;
                    const char *nl = static_cast<const char *>(std::memchr(from, '\n', static_cast<size_t>(end - from)));
                    if (!nl && !eof) return false;                 // incomplete line: wait for more input
                    lb = from; le = nl ? nl : end; next = nl ? nl + 1 : end;
                    return true;
                };
                const char *hb, *he, *nx;
                if (!nextLine(p, hb, he, nx)) break;
                if (detail::logicalLineLength(hb, he) == 0) { p = nx; consumed = p; continue; }   // blank line before a header
                const char *sb, *se, *pb, *pe, *qb, *qe, *n2, *n3, *n4;
                const bool l2 = nextLine(nx, sb, se, n2), l3 = l2 && nextLine(n2, pb, pe, n3), l4 = l3 && nextLine(n3, qb, qe, n4);
                if (!l4) {
                    if (eof) fail("truncated FASTQ record");
                    break;                                          // the record continues in the next block
                }
                if (he == hb || *hb != '@') fail("expected header line starting with '@'");
                if (pe == pb || *pb != '+') fail("expected separator line starting with '+'");
                if (detail::logicalLineLength(sb, se) != detail::logicalLineLength(qb, qe)) fail("sequence and quality length differ");
                ++recordNumber;
                batch.push_back(Rec{hb, sb, qe, static_cast<uint64_t>(se - sb), true});
                batchBytes += static_cast<size_t>(n4 - hb);
                p = n4;
                consumed = p;
            }
            runBatch();
        }
        return consumed;
    };
    if (seekable) {
        // A regular file is mapped and parsed where the page cache holds it: the only copy of a sequence is the one
        // into the pinned upload ring.
        struct stat sb;
        if (::fstat(src.fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
            const size_t size = static_cast<size_t>(sb.st_size);
            void *map = ::mmap(nullptr, size, PROT_READ, MAP_PRIVATE, src.fd, 0);
            if (map != MAP_FAILED) {
                (void)::madvise(map, size, MADV_SEQUENTIAL);
                struct Unmap { void *p; size_t n; ~Unmap() { ::munmap(p, n); } } unmap{map, size};
                const char *data = static_cast<const char *>(map);
                if (data[0] != '@') throw std::runtime_error("FASTQ input must start with '@'");
                // a large file's records are located by all host threads at once (one thread walks 3 GB of text in
                // ~0.5 s); anything that is not provably well-formed that way takes the sequential parser, which
                // reports errors the way the reference does
                std::vector<Rec> all;
                if (size >= (size_t(64) << 20) && detail::splitFastqRecords(data, size, all)) {
                    size_t i = 0;
                    while (i < all.size()) {
                        batch.clear();
                        size_t batchBytes = 0;
                        while (i < all.size() && batch.size() < readsPerBatch && batchBytes < bytesPerBatch) {
                            batchBytes += static_cast<size_t>(all[i].end - all[i].begin) + 1;
                            batch.push_back(all[i++]);
                        }
                        runBatch();
                    }
                    out.flush();
                    return res;
                }
                (void)processRange(data, data + size, true);
                out.flush();
                return res;
            }
        }
    }
    int cur = 0;
    fill(blk[0]);
    for (;;) {
        Block &B = blk[cur], &N = blk[cur ^ 1];
        const bool eof = B.last;
        // the next block is read while this one is parsed and filtered
        std::exception_ptr readError;
        std::thread reader;
        if (!eof) reader = std::thread([&] { try { fill(N); } catch (...) { readError = std::current_exception(); } });
        struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{reader};
        if (first) {
            if (B.end == B.begin) throw std::runtime_error("FASTQ input is empty");
            if (B.buf[B.begin] != '@') throw std::runtime_error("FASTQ input must start with '@'");
            first = false;
        }
        const char *const end = B.buf.data() + B.end;
        const char *const consumed = processRange(B.buf.data() + B.begin, end, eof);
        if (eof) break;
        reader.join();
        if (readError) std::rethrow_exception(readError);
        // what was not consumed (a partial record) goes in front of the next block's data
        const size_t left = static_cast<size_t>(end - consumed);
        if (left > N.begin) {                                   // more than the headroom (a record larger than a block)
            const size_t have = N.end - N.begin;
            std::vector<char> grown(left + std::max(blockBytes, have) + head);
            std::memcpy(grown.data() + left, N.buf.data() + N.begin, have);
            N.buf.swap(grown);
            N.begin = left; N.end = left + have;
            head = std::max(head, left);                        // later blocks leave that much room
        }
        if (left) std::memcpy(N.buf.data() + N.begin - left, consumed, left);
        N.begin -= left;
        cur ^= 1;
    }
    out.flush();
    return res;
}

// ---------------------------------------------------------------------------------------------
// --bam-subset (subsetBam, src/bam.cpp:188-259): copy the header and the alignment records whose read
// carries a terminal telomere block, byte for byte, into a new BAM.
//
// BGZF is a series of independent gzip members of at most 64 KB, each carrying its own size ("BC" extra subfield), so the
// input is not read as one stream: a few hundred MB of blocks are LOCATED by walking their headers (validated like
// BgzfReader::loadBlock, src/bgzf.cpp:57-196: FEXTRA set, reserved flag bits clear, exactly one BC subfield of length 2,
// block and uncompressed sizes <= 64 KB, optional file name / comment / header checksum), INFLATED by all host threads
// straight into their places of one buffer (raw inflate must use the whole payload and yield exactly ISIZE bytes, CRC32
// checked), the records in it are walked and validated, and their sequences (4-bit codes "=ACMGRSVTWYHKDBN") are decoded
// by all host threads into one arena that the GPU filters in batches of up to readsPerBatch records.  The output is
// written as BGZF blocks of <= 0xff00 payload bytes (raw deflate + the BC field + CRC32/ISIZE) closed by the EOF marker.
// (Round 2 until here read the input with zlib's gz* API on one thread: 0.14 Gbases/s on a HiFi BAM;
// profiles/r02/bam_subset_rate.txt.)
struct BamSubsetStats { uint64_t totalRecords = 0, passedRecords = 0, missingSequenceRecords = 0; bool missingEofBlock = false; };

namespace detail {
// the host-side helpers of this header: a bounded queue between pipeline stages, and a dynamic parallel-for
template <typename T>
class BoundedQueue {
public:
    explicit BoundedQueue(size_t cap) : cap_(cap) {}
    void push(T v) {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return q_.size() < cap_ || closed_; });
        q_.push_back(std::move(v));
        cv_.notify_all();
    }
    void close() { { std::lock_guard<std::mutex> g(m_); closed_ = true; } cv_.notify_all(); }
    bool pop(T &out) {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        cv_.notify_all();
        return true;
    }
private:
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<T> q_;
    size_t cap_;
    bool closed_ = false;
};

template <typename F>
inline void onThreads(size_t n, F &&f) {                        // f(i) for i in [0, n), dynamic, on up to 16 host threads
    const unsigned nt = static_cast<unsigned>(std::min<size_t>({size_t(16), n, size_t(std::max(1u, std::thread::hardware_concurrency()))}));
    if (nt <= 1) { for (size_t i = 0; i < n; ++i) f(i); return; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t) pool.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < n;) f(i); });
    for (std::thread &th : pool) th.join();
}

struct BgzfBlockRef { const unsigned char *payload; uint32_t payloadLen, isize, crc; size_t outOff; };

// The BGZF block that starts at p, of which n bytes are at hand: its total size, or 0 when it is not complete within n
// bytes (more input is needed).  Throws on anything that is not a BGZF block.
inline size_t parseBgzfBlock(const unsigned char *p, size_t n, BgzfBlockRef &ref, bool &eofMarker) {
    static const unsigned char kEof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    auto u16 = [](const unsigned char *q) { return static_cast<uint32_t>(q[0]) | (static_cast<uint32_t>(q[1]) << 8); };
    auto u32 = [&](const unsigned char *q) { return u16(q) | (u16(q + 2) << 16); };
    if (n < 12) return 0;
    if (p[0] != 31 || p[1] != 139 || p[2] != 8) throw std::runtime_error("input is not BGZF-compressed BAM (not a BAM file)");
    const unsigned flags = p[3];
    if ((flags & 0x04) == 0 || (flags & 0xe0) != 0) throw std::runtime_error("invalid BGZF gzip flags");
    const size_t xlen = u16(p + 10);
    if (n < 12 + xlen) return 0;
    bool found = false;
    size_t total = 0;
    for (size_t pos = 12; pos < 12 + xlen;) {
        if (12 + xlen - pos < 4) throw std::runtime_error("malformed BGZF extra field");
        const size_t slen = u16(p + pos + 2), end = pos + 4 + slen;
        if (end > 12 + xlen) throw std::runtime_error("malformed BGZF extra subfield");
        if (p[pos] == 'B' && p[pos + 1] == 'C') {
            if (slen != 2 || found) throw std::runtime_error("invalid BGZF BC subfield");
            total = static_cast<size_t>(u16(p + pos + 4)) + 1;
            found = true;
        }
        pos = end;
    }
    if (!found) throw std::runtime_error("BGZF block is missing the BC subfield");
    if (total > 65536 || total < 12 + xlen + 8) throw std::runtime_error("invalid BGZF block size");
    if (n < total) return 0;
    size_t at = 12 + xlen;
    const size_t footer = total - 8;
    auto skipText = [&](const char *what) {
        while (at < footer && p[at] != 0) ++at;
        if (at == footer) throw std::runtime_error(std::string("unterminated BGZF ") + what);
        ++at;
    };
    if (flags & 0x08) skipText("filename");
    if (flags & 0x10) skipText("comment");
    if (flags & 0x02) {
        if (footer - at < 2) throw std::runtime_error("truncated BGZF header checksum");
        const uLong crc = crc32(crc32(0L, Z_NULL, 0), p, static_cast<uInt>(at));
        if (static_cast<uint32_t>(crc & 0xffffu) != u16(p + at)) throw std::runtime_error("BGZF header checksum mismatch");
        at += 2;
    }
    ref.payload = p + at;
    ref.payloadLen = static_cast<uint32_t>(footer - at);
    ref.crc = u32(p + footer);
    ref.isize = u32(p + footer + 4);
    if (ref.isize > 65536) throw std::runtime_error("BGZF uncompressed block is too large");
    eofMarker = total == sizeof kEof && std::memcmp(p, kEof, sizeof kEof) == 0;
    return total;
}

inline void inflateBgzfBlock(const BgzfBlockRef &b, unsigned char *dst) {
    z_stream zs{};
    unsigned char dummy = 0;
    zs.next_in = const_cast<unsigned char *>(b.payload); zs.avail_in = b.payloadLen;
    zs.next_out = b.isize ? dst : &dummy; zs.avail_out = b.isize ? b.isize : 1;
    if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("could not initialize BGZF decompressor");
    const int rc = inflate(&zs, Z_FINISH);
    const bool whole = rc == Z_STREAM_END && zs.total_out == b.isize && zs.total_in == b.payloadLen;
    inflateEnd(&zs);
    if (!whole) throw std::runtime_error("invalid BGZF deflate payload");
    const uLong crc = crc32(crc32(0L, Z_NULL, 0), b.isize ? dst : Z_NULL, b.isize);
    if (static_cast<uint32_t>(crc) != b.crc) throw std::runtime_error("BGZF checksum mismatch");
}

// Uncompressed bytes of a BGZF file or stream, a few hundred MB per call, inflated by all host threads.
class BgzfParallelReader {
    int fd_;
    const unsigned char *map_ = nullptr;        // a regular file is mapped ...
    size_t mapSize_ = 0, mapPos_ = 0;
    std::vector<unsigned char> buf_;            // ... a pipe is read into a buffer (the incomplete block at its end is kept)
    bool streamEof_ = false;
    std::vector<BgzfBlockRef> blocks_;
public:
    bool sawEofMarker = false;
    explicit BgzfParallelReader(int fd) : fd_(fd) {
        struct stat st;
        if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            void *m = ::mmap(nullptr, static_cast<size_t>(st.st_size), PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                map_ = static_cast<const unsigned char *>(m); mapSize_ = static_cast<size_t>(st.st_size);
                ::madvise(m, mapSize_, MADV_SEQUENTIAL);
            }
        }
    }
    ~BgzfParallelReader() { if (map_) ::munmap(const_cast<unsigned char *>(map_), mapSize_); }
    BgzfParallelReader(const BgzfParallelReader &) = delete;
    BgzfParallelReader &operator=(const BgzfParallelReader &) = delete;

    // Inflates the next whole blocks into dst[0 .. cap) (cap >= 64 KB): the number of bytes produced; `more` = false when
    // the input is exhausted.
    size_t next(unsigned char *dst, size_t cap, bool &more) {
        const unsigned char *w;
        size_t wn;
        bool atEnd;
        if (map_) { w = map_ + mapPos_; wn = mapSize_ - mapPos_; atEnd = true; }
        else {
            while (!streamEof_ && buf_.size() < cap) {             // compressed bytes: never more than the uncompressed room
                const size_t at = buf_.size();
                buf_.resize(at + (8u << 20));
                const ssize_t r = ::read(fd_, buf_.data() + at, 8u << 20);
                if (r < 0) throw std::runtime_error("cannot read BAM input");
                buf_.resize(at + static_cast<size_t>(r));
                if (r == 0) streamEof_ = true;
            }
            w = buf_.data(); wn = buf_.size(); atEnd = streamEof_;
        }
        blocks_.clear();
        size_t used = 0, produced = 0;
        while (used < wn) {
            BgzfBlockRef ref{};
            bool eofm = false;
            const size_t total = parseBgzfBlock(w + used, wn - used, ref, eofm);
            if (total == 0) {
                if (atEnd) throw std::runtime_error(wn - used < 12 ? "truncated BGZF header" : "truncated BGZF block");
                break;
            }
            if (produced + ref.isize > cap) break;
            sawEofMarker = sawEofMarker || eofm;
            ref.outOff = produced;
            produced += ref.isize;
            blocks_.push_back(ref);
            used += total;
        }
        if (!blocks_.empty()) {
            const size_t per = 32, tasks = (blocks_.size() + per - 1) / per;
            std::mutex em;
            std::exception_ptr err;
            onThreads(tasks, [&](size_t t) {
                try {
                    for (size_t i = t * per; i < std::min(blocks_.size(), (t + 1) * per); ++i) inflateBgzfBlock(blocks_[i], dst + blocks_[i].outOff);
                } catch (...) { std::lock_guard<std::mutex> g(em); if (!err) err = std::current_exception(); }
            });
            if (err) std::rethrow_exception(err);
        }
        if (map_) mapPos_ += used; else buf_.erase(buf_.begin(), buf_.begin() + static_cast<std::ptrdiff_t>(used));
        more = used > 0 && (map_ ? mapPos_ < mapSize_ : !(streamEof_ && buf_.empty()));
        return produced;
    }
};
}  // namespace detail

namespace detail {
class BgzfWriter {
    static constexpr size_t kPayload = 0xff00, kSlot = 65536, kBlocksPerFlush = 64;
    std::ostream &out;
    std::vector<unsigned char> pending, comp;
    std::vector<size_t> sizes;
    static size_t block(const unsigned char *data, size_t n, unsigned char *slot) {         // one BGZF block into its 64 KB slot
        z_stream zs{};
        if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2 failed");
        zs.next_in = const_cast<unsigned char *>(data); zs.avail_in = static_cast<unsigned>(n);
        zs.next_out = slot + 18; zs.avail_out = kSlot - 18 - 8;
        const int rc = deflate(&zs, Z_FINISH);
        deflateEnd(&zs);
        if (rc != Z_STREAM_END) throw std::runtime_error("BGZF block does not fit");
        const size_t clen = zs.total_out, total = 18 + clen + 8;
        static const unsigned char head[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};
        std::memcpy(slot, head, 16);
        slot[16] = static_cast<unsigned char>((total - 1) & 0xff); slot[17] = static_cast<unsigned char>((total - 1) >> 8);
        const uint32_t crc = static_cast<uint32_t>(crc32(crc32(0L, Z_NULL, 0), data, static_cast<unsigned>(n)));
        unsigned char *tail = slot + 18 + clen;
        for (int i = 0; i < 4; ++i) { tail[i] = static_cast<unsigned char>(crc >> (8 * i)); tail[4 + i] = static_cast<unsigned char>(static_cast<uint32_t>(n) >> (8 * i)); }
        return total;
    }
    // the first `bytes` pending bytes leave as blocks of kPayload (the last one may be shorter), deflated by all host threads
    void flush(size_t bytes) {
        const size_t nb = (bytes + kPayload - 1) / kPayload;
        if (!nb) return;
        comp.resize(nb * kSlot);
        sizes.assign(nb, 0);
        std::mutex em;
        std::exception_ptr err;
        onThreads(nb, [&](size_t i) {
            try { sizes[i] = block(pending.data() + i * kPayload, std::min(kPayload, bytes - i * kPayload), comp.data() + i * kSlot); }
            catch (...) { std::lock_guard<std::mutex> g(em); if (!err) err = std::current_exception(); }
        });
        if (err) std::rethrow_exception(err);
        for (size_t i = 0; i < nb; ++i) out.write(reinterpret_cast<const char *>(comp.data() + i * kSlot), static_cast<std::streamsize>(sizes[i]));
        pending.erase(pending.begin(), pending.begin() + static_cast<std::ptrdiff_t>(bytes));
    }
public:
    explicit BgzfWriter(std::ostream &o) : out(o) {}
    void write(const unsigned char *data, size_t n) {
        pending.insert(pending.end(), data, data + n);
        if (pending.size() >= kBlocksPerFlush * kPayload) flush(pending.size() / kPayload * kPayload);      // whole blocks only
    }
    void finish() {
        flush(pending.size());
        static const unsigned char eof[28] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        out.write(reinterpret_cast<const char *>(eof), sizeof eof);
        out.flush();
        if (!out.good()) throw std::runtime_error("failed while writing BAM subset");
    }
};
}  // namespace detail

inline BamSubsetStats bamSubset(const std::string &inFile, std::ostream &out, ReadTelomereFilter &filter,
                                size_t readsPerBatch = 1u << 20, size_t bytesPerBatch = 256u << 20) {
    BamSubsetStats stats;
    detail::FilterWarmUp warmUp(filter);
    int fd = 0;
    if (inFile != "-") {
        fd = ::open(inFile.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open BAM input '" + inFile + "'");
    }
    struct Closer { int fd; ~Closer() { if (fd > 0) ::close(fd); } } closer{fd};
    detail::BgzfParallelReader reader(fd);
    auto le32 = [](const unsigned char *p) { return static_cast<uint32_t>(p[0]) | (static_cast<uint32_t>(p[1]) << 8) | (static_cast<uint32_t>(p[2]) << 16) | (static_cast<uint32_t>(p[3]) << 24); };

    detail::BgzfWriter writer(out);
    // Uncompressed bytes arrive in chunks of ~bytesPerBatch, inflated by a reader thread (which spreads the blocks over the
    // host threads) into one of two buffers while the other one is parsed and filtered.  What a chunk leaves unconsumed —
    // the head of a record that continues in the next chunk — is carried into the headroom in front of the next one.
    constexpr size_t kHeadroom = 1u << 20;
    const size_t chunkBytes = std::max<size_t>(bytesPerBatch, 1u << 20);
    struct Chunk { std::unique_ptr<unsigned char[]> buf; size_t n = 0; bool more = false; std::exception_ptr err; };
    Chunk chunks[2];
    for (Chunk &c : chunks) c.buf.reset(new unsigned char[kHeadroom + chunkBytes]);
    detail::BoundedQueue<Chunk *> ready(2), spare(2);
    spare.push(&chunks[0]); spare.push(&chunks[1]);
    double msInflate = 0;
    using Clock = std::chrono::steady_clock;
    std::thread inflater([&] {
        Chunk *c;
        while (spare.pop(c)) {
            const Clock::time_point t0 = Clock::now();
            try { c->err = nullptr; c->n = reader.next(c->buf.get() + kHeadroom, chunkBytes, c->more); }
            catch (...) { c->err = std::current_exception(); c->n = 0; c->more = false; }
            msInflate += std::chrono::duration<double, std::milli>(Clock::now() - t0).count();
            const bool last = !c->more;
            ready.push(c);
            if (last) break;
        }
        ready.close();
    });
    struct Joiner { std::thread &t; detail::BoundedQueue<Chunk *> &q; ~Joiner() { q.close(); if (t.joinable()) t.join(); } } joiner{inflater, spare};

    const unsigned char *plain = nullptr;       // the bytes being parsed: [plain + pos, plain + plainSize)
    size_t plainSize = 0, pos = 0;
    std::vector<unsigned char> carry, joined;   // unconsumed tail of the last chunk; (rare) a tail larger than the headroom + the next chunk
    // header: magic, text, reference list — validated like copyBamHeader (src/bam.cpp:75-120: text <= 1 GiB, names
    // 1 .. 1 MiB and NUL-terminated, no negative counts or lengths) and copied verbatim.  A step of the parse that lacks
    // bytes returns false and is retried when the next chunk has been inflated.
    constexpr uint32_t kMaxHeaderText = 1u << 30, kMaxReferenceName = 1u << 20;
    enum { Magic, Text, Refs, Records } phase = Magic;
    uint32_t ltext = 0, nref = 0, refsDone = 0;
    auto have = [&](size_t n) { return plainSize - pos >= n; };
    auto parseHeader = [&]() -> bool {                      // true when the header is complete
        for (;;) {
            if (phase == Magic) {
                if (!have(8)) return false;
                if (std::memcmp(plain + pos, "BAM\1", 4) != 0) throw std::runtime_error("input is not a BAM file");
                ltext = le32(plain + pos + 4);
                if (ltext > kMaxHeaderText) throw std::runtime_error("invalid BAM header text length");
                phase = Text;
            } else if (phase == Text) {
                if (!have(8 + static_cast<size_t>(ltext) + 4)) return false;
                nref = le32(plain + pos + 8 + ltext);
                if (nref > 0x7fffffffu) throw std::runtime_error("invalid BAM reference count");
                writer.write(plain + pos, 8 + static_cast<size_t>(ltext) + 4);
                pos += 8 + static_cast<size_t>(ltext) + 4;
                phase = Refs;
            } else if (phase == Refs) {
                if (refsDone == nref) { phase = Records; return true; }
                if (!have(4)) return false;
                const uint32_t lname = le32(plain + pos);
                if (lname == 0 || lname > kMaxReferenceName) throw std::runtime_error("invalid BAM reference name length");
                if (!have(4 + static_cast<size_t>(lname) + 4)) return false;
                if (plain[pos + 4 + lname - 1] != 0) throw std::runtime_error("BAM reference name is not NUL-terminated");
                if (le32(plain + pos + 4 + lname) > 0x7fffffffu) throw std::runtime_error("invalid BAM reference length");
                writer.write(plain + pos, 4 + static_cast<size_t>(lname) + 4);
                pos += 4 + static_cast<size_t>(lname) + 4;
                ++refsDone;
            } else {
                return true;
            }
        }
    };

    struct Rec { size_t rawOff, rawLen, seqAt; uint32_t seqLen; size_t seqOff; };
    std::vector<Rec> recs;
    std::unique_ptr<char[]> seqs;               // the chunk's decoded sequences, back to back
    size_t seqsCap = 0;
    std::vector<const char *> ptr;
    std::vector<uint64_t> len;
    std::vector<uint8_t> pass;
    double msDecode = 0, msFilter = 0, msRecords = 0;
    // two bases per packed byte
    static const std::array<uint16_t, 256> pairs = [] {
        std::array<uint16_t, 256> t{};
        const char *bases = "=ACMGRSVTWYHKDBN";
        for (unsigned v = 0; v < 256; ++v) {
            const unsigned char two[2] = {static_cast<unsigned char>(bases[v >> 4]), static_cast<unsigned char>(bases[v & 15])};
            uint16_t w;
            std::memcpy(&w, two, 2);
            t[v] = w;
        }
        return t;
    }();
    auto processRecords = [&]() {
        if (recs.empty()) return;
        size_t total = 0;
        for (Rec &r : recs) { r.seqOff = total; total += (static_cast<size_t>(r.seqLen) + 1) & ~size_t(1); }      // even lengths: whole pairs are stored
        if (total > seqsCap) { seqsCap = total + total / 8; seqs.reset(new char[seqsCap]); }
        // decode on all host threads: tasks of ~2 MB of bases
        std::vector<size_t> cut{0};
        for (size_t i = 0, acc = 0; i < recs.size(); ++i) { acc += recs[i].seqLen; if (acc >= (2u << 20)) { cut.push_back(i + 1); acc = 0; } }
        if (cut.back() != recs.size()) cut.push_back(recs.size());
        const Clock::time_point td = Clock::now();
        detail::onThreads(cut.size() - 1, [&](size_t t) {
            for (size_t i = cut[t]; i < cut[t + 1]; ++i) {
                const Rec &r = recs[i];
                const unsigned char *packed = plain + r.seqAt;
                char *dst = seqs.get() + r.seqOff;
                for (size_t j = 0, n = (static_cast<size_t>(r.seqLen) + 1) / 2; j < n; ++j) std::memcpy(dst + 2 * j, &pairs[packed[j]], 2);
            }
        });
        msDecode += std::chrono::duration<double, std::milli>(Clock::now() - td).count();
        for (size_t a = 0; a < recs.size(); a += readsPerBatch) {
            const size_t z = std::min(recs.size(), a + readsPerBatch);
            ptr.clear(); len.clear();
            for (size_t i = a; i < z; ++i)
                if (recs[i].seqLen) { ptr.push_back(seqs.get() + recs[i].seqOff); len.push_back(recs[i].seqLen); }
            pass.assign(ptr.size(), 0);
            const Clock::time_point tf = Clock::now();
            warmUp.join();
            if (!ptr.empty()) filter.matchesPointers(ptr.data(), len.data(), ptr.size(), pass.data());
            msFilter += std::chrono::duration<double, std::milli>(Clock::now() - tf).count();
            size_t k = 0;
            for (size_t i = a; i < z; ++i) {
                const Rec &r = recs[i];
                ++stats.totalRecords;
                if (!r.seqLen) { ++stats.missingSequenceRecords; continue; }
                if (pass[k++]) { writer.write(plain + r.rawOff, r.rawLen); ++stats.passedRecords; }
            }
        }
        recs.clear();
    };

    bool more = true;
    while (more) {
        Chunk *c = nullptr;
        if (!ready.pop(c)) break;
        if (c->err) std::rethrow_exception(c->err);
        more = c->more;
        const Clock::time_point t1 = Clock::now();
        // the unconsumed tail of the chunk before goes in front of this one
        if (carry.size() <= kHeadroom) {
            unsigned char *start = c->buf.get() + kHeadroom - carry.size();
            if (!carry.empty()) std::memcpy(start, carry.data(), carry.size());
            plain = start; plainSize = carry.size() + c->n;
        } else {                                            // a record of many MB that spans chunks
            joined.assign(carry.begin(), carry.end());
            joined.insert(joined.end(), c->buf.get() + kHeadroom, c->buf.get() + kHeadroom + c->n);
            plain = joined.data(); plainSize = joined.size();
        }
        pos = 0;
        if (phase == Records || parseHeader()) {
            while (have(4)) {
                const int32_t blockSize = static_cast<int32_t>(le32(plain + pos));
                if (blockSize < 32 || static_cast<uint32_t>(blockSize) > (256u << 20)) throw std::runtime_error("invalid BAM record block_size");   // BAM_MAX_RECORD_SIZE, src/bam.cpp:29
                if (!have(4 + static_cast<size_t>(blockSize))) break;
                const unsigned char *core = plain + pos + 4;
                const uint32_t lname = core[8], ncigar = static_cast<uint32_t>(core[12]) | (static_cast<uint32_t>(core[13]) << 8), lseq = le32(core + 16);
                const uint64_t seqAt = 32ull + lname + 4ull * ncigar;
                if (lname == 0 || lseq > 0x7fffffffu) throw std::runtime_error("invalid BAM record lengths");          // decodeSequence, src/bam.cpp:122-163
                if (seqAt + (static_cast<uint64_t>(lseq) + 1) / 2 + lseq > static_cast<uint64_t>(blockSize))
                    throw std::runtime_error("BAM record fields exceed block_size");
                if (core[32 + lname - 1] != 0) throw std::runtime_error("BAM read name is not NUL-terminated");
                recs.push_back(Rec{pos, 4 + static_cast<size_t>(blockSize), pos + 4 + static_cast<size_t>(seqAt), lseq, 0});
                pos += 4 + static_cast<size_t>(blockSize);
            }
            processRecords();                   // (record offsets point into `plain`: used before the chunk goes back)
        }
        carry.assign(plain + pos, plain + plainSize);
        msRecords += std::chrono::duration<double, std::milli>(Clock::now() - t1).count();
        if (more) spare.push(c);
    }
    if (phase != Records) throw std::runtime_error(phase == Refs ? "truncated BAM reference" : "truncated BAM header");
    if (!carry.empty()) throw std::runtime_error(carry.size() < 4 ? "truncated BAM record size" : "truncated BAM record");
    spare.close();
    if (inflater.joinable()) inflater.join();
    stats.missingEofBlock = !reader.sawEofMarker;
    writer.finish();
    if (std::getenv("TS_TIMING"))
        std::fprintf(stderr, "bamSubset: locate + inflate %.0f ms (reader thread), records (walk, decode, filter, write) %.0f ms of which decode %.0f ms, filter %.0f ms\n",
                     msInflate, msRecords, msDecode, msFilter);
    return stats;
}

inline const char *scaffoldTypeToString(ScaffoldType t) {       // src/tools.cpp
    static const char *names[] = {"t2t", "gapped_t2t", "misassembly", "gapped_misassembly", "incomplete",
                                  "gapped_incomplete", "none", "gapped_none", "discordant", "gapped_discordant"};
    return names[static_cast<int>(t)];
}

// Totals that writeBEDFile accumulates and printSummary reports.
struct AssemblySummary {
    uint32_t totalPaths = 0, totalNWindows = 0, totalITS = 0, totalCanMatches = 0, totalTelomeres = 0, totalGaps = 0;
    uint32_t byType[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t scaffoldN50 = 0, contigN50 = 0;
    float teloMean = 0.0f, teloMedian = 0.0f, teloMin = 0.0f, teloMax = 0.0f;
};

namespace detail {

// the text operator<< produces, appended to a string without a stream
inline void put(std::string &s, uint64_t v) {
    char b[24];
    s.append(b, std::to_chars(b, b + sizeof b, v).ptr);
}
inline void put(std::string &s, float v) {                      // default ostream float: %g, precision 6
    // The float columns of the window files take few distinct values (ratios of small integers, entropy rounded to
    // three decimals), and shortest-round-trip formatting is the costly part of a line: every thread remembers the
    // text of the values it has formatted (4096-entry direct-mapped table keyed by the float's bits).
    struct Memo { uint32_t key; uint8_t len; char text[19]; };
    thread_local std::vector<Memo> memo;
    if (memo.empty()) { memo.resize(4096); for (Memo &m : memo) { m.key = 0; m.len = 0; } }
    uint32_t bits;
    std::memcpy(&bits, &v, 4);
    Memo &m = memo[(bits * 2654435761u) >> 20];
    if (m.len == 0 || m.key != bits) {
        char b[48];
        const size_t n = static_cast<size_t>(std::to_chars(b, b + sizeof b, v, std::chars_format::general, 6).ptr - b);
        if (n > sizeof m.text) { s.append(b, n); return; }
        m.key = bits; m.len = static_cast<uint8_t>(n);
        std::memcpy(m.text, b, n);
    }
    s.append(m.text, m.len);
}
inline void put(std::string &s, const std::string &v) { s += v; }
inline void put(std::string &s, const char *v) { s += v; }
inline void put(std::string &s, char v) { s += v; }

template <typename... A>
inline void line(std::string &s, const A &...a) { (put(s, a), ...); }

inline uint64_t computeN50(std::vector<uint64_t> v) {           // include/teloscope.h:224-235
    if (v.empty()) return 0;
    std::sort(v.begin(), v.end(), [](uint64_t a, uint64_t b) { return a > b; });
    uint64_t total = 0, cum = 0;
    for (uint64_t l : v) total += l;
    for (uint64_t l : v) { cum += l; if (cum * 2 >= total) return l; }
    return v.back();
}

enum File { DENSITY, CANON_RATIO, STRAND_RATIO, GC, ENTROPY, CAN_MATCH, NONCAN_MATCH, TERMINAL, ITS, GAPS, REPORT, CONSOLE, NFILES };

struct Task {                                  // a path's blocks/gaps/matches/report row, or a run of its windows
    uint32_t path;
    bool windows;
    size_t w0, w1;
};

}  // namespace detail

// handleBEDFile + writeBEDFile: writes <outBase>_*.bed / .bedgraph / _report.tsv and the console path
// report; fills the totals printSummary needs.  Incremental: paths are added in seqPos order, all at once
// (writeBEDFiles) or group by group while later records are still being read and scanned (scanFastaToFiles).
class BedWriter {
    std::string outBase;
    const UserInputTeloscope &ui;
    std::ostream &console;
    bool manualCuration;
    unsigned threads;
    bool on[detail::NFILES];
    std::ofstream files[detail::NFILES];
    std::vector<float> telomereLengths;
    std::vector<uint64_t> scaffoldLens, contigLens;
    AssemblySummary sum;

    void format(const PathData &pd, const detail::Task &t, std::array<std::string, detail::NFILES> &o) const {
        using namespace detail;
        const std::string &h = pd.header;
        if (t.windows) {
            // the five window files share "header \t start \t end \t" per window: it is formatted once, and every file's
            // buffer is sized up front (30 M lines for a 3 Gb assembly: string growth and per-field appends were half the cost)
            const size_t nwin = t.w1 - t.w0, per = h.size() + 48;
            for (int f : {DENSITY, CANON_RATIO, STRAND_RATIO})
                if (ui.outWinRepeats) o[f].reserve(o[f].size() + nwin * per);
            if (ui.outEntropy) o[ENTROPY].reserve(o[ENTROPY].size() + nwin * per);
            if (ui.outGC) o[GC].reserve(o[GC].size() + nwin * per);
            std::string prefix;
            prefix.reserve(per);
            auto emit = [&](std::string &dst, float v) {
                dst.append(prefix);
                put(dst, v);
                dst.push_back('\n');
            };
            for (size_t i = t.w0; i < t.w1; ++i) {
                const WindowData &w = pd.windows[i];
                const uint64_t end = w.windowStart + w.currentWindowSize;
                prefix.assign(h);
                prefix.push_back('\t');
                put(prefix, w.windowStart);
                prefix.push_back('\t');
                put(prefix, end);
                prefix.push_back('\t');
                if (ui.outWinRepeats) {
                    const uint32_t covered = w.fwdCovered + w.revCovered;
                    const float density = static_cast<float>(covered) / w.currentWindowSize;
                    const float canon = covered > 0 ? static_cast<float>(w.canonicalCovered) / (w.canonicalCovered + w.nonCanonicalCovered) : -1.0f;
                    const float strand = covered > 0 ? static_cast<float>(w.fwdCovered) / (w.fwdCovered + w.revCovered) : -1.0f;
                    emit(o[DENSITY], density);
                    emit(o[CANON_RATIO], canon);
                    emit(o[STRAND_RATIO], strand);
                }
                if (ui.outEntropy) emit(o[ENTROPY], w.shannonEntropy);
                if (ui.outGC) emit(o[GC], w.gcContent);
            }
            return;
        }
        for (const TelomereBlock &b : pd.terminalBlocks) {
            const uint64_t end = b.start + b.blockLen;
            const bool scaffoldTerminal = b.start < ui.terminalLimit || end > pd.pathSize - ui.terminalLimit;
            if (scaffoldTerminal || manualCuration)
                line(o[TERMINAL], h, '\t', b.start, '\t', end, '\t', uint64_t(b.blockLen), '\t', b.blockLabel, '\t',
                     uint64_t(b.forwardCount), '\t', uint64_t(b.reverseCount), '\t', uint64_t(b.canonicalCount), '\t',
                     uint64_t(b.nonCanonicalCount), '\t', pd.pathSize, '\t', scaffoldTerminal ? "scaffold" : "contig", '\n');
        }
        if (ui.outITS)
            for (const TelomereBlock &b : pd.interstitialBlocks)
                line(o[ITS], h, '\t', b.start, '\t', b.start + b.blockLen, '\t', uint64_t(b.blockLen), '\t', b.blockLabel, '\t',
                     uint64_t(b.forwardCount), '\t', uint64_t(b.reverseCount), '\t', uint64_t(b.canonicalCount), '\t',
                     uint64_t(b.nonCanonicalCount), '\t', pd.pathSize, '\n');
        for (const GapInfo &g : pd.gapInfos) line(o[GAPS], h, '\t', g.start, '\t', g.start + g.length, '\n');
        if (ui.outMatches) {
            for (const MatchInfo &m : pd.canonicalMatches)
                line(o[CAN_MATCH], h, '\t', m.position, '\t', m.position + m.matchSize, '\t', m.matchSeq, '\n');
            for (const MatchInfo &m : pd.nonCanonicalMatches)
                line(o[NONCAN_MATCH], h, '\t', m.position, '\t', m.position + m.matchSize, '\t', m.matchSeq, '\n');
        }
        uint64_t longest = 0;
        std::string labels;
        for (const TelomereBlock &b : pd.terminalBlocks)
            if (b.isLongest) { ++longest; labels += b.blockLabel; }
        std::string row;
        line(row, uint64_t(pd.seqPos) + 1, '\t', h, '\t', longest, '\t', labels.empty() ? std::string("none") : labels, '\t',
             uint64_t(static_cast<uint16_t>(pd.gapInfos.size())), '\t', scaffoldTypeToString(pd.scaffoldType), '\t', pd.terminalLabel);
        if (!ui.ultraFastMode)
            line(row, '\t', uint64_t(pd.interstitialBlocks.size()), '\t', pd.canonicalMatchCount, '\t',
                 uint64_t(pd.windows.size()));
        row += '\n';
        o[REPORT] += row;
        o[CONSOLE] += row;
    }

public:
    BedWriter(const std::string &outBase_, const UserInputTeloscope &ui_, std::ostream &console_, bool manualCuration_ = false,
              unsigned threads_ = 0)
        : outBase(outBase_), ui(ui_), console(console_), manualCuration(manualCuration_), threads(threads_) {
        using namespace detail;
        static const char *suffix[NFILES] = {"_window_repeat_density.bedgraph", "_window_canonical_ratio.bedgraph",
                                             "_window_strand_ratio.bedgraph", "_window_gc.bedgraph", "_window_entropy.bedgraph",
                                             "_canonical_matches.bed", "_noncanonical_matches.bed", "_terminal_telomeres.bed",
                                             "_interstitial_telomeres.bed", "_gaps.bed", "_report.tsv", ""};
        const bool want[NFILES] = {ui.outWinRepeats, ui.outWinRepeats, ui.outWinRepeats, ui.outGC, ui.outEntropy, ui.outMatches,
                                   ui.outMatches, true, ui.outITS, true, true, false};
        for (int f = 0; f < NFILES; ++f) {
            on[f] = want[f];
            if (!on[f]) continue;
            files[f].open(outBase + suffix[f], std::ios::binary);
            if (!files[f]) throw std::runtime_error("Could not open '" + outBase + suffix[f] + "' for writing.");
        }
        if (ui.outWinRepeats) {
            files[DENSITY] << "track type=bedGraph name=\"Repeat Density\" description=\"Total repeat density per window\"\n";
            files[CANON_RATIO] << "track type=bedGraph name=\"Canonical Ratio\" description=\"Canonical fraction of repeat density per window\"\n";
            files[STRAND_RATIO] << "track type=bedGraph name=\"Strand Ratio\" description=\"Forward-strand fraction of repeat density per window\"\n";
        }
        if (ui.outEntropy) files[ENTROPY] << "track type=bedGraph name=\"Shannon Entropy\" description=\"Shannon entropy per window\"\n";
        if (ui.outGC) files[GC] << "track type=bedGraph name=\"GC Content\" description=\"GC content per window\"\n";
        console << "\n+++ Path Summary Report +++\n";
        const char *cols = ui.ultraFastMode ? "pos\theader\ttelomeres\tlabels\tgaps\ttype\tgranular\n"
                                            : "pos\theader\ttelomeres\tlabels\tgaps\ttype\tgranular\tits\tcanonical\twindows\n";
        console << cols;
        files[REPORT] << cols;
        if (threads == 0) threads = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    }

    // formats and writes these paths (which follow, in seqPos order, the ones added before)
    void add(const std::vector<PathData> &paths) {
        using namespace detail;
        // work list in output order: per path its row/blocks/gaps/matches, then its windows in runs
        constexpr size_t kRun = 1u << 16;
        std::vector<Task> tasks;
        for (uint32_t p = 0; p < paths.size(); ++p) {
            tasks.push_back(Task{p, false, 0, 0});
            for (size_t w = 0; w < paths[p].windows.size(); w += kRun)
                tasks.push_back(Task{p, true, w, std::min(paths[p].windows.size(), w + kRun)});
        }
        // batches of tasks are formatted in parallel and written in order
        const size_t batch = static_cast<size_t>(threads) * 4;
        std::vector<std::array<std::string, NFILES>> outs(batch);
        for (size_t b0 = 0; b0 < tasks.size(); b0 += batch) {
            const size_t nb = std::min(batch, tasks.size() - b0);
            for (size_t i = 0; i < nb; ++i)
                for (std::string &s : outs[i]) s.clear();
            std::atomic<size_t> next{0};
            auto worker = [&]() {
                for (size_t i; (i = next.fetch_add(1)) < nb;) format(paths[tasks[b0 + i].path], tasks[b0 + i], outs[i]);
            };
            const unsigned nt = static_cast<unsigned>(std::min<size_t>(threads, nb));
            if (nt <= 1) {
                worker();
            } else {
                std::vector<std::thread> pool;
                for (unsigned i = 0; i < nt; ++i) pool.emplace_back(worker);
                for (std::thread &th : pool) th.join();
            }
            // The files are independent of each other: a file's pieces go out in task order, different files side by side — the
            // five window tracks carry nearly all the bytes (2.6 GB per 3 Gb), and one thread copying them into the page cache
            // file after file was the write stage's bound (round 5: 12 GB/s).  Small shares are written here.
            auto write_file = [&](int f) {
                for (size_t i = 0; i < nb; ++i)
                    if (!outs[i][f].empty()) files[f].write(outs[i][f].data(), static_cast<std::streamsize>(outs[i][f].size()));
            };
            std::vector<std::thread> writers;
            for (int f = 0; f < NFILES; ++f) {
                if (f == CONSOLE || !on[f]) continue;
                size_t bytes = 0;
                for (size_t i = 0; i < nb; ++i) bytes += outs[i][f].size();
                if (bytes >= (size_t(1) << 20) && threads > 1) writers.emplace_back(write_file, f);
                else if (bytes) write_file(f);
            }
            for (size_t i = 0; i < nb; ++i) console << outs[i][CONSOLE];
            for (std::thread &th : writers) th.join();
        }
        // totals (what writeBEDFile accumulates while writing) and summary counts (computeSummaryCounts)
        sum.totalPaths += static_cast<uint32_t>(paths.size());
        for (const PathData &pd : paths) {
            for (const TelomereBlock &b : pd.terminalBlocks)
                if (b.isLongest) { ++sum.totalTelomeres; telomereLengths.push_back(static_cast<float>(b.blockLen)); }
            sum.totalGaps += static_cast<uint16_t>(pd.gapInfos.size());
            if (!ui.ultraFastMode) {
                sum.totalNWindows += static_cast<uint32_t>(pd.windows.size());
                sum.totalITS += static_cast<uint32_t>(pd.interstitialBlocks.size());
                sum.totalCanMatches += static_cast<uint32_t>(pd.canonicalMatchCount);
            }
            sum.byType[static_cast<int>(pd.scaffoldType)]++;
            scaffoldLens.push_back(pd.pathSize);
            std::vector<GapInfo> gaps = pd.gapInfos;
            std::sort(gaps.begin(), gaps.end(), [](const GapInfo &a, const GapInfo &b) { return a.start < b.start; });
            uint64_t prevEnd = 0;
            for (const GapInfo &g : gaps) {
                if (g.start > prevEnd) contigLens.push_back(g.start - prevEnd);
                prevEnd = g.start + g.length;
            }
            if (pd.pathSize > prevEnd) contigLens.push_back(pd.pathSize - prevEnd);
        }
    }

    // closes the files and returns the totals printSummary reports
    AssemblySummary finish() {
        using namespace detail;
        // (flushed and closed side by side: eleven closes one after the other were 57 ms behind a 3 Gb assembly's last group)
        {
            std::atomic<bool> bad{false};
            std::vector<std::thread> closers;
            for (int f = 0; f < NFILES; ++f)
                if (on[f]) closers.emplace_back([this, f, &bad] {
                    files[f].flush();
                    if (!files[f].good()) bad.store(true);
                    files[f].close();
                });
            const auto tc0 = std::chrono::steady_clock::now();
            for (std::thread &th : closers) th.join();
            if (std::getenv("TS_MIRROR_TRACE"))
                std::fprintf(stderr, "trace closes %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count());
            if (bad.load()) throw std::runtime_error("failed while writing the output files of " + outBase);
        }
        sum.scaffoldN50 = computeN50(scaffoldLens);
        sum.contigN50 = computeN50(contigLens);
        if (!telomereLengths.empty()) {                              // getStats, src/tools.cpp:23-51
            std::vector<float> &v = telomereLengths;
            float total = 0.0f;
            sum.teloMin = sum.teloMax = v[0];
            for (float x : v) { sum.teloMin = std::min(sum.teloMin, x); sum.teloMax = std::max(sum.teloMax, x); total += x; }
            sum.teloMean = total / v.size();
            std::sort(v.begin(), v.end());
            const size_t mid = v.size() / 2;
            sum.teloMedian = v.size() % 2 ? v[mid] : (v[mid] + v[mid - 1]) / 2;
        }
        return sum;
    }
};

// `paths` must be in seqPos order (walkPaths' order).
inline void writeBEDFiles(const std::string &outBase, const std::vector<PathData> &paths, const UserInputTeloscope &ui,
                          std::ostream &console, AssemblySummary &sum, bool manualCuration = false, unsigned threads = 0) {
    BedWriter w(outBase, ui, console, manualCuration, threads);
    w.add(paths);
    sum = w.finish();
}

// printSummary: the same text to the console and (appended) to <outBase>_report.tsv when given.
inline void printSummary(std::ostream &console, const AssemblySummary &s, bool ultraFastMode, const std::string &reportFile = "") {
    using detail::line;
    std::string t;
    line(t, "\n+++ Assembly Summary Report +++\n", "Total paths:\t", uint64_t(s.totalPaths), '\n', "Total gaps:\t", uint64_t(s.totalGaps), '\n',
         "Scaffold N50:\t", s.scaffoldN50, '\n', "Contig N50:\t", s.contigN50, '\n', "Total telomeres:\t", uint64_t(s.totalTelomeres), '\n');
    if (!ultraFastMode)
        line(t, "Total ITS blocks:\t", uint64_t(s.totalITS), '\n', "Total canonical matches:\t", uint64_t(s.totalCanMatches), '\n',
             "Total windows analyzed:\t", uint64_t(s.totalNWindows), '\n');
    line(t, "\n+++ Telomere Statistics +++\n");
    if (s.totalTelomeres > 0)
        line(t, "Mean length:\t", s.teloMean, '\n', "Median length:\t", s.teloMedian, '\n', "Min length:\t", s.teloMin, '\n',
             "Max length:\t", s.teloMax, '\n');
    else
        line(t, "No telomeres found for statistics.\n");
    const uint32_t *c = s.byType;
    line(t, "\n+++ Chromosome Telomere Counts+++\n", "Two telomeres:\t", uint64_t(c[0] + c[1] + c[2] + c[3]), '\n',
         "One telomere:\t", uint64_t(c[4] + c[5]), '\n', "Zero telomeres:\t", uint64_t(c[6] + c[7]), '\n');
    line(t, "\n+++ Chromosome Telomere/Gap Completeness+++\n", "T2T:\t", uint64_t(c[0]), '\n', "Gapped T2T:\t", uint64_t(c[1]), '\n',
         "Misassembled:\t", uint64_t(c[2]), '\n', "Gapped misassembled:\t", uint64_t(c[3]), '\n', "Incomplete:\t", uint64_t(c[4]), '\n',
         "Gapped incomplete:\t", uint64_t(c[5]), '\n', "No telomeres:\t", uint64_t(c[6]), '\n', "Gapped no telomeres:\t", uint64_t(c[7]), '\n',
         "Discordant:\t", uint64_t(c[8]), '\n', "Gapped discordant:\t", uint64_t(c[9]), '\n');
    console << t;
    if (!reportFile.empty()) {
        std::ofstream f(reportFile, std::ios::binary | std::ios::app);
        if (!f) throw std::runtime_error("Could not open '" + reportFile + "' for writing.");
        f << t;
    }
}

// ---------------------------------------------------------------------------------------------
// FASTA text in, all output files out, as ONE pipeline (rows f2 + f3 together).  The reference loads the whole
// genome, runs one job per path, sorts and then writes (Input::read, src/input.cpp:575-734); with the scan in
// milliseconds that order leaves three whole-genome phases — parse, scan, format — one after the other on the
// host.  Here records flow in groups of ~256 MB through three stages that overlap:
//
//   read    a mapped file's records are located by one parallel pass over the text; a group's body lines are
//           joined into one buffer per record by all host threads (two passes: count, copy — no zero-fill, no
//           per-line allocation)                                                    [reader thread + pool]
//   scan    splitPath + ONE batched scan of the group's segments + labelTerminalBlocks (walkRecordViews)
//   write   BedWriter::add formats the group's lines on the host threads and appends them to the files
//
// Groups are consecutive runs of records, so everything leaves in seqPos order: the files and the console text
// are byte-identical to readFasta + walkPaths + writeBEDFiles (tests/test_writers.py runs both).
namespace detail {

// bases of [p, end) without the line ends ('\n', and a '\r' right before it or at the very end); [p, end) starts
// at a line start.  Eight bytes at a time: line ends are counted, not searched for.
inline size_t countFastaBases(const char *p, const char *end) {
    const size_t len = static_cast<size_t>(end - p);
    const uint64_t ones = 0x0101010101010101ull, low7 = 0x7F7F7F7F7F7F7F7Full;
    auto zeroBytes = [&](uint64_t v) { return ~(((v & low7) + low7) | v | low7); };            // 0x80 in every zero byte, exactly
    size_t nl = 0, cr = 0, i = 0;
    for (; i + 8 <= len; i += 8) {
        uint64_t v;
        std::memcpy(&v, p + i, 8);
        nl += static_cast<size_t>(__builtin_popcountll(zeroBytes(v ^ (ones * 0x0A))));
        cr += static_cast<size_t>(__builtin_popcountll(zeroBytes(v ^ (ones * 0x0D))));
    }
    for (; i < len; ++i) { nl += p[i] == '\n'; cr += p[i] == '\r'; }
    if (cr == 0) return len - nl;
    // carriage returns: only those that end a line are dropped (rare input: count them the careful way)
    size_t n = 0;
    while (p < end) {
        const char *q = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
        const char *stop = q ? q : end;
        if (stop > p) n += static_cast<size_t>((stop[-1] == '\r' ? stop - 1 : stop) - p);
        p = q ? q + 1 : end;
    }
    return n;
}
inline char *copyFastaBases(const char *p, const char *end, char *dst) {
    while (p < end) {
        const char *nl = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
        const char *stop = nl ? nl : end;
        if (stop > p) {
            const size_t n = static_cast<size_t>((stop[-1] == '\r' ? stop - 1 : stop) - p);
            std::memcpy(dst, p, n);
            dst += n;
        }
        p = nl ? nl + 1 : end;
    }
    return dst;
}

struct RawRecord {                                              // a record of a streamed group
    std::string header;
    std::unique_ptr<char[]> data;                               // joined bases (new char[]: not zero-filled) — or,
    std::vector<ts_text_piece> pieces;                          // text mode: the record's body text where the file is mapped
    std::vector<TextLines> lines;                               //            and how each piece's lines lie (for matchSeq)
    size_t size = 0;                                            // bases
};

struct FastaGroup {
    size_t firstRecord = 0;
    std::vector<RawRecord> records;
    std::vector<PathComponents> comps;                          // of `records`, found while their bases were copied
    std::shared_ptr<void> mapping;                              // text mode: the records' pieces point into the mapped file
    std::vector<FastaRecord> owned;                             // gzip / stdin input: records read the plain way
};

}  // namespace detail

// Groups of consecutive records of a FASTA file, each with its records' lines joined.
class FastaGroupReader {
    struct Span { const char *head, *body, *stop; };
    int fd = -1;
    void *map = nullptr;
    size_t mapSize = 0;
    std::shared_ptr<void> mapping;                              // unmaps when the reader AND every group that points into it are gone
    std::vector<Span> spans;
    size_t nextSpan = 0;
    size_t groupBytes, pieceBytes;
    bool textPieces;                                            // records as text pieces (line ends skipped by the library's staging)
    std::vector<FastaRecord> all;                               // not a mapped plain file: everything was read up front
    bool mapped = false;

public:
    // pieceBytes: text bytes a host thread handles at a time (a record's lines are counted / joined by pieces, in parallel).
    // textPieces: do not join the lines at all — a record is the list of its text pieces in the mapped file, with their
    // base counts and N-runs (one pass over the text); the library strips the line ends while it stages the upload
    // (TS_INPUT_TEXT_PIECES).  Only for a mapped plain file; gzip / stdin input is joined by zlib's reader anyway.
    explicit FastaGroupReader(const std::string &file, size_t groupBytes_ = size_t(256) << 20, size_t pieceBytes_ = size_t(4) << 20,
                              bool textPieces_ = false)
        : groupBytes(std::max<size_t>(groupBytes_, 1)), pieceBytes(std::min<size_t>(std::max<size_t>(pieceBytes_, 1), size_t(16) << 20)),
          textPieces(textPieces_) {
        fd = ::open(file.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open " + file);
        struct stat sb;
        unsigned char magic[2] = {0, 0};
        const bool gz = ::pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        if (!gz && ::fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
            mapSize = static_cast<size_t>(sb.st_size);
            map = ::mmap(nullptr, mapSize, PROT_READ, MAP_PRIVATE, fd, 0);
            if (map == MAP_FAILED) map = nullptr;
        }
        if (!map) {                                              // gzip (zlib is one serial stream anyway), FIFOs, empty files
            ::close(fd); fd = -1;
            all = readFasta(file);
            return;
        }
        mapped = true;
        {
            const size_t n = mapSize;
            mapping = std::shared_ptr<void>(map, [n](void *p) { ::munmap(p, n); });
        }
        (void)::madvise(map, mapSize, MADV_SEQUENTIAL);
        const char *data = static_cast<const char *>(map), *end = data + mapSize;
        // record starts = '>' at a line start; found by slices of the text in parallel, then put in order
        const size_t nslice = std::max<size_t>(1, std::min<size_t>(64, mapSize >> 24));
        std::vector<std::vector<const char *>> found(nslice);
        detail::onThreads(nslice, [&](size_t k) {
            const char *lo = data + mapSize / nslice * k, *hi = k + 1 == nslice ? end : data + mapSize / nslice * (k + 1);
            for (const char *p = lo; p < hi;) {
                const char *gt = static_cast<const char *>(std::memchr(p, '>', static_cast<size_t>(hi - p)));
                if (!gt) break;
                if (gt == data || gt[-1] == '\n') found[k].push_back(gt);
                p = gt + 1;
            }
        });
        std::vector<const char *> starts;
        for (const auto &v : found) starts.insert(starts.end(), v.begin(), v.end());
        for (size_t i = 0; i < starts.size(); ++i) {
            const char *gt = starts[i];
            const char *lim = i + 1 < starts.size() ? starts[i + 1] : end;
            const char *nl = static_cast<const char *>(std::memchr(gt, '\n', static_cast<size_t>(lim - gt)));
            spans.push_back(Span{gt + 1, nl ? nl + 1 : lim, lim});
        }
    }
    ~FastaGroupReader() {
        if (map && !mapping) ::munmap(map, mapSize);
        if (fd >= 0) ::close(fd);
    }
    FastaGroupReader(const FastaGroupReader &) = delete;
    FastaGroupReader &operator=(const FastaGroupReader &) = delete;

    // the next group (false at the end of the file)
    bool next(detail::FastaGroup &g) {
        g = detail::FastaGroup{};
        if (!mapped) {
            if (nextSpan >= all.size()) return false;
            g.firstRecord = nextSpan;
            size_t bytes = 0;
            while (nextSpan < all.size() && (g.owned.empty() || bytes + all[nextSpan].sequence.size() <= groupBytes)) {
                bytes += all[nextSpan].sequence.size();
                g.owned.push_back(std::move(all[nextSpan++]));
            }
            return true;
        }
        if (nextSpan >= spans.size()) return false;
        g.firstRecord = nextSpan;
        size_t first = nextSpan, bytes = 0;
        while (nextSpan < spans.size() && (nextSpan == first || bytes + static_cast<size_t>(spans[nextSpan].stop - spans[nextSpan].body) <= groupBytes)) {
            bytes += static_cast<size_t>(spans[nextSpan].stop - spans[nextSpan].body);
            ++nextSpan;
        }
        const size_t nrec = nextSpan - first;
        g.records.resize(nrec);
        // pieces of ~4 MB of text, cut at line starts; pass 1 counts every piece's bases, pass 2 copies them to
        // their place in the record's buffer
        struct Piece { size_t rec; const char *a, *z; size_t bases, at; PathComponents pc; bool firstIsGap, lastIsGap; TextLines lines; };
        std::vector<Piece> pieces;
        const size_t kPiece = pieceBytes;
        for (size_t r = 0; r < nrec; ++r) {
            const Span &sp = spans[first + r];
            const char *he = sp.body > sp.head && sp.body[-1] == '\n' ? sp.body - 1 : sp.body;
            g.records[r].header = detail::fastaHeaderWord(sp.head, he);
            for (const char *a = sp.body; a < sp.stop;) {
                const char *z = sp.stop - a > static_cast<ptrdiff_t>(kPiece) ? a + kPiece : sp.stop;
                if (z < sp.stop) {
                    const char *nl = static_cast<const char *>(std::memchr(z, '\n', static_cast<size_t>(sp.stop - z)));
                    z = nl ? nl + 1 : sp.stop;
                }
                pieces.push_back(Piece{r, a, z, 0, 0, {}, false, false, TextLines{}});
                a = z;
            }
        }
        if (textPieces) {
            g.mapping = mapping;
            // ONE pass over the text: every piece's base count and N-runs; nothing is copied
            detail::onThreads(pieces.size(), [&](size_t i) {
                Piece &p = pieces[i];
                uint64_t nb = 0;
                p.pc = detail::splitPathText(p.a, p.z, &nb, &p.firstIsGap, &p.lastIsGap, &p.lines);
                p.bases = static_cast<size_t>(nb);
            });
            for (size_t i = 0; i < pieces.size(); ++i) {
                Piece &p = pieces[i];
                p.at = g.records[p.rec].size;
                g.records[p.rec].size += p.bases;
                for (auto &sg : p.pc.segments) sg.first += p.at;
                for (GapInfo &gp : p.pc.gaps) gp.start += p.at;
                if (p.bases) {
                    g.records[p.rec].pieces.push_back(ts_text_piece{p.a, static_cast<uint64_t>(p.z - p.a), static_cast<uint64_t>(p.bases)});
                    g.records[p.rec].lines.push_back(p.lines);
                }
            }
        } else {
        detail::onThreads(pieces.size(), [&](size_t i) { pieces[i].bases = detail::countFastaBases(pieces[i].a, pieces[i].z); });
        for (size_t i = 0; i < pieces.size(); ++i) {
            pieces[i].at = g.records[pieces[i].rec].size;
            g.records[pieces[i].rec].size += pieces[i].bases;
        }
        for (detail::RawRecord &rr : g.records) rr.data.reset(new char[rr.size + 1]);
        detail::onThreads(pieces.size(), [&](size_t i) {
            Piece &p = pieces[i];
            char *dst = g.records[p.rec].data.get() + p.at;
            (void)detail::copyFastaBases(p.a, p.z, dst);
            if (!p.bases) return;
            // N-runs of the piece, while its bases are still in this core's cache (in record coordinates)
            p.pc = splitPath(dst, p.bases);
            for (auto &sg : p.pc.segments) sg.first += p.at;
            for (GapInfo &gp : p.pc.gaps) gp.start += p.at;
            auto isGap = [](char c) { return c == 'N' || c == 'n' || c == 'X' || c == 'x'; };
            p.firstIsGap = isGap(dst[0]);
            p.lastIsGap = isGap(dst[p.bases - 1]);
        });
        }
        g.comps.resize(nrec);
        std::vector<int> prevLast(nrec, -1);                      // -1: no bases of the record yet
        for (const Piece &p : pieces) {
            if (!p.bases) continue;
            detail::appendPieceRuns(g.comps[p.rec], p.pc, prevLast[p.rec] >= 0, prevLast[p.rec] == 1, p.firstIsGap);
            prevLast[p.rec] = p.lastIsGap ? 1 : 0;
        }
        return true;
    }
};

struct ScanFastaTimes { double read_ms = 0, scan_ms = 0, write_ms = 0, wall_ms = 0; uint64_t bases = 0, windows = 0; size_t groups = 0; };

// FASTA file -> the eleven output files + console path report; returns the totals for printSummary.
inline AssemblySummary scanFastaToFiles(Teloscope &teloscope, const std::string &fastaFile, const std::string &outBase,
                                        std::ostream &console, bool manualCuration = false,
                                        size_t groupBytes = size_t(256) << 20, ScanFastaTimes *times = nullptr,
                                        size_t pieceBytes = size_t(4) << 20, int textPieces = -1) {
    // textPieces: -1 = whenever the library takes text input for this parameter set (the tiled kernel's), 0 / 1 = forced
    using Clock = std::chrono::steady_clock;
    auto ms = [](Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t_begin = Clock::now();
    struct Scanned { detail::FastaGroup group; std::vector<PathData> paths; };
    detail::BoundedQueue<detail::FastaGroup> toScan(2);
    detail::BoundedQueue<Scanned> toWrite(2);
    std::exception_ptr readError, scanError;
    ScanFastaTimes T;
    const bool trace = std::getenv("TS_MIRROR_TRACE") != nullptr;      // per-group stage intervals on stderr (ms since the call began)

    std::thread reader([&] {
        try {
            const bool text = textPieces < 0 ? teloscope.takesTextPieces() : textPieces != 0;
            FastaGroupReader rd(fastaFile, groupBytes, pieceBytes, text);
            detail::FastaGroup g;
            for (;;) {
                const auto t0 = Clock::now();
                if (!rd.next(g)) break;
                T.read_ms += ms(t0, Clock::now());
                toScan.push(std::move(g));
            }
        } catch (...) { readError = std::current_exception(); }
        toScan.close();
    });
    std::thread scanner([&] {
        try {
            detail::FastaGroup g;
            // The library's first call in a process pays for its streams, its kernels' code, the device pool and the pinned staging
            // (81-115 ms against 12-15 ms for a later group of 256 MB): made here, on a dummy record, while the reader is still
            // parsing its first group.  TS_MIRROR_WARMUP=0 leaves it out (A/B).
            if (const char *wu = std::getenv("TS_MIRROR_WARMUP"); !(wu && wu[0] == '0')) {
                const auto tw = Clock::now();
                const std::string hdr = "warm-up", seq(size_t(1) << 20, 'A');
                std::vector<RecordView> one{RecordView{&hdr, seq.data(), seq.size(), nullptr, 0, nullptr}};
                try { (void)walkRecordViews(teloscope, one, 0, nullptr); } catch (...) {}
                if (trace) std::fprintf(stderr, "trace warm  %7.1f .. %7.1f\n", ms(t_begin, tw), ms(t_begin, Clock::now()));
            }
            while (toScan.pop(g)) {
                const auto t0 = Clock::now();
                std::vector<RecordView> views;
                for (const detail::RawRecord &r : g.records)
                    views.push_back(r.pieces.empty() && r.size ? RecordView{&r.header, r.data.get(), r.size, nullptr, 0, nullptr}
                                                               : RecordView{&r.header, nullptr, r.size, r.pieces.data(), r.pieces.size(), r.lines.data()});
                for (const FastaRecord &r : g.owned) views.push_back(RecordView{&r.header, r.sequence.data(), r.sequence.size(), nullptr, 0, nullptr});
                Scanned s;
                s.paths = walkRecordViews(teloscope, views, g.firstRecord, g.owned.empty() ? &g.comps : nullptr);
                s.group = std::move(g);                           // (-m: matchSeq was copied out of the bases already)
                T.scan_ms += ms(t0, Clock::now());
                if (trace) std::fprintf(stderr, "trace scan  %7.1f .. %7.1f\n", ms(t_begin, t0), ms(t_begin, Clock::now()));
                toWrite.push(std::move(s));
            }
        } catch (...) {
            scanError = std::current_exception();
            detail::FastaGroup drop;
            while (toScan.pop(drop)) {}
        }
        toWrite.close();
    });
    AssemblySummary sum;
    std::exception_ptr writeError;
    try {
        BedWriter writer(outBase, teloscope.input(), console, manualCuration);
        Scanned s;
        while (toWrite.pop(s)) {
            const auto t0 = Clock::now();
            writer.add(s.paths);
            for (const PathData &pd : s.paths) { T.bases += pd.pathSize; T.windows += pd.windows.size(); }
            ++T.groups;
            T.write_ms += ms(t0, Clock::now());
            if (trace) std::fprintf(stderr, "trace write %7.1f .. %7.1f\n", ms(t_begin, t0), ms(t_begin, Clock::now()));
        }
        const auto tf = Clock::now();
        sum = writer.finish();
        if (trace) std::fprintf(stderr, "trace finish %7.1f .. %7.1f\n", ms(t_begin, tf), ms(t_begin, Clock::now()));
    } catch (...) {
        writeError = std::current_exception();
        Scanned drop;
        while (toWrite.pop(drop)) {}
    }
    reader.join();
    scanner.join();
    if (readError) std::rethrow_exception(readError);
    if (scanError) std::rethrow_exception(scanError);
    if (writeError) std::rethrow_exception(writeError);
    T.wall_ms = ms(t_begin, Clock::now());
    if (times) *times = T;
    return sum;
}

}  // namespace teloscope_mi355x

#endif
