// Host-only self-test of the FASTA reader and the path splitter of include/teloscope_mi355x_io.hpp (no GPU
// call is made): the mapped, multi-threaded reader of plain files must return what the zlib stream reader
// returns for the same text gzip-compressed, and splitPath's 8-bytes-at-a-time scan must equal a per-character
// walk.  Usage: io_selftest <scratch directory>
#include "teloscope_mi355x_io.hpp"

#include <cstdlib>
#include <random>

using namespace teloscope_mi355x;

static void check(bool ok, const char *what) {
    if (!ok) { fprintf(stderr, "io_selftest: FAILED: %s\n", what); std::exit(1); }
}

int main(int argc, char **argv) {
    check(argc == 2, "usage: io_selftest <scratch directory>");
    const std::string dir = argv[1];
    std::mt19937_64 rng(11);
    // ---- splitPath against a per-character walk
    const char alpha[] = "ACGTNnXxYyOoacgtRWm";
    for (int it = 0; it < 20000; ++it) {
        const size_t n = rng() % 200;
        const int mode = static_cast<int>(rng() % 3);
        std::string s(n, 'A');
        for (size_t i = 0; i < n; ++i)
            s[i] = mode == 0 ? alpha[rng() % (sizeof(alpha) - 1)]
                             : (rng() % (mode == 1 ? 7 : 40) == 0 ? "NnXx"[rng() % 4] : alpha[rng() % (sizeof(alpha) - 1)]);
        if (mode == 2 && n > 30) for (size_t i = 5; i < 29; ++i) s[i] = "Nn"[i & 1];
        const std::string view = (std::string(rng() % 8, 'A') + s).substr(0);     // varying alignment of the data
        const PathComponents pc = splitPath(view);
        auto isGap = [](char c) { return c == 'N' || c == 'n' || c == 'X' || c == 'x'; };
        size_t i = 0, gi = 0, si = 0;
        while (i < view.size()) {
            const bool g = isGap(view[i]);
            size_t j = i;
            while (j < view.size() && isGap(view[j]) == g) ++j;
            if (g) { check(gi < pc.gaps.size() && pc.gaps[gi].start == i && pc.gaps[gi].length == j - i, "gap run"); ++gi; }
            else { check(si < pc.segments.size() && pc.segments[si].first == i && pc.segments[si].second == j - i, "segment run"); ++si; }
            i = j;
        }
        check(gi == pc.gaps.size() && si == pc.segments.size(), "run counts");
        // the same record cut into pieces that are scanned independently and stitched (splitPaths): any piece size
        const std::string header = "h";
        for (size_t piece : {size_t(1), size_t(7), size_t(64), size_t(1000), size_t(1) << 20}) {
            if (it % 16 != 0) break;                                   // (each call starts a thread pool)
            const std::vector<PathComponents> got = splitPaths({RecordView{&header, view.data(), view.size()}}, piece);
            check(got.size() == 1 && got[0].segments == pc.segments && got[0].gaps.size() == pc.gaps.size(), "stitched runs");
            for (size_t g = 0; g < pc.gaps.size() && g < got[0].gaps.size(); ++g)
                check(got[0].gaps[g].start == pc.gaps[g].start && got[0].gaps[g].length == pc.gaps[g].length, "stitched gap");
        }
    }
    // ---- readFasta: plain (mapped) vs gzip (stream) on awkward texts
    for (int it = 0; it < 40; ++it) {
        std::string text;
        if (it % 5 == 1) text += "stray line before the first header\n";
        const int nrec = 1 + static_cast<int>(rng() % 6);
        const bool crlf = it % 3 == 1;
        for (int r = 0; r < nrec; ++r) {
            text += ">rec" + std::to_string(r) + (r % 2 ? " description > with a bracket" : "") + (crlf ? "\r\n" : "\n");
            const size_t len = rng() % 5000, width = 1 + rng() % 90;
            for (size_t i = 0; i < len; ++i) {
                text += "ACGTNacgtn>"[rng() % (i % width == 0 ? 10 : 11)];      // '>' never at a line start
                if ((i + 1) % width == 0) text += crlf ? "\r\n" : "\n";
            }
            if (rng() % 3) text += crlf ? "\r\n" : "\n";
            if (rng() % 4 == 0) text += "\n";
        }
        if (it % 7 == 3 && !text.empty() && text.back() == '\n') text.pop_back();     // no newline at the end of the file
        const std::string plain = dir + "/t.fa", packed = dir + "/t.fa.gz";
        { std::ofstream f(plain, std::ios::binary); f << text; }
        { gzFile g = gzopen(packed.c_str(), "wb"); check(g != nullptr, "gzopen"); gzwrite(g, text.data(), static_cast<unsigned>(text.size())); gzclose(g); }
        const std::vector<FastaRecord> a = readFasta(plain), b = readFasta(packed);
        check(a.size() == b.size(), "record count");
        for (size_t i = 0; i < a.size(); ++i) check(a[i].header == b[i].header && a[i].sequence == b[i].sequence, "record content");
        // the streaming reader (groups of records, lines joined in pieces by the thread pool): any group size must
        // give the same records in the same order, from the mapped file and from the gzip stream
        for (const std::string &file : {plain, packed})
            for (size_t groupBytes : {size_t(1), size_t(700), size_t(1) << 20}) {
                FastaGroupReader rd(file, groupBytes, groupBytes == 700 ? size_t(37) : size_t(4) << 20);     // (tiny pieces: every record is stitched)
                detail::FastaGroup g;
                size_t at = 0;
                while (rd.next(g)) {
                    check(g.firstRecord == at, "group order");
                    check(g.comps.size() == g.records.size(), "components per record");
                    for (size_t ri = 0; ri < g.records.size(); ++ri) {
                        const detail::RawRecord &r = g.records[ri];
                        check(at < a.size() && r.header == a[at].header && std::string(r.data.get(), r.size) == a[at].sequence, "streamed record");
                        const PathComponents want = splitPath(a[at].sequence);
                        check(ri < g.comps.size() && g.comps[ri].segments == want.segments && g.comps[ri].gaps.size() == want.gaps.size(), "streamed components");
                        for (size_t q = 0; ri < g.comps.size() && q < want.gaps.size() && q < g.comps[ri].gaps.size(); ++q)
                            check(g.comps[ri].gaps[q].start == want.gaps[q].start && g.comps[ri].gaps[q].length == want.gaps[q].length, "streamed gap");
                        ++at;
                    }
                    for (const FastaRecord &r : g.owned) {
                        check(at < a.size() && r.header == a[at].header && r.sequence == a[at].sequence, "streamed record (zlib)");
                        ++at;
                    }
                }
                check(at == a.size(), "streamed record count");
            }
        // text mode (no lines joined: a record = its text pieces in the mapped file + base counts + N-runs): the pieces
        // must hold exactly the record's bases, the components must equal splitPath of the joined record, and any base
        // range must read back out of the pieces (what -m's matchSeq and the N-cut segments rely on)
        for (size_t pieceBytes : {size_t(1), size_t(53), size_t(4) << 20}) {
            FastaGroupReader rd(plain, 900, pieceBytes, true);
            detail::FastaGroup g;
            size_t at = 0;
            while (rd.next(g)) {
                for (size_t ri = 0; ri < g.records.size(); ++ri, ++at) {
                    const detail::RawRecord &r = g.records[ri];
                    check(at < a.size() && r.header == a[at].header && r.size == a[at].sequence.size(), "text record size");
                    if (at >= a.size()) continue;
                    uint64_t sum = 0;
                    for (const ts_text_piece &t : r.pieces) sum += t.n_bases;
                    check(sum == r.size, "text pieces hold the record's bases");
                    const PathComponents want = splitPath(a[at].sequence);
                    check(g.comps[ri].segments == want.segments && g.comps[ri].gaps.size() == want.gaps.size(), "text components");
                    for (size_t q = 0; q < want.gaps.size() && q < g.comps[ri].gaps.size(); ++q)
                        check(g.comps[ri].gaps[q].start == want.gaps[q].start && g.comps[ri].gaps[q].length == want.gaps[q].length, "text gap");
                    if (r.size) {
                        check(r.lines.size() == r.pieces.size(), "line structure per piece");
                        const Teloscope::Segment seg(r.pieces.data(), r.size, 0, false, (at % 2) ? r.lines.data() : nullptr);
                        for (int probe = 0; probe < 12; ++probe) {
                            const uint64_t pos = rng() % r.size, n = 1 + rng() % std::min<uint64_t>(r.size - pos, 200);
                            std::string want2 = a[at].sequence.substr(pos, n);
                            for (char &ch : want2) if (ch >= 'a' && ch <= 'z') ch = static_cast<char>(ch - 32);
                            check(seg.bases(pos, n) == want2, "bases out of text pieces");
                            check(*detail::textLocate(r.pieces[0].text, r.pieces[0].text_len, 0) == a[at].sequence[0] || r.pieces[0].n_bases == 0, "textLocate");
                        }
                    }
                }
            }
            check(at == a.size(), "text record count");
        }
    }
    // ---- splitFastqRecords (pieces parsed in parallel) against the sequential piece parser on awkward FASTQ texts
    size_t accepted = 0, offered = 0;
    for (int it = 0; it < 24; ++it) {
        std::string text;
        const int nrec = 50 + static_cast<int>(rng() % 400);
        const bool crlf = it % 4 == 1;
        const char *eol = crlf ? "\r\n" : "\n";
        for (int r = 0; r < nrec; ++r) {
            if (rng() % 11 == 0) text += eol;                                   // blank line before a header
            const size_t len = rng() % 300;
            text += "@r" + std::to_string(r) + (r % 3 ? " extra @ + words" : "") + eol;
            for (size_t i = 0; i < len; ++i) text += "ACGTN"[rng() % 5];
            text += eol;
            text += (r % 5 == 0 ? std::string("+r") + std::to_string(r) : std::string("+")) + eol;
            for (size_t i = 0; i < len; ++i) text += (i == 0 && rng() % 3 == 0) ? '@' : static_cast<char>('!' + rng() % 60);   // quality lines may begin with '@' or '+'
            text += eol;
        }
        if (it % 5 == 2) { text.pop_back(); if (crlf) text.pop_back(); }         // no newline at the end of the file
        if (it % 13 == 7) text.insert(text.size() / 2, "garbage line\n");         // malformed: must be refused, not mis-split
        std::vector<detail::FastqRec> seq, par;
        const char *stop = detail::parseFastqPiece(text.data(), text.data() + text.size(), text.data() + text.size(), seq);
        const bool wellFormed = stop == text.data() + text.size();
        for (size_t piece : {size_t(200), size_t(1500), size_t(20000)}) {
            const bool ok = detail::splitFastqRecords(text.data(), text.size(), par, piece);
            if (!wellFormed) { check(!ok, "malformed FASTQ accepted by the parallel splitter"); continue; }
            ++offered;
            if (!ok) continue;                                                  // refusing is always allowed (sequential fallback)
            ++accepted;
            check(par.size() == seq.size(), "parallel FASTQ record count");
            for (size_t i = 0; i < seq.size() && i < par.size(); ++i)
                check(par[i].begin == seq[i].begin && par[i].seq == seq[i].seq && par[i].end == seq[i].end && par[i].seqLen == seq[i].seqLen, "parallel FASTQ record");
        }
    }
    check(offered > 20 && accepted * 10 >= offered * 9, "the parallel splitter refuses well-formed FASTQ too often");
    puts("io_selftest ok");
    return 0;
}
