// manifest_cli.cpp — TEST INFRASTRUCTURE: a C++17 driver that exercises the host-side mirror
// (include/teloscope_mi355x.hpp) the way the reference's front-end exercises its Teloscope class, so
// that the reference's own `.tst` manifests can be replayed through the GPU from C++:
//   option loop of src/main.cpp:186-565 (flags the manifests use), FASTA -> '+' segments / N-gaps
//   (gfalibs behaviour pinned by testFiles/expected/*_gaps.bed), walkPath (src/input.cpp:942-1041)
//   with ONE batched scanSegments call for all segments, and the stdout of writeBEDFile /
//   printSummary (src/teloscope.cpp:687-694, 815-857, 959-1055).
// Usage: manifest_cli <flags as in the manifest's first line, input path already resolved>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "teloscope_mi355x.hpp"

using namespace teloscope_mi355x;

struct Path {
    std::string header, seq;
    std::vector<std::pair<uint64_t, uint32_t>> gaps;
    std::vector<std::pair<uint64_t, std::string>> segs;      // (absPos, upper-cased sequence)
};

static std::vector<Path> readFasta(const std::string &file) {
    std::ifstream in(file);
    if (!in) { fprintf(stderr, "Error: cannot open %s\n", file.c_str()); exit(EXIT_FAILURE); }
    std::vector<Path> paths;
    std::string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (!line.empty() && line[0] == '>') {
            Path p;
            std::istringstream hs(line.substr(1));
            hs >> p.header;
            paths.push_back(p);
        } else if (!paths.empty()) {
            paths.back().seq += line;
        }
    }
    for (Path &p : paths) {
        size_t i = 0, n = p.seq.size();
        while (i < n) {
            const bool gap = p.seq[i] == 'N' || p.seq[i] == 'n' || p.seq[i] == 'X' || p.seq[i] == 'x';
            size_t j = i;
            while (j < n && ((p.seq[j] == 'N' || p.seq[j] == 'n' || p.seq[j] == 'X' || p.seq[j] == 'x') == gap)) ++j;
            if (gap) p.gaps.emplace_back(i, static_cast<uint32_t>(j - i));
            else {
                std::string s = p.seq.substr(i, j - i);
                for (char &c : s) c = static_cast<char>(toupper(static_cast<unsigned char>(c)));   // unmaskSequence
                p.segs.emplace_back(i, std::move(s));
            }
            i = j;
        }
    }
    return paths;
}

static uint64_t n50(std::vector<uint64_t> v) {                // Teloscope::computeN50, include/teloscope.h:224-235
    if (v.empty()) return 0;
    std::sort(v.begin(), v.end(), [](uint64_t a, uint64_t b) { return a > b; });
    uint64_t total = 0, cum = 0;
    for (uint64_t l : v) total += l;
    for (uint64_t l : v) { cum += l; if (cum * 2 >= total) return l; }
    return v.back();
}

int main(int argc, char **argv) {
    UserInputTeloscope ui;
    std::string input, canonical;
    std::vector<std::string> rawPatterns;
    bool hasPatterns = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) exit(EXIT_FAILURE); return argv[++i]; };
        if (a == "-f") input = val();
        else if (a == "-o" || a == "-j") (void)val();
        else if (a == "-c") canonical = val();
        else if (a == "-p") {
            hasPatterns = true;
            std::istringstream ps(val());
            std::string p;
            while (std::getline(ps, p, ',')) if (!p.empty()) rawPatterns.push_back(p);
        }
        else if (a == "-w") ui.windowSize = std::stoi(val());
        else if (a == "-s") ui.step = std::stoi(val());
        else if (a == "-t") ui.terminalLimit = std::stoi(val());
        else if (a == "-k") ui.maxMatchDist = static_cast<unsigned short>(std::stoi(val()));
        else if (a == "-d") ui.maxBlockDist = static_cast<unsigned short>(std::stoi(val()));
        else if (a == "-l") { ui.minBlockLen = static_cast<unsigned short>(std::stoi(val())); ui.minBlockLenSet = true; }
        else if (a == "-y") ui.minBlockDensity = std::stof(val());
        else if (a == "-x") ui.editDistance = static_cast<uint8_t>(std::stoi(val()));
        else if (a == "-r") { ui.outWinRepeats = true; ui.ultraFastMode = false; }
        else if (a == "-g") { ui.outGC = true; ui.ultraFastMode = false; }
        else if (a == "-e") { ui.outEntropy = true; ui.ultraFastMode = false; }
        else if (a == "-m") { ui.outMatches = true; ui.ultraFastMode = false; }
        else if (a == "-i") { ui.outITS = true; ui.ultraFastMode = false; }
        else if (a == "-a") ui.ultraFastMode = false;
        else if (a == "-u") ui.ultraFastMode = !(ui.outWinRepeats || ui.outGC || ui.outEntropy || ui.outITS || ui.outMatches);
        else if (a == "-n") {}
        else if (!a.empty() && a[0] != '-' && input.empty()) input = a;
    }
    if (ui.step > ui.windowSize) { fprintf(stderr, "Error: Step size cannot be larger than window size.\n"); return EXIT_FAILURE; }
    try {
        if (!canonical.empty()) setCanonical(ui, canonical);
        ui.rawPatterns = (hasPatterns && !rawPatterns.empty()) ? rawPatterns
                       : std::vector<std::string>{ui.canonicalFwd, ui.canonicalRev};
        ui.patternInfo = expandPatternsWithOrientation(ui.rawPatterns, ui.editDistance, ui.canonicalFwd);
        Teloscope teloscope(ui);

        std::vector<Path> paths = readFasta(input);
        std::vector<Teloscope::Segment> batch;                 // every '+' segment of every path, one launch
        for (const Path &p : paths)
            for (const auto &sg : p.segs) batch.push_back(Teloscope::Segment{&sg.second, sg.first, ui.ultraFastMode});
        std::vector<SegmentData> scanned = teloscope.scanSegments(batch);

        std::cout << "\n+++ Path Summary Report +++\n";
        std::cout << (ui.ultraFastMode ? "pos\theader\ttelomeres\tlabels\tgaps\ttype\tgranular\n"
                                        : "pos\theader\ttelomeres\tlabels\tgaps\ttype\tgranular\tits\tcanonical\twindows\n");
        static const char *typeNames[] = {"t2t", "gapped_t2t", "misassembly", "gapped_misassembly", "incomplete",
                                          "gapped_incomplete", "none", "gapped_none", "discordant", "gapped_discordant"};
        uint32_t counts[10] = {0}, totalTelomeres = 0, totalGaps = 0, totalITS = 0, totalCan = 0, totalWin = 0;
        std::vector<float> teloLens;
        std::vector<uint64_t> scafLens, contigLens;
        size_t si = 0;
        for (size_t pi = 0; pi < paths.size(); ++pi) {
            const Path &p = paths[pi];
            std::vector<TelomereBlock> terminal;
            size_t its = 0, can = 0, win = 0;
            for (size_t k = 0; k < p.segs.size(); ++k, ++si) {
                const SegmentData &sd = scanned[si];
                terminal.insert(terminal.end(), sd.terminalBlocks.begin(), sd.terminalBlocks.end());
                its += sd.interstitialBlocks.size(); can += sd.canonicalMatches.size(); win += sd.windows.size();
            }
            std::string label;
            ScaffoldType type;
            teloscope.labelTerminalBlocks(terminal, static_cast<uint16_t>(p.gaps.size()), label, type, p.seq.size(), ui.terminalLimit);
            int longest = 0;
            std::string labels;
            for (const TelomereBlock &b : terminal)
                if (b.isLongest) { ++longest; labels += b.blockLabel; teloLens.push_back(static_cast<float>(b.blockLen)); }
            std::cout << pi + 1 << "\t" << p.header << "\t" << longest << "\t" << (labels.empty() ? "none" : labels) << "\t"
                      << static_cast<uint16_t>(p.gaps.size()) << "\t" << typeNames[static_cast<int>(type)] << "\t" << label;
            if (!ui.ultraFastMode) std::cout << "\t" << its << "\t" << can << "\t" << win;
            std::cout << "\n";
            totalTelomeres += longest; totalGaps += static_cast<uint16_t>(p.gaps.size());
            totalITS += its; totalCan += can; totalWin += win;
            counts[static_cast<int>(type)]++;
            scafLens.push_back(p.seq.size());
            uint64_t prevEnd = 0;
            for (const auto &g : p.gaps) { if (g.first > prevEnd) contigLens.push_back(g.first - prevEnd); prevEnd = g.first + g.second; }
            if (p.seq.size() > prevEnd) contigLens.push_back(p.seq.size() - prevEnd);
        }
        std::cout << "\n+++ Assembly Summary Report +++\n";
        std::cout << "Total paths:\t" << paths.size() << "\n" << "Total gaps:\t" << totalGaps << "\n"
                  << "Scaffold N50:\t" << n50(scafLens) << "\n" << "Contig N50:\t" << n50(contigLens) << "\n"
                  << "Total telomeres:\t" << totalTelomeres << "\n";
        if (!ui.ultraFastMode)
            std::cout << "Total ITS blocks:\t" << totalITS << "\n" << "Total canonical matches:\t" << totalCan << "\n"
                      << "Total windows analyzed:\t" << totalWin << "\n";
        std::cout << "\n+++ Telomere Statistics +++\n";
        if (totalTelomeres > 0) {                                // getStats, src/tools.cpp:23-51
            float sum = 0.0f, mn = teloLens[0], mx = teloLens[0];
            for (float v : teloLens) { mn = std::min(mn, v); mx = std::max(mx, v); sum += v; }
            const float mean = sum / teloLens.size();
            std::vector<float> srt = teloLens;
            std::sort(srt.begin(), srt.end());
            const size_t mid = srt.size() / 2;
            const float median = srt.size() % 2 ? srt[mid] : (srt[mid] + srt[mid - 1]) / 2;
            std::cout << "Mean length:\t" << mean << "\n" << "Median length:\t" << median << "\n"
                      << "Min length:\t" << mn << "\n" << "Max length:\t" << mx << "\n";
        } else {
            std::cout << "No telomeres found for statistics.\n";
        }
        std::cout << "\n+++ Chromosome Telomere Counts+++\n"
                  << "Two telomeres:\t" << counts[0] + counts[1] + counts[2] + counts[3] << "\n"
                  << "One telomere:\t" << counts[4] + counts[5] << "\n" << "Zero telomeres:\t" << counts[6] + counts[7] << "\n";
        std::cout << "\n+++ Chromosome Telomere/Gap Completeness+++\n"
                  << "T2T:\t" << counts[0] << "\n" << "Gapped T2T:\t" << counts[1] << "\n"
                  << "Misassembled:\t" << counts[2] << "\n" << "Gapped misassembled:\t" << counts[3] << "\n"
                  << "Incomplete:\t" << counts[4] << "\n" << "Gapped incomplete:\t" << counts[5] << "\n"
                  << "No telomeres:\t" << counts[6] << "\n" << "Gapped no telomeres:\t" << counts[7] << "\n"
                  << "Discordant:\t" << counts[8] << "\n" << "Gapped discordant:\t" << counts[9] << "\n";
    } catch (const std::exception &e) {
        fprintf(stderr, "Error: %s\n", e.what());
        return EXIT_FAILURE;
    }
    return 0;
}
