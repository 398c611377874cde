// manifest_cli.cpp — TEST INFRASTRUCTURE: a C++17 driver that exercises the host-side mirror
// (include/teloscope_mi355x.hpp) the way the reference's front-end exercises its Teloscope class, so
// that the reference's own `.tst` manifests can be replayed through the GPU from C++:
//   option loop of src/main.cpp:186-565 (flags the manifests use), FASTA -> '+' segments / N-gaps
//   (gfalibs behaviour pinned by testFiles/expected/*_gaps.bed), walkPath (src/input.cpp:942-1041)
//   with ONE batched scanSegments call for all segments, and writeBEDFile / printSummary
//   (src/teloscope.cpp:661-1055) — the last three through include/teloscope_mi355x_io.hpp.
// Usage: manifest_cli <flags as in the manifest's first line, input path already resolved>
//        [--out-base <prefix>]   where the eleven output files go (default: a scratch prefix)
//        [--read-devices 0,0]    read filter over several contexts (one per listed HIP ordinal; repeats allowed)
//        [--devices 0,0]         the assembly scan over several contexts: every batch of segments cut into one shard per context
//        [--reads-per-batch n]   reads per GPU batch of --fastq-subset / --bam-subset (test hook)
//        [--bam-chunk-bytes n]   uncompressed BAM bytes inflated at a time by --bam-subset (test hook; at least 1 MiB)
//        [--no-stream]           readFasta, then walkPaths, then writeBEDFiles (default: the three overlap in
//                                scanFastaToFiles, records flowing in groups)   [--group-bytes n: group size, test hook]
//        [--join-lines]          streaming, but a record's lines are joined on the host (default: text pieces — the
//                                library strips the line ends while staging)     [--piece-bytes n: test hook]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <unistd.h>
#include <sstream>
#include <string>
#include <vector>

#include "teloscope_mi355x_io.hpp"

using namespace teloscope_mi355x;

int main(int argc, char **argv) {
    UserInputTeloscope ui;
    std::string input, canonical, outBase;
    bool scratch = false, manualCuration = false, fastqSubsetMode = false, bamSubsetMode = false;
    size_t fastqBlock = 512u << 20, readsPerBatch = 1u << 20, groupBytes = size_t(256) << 20, pieceBytes = size_t(4) << 20, bamChunk = size_t(256) << 20;
    bool stream = true;
    int textPieces = -1;
    std::vector<int> readDevices, scanDevices;
    std::string bamList;
    std::vector<std::string> rawPatterns;
    bool hasPatterns = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) exit(EXIT_FAILURE); return argv[++i]; };
        if (a == "-f") input = val();
        else if (a == "--fastq-subset") fastqSubsetMode = true;
        else if (a == "--bam-subset") bamSubsetMode = true;
        else if (a == "--bam-subset-each") { bamSubsetMode = true; bamList = val(); }   // test hook: one filter, many inputs (below)
        else if (a == "--fastq-block") fastqBlock = static_cast<size_t>(std::stoull(val()));   // test hook: arena size in bytes
        else if (a == "--out-base") outBase = val();
        else if (a == "--no-stream") stream = false;
        else if (a == "--group-bytes") groupBytes = static_cast<size_t>(std::stoull(val()));
        else if (a == "--piece-bytes") pieceBytes = static_cast<size_t>(std::stoull(val()));   // test hook: text bytes per piece
        else if (a == "--join-lines") textPieces = 0;                                            // streaming, but records joined on the host
        else if (a == "--reads-per-batch") readsPerBatch = static_cast<size_t>(std::stoull(val()));
        else if (a == "--bam-chunk-bytes") bamChunk = static_cast<size_t>(std::stoull(val()));    // test hook: uncompressed bytes inflated at a time
        else if (a == "--read-devices") {
            std::istringstream ds(val());
            std::string d;
            while (std::getline(ds, d, ',')) if (!d.empty()) readDevices.push_back(std::stoi(d));
        }
        else if (a == "--devices") {
            std::istringstream ds(val());
            std::string d;
            while (std::getline(ds, d, ',')) if (!d.empty()) scanDevices.push_back(std::stoi(d));
        }
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
        else if (a == "-n") manualCuration = true;              // --manual-curation, src/main.cpp:510
        else if (!a.empty() && a[0] != '-' && input.empty()) input = a;
    }
    if (ui.step > ui.windowSize) { fprintf(stderr, "Error: Step size cannot be larger than window size.\n"); return EXIT_FAILURE; }
    try {
        if (!canonical.empty()) setCanonical(ui, canonical);
        ui.rawPatterns = (hasPatterns && !rawPatterns.empty()) ? rawPatterns
                       : std::vector<std::string>{ui.canonicalFwd, ui.canonicalRev};
        ui.patternInfo = expandPatternsWithOrientation(ui.rawPatterns, ui.editDistance, ui.canonicalFwd);
        if (bamSubsetMode && !bamList.empty()) {
            // the mutation suite: every file named in the list through bamSubset with ONE filter (device start-up once);
            // <file>.out = the subset, or <file>.err = the error message when bamSubset threw
            ReadTelomereFilter filter(ui, readDevices);
            filter.bindThreadToDevice();
            std::ifstream list(bamList);
            std::string path;
            while (std::getline(list, path)) {
                if (path.empty()) continue;
                try {
                    std::ofstream out(path + ".out", std::ios::binary);
                    const BamSubsetStats st = bamSubset(path, out, filter, readsPerBatch, bamChunk);
                    out.close();
                    std::ofstream(path + ".ok") << st.passedRecords << " " << st.totalRecords << " " << (st.missingEofBlock ? 1 : 0) << "\n";
                } catch (const std::exception &e) {
                    std::remove((path + ".out").c_str());
                    std::ofstream(path + ".err") << e.what() << "\n";
                }
            }
            return 0;
        }
        if (bamSubsetMode) {                                    // runBamSubsetMode, src/bam.cpp:262-316
            ReadTelomereFilter filter(ui, readDevices);
            filter.bindThreadToDevice();                          // this thread and the ones bamSubset starts: the GPU's NUMA node
            const BamSubsetStats st = bamSubset(input.empty() ? "-" : input, std::cout, filter, readsPerBatch, bamChunk);
            if (st.missingEofBlock) fprintf(stderr, "Warning: BAM input is missing the BGZF EOF marker.\n");
            if (st.missingSequenceRecords)
                fprintf(stderr, "BAM subset: skipped %llu record%s without SEQ.\n", (unsigned long long)st.missingSequenceRecords,
                        st.missingSequenceRecords == 1 ? "" : "s");
            fprintf(stderr, "BAM subset: kept %llu of %llu records.\n", (unsigned long long)st.passedRecords, (unsigned long long)st.totalRecords);
            return 0;
        }
        if (fastqSubsetMode) {                                  // src/main.cpp:699-716: reads in, telomeric reads out
            const auto f0 = std::chrono::steady_clock::now();
            ReadTelomereFilter filter(ui, readDevices);
            filter.bindThreadToDevice();
            const auto f1 = std::chrono::steady_clock::now();
            const FastqSubsetResult r = fastqSubset(input.empty() ? "-" : input, std::cout, filter, readsPerBatch, fastqBlock);
            if (getenv("TS_TIMING"))
                fprintf(stderr, "manifest_cli: filter construction (device start-up) %.0f ms, fastqSubset %.0f ms\n",
                        std::chrono::duration<double, std::milli>(f1 - f0).count(),
                        std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - f1).count());
            fprintf(stderr, "FASTQ subset: kept %llu of %llu reads.\n", (unsigned long long)r.kept, (unsigned long long)r.total);
            return 0;
        }
        Teloscope teloscope(ui, scanDevices);
        teloscope.bindThreadToDevice();                           // reader, scan and writer threads start from here: the GPU's NUMA node

        const bool timing = getenv("TS_TIMING") != nullptr;      // stage times to stderr
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms = [](auto x, auto y) { return std::chrono::duration<double, std::milli>(y - x).count(); };
        if (outBase.empty()) {
            const char *tmp = getenv("TMPDIR");
            outBase = std::string(tmp ? tmp : "/tmp") + "/manifest_cli_" + std::to_string(static_cast<long>(getpid()));
            scratch = true;
        }
        const auto t0 = now();
        if (stream) {
            ScanFastaTimes T;
            const AssemblySummary summary = scanFastaToFiles(teloscope, input, outBase, std::cout, manualCuration, groupBytes, &T, pieceBytes, textPieces);
            printSummary(std::cout, summary, ui.ultraFastMode, outBase + "_report.tsv");
            if (timing)
                fprintf(stderr, "manifest_cli (streaming): %.1f Mb, %llu windows, %zu groups: wall %.0f ms = %.2f Gbases/s; stage sums (the stages "
                                "overlap): read + join lines %.0f ms, scan + host post-processing %.0f ms, format + write %.0f ms\n",
                        T.bases / 1e6, (unsigned long long)T.windows, T.groups, ms(t0, now()), T.bases / 1e6 / ms(t0, now()), T.read_ms, T.scan_ms, T.write_ms);
        } else {
            std::vector<FastaRecord> records = readFasta(input);
            const auto t1 = now();
            std::vector<PathData> paths = walkPaths(teloscope, records);
            const auto t2 = now();
            AssemblySummary summary;
            writeBEDFiles(outBase, paths, ui, std::cout, summary, manualCuration);
            printSummary(std::cout, summary, ui.ultraFastMode, outBase + "_report.tsv");
            if (timing) {
                uint64_t bases = 0, windows = 0;
                for (const PathData &pd : paths) { bases += pd.pathSize; windows += pd.windows.size(); }
                fprintf(stderr, "manifest_cli: %.1f Mb, %llu windows: read %.0f ms, walkPaths (scan + host post-processing) %.0f ms, "
                                "writeBEDFiles + printSummary %.0f ms\n", bases / 1e6, (unsigned long long)windows, ms(t0, t1), ms(t1, t2), ms(t2, now()));
            }
        }
        if (scratch)
            for (const char *sfx : {"_window_repeat_density.bedgraph", "_window_canonical_ratio.bedgraph", "_window_strand_ratio.bedgraph",
                                    "_window_gc.bedgraph", "_window_entropy.bedgraph", "_canonical_matches.bed", "_noncanonical_matches.bed",
                                    "_terminal_telomeres.bed", "_interstitial_telomeres.bed", "_gaps.bed", "_report.tsv"})
                std::remove((outBase + sfx).c_str());
    } catch (const std::exception &e) {
        fprintf(stderr, "Error: %s\n", e.what());
        return EXIT_FAILURE;
    }
    return 0;
}
