// rank_scan.cpp — one RANK of a sharded assembly scan written against the C-ABI alone (no Python, no torch): what a C++
// host that runs one process per GPU does per batch of segments.  Every rank makes the same synthetic assembly (seeded)
// and the same plan, scans and block-calls ITS shard on its own GPU, packs one message; ts_exchange_gather moves the
// messages to rank 0 in one grouped RCCL send / recv; rank 0 merges them (ts_shards_finalize) and compares every segment
// with what it gets from scanning the whole batch alone (ts_scan_segments_blocks on the same context).
//
//   rank_scan --rank R --ranks N --id-file PATH [--device D] [--mbases M] [--segments S]
//
// Rank 0 writes the 128-byte RCCL id to PATH (atomically, via rename); the others wait for the file.  With N = 1 the
// message goes through RCCL to the same rank (the loop-back form), so that a one-GPU box runs the same calls.
#include <hip/hip_runtime_api.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include "teloscope_mi355x.hpp"

using namespace teloscope_mi355x;

#define CHECK(expr, what) do { if (!(expr)) { fprintf(stderr, "rank %d: %s failed: %s\n", rank, what, ctx ? ts_last_error(ctx) : ts_last_error(nullptr)); return 1; } } while (0)
#define HIPCHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "rank %d: %s: %s\n", rank, #expr, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv) {
    int rank = 0, ranks = 1, device = -1;
    uint64_t mbases = 40, nseg = 9;
    std::string idFile;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) exit(2); return argv[++i]; };
        if (a == "--rank") rank = std::stoi(val());
        else if (a == "--ranks") ranks = std::stoi(val());
        else if (a == "--id-file") idFile = val();
        else if (a == "--device") device = std::stoi(val());
        else if (a == "--mbases") mbases = std::stoull(val());
        else if (a == "--segments") nseg = std::stoull(val());
    }
    ts_ctx *ctx = nullptr;
    if (idFile.empty() || rank < 0 || rank >= ranks) { fprintf(stderr, "usage: rank_scan --rank R --ranks N --id-file PATH [--device D] [--mbases M]\n"); return 2; }
    const int ndev = ts_device_count();
    if (ndev <= 0) { fprintf(stderr, "no usable HIP device\n"); return 1; }
    if (device < 0) device = rank % ndev;

    // ---- the context: configs[1]'s flags (-c TTAGGG -p TTAGGG,TCAGGG,TGAGGG,TTGGGG -w 1000 -s 500 -r -g -e -m -i -t 3000)
    UserInputTeloscope ui;
    setCanonical(ui, "TTAGGG");
    ui.rawPatterns = {"TTAGGG", "TCAGGG", "TGAGGG", "TTGGGG"};
    ui.windowSize = 1000; ui.step = 500; ui.terminalLimit = 3000;
    ui.outWinRepeats = ui.outGC = ui.outEntropy = ui.outMatches = ui.outITS = true; ui.ultraFastMode = false;
    ui.device = device;
    std::vector<ts_pattern> pats = detail::makePatterns(ui);
    ts_params prm = detail::makeParams(ui);
    ctx = ts_create(&prm, pats.data(), pats.size());
    CHECK(ctx, "ts_create");
    HIPCHECK(hipSetDevice(device));

    // ---- the same assembly on every rank: random bases, telomeres at both ends of every segment, a few interstitial arrays
    std::vector<uint64_t> lens(nseg), abs(nseg);
    uint64_t total = mbases * 1000000ull, left = total;
    for (uint64_t i = 0; i < nseg; ++i) { lens[i] = i + 1 == nseg ? left : (left / (nseg - i)) * (i % 3 + 1) / 2; left -= lens[i]; abs[i] = 17 * i; }
    std::vector<std::string> seqs(nseg);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (uint64_t i = 0; i < nseg; ++i) {
        std::string &s = seqs[i];
        s.resize(lens[i]);
        for (uint64_t j = 0; j < lens[i]; ++j) s[j] = "ACGT"[rnd() & 3];
        const uint64_t tel = std::min<uint64_t>(lens[i] / 4, 1800 + 200 * i);
        for (uint64_t j = 0; j + 6 <= tel; j += 6) { std::memcpy(&s[j], "CCCTAA", 6); std::memcpy(&s[lens[i] - 6 - j], "TTAGGG", 6); }
        for (uint64_t q = 1; q <= 3 && lens[i] > 40000; ++q) {
            const uint64_t at = lens[i] * q / 4;
            for (uint64_t j = 0; j < 600; j += 6) std::memcpy(&s[at + j], "TTAGGG", 6);
        }
        if (lens[i] > 1000) s[lens[i] / 3] = 'N';
    }

    // ---- plan, shard, upload the bytes this rank's shard reads
    ts_batch *b = ts_batch_create(ctx, lens.data(), abs.data(), nseg, 0, 0);
    CHECK(b, "ts_batch_create");
    std::vector<ts_shard_info> info(ranks);
    for (int p = 0; p < ranks; ++p) CHECK(ts_batch_shard_info(b, ranks, p, 1, &info[p]) == TS_OK, "ts_batch_shard_info");
    CHECK(ts_batch_restrict_shard(b, ranks, rank, 1) == TS_OK, "ts_batch_restrict_shard");
    CHECK(ts_batch_set_emit(b, 1) == TS_OK, "ts_batch_set_emit");
    const ts_shard_info &me = info[rank];
    const uint64_t in_bytes = std::max<uint64_t>(me.input_end - me.input_begin, 64) + 4096;
    std::vector<char> host_in(in_bytes, 0);
    for (uint64_t i = 0; i < nseg; ++i) {
        const uint64_t off = ts_batch_segment_offset(b, i);
        const uint64_t a = std::max(off, me.input_begin), z = std::min(off + lens[i], me.input_end);
        if (z > a) std::memcpy(&host_in[a - me.input_begin], &seqs[i][a - off], z - a);
    }
    void *d_in = nullptr, *d_msg = nullptr;
    HIPCHECK(hipMalloc(&d_in, in_bytes));
    HIPCHECK(hipMemcpy(d_in, host_in.data(), in_bytes, hipMemcpyHostToDevice));
    HIPCHECK(hipMalloc(&d_msg, me.msg_bytes));
    hipStream_t st;
    HIPCHECK(hipStreamCreate(&st));

    // ---- the communicator
    char id[TS_EXCHANGE_ID_BYTES];
    if (rank == 0) {
        if (ts_exchange_unique_id(id) != TS_OK) { fprintf(stderr, "ts_exchange_unique_id: %s\n", ts_exchange_last_error()); return 1; }
        std::ofstream(idFile + ".tmp", std::ios::binary).write(id, sizeof id);
        std::rename((idFile + ".tmp").c_str(), idFile.c_str());
    } else {
        for (int tries = 0;; ++tries) {
            std::ifstream f(idFile, std::ios::binary);
            if (f && f.read(id, sizeof id)) break;
            if (tries > 600) { fprintf(stderr, "rank %d: no id file\n", rank); return 1; }
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
    }
    ts_exchange *ex = ts_exchange_create(ctx, id, rank, ranks);
    CHECK(ex, "ts_exchange_create");

    // ---- receive buffers on rank 0 (its own message stays where it was packed; one rank alone loops it back)
    std::vector<void *> d_recv(ranks, nullptr);
    std::vector<uint64_t> sizes(ranks);
    for (int p = 0; p < ranks; ++p) sizes[p] = info[p].msg_bytes;
    if (rank == 0) for (int p = 0; p < ranks; ++p) if (p != 0 || ranks == 1) HIPCHECK(hipMalloc(&d_recv[p], sizes[p]));

    // ---- a step: scan, block calling + message on the device, the exchange — nothing read back in between
    // (a message that reports an overflow — ts_shard_peek on rank 0 — would be answered with ts_batch_sync or a larger scale on
    // every rank and a second round; this rehearsal's input is sized so that none does, and ts_shards_finalize says so)
    CHECK(ts_batch_scan(b, d_in, st) == TS_OK, "ts_batch_scan");
    CHECK(ts_batch_pack_shard(b, d_msg, me.msg_bytes, st) == TS_OK, "ts_batch_pack_shard");
    CHECK(ts_exchange_gather(ex, 0, d_msg, me.msg_bytes, d_recv.data(), sizes.data(), st) == TS_OK, "ts_exchange_gather");
    HIPCHECK(hipStreamSynchronize(st));
    int rc = 0;
    if (rank == 0) {
        std::vector<std::vector<char>> msgs(ranks);
        std::vector<const void *> ptrs(ranks);
        for (int p = 0; p < ranks; ++p) {
            msgs[p].resize(sizes[p]);
            HIPCHECK(hipMemcpy(msgs[p].data(), (p == 0 && ranks > 1) ? d_msg : d_recv[p], sizes[p], hipMemcpyDeviceToHost));
            ptrs[p] = msgs[p].data();
        }
        ts_batch *plan = ts_batch_create(ctx, lens.data(), abs.data(), nseg, 0, 0);
        CHECK(plan, "ts_batch_create (plan)");
        std::vector<ts_segment_out> merged(nseg), alone(nseg);
        std::vector<ts_segment_counts> mc(nseg), ac(nseg);
        const int frc = ts_shards_finalize(plan, ptrs.data(), sizes.data(), (uint32_t)ranks, merged.data(), mc.data());
        if (frc != 0) { fprintf(stderr, "ts_shards_finalize returned %d: %s\n", frc, ts_last_error(ctx)); return 1; }
        std::vector<ts_segment_in> in(nseg);
        for (uint64_t i = 0; i < nseg; ++i) { in[i] = ts_segment_in{}; in[i].seq = seqs[i].data(); in[i].len = lens[i]; in[i].abs_pos = abs[i]; }
        CHECK(ts_scan_segments_blocks(ctx, in.data(), nseg, alone.data(), ac.data()) == TS_OK, "ts_scan_segments_blocks");
        uint64_t nblocks = 0, nwin = 0;
        for (uint64_t i = 0; i < nseg && rc == 0; ++i) {
            const ts_segment_out &m = merged[i], &a = alone[i];
            if (m.n_windows != a.n_windows || m.n_terminal_blocks != a.n_terminal_blocks || m.n_interstitial_blocks != a.n_interstitial_blocks ||
                std::memcmp(&mc[i], &ac[i], sizeof mc[i]) != 0) rc = 3;
            else if (m.n_windows && std::memcmp(m.windows, a.windows, m.n_windows * sizeof(ts_window)) != 0) rc = 4;
            else if (m.n_terminal_blocks && std::memcmp(m.terminal_blocks, a.terminal_blocks, m.n_terminal_blocks * sizeof(ts_block)) != 0) rc = 5;
            else if (m.n_interstitial_blocks && std::memcmp(m.interstitial_blocks, a.interstitial_blocks, m.n_interstitial_blocks * sizeof(ts_block)) != 0) rc = 6;
            if (rc) fprintf(stderr, "segment %llu differs (%d)\n", (unsigned long long)i, rc);
            nblocks += m.n_terminal_blocks + m.n_interstitial_blocks; nwin += m.n_windows;
        }
        ts_free_segments(merged.data(), nseg);
        ts_free_segments(alone.data(), nseg);
        ts_batch_destroy(plan);
        if (rc == 0) printf("rank_scan ok: %d rank(s), %llu segments, %llu windows, %llu blocks, %llu message bytes over RCCL\n", ranks,
                            (unsigned long long)nseg, (unsigned long long)nwin, (unsigned long long)nblocks,
                            (unsigned long long)[&] { uint64_t t = 0; for (int p = ranks == 1 ? 0 : 1; p < ranks; ++p) t += sizes[p]; return t; }());
    }
    ts_exchange_destroy(ex);
    ts_batch_destroy(b);
    for (void *p : d_recv) if (p) (void)hipFree(p);
    (void)hipFree(d_in); (void)hipFree(d_msg);
    (void)hipStreamDestroy(st);
    ts_destroy(ctx);
    return rc;
}
