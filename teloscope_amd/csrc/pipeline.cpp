// pipeline.cpp — the host-buffer entry points of libteloscan: ts_scan_segments, ts_scan_segments_blocks,
// ts_filter_reads(_multi).  Host memory in, host results out.
//
// A call's segments (or reads) are cut into GROUPS of about 256 MB of input and the groups flow through three
// stages that run concurrently, each on its own host thread and HIP stream:
//
//   upload     plan the group's batch, stage its bases into a ring of pinned chunks (several memcpy threads)
//              and DMA them to the device (up_stream)                                 — bound by PCIe
//   scan       ts_scan_tiles on the group (scan_stream), read back the per-wave fill, regrow + rescan on
//              overflow; the group's input buffer goes back to the pool                — ~0.1 ms per group
//   download   mode-specific device work (block calling / read predicate / record compaction), D2H into a
//              pinned landing area, host post-processing on the host threads (down_stream)
//
// so that the wall time of a call is the upload plus the tail of the last group, instead of the sum of the
// three.  Device buffers come from the context's pool: a call neither allocates nor frees device memory once
// the pool is warm.  The reference's own decomposition is one thread-pool job per path (src/input.cpp:719-724)
// and one job per chunk of a 2048-record FASTQ batch (src/input.cpp:753-812); a group is the GPU-sized
// equivalent of such a job, and results come back in input order whatever the grouping.
#include <hip/hip_runtime.h>
#include <immintrin.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "capi_internal.hpp"

namespace {

using Clock = std::chrono::steady_clock;
double ms_between(Clock::time_point a, Clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); }

uint64_t group_target_bytes() {
    static const uint64_t v = [] {
        if (const char *e = getenv("TS_GROUP_MB")) { const long mb = atol(e); if (mb > 0) return (uint64_t)mb << 20; }
        // (with the packed upload a group's bases are staged in 128 MB chunks: larger groups, fewer hand-overs between the
        // stages — profiles/r03/packed_upload_rate.txt)
        const char *pk = getenv("TS_PACKED_UPLOAD");
        return (uint64_t)((pk && pk[0] == '0') ? 256 : 512) << 20;
    }();
    return v;
}

template <typename T>
class Channel {                                   // unbounded FIFO between two pipeline stages
public:
    void push(T v) { { std::lock_guard<std::mutex> g(m_); q_.push_back(std::move(v)); } cv_.notify_one(); }
    void close() { { std::lock_guard<std::mutex> g(m_); closed_ = true; } cv_.notify_all(); }
    bool pop(T &out) {
        std::unique_lock<std::mutex> g(m_);
        cv_.wait(g, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        return true;
    }
private:
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<T> q_;
    bool closed_ = false;
};

class Semaphore {
public:
    explicit Semaphore(int n) : n_(n) {}
    void acquire() { std::unique_lock<std::mutex> g(m_); cv_.wait(g, [&] { return n_ > 0; }); --n_; }
    void release() { { std::lock_guard<std::mutex> g(m_); ++n_; } cv_.notify_one(); }
private:
    std::mutex m_;
    std::condition_variable cv_;
    int n_;
};

enum class Mode { Matches, Blocks, ReadPass };

struct Item { const char *seq; uint64_t len, abs_pos; uint8_t format; uint32_t n_pieces; };     // format: TS_INPUT_BASES / TS_INPUT_TEXT_PIECES (seq = ts_text_piece[])

struct Group {
    size_t first = 0, count = 0;                  // items [first, first + count) of the call's item list
    ts_batch *b = nullptr;
    hipEvent_t uploaded = nullptr;
    double t_plan = 0, t_upload = 0, t_scan = 0, t_down = 0, t_fetch = 0, t_final = 0;
};

int ensure_streams(ts_ctx *c) {
    if (c->up_stream) return TS_OK;
    HIP_TRY(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->scan_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking));
    // the pinned staging ring is allocated — pinning touches the pages, which places them — by a thread on the device's NUMA node
    int rc = TS_OK;
    std::thread alloc([&] {
        c->bind_this_thread();
        DeviceGuard g(c->device);
        for (int i = 0; i < ts_ctx::kUpSlots && rc == TS_OK; ++i) {
            // (+ 4096: a packed chunk of exactly 128 Mi positions that starts off a 64-byte boundary packs to a few bytes more
            // than 32 MiB, and its DMA reads 8 bytes past the last code)
            if (c->pin_up[i].ensure((32u << 20) + 4096u) != hipSuccess) { rc = c->fail(TS_ERR_ALLOC, "cannot allocate the pinned staging ring"); break; }
            if (hipEventCreateWithFlags(&c->pin_up_ev[i], hipEventDisableTiming) != hipSuccess) rc = c->fail(TS_ERR_HIP, "hipEventCreate failed");
        }
    });
    alloc.join();
    return rc;
}

struct UpPiece { uint64_t off; const char *src; uint64_t len; uint64_t text_len;     // off: byte offset in the input layout; len: bases;
                                                                                     // text_len != 0: src is FASTA text (line ends to skip)
                 const ts_packed_seq *packed = nullptr; uint64_t packed_first = 0; };  // packed: the bases are codes [packed_first, + len) of *packed

// the bases of a run of FASTA body text, without its line ends ('\n', and a '\r' right before one or at the very end)
bool strip_copy(char *dst, const char *text, uint64_t text_len, uint64_t n_bases) {
    const char *p = text, *end = text + text_len;
    uint64_t left = n_bases;
    while (left && p < end) {
        const char *nl = (const char *)std::memchr(p, '\n', (size_t)(end - p));
        const char *stop = nl ? nl : end;
        uint64_t line = (uint64_t)(stop - p);
        if (line && stop[-1] == '\r') --line;
        const uint64_t take = std::min(line, left);
        std::memcpy(dst, p, take);
        dst += take; left -= take;
        p = nl ? nl + 1 : end;
    }
    return left == 0;
}

// The two walks over FASTA body text below — find the text position of a base, copy bases without their line ends — took a
// memchr and a memcpy per 81-byte LINE (37 M lines per 3 Gb: FASTA text in ran at half the rate of joined bases).  With AVX2 they
// go 32 bytes at a time.  A byte is a base unless it is a line feed, or a carriage return right before one (the byte behind the
// chunk is looked at for the chunk's last byte, so a chunk's count is exact by itself).  The scalar loops finish the last bytes of
// a piece and are the whole path without AVX2.
__attribute__((target("avx2")))
inline uint32_t line_end_mask32(const char *p) {               // bit i: byte i of the chunk is (part of) a line end; p[32] is readable
    const __m256i v = _mm256_loadu_si256((const __m256i *)p);
    const uint32_t lf = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('\n')));
    const uint32_t cr = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(v, _mm256_set1_epi8('\r')));
    const uint32_t lf_next = (lf >> 1) | (p[32] == '\n' ? 0x80000000u : 0u);     // bit i: byte i + 1 is a line feed
    return lf | (cr & lf_next);
}

// skips whole 32-byte chunks of [p, end) while base `*skip` lies behind them; returns the chunk it lies in (or the last bytes)
__attribute__((target("avx2")))
const char *text_locate_avx2(const char *p, const char *end, uint64_t *skip) {
    while (end - p >= 33) {
        const uint32_t bases = 32u - (uint32_t)__builtin_popcount(line_end_mask32(p));
        if (*skip < bases) break;
        *skip -= bases;
        p += 32;
    }
    return p;
}

// copies bases of [*pp, end) to dst, 32 text bytes at a time, while at least 32 more are wanted; the cursor stays at a place the
// scalar walk can go on from (never between a carriage return and its line feed).  dst must have 32 bytes of slack.
__attribute__((target("avx2")))
uint64_t strip_take_avx2(char *dst, uint64_t n, const char **pp, const char *end) {
    const char *p = *pp;
    uint64_t left = n;
    while (left >= 32 && end - p >= 33) {
        const uint32_t m = line_end_mask32(p);
        _mm256_storeu_si256((__m256i *)dst, _mm256_loadu_si256((const __m256i *)p));
        if (m == 0u) { dst += 32; p += 32; left -= 32; continue; }
        const uint32_t pos = (uint32_t)__builtin_ctz(m);             // the bases before the chunk's first line end are in place
        dst += pos; left -= pos;
        p += pos;
        if (*p == '\r') ++p;                                        // (followed by a line feed: that is what the mask says)
        ++p;                                                        // the line feed
    }
    *pp = p;
    return n - left;
}

// text position of base `skip` of a text piece (skip < its n_bases)
const char *text_locate(const char *text, uint64_t text_len, uint64_t skip) {
    const char *p = text, *end = text + text_len;
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    if (have_avx2) p = text_locate_avx2(p, end, &skip);          // (chunks begin anywhere in a line: the walk below counts from any byte)
    while (p < end) {
        const char *nl = (const char *)std::memchr(p, '\n', (size_t)(end - p));
        const char *stop = nl ? nl : end;
        uint64_t line = (uint64_t)(stop - p);
        if (line && stop[-1] == '\r') --line;
        if (skip < line) return p + skip;
        skip -= line;
        p = nl ? nl + 1 : end;
    }
    return end;
}

// Up to n bases of FASTA body text from the cursor *pp on (line ends skipped), cursor advanced; returns the bases taken
// (fewer than n only when the text ends).
uint64_t strip_take(char *dst, uint64_t n, const char **pp, const char *end) {
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    uint64_t left = n;
    if (have_avx2) { const uint64_t got = strip_take_avx2(dst, n, pp, end); dst += got; left -= got; }
    const char *p = *pp;
    while (left && p < end) {
        const char *nl = (const char *)std::memchr(p, '\n', (size_t)(end - p));
        const char *stop = nl ? nl : end;
        uint64_t line = (uint64_t)(stop - p);
        const bool cr = line && stop[-1] == '\r';
        if (cr) --line;
        const uint64_t take = std::min(line, left);
        std::memcpy(dst, p, take);
        dst += take; left -= take;
        if (take < line) { p += take; break; }             // stopped inside the line
        p = nl ? nl + 1 : end;
    }
    *pp = p;
    return n - left;
}

constexpr uint32_t kPackRunCap = 1u << 18;                 // invalid runs a packed chunk may carry (more: the chunk goes as ASCII)


// Uploads pieces of an input layout to the device buffer that holds its bytes from lo_all on (din = address of byte
// lo_all).  Consecutive pieces that lie close together in the layout (full scans, reads: a few padding bytes apart) are
// mirrored together in one of three pinned 32 MB buffers, filled by several host threads — plain bases by memcpy, FASTA
// text by a copy that skips the line ends — and leave by ONE DMA while the next buffer is being filled (one memcpy
// stream fills pinned memory at ~10 GB/s, a fraction of what the link moves; one copy per read would cost ~10 us each,
// one pageable 3 GB copy ~0.5 s).  Pieces far apart (the two terminal regions of a long contig in tips-only mode) go
// separately.  Bytes between pieces are never read as bases (the kernels mask everything past a region's end).
// Asynchronous: the DMAs are queued on up_stream.  `pieces` ascend by offset and do not overlap.
int upload_pieces(ts_ctx *c, const std::vector<UpPiece> &pieces_in, void *din, uint64_t lo_all, int &slot, bool used[]) {
    constexpr uint64_t kChunk = 32u << 20, kMaxGap = 64u << 10;
    constexpr uint64_t kChunkPacked = 4 * kChunk;            // a packed chunk fills the same 32 MB pinned slot with 128 MB of layout
    std::vector<UpPiece> pieces;
    pieces.reserve(pieces_in.size());
    bool any_text = false, any_packed = false;
    for (const UpPiece &pc : pieces_in) {
        if (pc.text_len) {
            if (pc.len > kChunk) return c->fail(TS_ERR_INVALID_ARG, "a text piece holds more than 32 MiB of bases");
            pieces.push_back(pc);
            any_text = true;
        } else if (pc.packed) {
            for (uint64_t a = 0; a < pc.len; a += kChunk) {
                UpPiece q = pc;
                q.off = pc.off + a; q.len = std::min<uint64_t>(kChunk, pc.len - a); q.packed_first = pc.packed_first + a;
                pieces.push_back(q);
            }
            any_packed = true;
        } else {
            for (uint64_t a = 0; a < pc.len; a += kChunk) pieces.push_back({pc.off + a, pc.src + a, std::min<uint64_t>(kChunk, pc.len - a), 0});
        }
    }
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    unsigned nthr = std::min(8u, std::max(1u, hw / 2u));
    if (c->knobs.stage_threads) nthr = c->knobs.stage_threads;
    // The staging threads live for the whole call, not for one 32 MB chunk (a chunk is staged in ~0.6 ms: spawning and
    // joining eight threads for each cost a tenth of the upload): job(t) runs on worker t, the caller is worker 0.
    struct StagePool {
        std::mutex m;
        std::condition_variable cv_go, cv_done;
        std::function<void(unsigned)> job;
        uint64_t generation = 0;
        unsigned active = 0, pending = 0;
        bool quit = false;
        std::vector<std::thread> threads;
        explicit StagePool(unsigned n) {
            for (unsigned t = 1; t < n; ++t)
                threads.emplace_back([this, t] {
                    uint64_t seen = 0;
                    for (;;) {
                        std::function<void(unsigned)> f;
                        {
                            std::unique_lock<std::mutex> g(m);
                            cv_go.wait(g, [&] { return quit || generation != seen; });
                            if (quit) return;
                            seen = generation;
                            if (t >= active) continue;
                            f = job;
                        }
                        f(t);
                        { std::lock_guard<std::mutex> g(m); if (--pending == 0) cv_done.notify_one(); }
                    }
                });
        }
        void run(unsigned n, const std::function<void(unsigned)> &f) {       // f(0 .. n-1), n <= workers
            if (n <= 1 || threads.empty()) { for (unsigned t = 0; t < n; ++t) f(t); return; }
            { std::lock_guard<std::mutex> g(m); job = f; active = n; pending = n - 1; ++generation; }
            cv_go.notify_all();
            f(0);
            std::unique_lock<std::mutex> g(m);
            cv_done.wait(g, [&] { return pending == 0; });
        }
        ~StagePool() {
            { std::lock_guard<std::mutex> g(m); quit = true; }
            cv_go.notify_all();
            for (std::thread &th : threads) th.join();
        }
    };
    uint64_t total_bytes = 0;
    for (const UpPiece &pc : pieces) total_bytes += pc.len;
    StagePool stage_pool(total_bytes >= (8u << 20) ? nthr : 1u);
    std::atomic<int> bad_text{0};
    const bool fold = c->params.fold_case != 0;
    const uint64_t packed_min = c->knobs.packed_min_bytes;
    bool use_packed = (c->knobs.packed_upload && total_bytes >= packed_min) || any_packed;      // (bases that arrive packed leave packed)
    if (use_packed && !c->d_pack[0].p) {                       // the device side of the ring, once per context
        for (int q = 0; q < ts_ctx::kUpSlots && use_packed; ++q) {
            if (c->d_pack[q].ensure((kChunkPacked >> 2) + 4096) != hipSuccess || c->d_runs[q].ensure((size_t)kPackRunCap * 8) != hipSuccess ||
                c->pin_runs[q].ensure((size_t)kPackRunCap * 8) != hipSuccess) { (void)hipGetLastError(); use_packed = false; }
        }
    }
    if (any_packed && !use_packed) return c->fail(TS_ERR_ALLOC, "packed input needs the packed upload's device buffers");
    static const bool stage_timing = getenv("TS_TIMING") != nullptr && getenv("TS_STAGE_TIMING") != nullptr;
    double t_wait = 0, t_pack = 0, t_issue = 0;
    size_t n_chunks = 0;
    struct Report { const bool on; double &w, &p, &i; size_t &n; ~Report() { if (on && n) fprintf(stderr, "  upload_pieces: %zu chunks, waiting for a free slot %.2f ms, packing %.2f ms, runs + DMA + unpack enqueue %.2f ms\n", n, w, p, i); } }
        report{stage_timing, t_wait, t_pack, t_issue, n_chunks};
    size_t i = 0;
    while (i < pieces.size()) {
        const uint64_t c0 = pieces[i].off;
        size_t j = i + 1;
        uint64_t bytes = pieces[i].len;
        const uint64_t chunk_limit = use_packed ? kChunkPacked : kChunk;
        while (j < pieces.size() && pieces[j].off + pieces[j].len - c0 <= chunk_limit &&
               pieces[j].off - (pieces[j - 1].off + pieces[j - 1].len) <= kMaxGap) { bytes += pieces[j].len; ++j; }
        const auto tw0 = Clock::now();
        if (used[slot]) HIP_TRY(c, hipEventSynchronize(c->pin_up_ev[slot]));
        if (stage_timing) t_wait += ms_between(tw0, Clock::now());
        ++n_chunks;
        char *dst = (char *)c->pin_up[slot].p;
        const unsigned nt = bytes >= (4u << 20) ? nthr : 1u;
        const uint64_t hi_pos = pieces[j - 1].off + pieces[j - 1].len;
        // ---- packed: the chunk leaves as 2-bit codes + invalid runs (pack.cpp), a quarter of the bytes; unpack.hip
        // restores the byte layout on the device.  Worker t packs the layout range [t, t + 1) x share of the chunk
        // (ranges end at multiples of 4096 positions: no two workers write the same byte), pulling its bases out of the
        // pieces that overlap it — plain pieces in place, FASTA text through a small buffer that the line ends are
        // stripped into — and 'A' into the padding between pieces, which nobody reads.
        if (use_packed && (bytes >= 4096 || any_packed)) {
            const uint64_t c0a = c0 & ~63ull;
            const uint64_t P = hi_pos - c0a;                                  // chunk positions, incl. the lead before c0
            const uint64_t share = ((P + nt - 1) / nt + 4095) & ~4095ull;
            std::vector<ts::PackRuns> wr(nt);
            std::atomic<int> short_text{0};
            // the workers' ranges: equal shares — but where FASTA text pieces lie in the chunk, a range begins where a piece begins
            // (rounded up to a whole byte of codes) when one does within half a share of the even cut: a worker that enters a text
            // piece in its middle has to find the text position of its first base, i.e. read the piece's text up to there, and the
            // same text is read again when it is packed (text in ran memory-bound at 1.5 x the traffic of joined bases)
            std::vector<uint64_t> cut(nt + 1);
            for (unsigned t = 0; t <= nt; ++t) cut[t] = std::min<uint64_t>(P, (uint64_t)t * share);
            {
                size_t q = i;
                for (unsigned t = 1; t < nt; ++t) {
                    const uint64_t ideal = (uint64_t)t * share;
                    if (ideal >= P) break;
                    while (q < j && pieces[q].off < c0a + ideal) ++q;                  // first piece that begins at or behind the even cut
                    uint64_t best = ideal, dist = share / 2;
                    for (size_t c = (q > i ? q - 1 : q); c < j && c <= q; ++c) {       // the piece starts either side of it
                        if (!pieces[c].text_len || pieces[c].off < c0a) continue;
                        const uint64_t at = (pieces[c].off - c0a + 3) & ~3ull;
                        const uint64_t d = at > ideal ? at - ideal : ideal - at;
                        if (d < dist && at > cut[t - 1] && at < P) { best = at; dist = d; }
                    }
                    cut[t] = best;
                }
                for (unsigned t = 1; t <= nt; ++t) cut[t] = std::max(cut[t], cut[t - 1]);
            }
            auto pack_range = [&](unsigned t) {
                const uint64_t a0 = cut[t], z0 = cut[t + 1];
                if (a0 >= z0) return;
                constexpr uint64_t BK = 16384;
                alignas(64) unsigned char buf[BK];
                ts::PackRuns &R = wr[t];
                size_t k = i;
                while (k < j && pieces[k].off + pieces[k].len <= c0a + a0) ++k;      // first piece that reaches into the range
                const char *tcur = nullptr, *tend = nullptr;                         // text cursor inside pieces[tk]
                size_t tk = (size_t)-1;
                for (uint64_t a = a0; a < z0; a += BK) {
                    const uint64_t z = std::min(z0, a + BK), la = c0a + a, lz = c0a + z;   // layout range of the block
                    while (k < j && pieces[k].off + pieces[k].len <= la) ++k;
                    // wholly inside one plain piece: packed from where it lies
                    if (k < j && !pieces[k].text_len && !pieces[k].packed && pieces[k].off <= la && pieces[k].off + pieces[k].len >= lz) {
                        ts::pack_bases((const unsigned char *)pieces[k].src + (la - pieces[k].off), z - a, (unsigned char *)dst + (a >> 2), fold, (uint32_t)a, R);
                        continue;
                    }
                    // wholly inside one FASTA text piece: packed straight from the text (ts::pack_text: 32 text bytes at a time, the line
                    // ends' slots taken out of the codes) — the stripped copy in between made text in half as fast as joined bases
                    if (k < j && pieces[k].text_len && !pieces[k].packed && pieces[k].off <= la && pieces[k].off + pieces[k].len >= lz) {
                        const UpPiece &pc = pieces[k];
                        if (tk != k) {                                               // enter this text piece (at base la - pc.off)
                            tk = k;
                            tend = pc.src + pc.text_len;
                            tcur = la > pc.off ? text_locate(pc.src, pc.text_len, la - pc.off) : pc.src;
                        }
                        if (ts::pack_text(&tcur, tend, z - a, (unsigned char *)dst + (a >> 2), fold, (uint32_t)a, R) != z - a) short_text.store(1);
                        continue;
                    }
                    // bases that arrive packed (TS_INPUT_PACKED2): when every piece the block touches is such, their codes are
                    // copied — whole bytes where source and destination agree on the phase within a byte (full scans: segments
                    // start on 16-byte boundaries of the layout), else shifted — and their invalid runs translated; a block
                    // that mixes packed and ASCII pieces (the seam between two segments of different formats) takes the
                    // ASCII path below with the codes spelled out as letters
                    {
                        bool all_packed = true, any_here = false;
                        for (size_t q = k; q < j && pieces[q].off < lz; ++q) {
                            if (pieces[q].off + pieces[q].len <= la) continue;
                            any_here = true;
                            if (!pieces[q].packed) all_packed = false;
                        }
                        if (any_here && all_packed) {
                            unsigned char *out = (unsigned char *)dst + (a >> 2);              // a is a multiple of 4 (blocks of 16384 from a multiple of 4096)
                            std::memset(out, 0, (z - a + 3) >> 2);
                            for (size_t q = k; q < j && pieces[q].off < lz; ++q) {
                                const UpPiece &pc = pieces[q];
                                const uint64_t s0 = std::max(pc.off, la), s1 = std::min(pc.off + pc.len, lz);
                                if (s1 <= s0) continue;
                                const uint64_t d0 = s0 - la, n = s1 - s0, i0 = pc.packed_first + (s0 - pc.off);   // block position, bases, source base index
                                const uint8_t *codes = pc.packed->codes;
                                if (((d0 ^ i0) & 3u) == 0u) {
                                    uint64_t x = 0;
                                    for (; x < n && ((d0 + x) & 3u); ++x)                                        // up to the first whole byte
                                        out[(d0 + x) >> 2] |= (unsigned char)(((codes[(i0 + x) >> 2] >> (2u * ((i0 + x) & 3u))) & 3u) << (2u * ((d0 + x) & 3u)));
                                    const uint64_t whole = (n - x) >> 2;
                                    std::memcpy(out + ((d0 + x) >> 2), codes + ((i0 + x) >> 2), whole);
                                    for (x += whole * 4; x < n; ++x)
                                        out[(d0 + x) >> 2] |= (unsigned char)(((codes[(i0 + x) >> 2] >> (2u * ((i0 + x) & 3u))) & 3u) << (2u * ((d0 + x) & 3u)));
                                } else {
                                    for (uint64_t x = 0; x < n; ++x)
                                        out[(d0 + x) >> 2] |= (unsigned char)(((codes[(i0 + x) >> 2] >> (2u * ((i0 + x) & 3u))) & 3u) << (2u * ((d0 + x) & 3u)));
                                }
                                // the piece's invalid runs that reach into [i0, i0 + n): chunk positions; the codes under them are zeroed
                                // (what pack_bases leaves there: the unpack kernel writes 'N' over them anyway)
                                const ts_packed_run *pr = pc.packed->runs;
                                const uint64_t nr = pr ? pc.packed->n_runs : 0;
                                size_t r0 = (size_t)(std::upper_bound(pr, pr + nr, i0, [](uint64_t v, const ts_packed_run &r) { return v < r.start + r.len; }) - pr);
                                for (; r0 < nr && pr[r0].start < i0 + n; ++r0) {
                                    const uint64_t ra = std::max<uint64_t>(pr[r0].start, i0), rz = std::min<uint64_t>(pr[r0].start + pr[r0].len, i0 + n);
                                    if (rz <= ra) continue;
                                    R.finish();
                                    const uint32_t cs = (uint32_t)(a + d0 + (ra - i0));
                                    if (!R.runs.empty() && R.runs.back().start + R.runs.back().len == cs) R.runs.back().len += (uint32_t)(rz - ra);
                                    else R.runs.push_back({cs, (uint32_t)(rz - ra)});
                                }
                            }
                            continue;
                        }
                    }
                    std::memset(buf, 'A', z - a);
                    for (size_t q = k; q < j && pieces[q].off < lz; ++q) {
                        const UpPiece &pc = pieces[q];
                        const uint64_t s0 = std::max(pc.off, la), s1 = std::min(pc.off + pc.len, lz);
                        if (s1 <= s0) continue;
                        if (pc.packed) {                                                             // (a mixed block: letters, 'N' under the runs)
                            const uint64_t i0 = pc.packed_first + (s0 - pc.off);
                            for (uint64_t x = 0; x < s1 - s0; ++x)
                                buf[s0 - la + x] = "ACTG"[(pc.packed->codes[(i0 + x) >> 2] >> (2u * ((i0 + x) & 3u))) & 3u];
                            const ts_packed_run *pr = pc.packed->runs;
                            for (uint64_t r = 0; pr && r < pc.packed->n_runs; ++r) {
                                const uint64_t ra = std::max<uint64_t>(pr[r].start, i0), rz = std::min<uint64_t>(pr[r].start + pr[r].len, i0 + (s1 - s0));
                                for (uint64_t y = ra; y < rz; ++y) buf[s0 - la + (y - i0)] = 'N';
                            }
                            continue;
                        }
                        if (!pc.text_len) { std::memcpy(buf + (s0 - la), pc.src + (s0 - pc.off), s1 - s0); continue; }
                        if (tk != q) {                                               // enter this text piece (at base s0 - pc.off)
                            tk = q;
                            tend = pc.src + pc.text_len;
                            tcur = s0 > pc.off ? text_locate(pc.src, pc.text_len, s0 - pc.off) : pc.src;
                        }
                        if (strip_take((char *)buf + (s0 - la), s1 - s0, &tcur, tend) != s1 - s0) short_text.store(1);
                    }
                    ts::pack_bases(buf, z - a, (unsigned char *)dst + (a >> 2), fold, (uint32_t)a, R);
                }
                R.finish();
            };
            const auto tp0 = Clock::now();
            if (nt == 1u) pack_range(0); else stage_pool.run(nt, pack_range);
            const auto tp1 = Clock::now();
            if (stage_timing) t_pack += ms_between(tp0, tp1);
            if (short_text.load()) bad_text.store(1);
            size_t nruns = 0;
            for (const ts::PackRuns &R : wr) nruns += R.runs.size();
            if (nruns <= kPackRunCap) {
                ts::InvalidRun *hr = (ts::InvalidRun *)c->pin_runs[slot].p;
                size_t at = 0;
                for (const ts::PackRuns &R : wr) { if (!R.runs.empty()) std::memcpy(hr + at, R.runs.data(), R.runs.size() * sizeof(ts::InvalidRun)); at += R.runs.size(); }
                const uint64_t pbytes = (((P + 3) >> 2) + 3) & ~3ull;
                if (pbytes + 8 > c->pin_up[slot].bytes || pbytes + 8 > c->d_pack[slot].bytes)
                    return c->fail(TS_ERR_STATE, "packed upload: a chunk does not fit its staging slot");
                std::memset(dst + ((P + 3) >> 2), 0, pbytes - ((P + 3) >> 2) + 8);     // (the kernel reads one dword past the last code)
                HIP_TRY(c, hipMemcpyAsync(c->d_pack[slot].p, dst, pbytes + 8, hipMemcpyHostToDevice, c->up_stream));
                if (nruns) HIP_TRY(c, hipMemcpyAsync(c->d_runs[slot].p, hr, nruns * sizeof(ts::InvalidRun), hipMemcpyHostToDevice, c->up_stream));
                if (ts_k_launch_unpack(c->d_pack[slot].p, (uint32_t)(c0 - c0a), (char *)din + (c0 - lo_all), hi_pos - c0, c->d_runs[slot].p, (uint32_t)nruns,
                                       (char *)din + (c0 - lo_all) - (c0 - c0a), c->up_stream) != 0)
                    return c->fail(TS_ERR_HIP, "unpack kernel launch failed");
                HIP_TRY(c, hipEventRecord(c->pin_up_ev[slot], c->up_stream));
                if (stage_timing) t_issue += ms_between(tp1, Clock::now());
                used[slot] = true;
                slot = (slot + 1) % ts_ctx::kUpSlots;
                i = j;
                continue;
            }
            // a chunk with more invalid runs than the list holds is not sequence data: the rest of the call goes the plain way
            if (any_packed) return c->fail(TS_ERR_INVALID_ARG, "packed input with more invalid runs per 128 Mi bases than the upload carries (2^18)");
            use_packed = false;
            continue;
        }
        auto copy_part = [&](size_t k) {
            const UpPiece &pc = pieces[k];
            if (pc.text_len) { if (!strip_copy(dst + (pc.off - c0), pc.src, pc.text_len, pc.len)) bad_text.store(1); }
            else std::memcpy(dst + (pc.off - c0), pc.src, pc.len);
        };
        if (nt == 1u) {
            for (size_t k = i; k < j; ++k) copy_part(k);
        } else if (any_text) {                                   // whole parts, handed out dynamically
            std::atomic<size_t> next{i};
            stage_pool.run(nt, [&](unsigned) { for (size_t k; (k = next.fetch_add(1)) < j;) copy_part(k); });
        } else {                                                 // worker t copies the bytes [t, t+1) * share of the concatenated parts
            const size_t share = (bytes + nt - 1) / nt;
            stage_pool.run(nt, [&](unsigned t) {
                const size_t lo = (size_t)t * share, hi = std::min<size_t>(bytes, lo + share);
                size_t at = 0;
                for (size_t k = i; k < j; ++k) {
                    const UpPiece &pc = pieces[k];
                    const size_t a = std::max(lo, at), z = std::min<size_t>(hi, at + pc.len);
                    if (z > a) std::memcpy(dst + (pc.off - c0) + (a - at), pc.src + (a - at), z - a);
                    at += pc.len;
                    if (at >= hi) break;
                }
            });
        }
        const uint64_t hi = pieces[j - 1].off + pieces[j - 1].len;
        HIP_TRY(c, hipMemcpyAsync((char *)din + (c0 - lo_all), dst, hi - c0, hipMemcpyHostToDevice, c->up_stream));
        HIP_TRY(c, hipEventRecord(c->pin_up_ev[slot], c->up_stream));
        used[slot] = true;
        slot = (slot + 1) % ts_ctx::kUpSlots;
        i = j;
    }
    if (bad_text.load()) return c->fail(TS_ERR_INVALID_ARG, "a text piece holds fewer bases than it declares");
    return TS_OK;
}

// The upload pieces of one scanned region of an item: its bases [rg_start, rg_start + rg_len), whose first lies at byte
// layout_off of the input layout, clipped to the layout range [lo, hi) the caller uploads — from whichever of the three
// input formats the item arrived in.
int region_pieces(ts_ctx *c, const Item &it, uint64_t seg_len, uint64_t rg_start, uint64_t rg_len, uint64_t layout_off,
                  uint64_t lo, uint64_t hi, std::vector<UpPiece> &pieces) {
    if (it.format == TS_INPUT_TEXT_PIECES) {
        // the region's bases out of the segment's text pieces
        const ts_text_piece *tp = (const ts_text_piece *)it.seq;
        uint64_t cum = 0, want = rg_start, left = rg_len, off = layout_off;
        for (size_t k = 0; left; ++k) {
            if (k >= it.n_pieces || cum >= seg_len)
                return c->fail(TS_ERR_INVALID_ARG, "text pieces hold fewer bases than the segment's length (or n_pieces is not set)");
            const ts_text_piece &t = tp[k];
            if (t.text_len > (16ull << 20) + 4096) return c->fail(TS_ERR_INVALID_ARG, "a text piece is larger than 16 MiB");
            if (want >= cum + t.n_bases) { cum += t.n_bases; continue; }      // wholly before the region
            const uint64_t skip = want - cum;
            const uint64_t n = std::min<uint64_t>(t.n_bases - skip, left);
            // of these n bases, those the range reads (a shard uploads only its part of a segment)
            const uint64_t a = std::max<uint64_t>(off, lo), z = std::min<uint64_t>(off + n, hi);
            if (z > a) {
                const uint64_t skip2 = skip + (a - off);
                const char *from = skip2 ? text_locate(t.text, t.text_len, skip2) : t.text;
                pieces.push_back({a, from, z - a, (uint64_t)(t.text + t.text_len - from)});
            }
            off += n; left -= n; want += n; cum += t.n_bases;
        }
        return TS_OK;
    }
    const uint64_t s0 = std::max<uint64_t>(layout_off, lo), s1 = std::min<uint64_t>(layout_off + rg_len, hi);
    if (it.format == TS_INPUT_PACKED2) {
        const ts_packed_seq *ps = (const ts_packed_seq *)it.seq;
        if (!ps || (!ps->codes && seg_len) || (ps->n_runs && !ps->runs)) return c->fail(TS_ERR_INVALID_ARG, "packed input: null codes or runs");
        if (s1 > s0) {
            UpPiece pc{s0, nullptr, s1 - s0, 0};
            pc.packed = ps; pc.packed_first = rg_start + (s0 - layout_off);
            pieces.push_back(pc);
        }
        return TS_OK;
    }
    if (s1 > s0) pieces.push_back({s0, it.seq + rg_start + (s0 - layout_off), s1 - s0, 0});
    return TS_OK;
}

// Uploads the bases a batch reads: whole segments of a full scan; only the two terminal regions of a long segment
// in tips-only mode — the rest of the layout is never read.
int upload_batch(ts_batch *b, const Item *items, int &slot, bool used[]) {
    ts_ctx *c = b->ctx;
    void *din = ts_batch_input_ptr_nozero(b);
    if (!din) return c->fail(TS_ERR_ALLOC, "cannot allocate device input buffer");
    std::vector<UpPiece> pieces;
    for (size_t i = 0; i < b->segs.size(); ++i) {
        const SegPlan &sp = b->segs[i];
        for (const Region &rg : sp.regions) {
            // (consecutive regions of one segment never overlap: tips regions are [0,t) and [N-t,N) with N > 2t)
            const int rc = region_pieces(c, items[i], sp.len, rg.start, rg.len, sp.in_off + rg.start, b->in_lo, b->in_hi, pieces);
            if (rc != TS_OK) return rc;
        }
    }
    return upload_pieces(c, pieces, din, b->in_lo, slot, used);
}

// The terminal-block predicate of a scanned tips batch on the device: one byte per read
// (ReadTelomereFilter::matches, src/read-filter.cpp:37-45, reduced to !terminalBlocks.empty()), written to
// d_pass (device).  The per-read table the kernel walks is built once per batch.
int batch_read_pass_device(ts_batch *b, unsigned char *d_pass, hipStream_t st) {
    ts_ctx *c = b->ctx;
    const size_t ns = b->segs.size();
    if (!ns) return TS_OK;
    const size_t off_in = (((ns + 1) * 4 + 15) & ~(size_t)15), off_len = off_in + ns * 8, off_long = off_len + ns * 8,
                 off_count = (off_long + ns * 4 + 15) & ~(size_t)15, off_flag = off_count + 16, bytes = off_flag + 16;   // + the list of long reads, its counter, the overflow flag
    if (!b->d_readtab.p) {
        std::vector<char> tab(bytes);
        for (size_t i = 0; i < ns; ++i) {
            ((uint32_t *)tab.data())[i] = b->segs[i].first_tile;
            ((unsigned long long *)(tab.data() + off_in))[i] = b->segs[i].in_off;
            ((unsigned long long *)(tab.data() + off_len))[i] = b->segs[i].len;
        }
        ((uint32_t *)tab.data())[ns] = (uint32_t)b->tiles.size();
        b->all_terminal = true;                                   // every segment terminal zone as a whole: the lean predicate kernel
        for (size_t i = 0; i < ns; ++i) if (b->segs[i].len > c->params.terminal_limit) b->all_terminal = false;
        HIP_TRY(c, c->pool.take(bytes, b->d_readtab));
        HIP_TRY(c, hipMemcpyAsync(b->d_readtab.p, tab.data(), bytes, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipStreamSynchronize(st));                    // (tab is a local)
    }
    char *const dt = (char *)b->d_readtab.p;
    TsPredParams Q{};
    Q.terminal_limit = c->params.terminal_limit;
    Q.max_match_dist = c->params.max_match_dist;
    Q.min_block_len = c->params.min_block_len;
    Q.max_block_dist = c->params.max_block_dist;
    Q.min_block_counts = c->params.min_block_counts;
    Q.min_block_density = c->params.min_block_density;
    Q.k = c->k;
    Q.long_list = 128;                                        // floor of the per-wave threshold, see ts_terminal_predicate
    const bool canon = b->emitted && b->kp.emit == 2u && !b->dense && b->all_terminal && b->d_chain.p && b->d_vis.p && !b->kp.vis_wide;
    int e = ts_k_launch_predicate((const TsTile *)b->d_tiles.p, (const unsigned long long *)b->d_tile_off.p,
                                  b->stats_ptr(), b->records_ptr(), b->records_limit(), (const uint32_t *)dt,
                                  (const unsigned long long *)(dt + off_in), (const unsigned long long *)(dt + off_len),
                                  (uint32_t)ns, &Q, d_pass, (uint32_t *)(dt + off_long), (uint32_t *)(dt + off_count),
                                  // the lean kernel: every segment terminal zone as a whole, and the records in the batch's own
                                  // regions (16-byte aligned, 16 bytes of slack behind them: whole aligned blocks can be fetched)
                                  (b->all_terminal && !b->dense && ((uintptr_t)b->records_ptr() & 15u) == 0) ? 1 : 0,
                                  (const uint32_t *)b->d_fill.p, b->dense ? 0xFFFFFFFFu : b->region_cap, b->dense ? 0u : b->total_waves,
                                  (uint32_t *)(dt + off_flag),
                                  // (the scan left the canonical records' indices: the predicate visits only the chains that hold one)
                                  canon ? (const uint32_t *)b->d_chain.p : nullptr, canon ? b->d_vis.p : nullptr, canon ? b->vis_cap : 0u,
                                  b->records16() ? 1 : 0, st);
    if (e != 0) return c->fail(TS_ERR_HIP, "predicate kernel launch failed");
    return TS_OK;
}

int batch_read_pass(ts_batch *b, uint8_t *pass_out, hipStream_t st) {
    ts_ctx *c = b->ctx;
    const size_t ns = b->segs.size();
    if (!ns) return TS_OK;
    DevBuf d_pass;
    struct Return { ts_ctx *c; DevBuf &a; ~Return() { c->pool.give(std::move(a)); } } give_back{c, d_pass};
    HIP_TRY(c, c->pool.take(ns + 16, d_pass));
    int rc = batch_read_pass_device(b, (unsigned char *)d_pass.p, st);
    if (rc != TS_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(pass_out, d_pass.p, ns, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    return TS_OK;
}

int batch_counts(ts_batch *b, ts_segment_counts *counts, bool tips, hipStream_t st) {
    ts_ctx *c = b->ctx;
    const size_t ns = b->segs.size();
    std::vector<unsigned long long> summary(4 * ns);
    DevBuf d_sum;
    struct Return { ts_ctx *c; DevBuf &a; ~Return() { c->pool.give(std::move(a)); } } give_back{c, d_sum};
    HIP_TRY(c, c->pool.take(summary.size() * 8 + 16, d_sum));
    int rc = ts_batch_segment_summary(b, d_sum.p, st);
    if (rc != TS_OK) return rc;
    HIP_TRY(c, hipMemcpyAsync(summary.data(), d_sum.p, summary.size() * 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
    for (size_t i = 0; i < ns; ++i)
        counts[i] = ts_segment_counts{tips ? 0 : summary[4 * i], summary[4 * i + 1], summary[4 * i + 2], summary[4 * i + 3]};
    return TS_OK;
}

// The three-stage pipeline over the groups of one call.  items: the call's segments or reads in input order;
// results go to out / counts / pass at the same indices.
int run_pipeline(ts_ctx *ctx, Mode mode, bool tips, const std::vector<Item> &items, ts_segment_out *out,
                 ts_segment_counts *counts, uint8_t *pass) {
    if (items.empty()) return TS_OK;
    if (ctx->device == kNoDevice) return ctx->fail(TS_ERR_NO_DEVICE, "planning-only context: no HIP device behind it");
    const auto t_begin = Clock::now();
    const bool timing = ctx->knobs.timing;                       // stage times to stderr
    {
        DeviceGuard g(ctx->device);
        if (g.error() != hipSuccess) return ctx->fail(TS_ERR_HIP, "hipSetDevice failed");
        int rc = ensure_streams(ctx);
        if (rc != TS_OK) return rc;
    }
    // groups of consecutive items, ~256 MB of input layout each (an item larger than that is its own group)
    std::vector<Group> groups;
    {
        const uint64_t target = group_target_bytes();
        uint64_t acc = 0;
        Group g;
        for (size_t i = 0; i < items.size(); ++i) {
            const uint64_t bytes = (items[i].len + 15) & ~15ull;
            if (g.count && acc + bytes > target) { groups.push_back(g); g = Group{}; g.first = i; acc = 0; }
            ++g.count;
            acc += bytes;
        }
        if (g.count) groups.push_back(g);
    }
    std::atomic<int> first_err{TS_OK};
    auto set_err = [&](int rc) { int expected = TS_OK; first_err.compare_exchange_strong(expected, rc); };
    Channel<Group *> to_scan, to_down;
    Semaphore inputs_in_flight(2);                               // device input buffers alive at a time

    // planning (host only: tiles, windows, the input layout of a group) runs a group or more ahead of the upload on a thread
    // of its own: for read batches — hundreds of thousands of segments — it takes as long as the upload itself
    Channel<Group *> to_upload;
    std::thread planner([&] {
        ctx->bind_this_thread();
        DeviceGuard g(ctx->device);
        std::vector<uint64_t> lens, abs;
        for (Group &gr : groups) {
            if (first_err.load() != TS_OK) break;
            const auto t0 = Clock::now();
            lens.resize(gr.count); abs.resize(gr.count);
            for (size_t i = 0; i < gr.count; ++i) { lens[i] = items[gr.first + i].len; abs[i] = items[gr.first + i].abs_pos; }
            gr.b = ts_batch_create(ctx, lens.data(), abs.data(), gr.count, tips ? 1 : 0, 0);
            if (gr.b) (void)ts_batch_set_emit(gr.b, 1);      // every download calls blocks on the device
            gr.t_plan = ms_between(t0, Clock::now());
            if (!gr.b) { set_err(ctx->error.rfind("unsupported", 0) == 0 ? TS_ERR_UNSUPPORTED : TS_ERR_HIP); break; }
            to_upload.push(&gr);
        }
        to_upload.close();
    });

    std::thread uploader([&] {
        ctx->bind_this_thread();
        DeviceGuard g(ctx->device);
        int slot = 0;
        bool used[ts_ctx::kUpSlots] = {false, false, false};
        Group *grp;
        while (to_upload.pop(grp)) {
            Group &gr = *grp;
            if (first_err.load() != TS_OK) { ts_batch_destroy(gr.b); gr.b = nullptr; continue; }
            const auto t0 = Clock::now();
            inputs_in_flight.acquire();
            int rc = ts_batch_ensure_device(gr.b);
            const auto t1 = Clock::now();
            if (rc == TS_OK) rc = upload_batch(gr.b, items.data() + gr.first, slot, used);
            if (rc == TS_OK && hipEventCreateWithFlags(&gr.uploaded, hipEventDisableTiming) != hipSuccess) rc = ctx->fail(TS_ERR_HIP, "hipEventCreate failed");
            if (rc == TS_OK && hipEventRecord(gr.uploaded, ctx->up_stream) != hipSuccess) rc = ctx->fail(TS_ERR_HIP, "hipEventRecord failed");
            gr.t_plan += ms_between(t0, t1);
            gr.t_upload = ms_between(t1, Clock::now());
            if (rc != TS_OK) { set_err(rc); ts_batch_destroy(gr.b); gr.b = nullptr; inputs_in_flight.release(); continue; }
            to_scan.push(&gr);
        }
        (void)hipStreamSynchronize(ctx->up_stream);              // the pinned ring is free again when the call returns
        to_scan.close();
    });

    std::thread scanner([&] {
        ctx->bind_this_thread();
        DeviceGuard g(ctx->device);
        Group *gr;
        while (to_scan.pop(gr)) {
            const auto t0 = Clock::now();
            int rc = first_err.load();
            if (rc == TS_OK && hipStreamWaitEvent(ctx->scan_stream, gr->uploaded, 0) != hipSuccess) rc = ctx->fail(TS_ERR_HIP, "hipStreamWaitEvent failed");
            if (rc == TS_OK) rc = ts_batch_scan(gr->b, nullptr, ctx->scan_stream);
            if (rc == TS_OK) rc = ts_batch_sync(gr->b);          // waits for the scan; regrows + rescans on overflow
            ts_batch_release_input(gr->b);
            inputs_in_flight.release();
            gr->t_scan = ms_between(t0, Clock::now());
            if (rc != TS_OK) set_err(rc);
            to_down.push(gr);
        }
        to_down.close();
    });

    std::thread downloader([&] {
        ctx->bind_this_thread();
        DeviceGuard g(ctx->device);
        std::lock_guard<std::mutex> dl(ctx->down_mtx);           // the pinned landing areas are this call's
        Group *gr;
        std::thread post;                                        // host post-processing of the previous group
        int slot = 0;
        auto retire = [&](Group *g2, double t_from) {
            if (g2->uploaded) (void)hipEventDestroy(g2->uploaded);
            ts_batch_destroy(g2->b);
            g2->b = nullptr;
            g2->t_down += t_from;
        };
        while (to_down.pop(gr)) {
            const auto t0 = Clock::now();
            int rc = first_err.load();
            if (rc == TS_OK) gr->b->last_stream = ctx->down_stream;   // the scan is complete (synced): later work runs on this stage's stream
            if (rc == TS_OK && mode == Mode::ReadPass) {
                rc = batch_read_pass(gr->b, pass + gr->first, ctx->down_stream);
            } else if (rc == TS_OK) {
                // device work + D2H of this group while the previous group's records are expanded on the host threads
                ts_fetched *f = ts_batch_fetch(gr->b, mode == Mode::Matches, slot, &rc);
                if (rc == TS_OK && mode == Mode::Blocks && counts) rc = batch_counts(gr->b, counts + gr->first, tips, ctx->down_stream);
                gr->t_fetch = ms_between(t0, Clock::now());
                if (post.joinable()) post.join();
                if (rc == TS_OK) {
                    Group *g2 = gr;
                    post = std::thread([&, g2, f] {
                        DeviceGuard g3(ctx->device);
                        const auto p0 = Clock::now();
                        const int prc = ts_batch_finalize(g2->b, f, out + g2->first);
                        if (prc != TS_OK) set_err(prc);
                        g2->t_final = ms_between(p0, Clock::now());
                        retire(g2, g2->t_final);
                    });
                    slot ^= 1;
                    gr->t_down += ms_between(t0, Clock::now());
                    continue;
                }
                if (f) { std::vector<ts_segment_out> scratch(gr->count); (void)ts_batch_finalize(gr->b, f, scratch.data()); ts_free_segments(scratch.data(), scratch.size()); }
            }
            if (rc != TS_OK) set_err(rc);
            retire(gr, ms_between(t0, Clock::now()));
        }
        if (post.joinable()) post.join();
    });
    planner.join();
    uploader.join();
    scanner.join();
    downloader.join();
    if (timing) {
        double p = 0, u = 0, s = 0, d = 0, f = 0, z = 0;
        for (const Group &gr : groups) { p += gr.t_plan; u += gr.t_upload; s += gr.t_scan; d += gr.t_down; f += gr.t_fetch; z += gr.t_final; }
        const char *name = mode == Mode::Matches ? "ts_scan_segments" : mode == Mode::Blocks ? "ts_scan_segments_blocks" : "ts_filter_reads";
        fprintf(stderr, "%s: %zu items in %zu groups, wall %.1f ms; stage sums (concurrent): plan %.1f ms, stage+upload %.1f ms, "
                        "scan (incl. waiting for the upload) %.1f ms, download + host post-processing %.1f ms (device work + D2H %.1f ms, host expansion %.1f ms)\n",
                name, items.size(), groups.size(), ms_between(t_begin, Clock::now()), p, u, s, d, f, z);
    }
    return first_err.load();
}

// ---------------------------------------------------------------------------------------- coalescing of concurrent callers
}  // namespace
struct SubmitReq {
    int mode; bool tips;
    const std::vector<Item> *items;
    ts_segment_out *out; ts_segment_counts *counts; uint8_t *pass;
    int rc = TS_OK; bool done = false;
    std::string error;
};
namespace {

// One pipeline run for the requests of `group` (all of one mode and kind): their items back to back, every caller's slice of
// the results to its own arrays.  When the merged run fails and more than one caller is in it, each request is run on its
// own, so that only the call that brought the bad input fails.
void run_group(ts_ctx *ctx, std::vector<SubmitReq *> &group) {
    std::lock_guard<std::mutex> api(ctx->api_mtx);
    const Mode mode = (Mode)group[0]->mode;
    const bool tips = group[0]->tips;
    if (group.size() == 1) {
        SubmitReq &r = *group[0];
        r.rc = run_pipeline(ctx, mode, tips, *r.items, r.out, r.counts, r.pass);
        if (r.rc != TS_OK) r.error = ctx->error;
        return;
    }
    size_t total = 0;
    for (SubmitReq *r : group) total += r->items->size();
    std::vector<Item> items;
    items.reserve(total);
    for (SubmitReq *r : group) items.insert(items.end(), r->items->begin(), r->items->end());
    std::vector<ts_segment_out> out(mode == Mode::ReadPass ? 0 : total);
    std::vector<ts_segment_counts> counts(mode == Mode::Blocks ? total : 0);
    std::vector<uint8_t> pass(mode == Mode::ReadPass ? total : 0);
    const int rc = run_pipeline(ctx, mode, tips, items, out.empty() ? nullptr : out.data(), counts.empty() ? nullptr : counts.data(),
                                pass.empty() ? nullptr : pass.data());
    if (rc == TS_OK) {
        size_t at = 0;
        for (SubmitReq *r : group) {
            const size_t n = r->items->size();
            if (r->out) std::memcpy(r->out, out.data() + at, n * sizeof(ts_segment_out));
            if (r->counts) std::memcpy(r->counts, counts.data() + at, n * sizeof(ts_segment_counts));
            if (r->pass) std::memcpy(r->pass, pass.data() + at, n);
            r->rc = TS_OK;
            at += n;
        }
        return;
    }
    if (!out.empty()) ts_free_segments(out.data(), out.size());
    for (SubmitReq *r : group) {
        r->rc = run_pipeline(ctx, mode, tips, *r->items, r->out, r->counts, r->pass);
        if (r->rc != TS_OK) r->error = ctx->error;
    }
}

// run_pipeline for a caller that does not hold the context's call lock: alone, it runs at once; beside others, it is merged
// with them.  (A merged run is capped at ~8 GB of input: what is left waits for the next one.)
int submit_pipeline(ts_ctx *ctx, Mode mode, bool tips, const std::vector<Item> &items, ts_segment_out *out,
                    ts_segment_counts *counts, uint8_t *pass) {
    if (items.empty()) return TS_OK;
    SubmitReq req{(int)mode, tips, &items, out, counts, pass, TS_OK, false, std::string()};
    std::unique_lock<std::mutex> lk(ctx->sq_mtx);
    ctx->sq.push_back(&req);
    while (!req.done) {
        if (ctx->sq_leader) { ctx->sq_cv.wait(lk); continue; }
        ctx->sq_leader = true;
        while (!ctx->sq.empty()) {
            std::vector<SubmitReq *> group;
            const int m = ctx->sq.front()->mode;
            const bool t = ctx->sq.front()->tips;
            uint64_t bytes = 0;
            for (auto it = ctx->sq.begin(); it != ctx->sq.end();) {
                SubmitReq *r = *it;
                if (r->mode != m || r->tips != t) { ++it; continue; }
                uint64_t b = 0;
                for (const Item &x : *r->items) b += x.len;
                if (!group.empty() && bytes + b > (8ull << 30)) { ++it; continue; }
                bytes += b;
                group.push_back(r);
                it = ctx->sq.erase(it);
            }
            lk.unlock();
            // (a throw in here — std::bad_alloc on a merged multi-GB batch — must not leave the context with a leader that
            // no longer exists: every later caller would wait for it forever)
            try {
                run_group(ctx, group);
            } catch (const std::exception &e) {
                for (SubmitReq *r : group)
                    if (!r->done) { r->rc = TS_ERR_ALLOC; r->error = std::string("batched call failed: ") + e.what(); }
            } catch (...) {
                for (SubmitReq *r : group)
                    if (!r->done) { r->rc = TS_ERR_ALLOC; r->error = "batched call failed"; }
            }
            lk.lock();
            for (SubmitReq *r : group) r->done = true;
            ctx->sq_cv.notify_all();
        }
        ctx->sq_leader = false;
        ctx->sq_cv.notify_all();
    }
    if (req.rc != TS_OK && !req.error.empty()) ctx->fail(req.rc, req.error);
    return req.rc;
}

// =========================================================================== general path
// Parameter sets outside the tiled kernel's closed form (mixed-length pattern sets, pattern lengths above 8, or a
// longest pattern exceeding min(step, window-step) where the reference's uint32 start index wraps): the general
// kernels of generic.hip over groups of ~256 MB of regions at a time — match masks, the records the reference pushes
// and the window records all come off the device; the host orders the records by the window that pushes them (only
// mixed-length sets can be out of position order at all), expands them and calls blocks (ts_finalize_segment).
std::atomic<uint64_t> ts_gen_ns[2];       // TS_TIMING: job time in record expansion / in ts_finalize_segment

// blocks_only: the caller reads no match vectors (ts_scan_segments_blocks) — where the blocks can be called on the device
// the match records then never leave it; counts (nullable): the sizes the vectors would have had.
int scan_group_generic(ts_ctx *c, const ts_segment_in *segs, const std::vector<size_t> &which,
                       bool tips, ts_segment_out *out, bool blocks_only = false, ts_segment_counts *counts = nullptr) {
    if (which.empty()) return TS_OK;
    if (!c->generic_ok)
        return c->fail(TS_ERR_UNSUPPORTED, "unsupported parameter set: a non-ACGT pattern, or more than 63 pattern lengths / a pattern "
                                           "longer than 63 bases (more than a ts_pattern holds)");
    DEVICE_TRY(c);
    { int rc = ensure_streams(c); if (rc != TS_OK) return rc; }
    const ts_params &P = c->params;
    const uint32_t s = P.step, w = P.window_size, ov = w - s, L = c->longest;
    hipStream_t st = c->scan_stream;
    TsGenericGeom Q{};
    Q.s = s; Q.w = w; Q.longest = L; Q.nuc_on = (P.out_gc || P.out_entropy) ? 1u : 0u; Q.fold = P.fold_case;
    Q.s_magic = s >= 2u ? (uint32_t)((1ull << 32) / s + 1ull) : 0u;
    Q.cw = w / s; Q.rw = w - Q.cw * s;
    Q.abl = c->knobs.gen_abl;
    const bool timing = c->knobs.timing;
    const auto t_begin = Clock::now();
    double t_dbg[3] = {0, 0, 0};
    double t_up = 0, t_dev = 0, t_host = 0, t_take = 0, t_fused = 0, t_blk = 0, t_d2h = 0, t_wait_next = 0;
    float t_kern = 0;
    if (timing && !c->gen_ev[0]) { HIP_TRY(c, hipEventCreate(&c->gen_ev[0])); HIP_TRY(c, hipEventCreate(&c->gen_ev[1])); }

    struct RegionL { uint64_t seg_start, len, layout_off; };                   // a scanned region and where it lies in the layout
    struct SegL { size_t idx; uint64_t len, abs_pos, layout_off; std::vector<RegionL> regions; uint64_t first_tile = 0, n_tiles = 0, win_base = 0, n_windows = 0; };
    // the wide form (sets beyond 8 lengths / 32 bases): its own kernel, a 64-base halo, records with six bits of length index;
    // smaller groups, because a tile's slot may have to grow to a record per position AND length
    const bool wide = c->gen_wide;
    const uint32_t rec_shift = wide ? 8u : 5u, rec_li_mask = wide ? 63u : 7u;
    static const uint64_t wide_cap = []() -> uint64_t { if (const char *e = getenv("TS_WIDE_GROUP_MB")) { const long mb = atol(e); if (mb > 0) return (uint64_t)mb << 20; } return 256ull << 20; }();
    const uint64_t target = wide ? std::min<uint64_t>(group_target_bytes(), wide_cap) : group_target_bytes();   // (16 - 64 KB of slot per tile: 4 - 17 GB per group)
    int slot = 0;
    bool used[ts_ctx::kUpSlots] = {false, false, false};
    size_t wi = 0;
    // The host stage of group g (ordering + block calling on the host threads) runs on a background thread while the
    // device stage of group g + 1 (upload, kernels, D2H) runs here: what a group's host stage reads lives in a GroupHost.
    struct GroupHost { std::vector<SegL> G; std::vector<TsGeneralTile> tiles; std::vector<unsigned long long> tile_off;
                       std::vector<uint32_t> recs_heap, wins_heap; const uint32_t *recs = nullptr, *wins = nullptr;
                       bool dev_blocks = false; std::vector<TsDevBlock> blocks; std::vector<unsigned long long> sums; };
    // Block calling on the device (blockcall.hip with the general record formats).  The reference calls blocks over allMatches
    // as pushed (src/teloscope.cpp:485-509, :642-657): position order for tips-only scans, for w == s, and under w > s when the
    // pattern lengths differ by at most one (the record that ends later is never pushed earlier: end positions are monotone
    // in stream order, hence so is the pushing window).  Sets with a length gap of two or more under w > s are pushed not
    // quite in position order: for those the compaction writes the dense stream IN PUSH ORDER
    // (ts_general_compact_push) and block calling walks it as the reference does (blockcall.hip, MODE 1: the predecessor as
    // the stream lies, the search range by stream index from a bisection restated probe by probe).  The host's expansion
    // of such a stream needs no ordering pass either.  TS_GEN_HOST_BLOCKS=1 forces the round-4 route — position-ordered
    // stream, ordering and block calling on the host — for A/B and tests.
    const uint32_t len_spread = c->gen_wide ? (c->wide_lens.empty() ? 0u : c->wide_lens.back() - c->wide_lens.front())
                                            : (c->gpat.nlen ? c->gpat.len[c->gpat.nlen - 1] - c->gpat.len[0] : 0u);
    const bool position_order = tips || ov == 0 || len_spread <= 1u;      // position order IS push order
    unsigned long long gen_lens = 0;
    for (uint32_t li = 0; li < c->gpat.nlen && li < 8u; ++li) gen_lens |= (unsigned long long)(c->gpat.len[li] & 63u) << (6u * li);
    if (c->gpat.nlen && c->gpat.len[c->gpat.nlen - 1] > 63u) gen_lens = 0;
    if (c->gen_wide) gen_lens = 1ull;                                      // (wide records: the lengths come from wpat.len; non-zero = "general format")
    const bool dev_blocks_ok = !c->knobs.gen_host_blocks && gen_lens != 0 && (c->gen_wide ? c->wpat.nlen >= 1 : c->gpat.nlen >= 1);
    const bool push_compact = dev_blocks_ok && !position_order;           // the device orders the stream
    const bool known_order = position_order || push_compact;               // what the host stage receives is in push order
    const bool skip_records = blocks_only && dev_blocks_ok;
    size_t group_no = 0;
    std::thread host_job;
    std::atomic<int> host_err{TS_OK};
    struct JoinJob { std::thread &t; ~JoinJob() { if (t.joinable()) t.join(); } } join_job{host_job};
    // What the upload stage hands the device stage: a group planned, its input buffer, tile list and segment table on the device.
    // Group g + 1 is planned, staged and uploaded on a thread of its own while group g's kernels, block calling and
    // download run here (round 4: the two used to run one after the other, 32 + 22 ms per 3 Gb).
    struct Prepared {
        std::shared_ptr<GroupHost> gh;
        DevBuf d_in, d_tiles, d_tab;
        uint64_t nwin_total = 0;
        int rc = TS_OK;
        double ms = 0;
    };
    const bool prefetch = c->knobs.gen_prefetch;
    auto prepare = [&](Prepared &PR) -> int {
        // ---- a group of consecutive segments, ~256 MB of regions; layout = the regions back to back, 16-byte aligned
        PR.gh = std::make_shared<GroupHost>();
        std::vector<SegL> &G = PR.gh->G;
        std::vector<TsGeneralTile> &tiles = PR.gh->tiles;
        std::vector<UpPiece> pieces;
        uint64_t off = 0, nwin_total = 0;
        while (wi < which.size() && (G.empty() || off < target)) {
            const ts_segment_in &sg = segs[which[wi]];
            const Item it{sg.seq, sg.len, sg.abs_pos, sg.input_format, sg.n_pieces};
            SegL sl{which[wi], sg.len, sg.abs_pos, off, {}};
            // regions exactly as scanSegment picks them (src/teloscope.cpp:576-583; uint32 product)
            if (tips) {
                const uint32_t twice = 2u * P.terminal_limit;
                if (sg.len > twice) { sl.regions.push_back({0, P.terminal_limit, 0}); sl.regions.push_back({sg.len - P.terminal_limit, P.terminal_limit, 0}); }
                else if (sg.len) sl.regions.push_back({0, sg.len, 0});
            } else if (sg.len) {
                sl.regions.push_back({0, sg.len, 0});
                sl.n_windows = ceil_div(sg.len, s);
            }
            sl.first_tile = tiles.size();
            for (RegionL &rg : sl.regions) {
                rg.layout_off = off;
                const int rc = region_pieces(c, it, sg.len, rg.seg_start, rg.len, off, 0, ~0ull, pieces);
                if (rc != TS_OK) return rc;
                uint64_t kq = rg.seg_start / s, kr = rg.seg_start - kq * s;          // P0 = kq s + kr, walked from tile to tile
                for (uint64_t a = 0; a < rg.len; a += TS_GENERAL_TILE) {
                    TsGeneralTile T{};
                    T.in_off = off + a;
                    T.seg_rel = rg.seg_start + a;
                    T.k_p0 = kq; T.r_p0 = (uint32_t)kr;
                    T.n = (uint32_t)std::min<uint64_t>(TS_GENERAL_TILE, rg.len - a);
                    T.avail = (uint32_t)std::min<uint64_t>(rg.len - a, (uint64_t)T.n + (wide ? (uint32_t)TS_WIDE_HALO : 32u));
                    T.seg = (uint32_t)G.size();
                    tiles.push_back(T);
                    kr += TS_GENERAL_TILE;
                    if (kr >= s) { const uint64_t d = kr / s; kq += d; kr -= d * s; }
                }
                off += (rg.len + 15) & ~15ull;
            }
            sl.n_tiles = tiles.size() - sl.first_tile;
            sl.win_base = nwin_total;
            nwin_total += sl.n_windows;
            G.push_back(std::move(sl));
            ++wi;
        }
        PR.nwin_total = nwin_total;
        const uint64_t span = off + 64;
        const size_t ns = G.size(), nt = tiles.size();
        if (nt >= 0x7FFFFFFFull) return c->fail(TS_ERR_UNSUPPORTED, "too many tiles in one group");
        const size_t tab_bytes = 4 * ns * 8 + 8 + 16;
        HIP_TRY(c, c->pool.take(span, PR.d_in));
        HIP_TRY(c, c->pool.take(std::max<size_t>(nt, 1) * sizeof(TsGeneralTile), PR.d_tiles));
        HIP_TRY(c, c->pool.take(tab_bytes, PR.d_tab));
        std::vector<unsigned long long> tab(4 * ns + 3, 0ull);
        for (size_t i = 0; i < ns; ++i) { tab[i] = G[i].len; tab[ns + i] = G[i].layout_off; tab[2 * ns + i] = G[i].win_base; tab[3 * ns + i] = G[i].n_windows; }
        tab[4 * ns] = nwin_total;
        const auto t0 = Clock::now();
        { int rc = upload_pieces(c, pieces, PR.d_in.p, 0, slot, used); if (rc != TS_OK) return rc; }
        HIP_TRY(c, hipMemcpyAsync(PR.d_tiles.p, tiles.data(), nt * sizeof(TsGeneralTile), hipMemcpyHostToDevice, c->up_stream));
        HIP_TRY(c, hipMemcpyAsync(PR.d_tab.p, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, c->up_stream));
        HIP_TRY(c, hipStreamSynchronize(c->up_stream));
        PR.ms = ms_between(t0, Clock::now());
        return TS_OK;
    };
    // a tile's slot at the start of a group: what the groups before needed (a telomeric tile under a many-length set holds more
    // than a record per position; finding that out again for every group ran half the groups of a call twice)
    uint32_t slot_cap_call = wide ? TS_GENERAL_TILE * std::min<uint32_t>(4u, std::max<uint32_t>(1u, c->wpat.nlen)) : TS_GENERAL_TILE;
    std::unique_ptr<Prepared> cur(new Prepared), nxt;
    std::thread pf;
    struct JoinJob join_pf{pf};
    cur->rc = prepare(*cur);
    while (cur) {
        if (cur->rc != TS_OK) return cur->rc;
        if (pf.joinable()) pf.join();
        nxt.reset();
        if (wi < which.size()) {
            nxt.reset(new Prepared);
            Prepared *np = nxt.get();
            // (an exception on the thread — std::bad_alloc while planning a group — must come back as an error code, not end the process)
            if (prefetch) pf = std::thread([&, np] {
                try { c->bind_this_thread(); DeviceGuard g2(c->device); np->rc = prepare(*np); }
                catch (const std::exception &e) { np->rc = c->fail(TS_ERR_ALLOC, std::string("general path: planning / upload of a group failed: ") + e.what()); }
            });
        }
        const auto t_iter0 = Clock::now();
        std::shared_ptr<GroupHost> gh = cur->gh;
        std::vector<SegL> &G = gh->G;
        std::vector<TsGeneralTile> &tiles = gh->tiles;
        const uint64_t nwin_total = cur->nwin_total;
        const size_t ns = G.size(), nt = tiles.size();
        // ---- device buffers from the pool
        DevBuf &d_in = cur->d_in, &d_tiles = cur->d_tiles, &d_tab = cur->d_tab;
        DevBuf d_slots, d_stats, d_off, d_tmp, d_rec, d_win;
        struct Return { ts_ctx *c; std::vector<DevBuf *> v; ~Return() { for (DevBuf *d : v) c->pool.give(std::move(*d)); } }
            give_back{c, {&d_in, &d_slots, &d_tiles, &d_tab, &d_stats, &d_off, &d_tmp, &d_rec, &d_win}};
        const size_t tab_len = 0, tab_win = 2 * ns * 8, tab_nwin = 3 * ns * 8, tab_flag = 4 * ns * 8 + 8;
        // a tile's slot: one record per position — all a single-length set can produce; a mixed-length tile that holds
        // more says so, and the group runs again with slots that cannot overflow
        uint32_t slot_cap = slot_cap_call;
        // the list form of the fused pass (per-candidate work on full wavefronts) when a tile adds to few enough window
        // records for the accumulators it keeps in LDS; a tile dense enough to overflow a wave's candidate list sends the
        // group through the position-strided form instead
        bool use_list = !wide && c->knobs.gen_list && s >= 2u && w < (1u << 28) &&
                        (tips || ((uint64_t)TS_GENERAL_TILE + w) / s + 3 <= ts_k_general_list_max_records());
        HIP_TRY(c, c->pool.take(std::max<size_t>(nt, 1) * (size_t)slot_cap * 4, d_slots));
        HIP_TRY(c, c->pool.take((nt + 1) * 16, d_stats));
        HIP_TRY(c, c->pool.take((nt + 1) * 8, d_off));
        HIP_TRY(c, c->pool.take((size_t)ts_k_scan_tmp_bytes((uint32_t)nt), d_tmp));
        if (nwin_total) HIP_TRY(c, c->pool.take(nwin_total * 32, d_win));
        const auto t1 = Clock::now();
        t_up += cur->ms;
        t_take += ms_between(t_iter0, t1);
        // ---- kernels: the fused pass, a prefix sum over the tile counts, the slots into one dense stream
        char *const dt = (char *)d_tab.p;
        std::vector<unsigned long long> &tile_off = gh->tile_off;
        tile_off.assign(nt + 1, 0);
        for (int attempt = 0;; ++attempt) {
            uint32_t flag = 0;
            {
                std::lock_guard<std::mutex> lk(c->mtx);
                if (timing) HIP_TRY(c, hipEventRecord(c->gen_ev[0], st));
                HIP_TRY(c, hipMemsetAsync(dt + tab_flag, 0, 16, st));
                if (nwin_total) HIP_TRY(c, hipMemsetAsync(d_win.p, 0, nwin_total * 32, st));
                if (wide) {
                    if (ts_k_launch_general_wide((const unsigned char *)d_in.p, (const TsGeneralTile *)d_tiles.p, (uint32_t)nt,
                                                 (const unsigned long long *)(dt + tab_len), (const unsigned long long *)(dt + tab_win),
                                                 (const unsigned long long *)(dt + tab_nwin), &c->wpat, &Q,
                                                 tips ? 1 : 0, slot_cap, (uint32_t *)d_stats.p, (uint32_t *)d_slots.p, (uint32_t *)d_win.p,
                                                 (uint32_t *)(dt + tab_flag), st) != 0)
                        return c->fail(TS_ERR_HIP, "general wide kernel launch failed");
                } else
                if (ts_k_launch_general_fused((const unsigned char *)d_in.p, (const TsGeneralTile *)d_tiles.p, (uint32_t)nt,
                                              (const unsigned long long *)(dt + tab_len), (const unsigned long long *)(dt + tab_win),
                                              (const unsigned long long *)(dt + tab_nwin), &c->gpat, &Q, tips ? 1 : 0, slot_cap, (uint32_t *)d_stats.p, (uint32_t *)d_slots.p,
                                              (uint32_t *)d_win.p, (uint32_t *)(dt + tab_flag), use_list ? 1 : 0, c->num_cu, st) != 0)
                    return c->fail(TS_ERR_HIP, "general fused kernel launch failed");
                if (ts_k_launch_tile_offsets((const uint32_t *)d_stats.p, (uint32_t)nt, (unsigned long long *)d_off.p, d_tmp.p, st) != 0)
                    return c->fail(TS_ERR_HIP, "tile-offset kernel launch failed");
                if (timing) HIP_TRY(c, hipEventRecord(c->gen_ev[1], st));
            }
            const auto te0 = Clock::now();
            // (pinned landing: asynchronous for real, then one memcpy into the group's own vector)
            unsigned long long *land = nullptr;
            if (nt && c->pin_off.ensure(std::max<size_t>((nt + 1) * 8 + 64, 1u << 20)) == hipSuccess) land = (unsigned long long *)c->pin_off.p;
            else (void)hipGetLastError();
            uint32_t *flag_land = land ? (uint32_t *)(land + nt + 1) : &flag;
            if (nt) HIP_TRY(c, hipMemcpyAsync(land ? land : tile_off.data(), d_off.p, (nt + 1) * 8, hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipMemcpyAsync(flag_land, dt + tab_flag, 4, hipMemcpyDeviceToHost, st));
            const auto te1 = Clock::now();
            HIP_TRY(c, hipStreamSynchronize(st));
            if (land) { std::memcpy(tile_off.data(), land, (nt + 1) * 8); flag = *flag_land; }
            if (timing) { t_dbg[0] += ms_between(t1, te0); t_dbg[1] += ms_between(te0, te1); t_dbg[2] += ms_between(te1, Clock::now()); }
            if (timing) { float ms = 0; if (hipEventElapsedTime(&ms, c->gen_ev[0], c->gen_ev[1]) == hipSuccess) t_kern += ms; }
            if (!flag) break;
            const uint32_t slot_max = TS_GENERAL_TILE * std::max<uint32_t>(1u, wide ? c->wpat.nlen : c->gpat.nlen);
            if (attempt > (wide ? 4 : 1) || (slot_cap >= slot_max && !(flag & 2u)))
                return c->fail(TS_ERR_STATE, "general path: a tile overflowed a slot that holds every match it can have");
            if (flag & 2u) { use_list = false; continue; }            // a candidate list spilled: the strided form takes this group
            // (the wide form grows by fours: a slot for every position AND length — 63 of them — is 1 MB per tile)
            slot_cap = wide ? std::min<uint32_t>(slot_max, slot_cap * 4u) : slot_max;
            slot_cap_call = slot_cap;
            c->pool.give(std::move(d_slots));
            HIP_TRY(c, c->pool.take(std::max<size_t>(nt, 1) * (size_t)slot_cap * 4, d_slots));
        }
        const uint64_t nrec = tile_off[nt];
        const auto t_f = Clock::now();
        t_fused += ms_between(t1, t_f);
        // blocks only, over a stream in position order: block calling reads the records where the fused pass wrote them (it
        // addresses them through the tile directory), so there is no dense stream to make — every record used to be read and
        // written once more for nothing
        const bool in_place = skip_records && !push_compact && !c->knobs.gen_compact_always;
        if (!in_place) HIP_TRY(c, c->pool.take(std::max<uint64_t>(nrec, 1) * 4, d_rec));
        const uint32_t *const d_records = in_place ? (const uint32_t *)d_slots.p : (const uint32_t *)d_rec.p;
        {
            std::lock_guard<std::mutex> lk(c->mtx);
            if (timing) HIP_TRY(c, hipEventRecord(c->gen_ev[0], st));
            if (in_place) {
                if (ts_k_launch_general_slot_offsets((unsigned long long *)d_off.p, (uint32_t)nt, slot_cap, st) != 0)
                    return c->fail(TS_ERR_HIP, "slot-offset kernel launch failed");
            } else
            if (push_compact) {
                if (ts_k_launch_general_compact_push((const TsGeneralTile *)d_tiles.p, (const uint32_t *)d_stats.p, (unsigned long long *)d_off.p,
                                                     (const uint32_t *)d_slots.p, slot_cap, (uint32_t)nt, (const unsigned long long *)(dt + tab_len),
                                                     w, s, len_spread, wide ? 1 : 0, gen_lens, wide ? c->wpat.len : nullptr, (uint32_t *)d_rec.p, st) != 0)
                    return c->fail(TS_ERR_HIP, "general push-order compact kernel launch failed");
                // (records moved across tile borders: the host stage reads the offsets as they are now)
                if (!skip_records) HIP_TRY(c, hipMemcpyAsync(tile_off.data(), d_off.p, (nt + 1) * 8, hipMemcpyDeviceToHost, st));
            } else
            if (ts_k_launch_general_compact((const uint32_t *)d_stats.p, (const unsigned long long *)d_off.p, (const uint32_t *)d_slots.p,
                                            slot_cap, (uint32_t)nt, (uint32_t *)d_rec.p, st) != 0)
                return c->fail(TS_ERR_HIP, "general compact kernel launch failed");
            if (timing) HIP_TRY(c, hipEventRecord(c->gen_ev[1], st));
        }
        if (dev_blocks_ok) {
            // ---- blocks on the device: the tiles as blockcall.hip addresses them, the canonical / forward counts, then the walks
            DevBuf d_bct, d_sbase;
            struct Ret2 { ts_ctx *c; DevBuf &a, &b2; ~Ret2() { c->pool.give(std::move(a)); c->pool.give(std::move(b2)); } } give2{c, d_bct, d_sbase};
            HIP_TRY(c, c->pool.take(std::max<size_t>(nt, 1) * sizeof(TsTile), d_bct));
            HIP_TRY(c, c->pool.take(std::max<size_t>(ns, 1) * 8, d_sbase));
            std::vector<unsigned long long> sbase(std::max<size_t>(ns, 1), 0ull);
            std::vector<TsShardSegIn> segtab(ns);
            unsigned long long X = 0;
            for (size_t i = 0; i < ns; ++i) {
                sbase[i] = X;
                TsShardSegIn &S = segtab[i];
                S = TsShardSegIn{};
                S.in_off = X; S.len = G[i].len; S.abs_pos = G[i].abs_pos;
                S.t0 = S.o0 = (uint32_t)G[i].first_tile; S.t1 = S.o1 = (uint32_t)(G[i].first_tile + G[i].n_tiles);
                S.flags = TS_SEG_F_HAS_START | TS_SEG_F_HAS_END;
                S.lo_rel = 0; S.hi_rel = G[i].len; S.seg = (uint32_t)i;
                X += G[i].len + 64;
            }
            HIP_TRY(c, hipMemcpyAsync(d_sbase.p, sbase.data(), sbase.size() * 8, hipMemcpyHostToDevice, st));
            if (ts_k_launch_general_block_inputs((const TsGeneralTile *)d_tiles.p, (const unsigned long long *)d_off.p, d_records,
                                                 (const unsigned long long *)d_sbase.p, (uint32_t)nt, (TsTile *)d_bct.p, (uint32_t *)d_stats.p,
                                                 push_compact ? 1 : 0, st) != 0)
                return c->fail(TS_ERR_HIP, "general block-input kernel launch failed");
            int rc = ts_device_block_call_raw(c, (const TsTile *)d_bct.p, (const unsigned long long *)d_off.p, (const uint32_t *)d_stats.p,
                                              d_records, nrec, segtab, nt, tips, gen_lens, nullptr, nullptr, st, gh->blocks, &gh->sums, 0,
                                              wide ? c->wpat.len : nullptr, push_compact);
            if (rc != TS_OK) return rc;
            gh->dev_blocks = true;
        }
        const auto t_b = Clock::now();
        t_blk += ms_between(t_f, t_b);
        // landing area: the context's pinned download buffers, alternating by group (the host stage of group g reads
        // its buffer while group g + 1 lands in the other; it has been joined before group g + 2 arrives)
        {
            const uint64_t nrec_dl = skip_records ? 0 : nrec;
            const size_t rec_bytes = ((size_t)nrec_dl * 4 + 255) & ~(size_t)255, win_bytes = (size_t)nwin_total * 32;
            PinBuf &pb = c->pin_down[group_no & 1];
            ++group_no;
            uint32_t *hrec, *hwin;
            if (rec_bytes + win_bytes + 256 <= (768ull << 20) && pb.ensure(std::max<size_t>(rec_bytes + win_bytes + 256, 32u << 20)) == hipSuccess) {
                hrec = (uint32_t *)pb.p;
                hwin = (uint32_t *)((char *)pb.p + rec_bytes);
            } else {
                (void)hipGetLastError();
                gh->recs_heap.resize(nrec_dl + 1);
                gh->wins_heap.resize(nwin_total * 8 + 1);
                hrec = gh->recs_heap.data();
                hwin = gh->wins_heap.data();
            }
            if (nrec_dl) HIP_TRY(c, hipMemcpyAsync(hrec, d_rec.p, nrec_dl * 4, hipMemcpyDeviceToHost, st));
            if (nwin_total) HIP_TRY(c, hipMemcpyAsync(hwin, d_win.p, nwin_total * 32, hipMemcpyDeviceToHost, st));
            gh->recs = hrec;
            gh->wins = hwin;
        }
        HIP_TRY(c, hipStreamSynchronize(st));
        if (timing) { float ms = 0; if (hipEventElapsedTime(&ms, c->gen_ev[0], c->gen_ev[1]) == hipSuccess) t_kern += ms; }
        const auto t2 = Clock::now();
        t_d2h += ms_between(t_b, t2);
        t_dev += ms_between(t1, t2);
        // ---- host: records -> MatchInfo in the reference's push order, then block calling; one job per segment
        if (host_job.joinable()) host_job.join();                     // (one host stage at a time: it takes all the host threads)
        if (host_err.load() != TS_OK) return host_err.load();
        host_job = std::thread([c, gh, ns, tips, s, w, ov, out, counts, skip_records, timing, wide, rec_shift, rec_li_mask, known_order, &host_err, &t_host]() {
        try {
        const auto th0 = Clock::now();
        const std::vector<SegL> &G = gh->G;
        const std::vector<TsGeneralTile> &tiles = gh->tiles;
        const std::vector<unsigned long long> &tile_off = gh->tile_off;
        const uint32_t *const recs = gh->recs, *const wins = gh->wins;
        std::atomic<size_t> next{0};
        std::atomic<int> first_err{TS_OK};
        const unsigned hw_threads = std::max(1u, std::thread::hardware_concurrency());
        const unsigned spare = std::max(1u, std::min(16u, hw_threads) / (unsigned)std::max<size_t>(1, std::min<size_t>(ns, 16)));
        // the device-called blocks of the group, sorted by segment: where each segment's begin
        std::vector<size_t> blk_at(ns + 1, 0);
        if (gh->dev_blocks) {
            size_t q = 0;
            for (size_t gi = 0; gi < ns; ++gi) {
                blk_at[gi] = q;
                while (q < gh->blocks.size() && gh->blocks[q].seg == gi) ++q;
            }
            blk_at[ns] = q;
        }
        auto worker = [&]() {
            for (size_t gi; (gi = next.fetch_add(1)) < ns && first_err.load() == TS_OK;) {
                const SegL &sl = G[gi];
                const TsDevBlock *pre = gh->dev_blocks ? gh->blocks.data() + blk_at[gi] : nullptr;
                const size_t n_pre = gh->dev_blocks ? blk_at[gi + 1] - blk_at[gi] : 0;
                if (counts && gh->dev_blocks)
                    counts[sl.idx] = ts_segment_counts{tips ? 0 : sl.n_windows, gh->sums[5 * gi + 2], gh->sums[5 * gi + 3], gh->sums[5 * gi + 4]};
                if (skip_records) {
                    // windows + the device's blocks; the match records stayed on the device
                    const int rc = ts_finalize_segment(c, tips, sl.len, sl.abs_pos, sl.n_windows ? &wins[sl.win_base * 8] : nullptr,
                                                       tips ? 0 : sl.n_windows, nullptr, 0, out[sl.idx], spare, pre, n_pre, true);
                    if (rc != TS_OK) { int e = TS_OK; first_err.compare_exchange_strong(e, rc); return; }
                    continue;
                }
                const uint64_t r0 = tile_off[sl.first_tile], r1 = tile_off[sl.first_tile + sl.n_tiles], nm = r1 - r0;
                ts_match *arr = nm ? (ts_match *)ts_alloc_large(nm * sizeof(ts_match)) : nullptr;
                if (nm && !arr) { int e = TS_OK; first_err.compare_exchange_strong(e, c->fail(TS_ERR_ALLOC, "out of host memory")); return; }
                // the window that pushes a match (src/teloscope.cpp:485): records must be in that order
                auto push_window = [&](uint64_t p, uint32_t len) -> uint64_t {
                    const uint64_t e = p + len - 1;
                    return ov == 0 ? p / s : (e < std::min<uint64_t>(w, sl.len) ? 0 : (e - ov) / s);
                };
                // a segment's records are expanded by all the threads its job can spare (a group that holds ONE 250 Mb
                // contig has one job): tile ranges of equal records, each thread checks the push order inside its range
                const unsigned nth = nm >= (1u << 18) ? std::max(1u, std::min<unsigned>(spare, (unsigned)(nm >> 17))) : 1u;
                std::vector<uint64_t> cut(nth + 1, sl.n_tiles);
                cut[0] = 0;
                for (unsigned q = 1; q < nth; ++q)
                    cut[q] = (uint64_t)(std::lower_bound(tile_off.begin() + sl.first_tile, tile_off.begin() + sl.first_tile + sl.n_tiles,
                                                         r0 + nm * q / nth) - (tile_off.begin() + sl.first_tile));
                const auto tw0 = Clock::now();
                std::vector<char> part_sorted(nth, 1);
                std::vector<uint64_t> first_key(nth, 0), last_key(nth, 0);
                // (a stream that is in the reference's push order by construction — tips-only scans, w == s, pattern lengths that
                // differ by at most one: what device block calling relies on as well — is not checked at all)
                const bool check_order = !tips && !known_order;
                const uint64_t head_end = std::min<uint64_t>(w, sl.len);
                auto expand = [&](unsigned q) {
                    bool ok = true, any = false, have_base = false;
                    uint64_t prev_k = 0, cur_k = 0, cur_base = 0;
                    for (uint64_t t = cut[q]; t < cut[q + 1]; ++t) {
                        const TsGeneralTile &T = tiles[sl.first_tile + t];
                        uint64_t at = tile_off[sl.first_tile + t] - r0;
                        for (uint64_t ri = tile_off[sl.first_tile + t]; ri < tile_off[sl.first_tile + t + 1]; ++ri, ++at) {
                            const uint32_t rec = recs[ri];
                            const uint64_t p = T.seg_rel + (rec >> rec_shift);
                            const uint32_t len = wide ? c->wide_lens[(rec >> 2) & rec_li_mask] : c->gpat.len[(rec >> 2) & rec_li_mask];
                            ts_match &m = arr[at];
                            std::memset(&m, 0, sizeof m);
                            m.position = sl.abs_pos + p;
                            m.match_size = (uint16_t)len;
                            m.flags = (uint8_t)(((rec & 1u) ? TS_MATCH_FORWARD : 0u) | ((rec & 2u) ? TS_MATCH_CANONICAL : 0u));   // (general records: forward is bit 0)
                            if (check_order) {
                                // the pushing window without a division per record: the stream is in position order, so the quotient
                                // of the record before is at most a step or two away
                                uint64_t k = 0;
                                const uint64_t e = p + len - 1;
                                if (ov == 0 || e >= head_end) {
                                    const uint64_t x = ov == 0 ? p : e - ov;
                                    if (!have_base) { cur_k = x / s; cur_base = cur_k * s; have_base = true; }
                                    while (x >= cur_base + s) { ++cur_k; cur_base += s; }
                                    while (x < cur_base) { --cur_k; cur_base -= s; }
                                    k = cur_k;
                                }
                                if (!any) { first_key[q] = k; any = true; }
                                else if (k < prev_k) ok = false;
                                prev_k = k;
                            }
                        }
                    }
                    part_sorted[q] = ok ? 1 : 0;
                    last_key[q] = prev_k;
                    if (!any) part_sorted[q] = 2;                                     // (an empty part)
                };
                if (nth <= 1) expand(0);
                else {
                    std::vector<std::thread> ex;
                    for (unsigned q = 0; q < nth; ++q) ex.emplace_back(expand, q);
                    for (std::thread &th : ex) th.join();
                }
                bool sorted = true;
                {
                    bool have_prev = false;
                    uint64_t prev_last = 0;
                    for (unsigned q = 0; q < nth; ++q) {                                 // (and the order across the parts' seams)
                        if (part_sorted[q] == 2) continue;
                        if (!part_sorted[q] || (have_prev && first_key[q] < prev_last)) sorted = false;
                        prev_last = last_key[q];
                        have_prev = true;
                    }
                }
                if (!tips && !sorted) {
                    // mixed-length sets: a long match near a window start is pushed by the NEXT window, after shorter
                    // matches that begin behind it — order by pushing window, position order within it
                    // The stream is in position order and a record's pushing window grows with its END position, so a record is
                    // out of place by at most the few records that start within (longest - shortest) bases ahead of it: one
                    // insertion pass, stable by construction (a record only moves behind records with a LARGER key), instead
                    // of a stable_sort over an index array and a gather (0.7 s per 3 Gb on a nine-length set).
                    std::vector<uint64_t> key(nm);
                    for (uint64_t i = 0; i < nm; ++i) key[i] = push_window(arr[i].position - sl.abs_pos, arr[i].match_size);
                    for (uint64_t i = 1; i < nm; ++i) {
                        if (key[i] >= key[i - 1]) continue;
                        const ts_match t = arr[i];
                        const uint64_t kk = key[i];
                        uint64_t j = i;
                        while (j > 0 && key[j - 1] > kk) { arr[j] = arr[j - 1]; key[j] = key[j - 1]; --j; }
                        arr[j] = t; key[j] = kk;
                    }
                }
                const auto tw1 = Clock::now();
                const int rc = ts_finalize_segment(c, tips, sl.len, sl.abs_pos, sl.n_windows ? &wins[sl.win_base * 8] : nullptr,
                                                   tips ? 0 : sl.n_windows, arr, nm, out[sl.idx], spare, pre, n_pre, gh->dev_blocks);
                if (timing) { ts_gen_ns[0] += (uint64_t)(ms_between(tw0, tw1) * 1e6); ts_gen_ns[1] += (uint64_t)(ms_between(tw1, Clock::now()) * 1e6); }
                if (rc != TS_OK) { int e = TS_OK; first_err.compare_exchange_strong(e, rc); return; }
            }
        };
        const unsigned nthreads = (unsigned)std::min<size_t>({(size_t)16, ns, (size_t)hw_threads});
        if (nthreads <= 1) worker();
        else {
            std::vector<std::thread> pool;
            for (unsigned i = 0; i < nthreads; ++i) pool.emplace_back(worker);
            for (std::thread &th : pool) th.join();
        }
        if (first_err.load() != TS_OK) { int e = TS_OK; host_err.compare_exchange_strong(e, first_err.load()); }
        t_host += ms_between(th0, Clock::now());
        } catch (...) { int e = TS_OK; host_err.compare_exchange_strong(e, c->fail(TS_ERR_ALLOC, "general path: the host stage ran out of memory")); }
        });
        // ---- next group: already uploaded by the prefetch thread (or planned and uploaded here)
        const auto t_w = Clock::now();
        if (pf.joinable()) pf.join();
        t_wait_next += ms_between(t_w, Clock::now());
        if (nxt && !prefetch) nxt->rc = prepare(*nxt);
        cur = std::move(nxt);
    }
    if (host_job.joinable()) host_job.join();
    if (host_err.load() != TS_OK) return host_err.load();
    if (timing)
        fprintf(stderr, "general path: device stage: buffers %.1f ms, fused pass + tile offsets (synced) %.1f ms, compaction + block calling %.1f ms, D2H %.1f ms, waiting for the next group's upload %.1f ms (fused stage: enqueue %.1f, copies enqueue %.1f, sync %.1f)\n",
                t_take, t_fused, t_blk, t_d2h, t_wait_next, t_dbg[0], t_dbg[1], t_dbg[2]);
    if (timing)
        fprintf(stderr, "general path: route: %s form, blocks called on the %s, stream %s\n", wide ? "wide" : "table",
                dev_blocks_ok ? "device" : "host", position_order ? "in position order" : push_compact ? "written in push order by the device" : "ordered on the host");
    if (timing)
        fprintf(stderr, "general path: %zu segments, wall %.1f ms: upload %.1f ms, kernels + D2H %.1f ms (kernels alone, HIP events: %.2f ms), host ordering + block calling %.1f ms (on a thread of its own, one group behind; job time: expansion %.1f ms, windows + block calling %.1f ms)\n",
                which.size(), ms_between(t_begin, Clock::now()), t_up, t_dev, (double)t_kern, t_host, ts_gen_ns[0].exchange(0) / 1e6, ts_gen_ns[1].exchange(0) / 1e6);
    return TS_OK;
}

std::vector<Item> items_of(const ts_segment_in *segs, const std::vector<size_t> &which) {
    std::vector<Item> v(which.size());
    for (size_t i = 0; i < which.size(); ++i) v[i] = Item{segs[which[i]].seq, segs[which[i]].len, segs[which[i]].abs_pos, segs[which[i]].input_format, segs[which[i]].n_pieces};
    return v;
}

// scanSegment over the subset `which` (all full scans or all tips-only) on the tiled kernel
int scan_subset(ts_ctx *ctx, Mode mode, const ts_segment_in *segs, const std::vector<size_t> &which, bool tips,
                ts_segment_out *out, ts_segment_counts *counts, bool have_lock) {
    if (which.empty()) return TS_OK;
    const std::vector<Item> items = items_of(segs, which);
    // results land in arrays parallel to `which`, then move to their places
    std::vector<ts_segment_out> tmp(which.size());
    std::vector<ts_segment_counts> cnt(counts ? which.size() : 0);
    int rc = have_lock ? run_pipeline(ctx, mode, tips, items, tmp.data(), counts ? cnt.data() : nullptr, nullptr)
                       : submit_pipeline(ctx, mode, tips, items, tmp.data(), counts ? cnt.data() : nullptr, nullptr);
    if (rc != TS_OK) { ts_free_segments(tmp.data(), tmp.size()); return rc; }
    for (size_t i = 0; i < which.size(); ++i) {
        out[which[i]] = tmp[i];
        if (counts) counts[which[i]] = cnt[i];
    }
    return TS_OK;
}

// the general kernels run one call at a time (their groups are not merged across callers)
int generic_locked(ts_ctx *ctx, const ts_segment_in *segs, const std::vector<size_t> &which, bool tips, ts_segment_out *out, bool have_lock,
                   bool blocks_only = false, ts_segment_counts *counts = nullptr) {
    // (an exception — std::bad_alloc while a multi-GB group is planned — must leave as an error code: this is a C boundary)
    try {
        if (have_lock) return scan_group_generic(ctx, segs, which, tips, out, blocks_only, counts);
        std::lock_guard<std::mutex> api(ctx->api_mtx);
        return scan_group_generic(ctx, segs, which, tips, out, blocks_only, counts);
    } catch (const std::exception &e) {
        return ctx->fail(TS_ERR_ALLOC, std::string("general path: ") + e.what());
    }
}

// ts_scan_segments; have_lock: the caller holds the context's call lock (ts_filter_reads's general path, ts_scan_segments_multi)
int scan_segments_impl(ts_ctx *ctx, const ts_segment_in *segs, size_t n_segs, ts_segment_out *out, bool have_lock) {
    for (size_t i = 0; i < n_segs; ++i) {
        std::memset(&out[i], 0, sizeof out[i]);
        if (segs[i].len && !segs[i].seq) return ctx->fail(TS_ERR_INVALID_ARG, "null sequence pointer");
        if (segs[i].input_format > TS_INPUT_PACKED2) return ctx->fail(TS_ERR_INVALID_ARG, "unknown input_format");
    }
    std::vector<size_t> full, tips;
    for (size_t i = 0; i < n_segs; ++i) (segs[i].tips_only ? tips : full).push_back(i);
    std::string why;
    int rc = ts_full_scan_supported(ctx, why) ? scan_subset(ctx, Mode::Matches, segs, full, false, out, nullptr, have_lock)
                                              : generic_locked(ctx, segs, full, false, out, have_lock);
    if (rc == TS_OK) rc = ctx->fast_ok ? scan_subset(ctx, Mode::Matches, segs, tips, true, out, nullptr, have_lock)
                                       : generic_locked(ctx, segs, tips, true, out, have_lock);
    if (rc != TS_OK) ts_free_segments(out, n_segs);
    return rc;
}

}  // namespace

int ts_scan_segments_unlocked(ts_ctx *ctx, const ts_segment_in *segs, size_t n_segs, ts_segment_out *out) {
    return scan_segments_impl(ctx, segs, n_segs, out, true);
}

// the pieces of the pipeline that ts_scan_segments_multi (multi.cpp) runs per context
int ts_pipeline_ensure_streams(ts_ctx *c) { return ensure_streams(c); }
int ts_pipeline_upload_batch(ts_batch *b, const ts_segment_in *segs, int *slot, bool used[]) {
    std::vector<Item> items(b->segs.size());
    for (size_t i = 0; i < items.size(); ++i) items[i] = Item{segs[i].seq, segs[i].len, segs[i].abs_pos, segs[i].input_format, segs[i].n_pieces};
    return upload_batch(b, items.data(), *slot, used);
}

// =========================================================================== scanSegment, batched
extern "C" {

int ts_scan_segments(ts_ctx *ctx, const ts_segment_in *segs, size_t n_segs, ts_segment_out *out) {
    if (!ctx || (n_segs && (!segs || !out))) return TS_ERR_INVALID_ARG;
    return scan_segments_impl(ctx, segs, n_segs, out, false);
}

// scanSegment for callers that do not read the match vectors: scan, block calling and the per-segment
// counts all stay on the device; windows, blocks and four counters per segment cross PCIe.
int ts_scan_segments_blocks(ts_ctx *ctx, const ts_segment_in *segs, size_t n_segs, ts_segment_out *out,
                            ts_segment_counts *counts) {
    if (!ctx || (n_segs && (!segs || !out))) return TS_ERR_INVALID_ARG;
    for (size_t i = 0; i < n_segs; ++i) {
        std::memset(&out[i], 0, sizeof out[i]);
        if (counts) counts[i] = ts_segment_counts{0, 0, 0, 0};
        if (segs[i].len && !segs[i].seq) return ctx->fail(TS_ERR_INVALID_ARG, "null sequence pointer");
        if (segs[i].input_format > TS_INPUT_PACKED2) return ctx->fail(TS_ERR_INVALID_ARG, "unknown input_format");
    }
    std::vector<size_t> full, tips;
    for (size_t i = 0; i < n_segs; ++i) (segs[i].tips_only ? tips : full).push_back(i);
    // parameter sets outside the tiled kernel take the general path and drop the match vectors afterwards
    auto via_matches = [&](const std::vector<size_t> &which, bool tips_mode) -> int {
        // (blocks on the device and no record download where the match stream is in calling order; else the host path, whose
        // match vectors are counted and dropped here)
        if (counts) for (size_t i : which) counts[i] = ts_segment_counts{~0ull, 0, 0, 0};
        int rc = generic_locked(ctx, segs, which, tips_mode, out, false, true, counts);
        if (rc != TS_OK) return rc;
        for (size_t i : which) {
            if (counts && counts[i].n_windows == ~0ull) {
                ts_segment_counts cnt{tips_mode ? 0 : out[i].n_windows, out[i].n_matches, 0, 0};
                for (uint64_t m = 0; m < out[i].n_matches; ++m) {
                    cnt.n_canonical += (out[i].matches[m].flags & TS_MATCH_CANONICAL) ? 1 : 0;
                    cnt.n_forward += (out[i].matches[m].flags & TS_MATCH_FORWARD) ? 1 : 0;
                }
                counts[i] = cnt;
            }
            std::free(out[i].matches);
            out[i].matches = nullptr;
            out[i].n_matches = 0;
        }
        return TS_OK;
    };
    std::string why;
    int rc = ts_full_scan_supported(ctx, why) ? scan_subset(ctx, Mode::Blocks, segs, full, false, out, counts, false) : via_matches(full, false);
    if (rc == TS_OK) rc = ctx->fast_ok ? scan_subset(ctx, Mode::Blocks, segs, tips, true, out, counts, false) : via_matches(tips, true);
    if (rc != TS_OK) ts_free_segments(out, n_segs);
    return rc;
}

// =========================================================================== ReadTelomereFilter
int ts_filter_reads(ts_ctx *ctx, const char *const *seqs, const uint64_t *lens, size_t n_reads,
                    uint8_t *pass) {
    if (!ctx || (n_reads && (!seqs || !lens || !pass))) return TS_ERR_INVALID_ARG;
    if (!ctx->read_filter) return ctx->fail(TS_ERR_STATE, "context was not made by ts_create_read_filter");
    if (n_reads == 0) return TS_OK;
    std::vector<Item> items(n_reads);
    for (size_t i = 0; i < n_reads; ++i) {
        uint64_t n = lens[i];
        if (n && !seqs[i]) return ctx->fail(TS_ERR_INVALID_ARG, "null sequence pointer");
        if (n && seqs[i][n - 1] == '\r') --n;             // src/read-filter.cpp:38-40
        items[i] = Item{seqs[i], n, 0, TS_INPUT_BASES, 0};
    }
    if (!ctx->fast_ok) {
        // pattern sets outside the tiled kernel (mixed lengths, k > 8): the general kernels in their blocks-only form — a
        // tips-only scan's stream is in calling order, so the blocks are called on the device and no match record leaves it
        // (the wide form alone keeps host block calling); all the filter reads is whether a read has a terminal block
        std::vector<ts_segment_in> in(n_reads);
        std::vector<size_t> all(n_reads);
        for (size_t i = 0; i < n_reads; ++i) { in[i] = ts_segment_in{}; in[i].seq = items[i].seq; in[i].len = items[i].len; in[i].abs_pos = 0; in[i].tips_only = 1; all[i] = i; }
        std::vector<ts_segment_out> out(n_reads);
        for (size_t i = 0; i < n_reads; ++i) std::memset(&out[i], 0, sizeof out[i]);
        int rc = generic_locked(ctx, in.data(), all, true, out.data(), false, true, nullptr);
        if (rc != TS_OK) { ts_free_segments(out.data(), n_reads); return rc; }
        for (size_t i = 0; i < n_reads; ++i) pass[i] = out[i].n_terminal_blocks != 0;
        ts_free_segments(out.data(), n_reads);
        return TS_OK;
    }
    // tiled path: whole-read tips scan, then the terminal-block predicate on the device; only one
    // byte per read comes back
    return submit_pipeline(ctx, Mode::ReadPass, true, items, nullptr, nullptr, pass);
}

// ReadTelomereFilter::matches over a device-resident tips-only batch (reads already in HBM, scanned on `stream`):
// the pass byte of every read to d_pass, asynchronously on the same stream.
int ts_batch_read_pass(ts_batch *b, void *d_pass, void *stream) {
    if (!b || !d_pass) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    DEVICE_TRY(c);
    if (!b->tips || !b->whole() || !b->scanned) return c->fail(TS_ERR_STATE, "ts_batch_read_pass needs a scanned, unrestricted tips-only batch");
    return batch_read_pass_device(b, (unsigned char *)d_pass, (hipStream_t)stream);
}

int ts_batch_read_pass_status(ts_batch *b, int *overflowed) {
    if (!b || !overflowed) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    DEVICE_TRY(c);
    *overflowed = 0;
    if (!b->d_readtab.p) return TS_OK;                            // no pass was ever enqueued
    const size_t ns = b->segs.size();
    const size_t off_in = (((ns + 1) * 4 + 15) & ~(size_t)15), off_len = off_in + ns * 8, off_long = off_len + ns * 8,
                 off_count = (off_long + ns * 4 + 15) & ~(size_t)15, off_flag = off_count + 16;
    uint32_t flag = 0;
    char *const dt = (char *)b->d_readtab.p;
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(&flag, dt + off_flag, 4, hipMemcpyDeviceToHost));
    if (flag) HIP_TRY(c, hipMemset(dt + off_flag, 0, 4));
    *overflowed = flag ? 1 : 0;
    return TS_OK;
}

int ts_filter_reads_multi(ts_ctx *const *ctxs, size_t n_ctx, const char *const *seqs, const uint64_t *lens,
                          size_t n_reads, uint8_t *pass) {
    if (!ctxs || !n_ctx || (n_reads && (!seqs || !lens || !pass))) return TS_ERR_INVALID_ARG;
    for (size_t i = 0; i < n_ctx; ++i) if (!ctxs[i]) return TS_ERR_INVALID_ARG;
    if (n_ctx == 1 || n_reads < 2 * n_ctx) return ts_filter_reads(ctxs[0], seqs, lens, n_reads, pass);
    // consecutive shards of equal bases: shard d ends at the first read where the running total reaches
    // (d + 1) / n_ctx of all bases; pass[] is written in place, so the result is in input order by construction
    uint64_t total = 0;
    for (size_t i = 0; i < n_reads; ++i) total += lens[i];
    std::vector<size_t> cut(n_ctx + 1, n_reads);
    cut[0] = 0;
    {
        uint64_t acc = 0;
        size_t d = 1;
        for (size_t i = 0; i < n_reads && d < n_ctx; ++i) {
            acc += lens[i];
            while (d < n_ctx && (unsigned __int128)acc * n_ctx >= (unsigned __int128)total * d) cut[d++] = i + 1;
        }
    }
    std::vector<int> rcs(n_ctx, TS_OK);
    std::vector<std::thread> pool;
    for (size_t d = 0; d < n_ctx; ++d)
        pool.emplace_back([&, d] {
            const size_t a = cut[d], z = cut[d + 1];
            if (z > a) rcs[d] = ts_filter_reads(ctxs[d], seqs + a, lens + a, z - a, pass + a);
        });
    for (std::thread &th : pool) th.join();
    for (size_t d = 0; d < n_ctx; ++d)
        if (rcs[d] != TS_OK) {
            if (d != 0) ctxs[0]->fail(rcs[d], std::string("shard on context ") + std::to_string(d) + ": " + ts_last_error(ctxs[d]));
            return rcs[d];
        }
    return TS_OK;
}

}  // extern "C"
