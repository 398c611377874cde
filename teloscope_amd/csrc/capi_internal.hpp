// capi_internal.hpp — structures shared by the host translation units of libteloscan
// (capi.cpp: contexts and device-resident batches; pipeline.cpp: the host-buffer entry points).
#ifndef TS_CAPI_INTERNAL_HPP
#define TS_CAPI_INTERNAL_HPP

#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "host.hpp"
#include "ts_internal.h"

// A planning-only context (ts_params.device == TS_DEVICE_NONE) never touches HIP: it plans batches
// (tiles, windows, input layout) and every device entry point fails on it with TS_ERR_NO_DEVICE.
constexpr int kNoDevice = TS_DEVICE_NONE;

// Device memory block; move-only, frees itself.
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    hipError_t ensure(size_t need) {
        if (need <= bytes && p) return hipSuccess;
        release();
        hipError_t e = hipMalloc(&p, need ? need : 16);
        if (e == hipSuccess) bytes = need ? need : 16; else p = nullptr;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

// Pinned host block (hipHostMalloc); move-only, frees itself.
struct PinBuf {
    void *p = nullptr;
    size_t bytes = 0;
    PinBuf() = default;
    PinBuf(const PinBuf &) = delete;
    PinBuf &operator=(const PinBuf &) = delete;
    PinBuf(PinBuf &&o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    PinBuf &operator=(PinBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~PinBuf() { release(); }
    hipError_t ensure(size_t need) {
        if (need <= bytes && p) return hipSuccess;
        release();
        hipError_t e = hipHostMalloc(&p, need ? need : 16, hipHostMallocDefault);
        if (e == hipSuccess) bytes = need ? need : 16; else p = nullptr;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; bytes = 0; }
};

// Device blocks kept between calls: the host entry points plan a batch per group of segments, and a
// hipMalloc / hipFree pair per multi-GB buffer per call costs more than the scan (hipFree also waits for
// the device).  take() hands out the smallest free block that fits within 2x, else allocates.
class BufferPool {
public:
    hipError_t take(size_t need, DevBuf &out);
    void give(DevBuf &&b);
    void clear();
private:
    std::mutex m_;
    std::vector<DevBuf> free_;
    size_t held_ = 0;
    static constexpr size_t kMaxBlocks = 96;
    static constexpr size_t kMaxHeld = 48ull << 30;
};

// Sets the calling thread's HIP device for the scope (every entry point that touches the device: a pool
// thread's current device need not be the context's) and restores the previous one.
class DeviceGuard {
public:
    explicit DeviceGuard(int device) {
        if (device < 0) { err_ = hipErrorNoDevice; return; }
        if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
        err_ = prev_ == device ? hipSuccess : hipSetDevice(device);
        changed_ = err_ == hipSuccess && prev_ != device;
    }
    ~DeviceGuard() { if (changed_ && prev_ >= 0) (void)hipSetDevice(prev_); }
    hipError_t error() const { return err_; }
private:
    int prev_ = -1;
    bool changed_ = false;
    hipError_t err_ = hipSuccess;
};

struct ts_ctx {
    ts_params params{};
    std::vector<ts::Pattern> patterns;
    ts::BlockParams bp{};
    uint32_t k = 0;                 // uniform pattern length (0 = mixed)
    uint32_t longest = 0;
    bool fast_ok = false;           // table-driven tiled kernel usable for the pattern set
    std::string why_not;            // reason when a scan mode is unsupported
    int device = 0;                 // HIP ordinal, or kNoDevice for a planning-only context
    int num_cu = 0;
    uint32_t table_rows = 0, fc_bytes = 0;
    bool fc_byte_table = true, pair_byte_table = false;
    // general kernels (generic.hip): sorted 2-bit codes per pattern length
    bool generic_ok = false;
    TsGenericPatterns gpat{};
    DevBuf d_gcodes, d_gflags;
    // ... or, for sets beyond 8 lengths / 32 bases, the wide form's tables (128-bit codes, up to 63 lengths of up to 63 bases)
    bool gen_wide = false;
    TsWidePatterns wpat{};
    std::vector<uint32_t> wide_lens;        // host copy of wpat.len
    DevBuf d_wlo, d_whi, d_wflags, d_wlen, d_wfirst;
    DevBuf d_table;
    mutable std::mutex mtx;         // guards batch planning / launches that read the context's tables
    mutable std::string error;
    bool read_filter = false;
    // measurement / test knobs of the host entry points, read from the environment ONCE, when the context is made (never per call):
    // TS_TIMING (stage times to stderr), TS_GEN_HOST_BLOCKS=1 (general path: block calling on the host), TS_GEN_PREFETCH=0,
    // TS_GEN_LIST=0 (general path: the strided form), TS_GEN_ABL (profiling mask of the general kernels)
    // TS_PACKED_UPLOAD=0 (bases cross PCIe as ASCII), TS_PACKED_MIN_BYTES (calls below it go plain), TS_STAGE_THREADS
    struct Knobs {
        bool timing = false, gen_host_blocks = false, gen_prefetch = true, gen_list = true, packed_upload = true, gen_compact_always = false;
        uint32_t gen_abl = 0, stage_threads = 0;
        int side_priority = 0;                                // stream priority of the pack's side stream (0: the default priority)
        int scan_events = 2;                                  // events ts_batch_scan records around a scan: 2 both (kernel times), 1 the one behind it, 0 none (measurements)
        bool rec16 = true;                                    // read-filter batches keep 16-bit records where they can (TS_REC32=1: 32-bit, for A/B)
        bool side_probe = true;                               // try the side stream against the scan / pack streams it meets (shard.cpp)
        uint64_t packed_min_bytes = 1u << 20;                 // small calls are latency, not link time: they go plain
    } knobs;
    BufferPool pool;
    // host-buffer entry points (pipeline.cpp): pinned staging rings and their streams
    // CPUs of the NUMA node the device is attached to (empty: unknown / TS_NO_NUMA_BIND): the pipeline's own threads run
    // there, and the pinned staging buffers are allocated there — a DMA out of the other socket's memory, or staging
    // threads on the other socket, cost 15-20 % of the PCIe-inclusive rate on a two-socket host
    std::vector<float> entropy_term;      // ts::entropy_terms(window_size): windows of the full size look their terms up
    std::vector<int> node_cpus;
    void bind_this_thread() const;          // no-op when node_cpus is empty; never widens the thread's current mask
    static constexpr int kUpSlots = 3;
    hipStream_t side_stream = nullptr;          // ts_batch_pack_shard: the terminal walks of every batch's pack (created on first use)
    std::mutex side_mtx;
    // the streams (scan, pack) the side stream has been tried against: HIP puts a process's streams on a few hardware queues, and a
    // side stream that shares the scan stream's queue puts the terminal walks of step i in front of the scan of step i + 1
    // (shard.cpp: side_stream_for)
    std::vector<void *> side_tried;
    PinBuf pin_up[kUpSlots];
    hipEvent_t pin_up_ev[kUpSlots] = {nullptr, nullptr, nullptr};
    DevBuf d_pack[kUpSlots], d_runs[kUpSlots];   // packed upload: a chunk's 2-bit codes and invalid runs on the device
    PinBuf pin_runs[kUpSlots];
    hipEvent_t gen_ev[2] = {nullptr, nullptr};   // TS_TIMING: around the general path's kernels
    PinBuf pin_down[2];
    PinBuf pin_off;                              // general path: a group's tile directory lands here (a pageable landing cost 9 ms per MB-sized copy)
    hipStream_t up_stream = nullptr, scan_stream = nullptr, down_stream = nullptr;

    std::mutex down_mtx;            // one download (pinned landing area + its stream) at a time
    mutable std::mutex err_mtx;
    std::mutex api_mtx;             // one pipeline run at a time per context (the runs share the pinned rings and streams)
    // Concurrent callers are COALESCED, not queued one behind the other (the reference calls scanSegment / matches
    // concurrently from its pool workers, src/input.cpp:977, :786): a call that arrives while a run is in flight waits in
    // `sq`; whoever finds no run in flight becomes the leader, takes every waiting call of one kind, runs them as ONE
    // batch and hands each caller its slice (pipeline.cpp: submit_pipeline)
    std::mutex sq_mtx;
    std::condition_variable sq_cv;
    std::deque<struct SubmitReq *> sq;
    bool sq_leader = false;
    int fail(int code, const std::string &msg) const { std::lock_guard<std::mutex> g(err_mtx); error = msg; return code; }
};

struct Region {                     // one scanned interval of a segment
    uint64_t start, len;            // relative to the segment
    uint32_t first_tile, n_tiles;
    uint64_t tile_bases;            // owned bases per tile
};

struct SegPlan {
    uint64_t len = 0, abs_pos = 0;
    uint64_t in_off = 0;            // byte offset in the (whole-batch) input layout
    uint64_t win_base = 0, n_windows = 0;
    uint32_t first_tile = 0, n_tiles = 0;
    std::vector<Region> regions;
};

// A shard of a plan (shard.cpp): the tiles it owns, the tiles it scans (owned + context), the segments with an owned tile
struct ShardRange {
    uint64_t own_lo, own_hi, ext_lo, ext_hi;
    uint64_t seg_begin, n_segs;
};
// ... and the layout of its result message
struct ShardLayout {
    uint64_t off_segs, off_windows, off_tilevis, off_visible, off_blocks, bytes;
    uint64_t visible_capacity;
    uint32_t block_capacity, window_bytes, visible_bytes, field_bits;
    uint64_t n_windows;
};

// A batch is a PLAN over all its segments (tiles, window records, input layout) plus the device state of
// the tile range [tile_lo, tile_hi) it executes: the whole plan by default, one rank's shard after
// ts_batch_restrict, or — on the rank that assembles — results produced elsewhere (ts_batch_adopt).
struct ts_batch {
    ts_ctx *ctx = nullptr;
    bool tips = false;
    std::vector<SegPlan> segs;
    std::vector<TsTile> tiles;      // the whole plan
    TsScanParams kp{};
    uint32_t wpt = 1;               // windows per tile
    uint32_t grid = 0, lds_bytes = 0;
    uint64_t total_bases = 0, input_bytes = 0, n_windows = 0, match_cap = 0, match_cap_request = 0;
    uint64_t n_matches = 0;
    // executed range and what it covers
    uint64_t tile_lo = 0, tile_hi = 0, win_lo = 0, win_hi = 0, in_lo = 0, in_hi = 0, range_bases = 0;
    bool allocated = false;
    double last_ms = 0.0;
    bool scanned = false, synced = false;
    bool dense = false;             // match records form one dense stream in tile order (adopted / exported results)
    const void *last_input = nullptr;
    void *last_stream = nullptr;
    DevBuf d_in, d_tiles, d_windows, d_matches, d_tile_off, d_stats, d_fill, d_tickets, d_segtab, d_dense, d_dense_base, d_scan_tmp, d_readtab;
    // what the scan hands to block calling and to a shard's message (kp.emit): per-wave regions of visible records, the
    // per-tile chain summaries (TsTileChain) and, planned on the host, the terminal-zone word of every tile
    DevBuf d_vis, d_chain, d_zone;
    uint32_t vis_cap = 0;           // visible records per wave region
    bool emitted = false;           // the latest scan ran the emitting build: tile_stats word 3, d_chain and d_vis describe it
    bool chain_valid() const { return emitted && !dense && d_chain.p != nullptr; }
    // a shard (ts_batch_restrict_shard): the range the batch executes is its OWNED tiles [own_lo, own_hi) plus context tiles
    uint32_t shard_parts = 0, shard_part = 0, shard_scale = 1;
    uint64_t own_lo = 0, own_hi = 0;
    DevBuf d_shard_segs, d_shard_bounds, d_shard_tmp, d_shard_cand;
    ShardRange shard_r{};
    ShardLayout shard_L{};
    // ts_batch_bind_shard_message: the message buffer the shard's (emitting) scans pack their window records into themselves;
    // msg_windows: the buffer the LATEST scan did so for (null: it wrote 8 x u32 per window, the pack's own kernel packs them)
    void *bound_msg = nullptr;
    const void *msg_windows = nullptr;
    // (ts_batch_pack_shard runs the terminal walks beside the counting / packing kernels on the CONTEXT's side stream: HIP
    // streams share a handful of hardware queues, and kernels of two streams on one queue run one after the other — a side
    // stream per batch, four buffer slots = four more streams, pushed the scan stream onto a shared queue)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // caller-owned result buffers (ts_batch_bind_results / ts_batch_adopt); null = the batch's own
    uint32_t *ext_windows = nullptr, *ext_stats = nullptr;
    const uint32_t *ext_dense = nullptr;
    uint32_t total_waves = 0, region_cap = 0;
    bool all_terminal = false;              // (set with d_readtab) no segment longer than the terminal limit
    bool dealt_tiles = false;               // tiles dealt round-robin instead of taken on demand (see ts_batch_scan)
    uint64_t ticket_seq = 0;                // launches with on-demand tiles so far: parity = the ticket counter in use
    std::vector<uint32_t> wave_fill;
    std::vector<hipEvent_t> evs;                 // ring of {start, stop} pairs, one per enqueued scan
    uint64_t scan_seq = 0, harvested = 0;        // scans enqueued / scans whose time has been read
    uint32_t time_every = 1;                     // ts_batch_set_timing: every n-th scan has a start event too (0: none)
    uint64_t timed_mask = 0;                     // ring slots whose scan was timed (kEventRing = 64 bits)
    double avg_ms = 0.0;
    uint64_t avg_n = 0;

    uint32_t *windows_ptr() const { return ext_windows ? ext_windows : (uint32_t *)d_windows.p; }
    uint32_t *stats_ptr() const { return ext_stats ? ext_stats : (uint32_t *)d_stats.p; }
    bool records16() const { return kp.rec16 != 0u && !dense; }    // the scan's own regions hold 16-bit records (the dense / adopted stream never does)
    const uint32_t *records_ptr() const { return dense ? (ext_dense ? ext_dense : (const uint32_t *)d_dense.p) : (const uint32_t *)d_matches.p; }
    // records that may be read behind records_ptr(): the per-wave regions, or the dense stream
    unsigned long long records_limit() const { return dense ? n_matches : (unsigned long long)region_cap * total_waves; }
    uint64_t range_tiles() const { return tile_hi - tile_lo; }
    bool whole() const { return tile_lo == 0 && tile_hi == tiles.size(); }
};

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return (ctx)->fail(TS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

#define DEVICE_TRY(ctx)                                                                       \
    if ((ctx)->device == kNoDevice)                                                           \
        return (ctx)->fail(TS_ERR_NO_DEVICE, "planning-only context: no HIP device behind it"); \
    DeviceGuard _dev_guard((ctx)->device);                                                    \
    if (_dev_guard.error() != hipSuccess)                                                     \
        return (ctx)->fail(TS_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(_dev_guard.error()))

inline uint64_t ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

// shard.cpp: boundaries of the split of a plan into n_parts consecutive tile ranges (n_parts + 1 values)
void ts_shard_boundaries(const ts_batch *b, uint32_t n_parts, std::vector<uint64_t> &out);

// pipeline.cpp: per-context pieces of the host pipeline, for multi.cpp
int  ts_scan_segments_unlocked(ts_ctx *ctx, const ts_segment_in *segs, size_t n_segs, ts_segment_out *out);   // ts_scan_segments for a caller that holds api_mtx
int  ts_pipeline_ensure_streams(ts_ctx *c);
int  ts_pipeline_upload_batch(ts_batch *b, const ts_segment_in *segs, int *slot, bool used[]);   // segs: the batch's segments, in plan order

// capi.cpp internals used by pipeline.cpp
bool ts_full_scan_supported(const ts_ctx *c, std::string &why);
void *ts_alloc_large(size_t bytes);     // malloc-compatible; many-MB arrays on 2 MB pages when the kernel grants them
int  ts_finalize_segment(ts_ctx *c, bool tips, uint64_t seg_len, uint64_t abs_pos, const uint32_t *win_raw,
                         uint64_t n_windows, ts_match *matches, uint64_t nm, ts_segment_out &o, unsigned spare_threads,
                         const TsDevBlock *pre_blocks = nullptr, size_t n_pre = 0, bool have_pre = false);   // have_pre: blocks called on the device
int  ts_batch_ensure_device(ts_batch *b);        // allocates the range's device state (idempotent)
// block calling on the device over a resident match stream + tile directory (a batch's, or the general kernels' dense stream)
int  ts_device_block_call_raw(ts_ctx *c, const TsTile *d_tiles, const unsigned long long *d_tile_off, const uint32_t *d_stats,
                              const uint32_t *d_matches, uint64_t n_matches_hint, const std::vector<TsShardSegIn> &tab, size_t nt,
                              bool tips, unsigned long long gen_lens, const uint32_t *d_chain, uint32_t *d_work, hipStream_t st,
                              std::vector<TsDevBlock> &blocks, std::vector<unsigned long long> *sums_out, int rec16 = 0,
                              const uint32_t *d_wide_len = nullptr, bool unordered = false);
void ts_batch_release_input(ts_batch *b);        // returns the batch's input buffer to the context's pool
void *ts_batch_input_ptr_nozero(ts_batch *b);
struct ts_fetched;                               // what a download left in host memory, before post-processing
ts_fetched *ts_batch_fetch(ts_batch *b, bool with_matches, int slot, int *rc_out);   // device work + D2H (pinned slot 0/1)
int  ts_batch_finalize(ts_batch *b, ts_fetched *f, ts_segment_out *out);             // host post-processing; frees f
    // the batch's own input buffer, not zero-filled (every byte read is uploaded)

#endif
