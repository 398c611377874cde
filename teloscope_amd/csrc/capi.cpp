// capi.cpp — the C-ABI of libteloscan (include/teloscan.h): contexts, device-resident
// batches, the batched scanSegment / ReadTelomereFilter entry points and their host
// post-processing.  There is no CPU scan path in this library: without a HIP device
// ts_create() fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <limits>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "host.hpp"
#include "ts_internal.h"

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t ensure(size_t need) {
        if (need <= bytes) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; bytes = 0;
        hipError_t e = hipMalloc(&p, need ? need : 16);
        if (e == hipSuccess) bytes = need;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
};

}  // namespace

struct ts_ctx {
    ts_params params{};
    std::vector<ts::Pattern> patterns;
    ts::BlockParams bp{};
    uint32_t k = 0;                 // uniform pattern length (0 = mixed)
    uint32_t longest = 0;
    bool fast_ok = false;           // table-driven tiled kernel usable for the pattern set
    std::string why_not;            // reason when a scan mode is unsupported
    int device = 0;
    int num_cu = 0;
    uint32_t table_rows = 0, fc_bytes = 0;
    bool fc_byte_table = true, pair_byte_table = false;
    // general kernels (generic.hip): sorted 2-bit codes per pattern length
    bool generic_ok = false;
    TsGenericPatterns gpat{};
    DevBuf d_gcodes, d_gflags;
    DevBuf d_table;
    mutable std::mutex mtx;
    mutable std::string error;
    bool read_filter = false;
    // pinned staging ring for host -> device uploads (batch_upload_all)
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t pin_ev[2] = {nullptr, nullptr};
    hipStream_t up_stream = nullptr;

    mutable std::mutex err_mtx;
    std::mutex api_mtx;             // ts_scan_segments / ts_scan_segments_blocks / ts_filter_reads run one at a time per context
                                    // (they share the pinned upload ring; results never depend on call order)
    int fail(int code, const std::string &msg) const { std::lock_guard<std::mutex> g(err_mtx); error = msg; return code; }
};

namespace {

struct Region {                     // one scanned interval of a segment
    uint64_t start, len;            // relative to the segment
    uint32_t first_tile, n_tiles;
    uint64_t tile_bases;            // owned bases per tile
};

struct SegPlan {
    uint64_t len = 0, abs_pos = 0;
    uint64_t in_off = 0;            // byte offset in the device input buffer
    uint64_t win_base = 0, n_windows = 0;
    uint32_t first_tile = 0, n_tiles = 0;
    std::vector<Region> regions;
};

}  // namespace

struct ts_batch {
    ts_ctx *ctx = nullptr;
    bool tips = false;
    std::vector<SegPlan> segs;
    std::vector<TsTile> tiles;
    TsScanParams kp{};
    uint32_t grid = 0, lds_bytes = 0;
    uint64_t total_bases = 0, input_bytes = 0, n_windows = 0, match_cap = 0;
    uint64_t n_matches = 0;
    double last_ms = 0.0;
    bool scanned = false, synced = false;
    const void *last_input = nullptr;
    void *last_stream = nullptr;
    DevBuf d_in, d_tiles, d_windows, d_matches, d_tile_off, d_stats, d_fill, d_segtab, d_dense, d_dense_base;
    uint32_t total_waves = 0, region_cap = 0;
    std::vector<uint32_t> wave_fill;
    std::vector<hipEvent_t> evs;                 // ring of {start, stop} pairs, one per enqueued scan
    uint64_t scan_seq = 0, harvested = 0;        // scans enqueued / scans whose time has been read
    double avg_ms = 0.0;
    uint64_t avg_n = 0;
};

namespace {

#define HIP_TRY(ctx, expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return (ctx)->fail(TS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

uint64_t ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }

constexpr uint32_t kMaxLds = 160u * 1024u;
constexpr uint64_t kEventRing = 64;            // scans whose HIP-event times a batch remembers
constexpr uint32_t kMaxBlocksPerTile = 448;
constexpr uint32_t kMaxChunks = 8;             // tiles up to ~16 k positions per wave
constexpr uint32_t kTipsChunks = 4;            // tips-only / read batches: ~8 k positions per tile

// Chooses waves per workgroup, chunks per tile and windows per tile so that the match table plus
// one LDS slice per wave fit in 160 KB; false if even one wave cannot hold one window.
bool plan_geometry(const ts_ctx *c, bool tips, TsScanParams &kp, uint32_t &wpt, std::string &why) {
    const ts_params &P = c->params;
    const uint32_t k = c->k;
    kp.k = k;
    kp.table_rows = c->table_rows;
    kp.fc_bytes = c->fc_bytes;
    kp.fc_byte_table = c->fc_byte_table ? 1u : 0u;
    kp.pair_byte_table = c->pair_byte_table ? 1u : 0u;
    kp.fold_mask = P.fold_case ? 0xDFDFDFDFu : 0xFFFFFFFFu;
    if (tips) {
        kp.halo_blocks = 0;
        kp.straddle_fix = 0; kp.windows_on = 0; kp.nuc_on = 0; kp.block_sums = 0;
    } else {
        const uint32_t s = P.step, w = P.window_size;
        kp.s = s; kp.w = w;
        kp.s_inv = (65536u + s - 1u) / s;
        kp.halo_blocks = (w + s - 1) / s - 1;            // window i reaches into step blocks i .. i + ceil(w/s) - 1
        kp.straddle_fix = (w == s) ? 1u : 0u;
        kp.windows_on = 1;
        kp.nuc_on = (P.out_gc || P.out_entropy) ? 1u : 0u;
        kp.block_sums = (w % s == 0) ? 1u : 0u;
    }
    // Search (waves per workgroup, chunks per tile) for the best modelled throughput:
    //   owned bases per tile x occupancy factor / instructions per tile.
    // Instructions: ~470 per 2016-position chunk (decode, probes, count planes, per-match pass), ~170 per
    // pass of the window loop (64 window fields per pass), ~100 fixed per tile.  Occupancy factors are
    // measured (profiles/r01/geometry_sweep.txt): against 16 waves per CU the kernel loses 12 % at 12
    // waves and 31 % at 8; more than 16 is not reachable (two workgroups per CU did not co-reside).
    static const struct { uint32_t waves; double factor; } kOccupancy[] = {
        {16, 1.0}, {12, 0.88}, {8, 0.69}, {4, 0.43}, {2, 0.22}, {1, 0.11}};
    uint32_t nch_min = 1;
    if (!tips) nch_min = (uint32_t)ceil_div((uint64_t)(1 + kp.halo_blocks) * kp.s + 63, TS_CHUNK);
    const uint32_t nch_max = tips ? kTipsChunks : std::max(nch_min, kMaxChunks);
    double best = 0.0;
    TsScanParams best_kp{};
    uint32_t best_wpt = 0;
    // TS_GEOMETRY="waves,chunks" pins the search to one point (0 = any): a test knob for the contract that the
    // output does not depend on the tiling ("independent of GPU count and tile size")
    uint32_t pin_waves = 0, pin_nch = 0;
    uint32_t pin_stage = 0;
    if (const char *g = getenv("TS_GEOMETRY")) sscanf(g, "%u,%u,%u", &pin_waves, &pin_nch, &pin_stage);
    for (const auto &occ : kOccupancy) {
        if (pin_waves && occ.waves != pin_waves) continue;
        for (uint32_t nch = nch_min; nch <= nch_max; ++nch) {
            if (nch * TS_CHUNK + 64u > 65535u) break;        // the match queue holds 16-bit plane coordinates
            if (pin_nch && nch != pin_nch) continue;
            TsScanParams cand = kp;
            cand.waves_per_wg = occ.waves;
            cand.nch = nch;
            const uint32_t span_max = nch * TS_CHUNK - 63u;
            uint32_t cwpt;
            if (tips) {
                const uint32_t tb = span_max & ~15u;            // one pseudo block per tile
                cand.s = cand.w = tb;
                cand.s_inv = (65536u + tb - 1u) / tb;
                cand.max_windows = 1;
                cwpt = 1;
            } else {
                const uint32_t nblk = span_max / cand.s;
                cwpt = nblk > cand.halo_blocks ? std::min<uint32_t>(nblk - cand.halo_blocks, kMaxBlocksPerTile) : 0u;
                cand.max_windows = cwpt;
            }
            // staging room for match records: what is left of the CU's LDS, 64 .. 1024 records per wave
            cand.stage_cap = 128;
            cand.acc_copies = 4;             // (8 copies measured no better than 4; the LDS goes to the record stage)
            while (cand.acc_copies > 1 && (uint32_t)ts_k_lds_bytes(&cand) > kMaxLds) cand.acc_copies >>= 1;
            if (cwpt < 1 || (uint32_t)ts_k_lds_bytes(&cand) > kMaxLds) continue;
            {
                const uint32_t spare = (kMaxLds - (uint32_t)ts_k_lds_bytes(&cand)) / occ.waves / 4u;
                cand.stage_cap = std::min<uint32_t>(pin_stage ? pin_stage : 1024u, 128u + (spare & ~15u));
                while ((uint32_t)ts_k_lds_bytes(&cand) > kMaxLds) cand.stage_cap -= 16;
            }
            const double passes = tips ? 0.0 : (double)ceil_div((uint64_t)cwpt * 4, 64);
            // fewer accumulator copies serialise the window adds of a pass: 8 chunks with 2 copies measured 2.5 %
            // slower than 7 chunks with 4 on the headline configuration, where the model alone says 1 % faster
            const double copies = cand.acc_copies >= 4 ? 1.0 : cand.acc_copies == 2 ? 0.96 : 0.92;
            const double score = (double)cwpt * cand.s * occ.factor * copies / (470.0 * nch + 170.0 * passes + 100.0);
            if (score > best) { best = score; best_kp = cand; best_wpt = cwpt; }
        }
    }
    if (best > 0.0) { kp = best_kp; wpt = best_wpt; return true; }
    why = "window/step geometry does not fit the 160 KB LDS of a CU";
    return false;
}

void add_region_tiles(ts_batch *b, SegPlan &sp, uint32_t seg_index, uint64_t start, uint64_t len,
                      uint64_t tile_bases, uint64_t windows_per_tile, bool windows) {
    Region rg{start, len, (uint32_t)b->tiles.size(), 0, tile_bases};
    const uint64_t nt = ceil_div(len, tile_bases);
    for (uint64_t j = 0; j < nt; ++j) {
        TsTile t{};
        const uint64_t u0 = j * tile_bases;
        const uint64_t rem = len - u0;
        t.in_off = sp.in_off + start + u0;
        t.nrel = (uint32_t)std::min<uint64_t>(rem, 1u << 30);
        if (windows) {
            const uint64_t wins_left = sp.n_windows - j * windows_per_tile;
            t.nwin = (uint32_t)std::min<uint64_t>(wins_left, windows_per_tile);
            t.win_out = sp.win_base + j * windows_per_tile;
        } else {
            t.nwin = 1;
            t.win_out = 0;
        }
        t.own_len = (uint32_t)std::min<uint64_t>(rem, tile_bases);
        t.seg = seg_index;
        b->tiles.push_back(t);
    }
    rg.n_tiles = (uint32_t)nt;
    sp.regions.push_back(rg);
}

int batch_alloc_outputs(ts_batch *b) {
    ts_ctx *c = b->ctx;
    const size_t nt = b->tiles.size();
    HIP_TRY(c, b->d_tiles.ensure(nt * sizeof(TsTile)));
    HIP_TRY(c, b->d_windows.ensure(std::max<uint64_t>(b->n_windows, 1) * 32));
    HIP_TRY(c, b->d_matches.ensure((uint64_t)b->region_cap * b->total_waves * 4));
    HIP_TRY(c, b->d_tile_off.ensure((nt + 1) * 8));
    HIP_TRY(c, b->d_stats.ensure((nt + 1) * 16));
    HIP_TRY(c, b->d_fill.ensure((size_t)b->total_waves * 4 + 16));
    if (nt) HIP_TRY(c, hipMemcpy(b->d_tiles.p, b->tiles.data(), nt * sizeof(TsTile), hipMemcpyHostToDevice));
    return TS_OK;
}

}  // namespace

// =========================================================================== misc entry points
extern "C" {

int ts_abi_version(void) { return TELOSCAN_ABI_VERSION; }

const char *ts_last_error(const ts_ctx *ctx) {
    return ctx ? ctx->error.c_str() : g_create_error.c_str();
}

int ts_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ts_canonical_orientation(const char *canonical_in, char *fwd_out, char *rev_out) {
    if (!canonical_in || !fwd_out || !rev_out || std::strlen(canonical_in) > 62) return TS_ERR_INVALID_ARG;
    std::string f, r;
    ts::canonical_orientation(canonical_in, f, r);
    std::strcpy(fwd_out, f.c_str());
    std::strcpy(rev_out, r.c_str());
    return TS_OK;
}

int ts_expand_patterns(const char *raw_csv, int edit_distance, const char *canonical_fwd,
                       ts_pattern **out, size_t *n_out) {
    if (!raw_csv || !canonical_fwd || !out || !n_out || edit_distance < 0 || edit_distance > 2)
        return TS_ERR_INVALID_ARG;
    std::vector<ts::Pattern> v = ts::expand_patterns(raw_csv, edit_distance, canonical_fwd);
    ts_pattern *arr = (ts_pattern *)std::calloc(v.size() ? v.size() : 1, sizeof(ts_pattern));
    if (!arr) return TS_ERR_ALLOC;
    for (size_t i = 0; i < v.size(); ++i) {
        std::strncpy(arr[i].seq, v[i].seq.c_str(), 63);
        arr[i].len = (uint8_t)v[i].seq.size();
        arr[i].is_forward = v[i].is_forward;
        arr[i].is_canonical = v[i].is_canonical;
    }
    *out = arr;
    *n_out = v.size();
    return TS_OK;
}

void ts_free_patterns(ts_pattern *p) { std::free(p); }

float ts_gc_content(const uint32_t counts[4], uint32_t window_size) { return ts::gc_content(counts, window_size); }
float ts_shannon_entropy(const uint32_t counts[4], uint32_t window_size) { return ts::shannon_entropy(counts, window_size); }

int ts_label_terminal_blocks(ts_block *blocks, size_t n, uint16_t gaps, uint64_t path_size,
                             uint32_t terminal_limit, char *label_out, int *scaffold_type_out) {
    if ((n && !blocks) || !label_out || !scaffold_type_out) return TS_ERR_INVALID_ARG;
    std::string label;
    *scaffold_type_out = ts::label_terminal_blocks(blocks, n, gaps, path_size, terminal_limit, label);
    std::memcpy(label_out, label.c_str(), label.size() + 1);
    return TS_OK;
}

// =========================================================================== context
static ts_ctx *create_impl(const ts_params *params, const ts_pattern *patterns, size_t n_patterns,
                           bool read_filter, int min_block_len_set) {
    g_create_error.clear();
    if (!params || params->struct_size != sizeof(ts_params) || (n_patterns && !patterns)) {
        g_create_error = "ts_create: bad arguments (struct_size mismatch?)";
        return nullptr;
    }
    if (params->step == 0 || params->window_size == 0 || params->step > params->window_size) {
        g_create_error = "ts_create: need 0 < step <= window_size";
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_error = "ts_create: no usable HIP device (libteloscan has no CPU fallback)";
        return nullptr;
    }
    ts_ctx *c = new ts_ctx();
    c->params = *params;
    c->read_filter = read_filter;
    if (read_filter) {                                   // makeReadFilterInput, src/read-filter.cpp:10-30
        if (!min_block_len_set) c->params.min_block_len = 42;
        c->params.terminal_limit = std::numeric_limits<uint32_t>::max() / 2;
        c->params.out_gc = c->params.out_entropy = c->params.out_matches = c->params.out_its = 0;
        c->params.fold_case = 1;
    }
    int dev = params->device;
    if (dev < 0) { if (hipGetDevice(&dev) != hipSuccess) dev = 0; }
    if (dev >= ndev || hipSetDevice(dev) != hipSuccess) {
        g_create_error = "ts_create: cannot select HIP device";
        delete c;
        return nullptr;
    }
    c->device = dev;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) c->num_cu = prop.multiProcessorCount;
    if (c->num_cu <= 0) c->num_cu = 256;

    uint32_t kmin = 0xFFFFFFFFu, kmax = 0;
    for (size_t i = 0; i < n_patterns; ++i) {
        ts::Pattern p;
        p.seq.assign(patterns[i].seq, patterns[i].len);
        p.is_forward = patterns[i].is_forward;
        p.is_canonical = patterns[i].is_canonical;
        kmin = std::min<uint32_t>(kmin, patterns[i].len);
        kmax = std::max<uint32_t>(kmax, patterns[i].len);
        c->patterns.push_back(std::move(p));
    }
    c->longest = kmax;
    c->bp.terminal_limit = c->params.terminal_limit;
    c->bp.max_match_dist = c->params.max_match_dist;
    c->bp.min_block_len = c->params.min_block_len;
    c->bp.max_block_dist = c->params.max_block_dist;
    c->bp.min_block_counts = c->params.min_block_counts;
    c->bp.min_block_density = c->params.min_block_density;
    c->bp.first_pattern_len = n_patterns ? patterns[0].len : 0;

    // tables of the general kernels: usable for any set with <= 8 distinct lengths <= 32
    if (n_patterns) {
        std::vector<uint32_t> lens;
        for (const ts::Pattern &p : c->patterns) lens.push_back((uint32_t)p.seq.size());
        std::sort(lens.begin(), lens.end());
        lens.erase(std::unique(lens.begin(), lens.end()), lens.end());
        bool ok = lens.size() <= 8 && lens.back() <= 32 && lens.front() >= 1;
        std::vector<unsigned long long> codes;
        std::vector<uint8_t> flags;
        if (ok) {
            c->gpat.nlen = (uint32_t)lens.size();
            for (size_t li = 0; li < lens.size() && ok; ++li) {
                c->gpat.len[li] = lens[li];
                c->gpat.first[li] = (uint32_t)codes.size();
                std::vector<std::pair<unsigned long long, uint8_t>> v;
                for (const ts::Pattern &p : c->patterns) {
                    if (p.seq.size() != lens[li]) continue;
                    unsigned long long code = 0;
                    for (size_t i = 0; i < p.seq.size(); ++i) {
                        const int b = ts::base_code(p.seq[i]);
                        if (b < 0) { ok = false; break; }
                        code |= (unsigned long long)b << (2 * i);
                    }
                    v.emplace_back(code, (uint8_t)((p.is_forward ? 1 : 0) | (p.is_canonical ? 2 : 0)));
                }
                std::sort(v.begin(), v.end());
                for (const auto &e : v) { codes.push_back(e.first); flags.push_back(e.second); }
            }
            c->gpat.first[lens.size()] = (uint32_t)codes.size();
        }
        if (ok && c->d_gcodes.ensure(codes.size() * 8) == hipSuccess && c->d_gflags.ensure(flags.size() + 16) == hipSuccess &&
            hipMemcpy(c->d_gcodes.p, codes.data(), codes.size() * 8, hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(c->d_gflags.p, flags.data(), flags.size(), hipMemcpyHostToDevice) == hipSuccess) {
            c->gpat.codes = (const unsigned long long *)c->d_gcodes.p;
            c->gpat.flags = (const uint8_t *)c->d_gflags.p;
            c->generic_ok = true;
        }
    }

    if (n_patterns == 0) {
        c->why_not = "empty pattern set";
    } else if (kmin != kmax) {
        c->why_not = "mixed-length pattern set";
    } else {
        std::vector<uint32_t> table;
        c->k = kmin;
        bool built = false;
        if (kmin == 7) {
            // k = 7: the byte pair table (64 KB) makes every probe a single v_bfe, but leaves the wave slices
            // less LDS: it pays when tiles of >= 5 chunks still fit 16 waves (measured on -w 2000 -s 1000:
            // 5 chunks with bytes 0.836 ms, 7 chunks with the 2-bit table 0.863 ms, 4 chunks with bytes 0.892 ms);
            // read filters scan in tips mode, whose tiles are small either way
            built = ts::build_match_table(c->patterns, kmin, 7, table, c->table_rows, c->fc_bytes, c->fc_byte_table, c->pair_byte_table);
            if (built) {
                TsScanParams kp{};
                uint32_t wpt = 0;
                std::string why;
                // (windows the tiled kernel does not take — see full_scan_supported — go to the general kernels)
                const uint32_t s_ = c->params.step, w_ = c->params.window_size, ov = w_ - s_;
                const bool win_tiled = !read_filter && kmin <= w_ && (ov == 0 || kmin <= std::min(s_, ov)) && w_ <= 32768u;
                const bool tips = !win_tiled;
                const bool fits = plan_geometry(c, tips, kp, wpt, why) && kp.waves_per_wg == 16u && (tips || kp.nch >= 5u);
                if (!fits) built = false;
            }
        }
        if (!built)
            built = ts::build_match_table(c->patterns, kmin, 6, table, c->table_rows, c->fc_bytes, c->fc_byte_table, c->pair_byte_table);
        if (!built) {
            c->k = 0;
            c->why_not = "pattern length outside 3..8 or non-ACGT pattern";
        } else {
            if (c->d_table.ensure(table.size() * 4) != hipSuccess ||
                hipMemcpy(c->d_table.p, table.data(), table.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
                g_create_error = "ts_create: cannot upload match table";
                delete c;
                return nullptr;
            }
            c->fast_ok = true;
        }
    }
    return c;
}

ts_ctx *ts_create(const ts_params *params, const ts_pattern *patterns, size_t n_patterns) {
    return create_impl(params, patterns, n_patterns, false, 1);
}

ts_ctx *ts_create_read_filter(const ts_params *params, int min_block_len_set,
                              const ts_pattern *patterns, size_t n_patterns) {
    return create_impl(params, patterns, n_patterns, true, min_block_len_set);
}

void ts_destroy(ts_ctx *ctx) {
    if (!ctx) return;
    ctx->d_table.release();
    ctx->d_gcodes.release();
    ctx->d_gflags.release();
    for (int i = 0; i < 2; ++i) {
        if (ctx->pin[i]) (void)hipHostFree(ctx->pin[i]);
        if (ctx->pin_ev[i]) (void)hipEventDestroy(ctx->pin_ev[i]);
    }
    if (ctx->up_stream) (void)hipStreamDestroy(ctx->up_stream);
    delete ctx;
}

// The tiled kernel implements the closed form of the reference's carry loop, valid when
// w == s or longest <= min(s, w - s); outside it the reference's uint32
// arithmetic wraps and only a literal emulation reproduces it.
static bool full_scan_supported(const ts_ctx *c, std::string &why) {
    if (!c->fast_ok) { why = c->why_not; return false; }
    const uint32_t s = c->params.step, w = c->params.window_size, ov = w - s, L = c->longest;
    if (L > w) { why = "pattern longer than window"; return false; }
    if (ov != 0 && L > std::min(s, ov)) {
        why = "longest pattern exceeds min(step, window-step): reference start-index arithmetic wraps";
        return false;
    }
    // the tiled kernel accumulates a window's match counts in 16-bit lanes of one packed LDS word, and a
    // tile (one wave's LDS slice) has to hold a whole window: larger windows take the general kernels
    if (w > 32768u) { why = "window larger than 32768 bases"; return false; }
    return true;
}

int ts_uses_fast_path(const ts_ctx *ctx) {
    std::string why;
    return ctx && full_scan_supported(ctx, why) ? 1 : 0;
}

// =========================================================================== batches
ts_batch *ts_batch_create(ts_ctx *ctx, const uint64_t *seg_lens, const uint64_t *abs_pos,
                          size_t n_segs, int tips_only, uint64_t match_capacity) {
    if (!ctx) return nullptr;
    std::lock_guard<std::mutex> lk(ctx->mtx);
    if (n_segs && !seg_lens) { ctx->fail(TS_ERR_INVALID_ARG, "ts_batch_create: null seg_lens"); return nullptr; }
    std::string why;
    if (tips_only ? !ctx->fast_ok : !full_scan_supported(ctx, why)) {
        ctx->fail(TS_ERR_UNSUPPORTED, "unsupported parameter set: " + (tips_only ? ctx->why_not : why));
        return nullptr;
    }
    if (hipSetDevice(ctx->device) != hipSuccess) { ctx->fail(TS_ERR_HIP, "hipSetDevice failed"); return nullptr; }

    ts_batch *b = new ts_batch();
    b->ctx = ctx;
    b->tips = tips_only != 0;
    uint32_t wpt = 1;
    if (!plan_geometry(ctx, b->tips, b->kp, wpt, why)) {
        ctx->fail(TS_ERR_UNSUPPORTED, "unsupported parameter set: " + why);
        delete b;
        return nullptr;
    }
    if (getenv("TS_TIMING"))
        fprintf(stderr, "ts_batch_create: %s scan, k=%u (%s pair table), %u waves per workgroup, %u chunks and %u windows per tile, LDS %d B "
                        "(match queue %u, record stage %u, %u accumulator copies)\n", b->tips ? "tips-only" : "window", ctx->k,
                ctx->pair_byte_table ? "byte" : "2-bit",
                b->kp.waves_per_wg, b->kp.nch, wpt, ts_k_lds_bytes(&b->kp), (unsigned)TS_LIST, b->kp.stage_cap, b->kp.acc_copies);
    const uint64_t tile_bases = (uint64_t)wpt * b->kp.s;
    const uint32_t tl = ctx->params.terminal_limit;

    uint64_t off = 0, wins = 0;
    b->segs.resize(n_segs);
    for (size_t i = 0; i < n_segs; ++i) {
        SegPlan &sp = b->segs[i];
        sp.len = seg_lens[i];
        sp.abs_pos = abs_pos ? abs_pos[i] : 0;
        sp.in_off = off;
        off += (sp.len + 15) & ~15ull;
        sp.first_tile = (uint32_t)b->tiles.size();
        if (b->tips) {
            const uint32_t twice = 2u * tl;             // uint32 product, as src/teloscope.cpp:576
            if (sp.len > twice) {
                add_region_tiles(b, sp, (uint32_t)i, 0, tl, tile_bases, 1, false);
                add_region_tiles(b, sp, (uint32_t)i, sp.len - tl, tl, tile_bases, 1, false);
            } else if (sp.len > 0) {
                add_region_tiles(b, sp, (uint32_t)i, 0, sp.len, tile_bases, 1, false);
            }
        } else {
            sp.win_base = wins;
            sp.n_windows = ceil_div(sp.len, ctx->params.step);
            wins += sp.n_windows;
            if (sp.len > 0) add_region_tiles(b, sp, (uint32_t)i, 0, sp.len, tile_bases, wpt, true);
        }
        sp.n_tiles = (uint32_t)b->tiles.size() - sp.first_tile;
        b->total_bases += sp.len;
    }
    b->input_bytes = off + TS_IN_PAD;
    b->n_windows = wins;
    b->match_cap = match_capacity ? match_capacity : b->total_bases / 4 + 4096;
    if (b->tiles.size() >= 0x7FFFFFFFull) {
        ctx->fail(TS_ERR_UNSUPPORTED, "too many tiles in one batch");
        delete b;
        return nullptr;
    }

    b->lds_bytes = (uint32_t)ts_k_lds_bytes(&b->kp);
    if (ts_k_prepare(b->lds_bytes) != 0) {
        ctx->fail(TS_ERR_HIP, "cannot raise dynamic LDS limit");
        delete b;
        return nullptr;
    }
    // one persistent workgroup per CU; tiles are dealt round-robin to its waves
    const uint32_t wpw = b->kp.waves_per_wg;
    b->grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(b->tiles.size(), wpw), 1), (uint64_t)ctx->num_cu);
    b->total_waves = b->grid * wpw;
    // every wave appends to its own region of the match buffer; small batches get the worst case
    // (every base a match), large ones bases/4 spread evenly, grown on overflow by ts_batch_sync
    // (worst case for wave w = the owned bases of the tiles it is dealt: t = w, w + waves, ...)
    uint64_t worst = 256;
    {
        std::vector<uint64_t> per_wave(b->total_waves, 0);
        for (size_t t = 0; t < b->tiles.size(); ++t) per_wave[t % b->total_waves] += b->tiles[t].own_len;
        for (uint64_t v : per_wave) worst = std::max(worst, v);
    }
    uint64_t cap = ceil_div(b->match_cap, b->total_waves);
    if (b->total_bases <= (64ull << 20) && !match_capacity) cap = worst;
    cap = std::min<uint64_t>(std::max<uint64_t>(cap, 256), worst);
    b->region_cap = (uint32_t)((cap + 3) & ~3ull);
    b->match_cap = (uint64_t)b->region_cap * b->total_waves;

    bool ev_ok = true;
    b->evs.assign(2 * kEventRing, nullptr);
    for (hipEvent_t &e : b->evs) ev_ok = ev_ok && hipEventCreate(&e) == hipSuccess;
    if (batch_alloc_outputs(b) != TS_OK || !ev_ok) {
        if (ctx->error.empty()) ctx->fail(TS_ERR_HIP, "batch allocation failed");
        ts_batch_destroy(b);
        return nullptr;
    }
    return b;
}

void ts_batch_destroy(ts_batch *b) {
    if (!b) return;
    b->d_in.release(); b->d_tiles.release(); b->d_windows.release(); b->d_matches.release();
    b->d_tile_off.release(); b->d_stats.release(); b->d_fill.release(); b->d_dense.release(); b->d_dense_base.release();
    b->d_segtab.release();
    for (hipEvent_t e : b->evs)
        if (e) (void)hipEventDestroy(e);
    delete b;
}

uint64_t ts_batch_segment_offset(const ts_batch *b, size_t i) {
    return (b && i < b->segs.size()) ? b->segs[i].in_off : 0;
}

// The kernels never look past a segment's end (the padding between segments is masked by position), so
// zero-filling only serves callers that fill the buffer piecewise and later inspect it.
static void *batch_input(ts_batch *b, bool zero_fill) {
    if (!b) return nullptr;
    if (!b->d_in.p) {
        if (b->d_in.ensure(b->input_bytes) != hipSuccess) return nullptr;
        if (zero_fill) (void)hipMemset(b->d_in.p, 0, b->input_bytes);
    }
    return b->d_in.p;
}

void *ts_batch_input_ptr(ts_batch *b) { return batch_input(b, true); }

int ts_batch_upload(ts_batch *b, size_t i, const char *seq) {
    if (!b || i >= b->segs.size() || (!seq && b->segs[i].len)) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    if (!ts_batch_input_ptr(b)) return c->fail(TS_ERR_ALLOC, "cannot allocate device input buffer");
    if (b->segs[i].len)
        HIP_TRY(c, hipMemcpy((char *)b->d_in.p + b->segs[i].in_off, seq, b->segs[i].len, hipMemcpyHostToDevice));
    return TS_OK;
}

int ts_batch_scan(ts_batch *b, const void *d_input, void *stream) {
    if (!b) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    std::lock_guard<std::mutex> lk(c->mtx);
    HIP_TRY(c, hipSetDevice(c->device));
    if (!d_input) d_input = ts_batch_input_ptr(b);
    if (!d_input) return c->fail(TS_ERR_ALLOC, "no device input buffer");
    hipStream_t st = (hipStream_t)stream;
    const size_t nt = b->tiles.size();
    b->last_input = d_input;
    b->last_stream = stream;
    b->scanned = true;
    b->synced = false;

    TsScanParams &kp = b->kp;
    kp.in = (const uint8_t *)d_input;
    kp.tiles = (const TsTile *)b->d_tiles.p;
    kp.table = (const uint32_t *)c->d_table.p;
    kp.windows_out = (uint32_t *)b->d_windows.p;
    kp.matches_out = (uint32_t *)b->d_matches.p;
    kp.tile_off = (unsigned long long *)b->d_tile_off.p;
    kp.tile_stats = (uint32_t *)b->d_stats.p;
    kp.wave_fill = (uint32_t *)b->d_fill.p;
    kp.region_cap = b->region_cap;
    kp.ntiles = (uint32_t)nt;

    const size_t slot = (size_t)(b->scan_seq % kEventRing);
    HIP_TRY(c, hipEventRecord(b->evs[2 * slot], st));
    {
        int e = ts_k_launch_scan(&kp, b->grid, b->lds_bytes, stream);
        if (e != 0) return c->fail(TS_ERR_HIP, std::string("scan kernel launch: ") + hipGetErrorString((hipError_t)e));
    }
    HIP_TRY(c, hipEventRecord(b->evs[2 * slot + 1], st));
    ++b->scan_seq;
    return TS_OK;
}

int ts_batch_sync(ts_batch *b) {
    if (!b) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    if (!b->scanned) return c->fail(TS_ERR_STATE, "ts_batch_sync before ts_batch_scan");
    HIP_TRY(c, hipSetDevice(c->device));
    for (int attempt = 0; attempt < 3; ++attempt) {
        {   // kernel times of the scans since the previous sync (the ring keeps the latest kEventRing)
            const size_t last = (size_t)((b->scan_seq - 1) % kEventRing);
            HIP_TRY(c, hipEventSynchronize(b->evs[2 * last + 1]));
            const uint64_t from = std::max(b->harvested, b->scan_seq > kEventRing ? b->scan_seq - kEventRing : 0);
            double sum = 0.0;
            float ms = 0.f;
            for (uint64_t q = from; q < b->scan_seq; ++q) {
                const size_t s2 = (size_t)(q % kEventRing);
                HIP_TRY(c, hipEventElapsedTime(&ms, b->evs[2 * s2], b->evs[2 * s2 + 1]));
                sum += ms;
            }
            if (b->scan_seq > from) {
                b->avg_n = b->scan_seq - from;
                b->avg_ms = sum / (double)b->avg_n;
                b->last_ms = ms;
            }
            b->harvested = b->scan_seq;
        }
        b->wave_fill.assign(b->total_waves, 0u);
        HIP_TRY(c, hipMemcpy(b->wave_fill.data(), b->d_fill.p, (size_t)b->total_waves * 4, hipMemcpyDeviceToHost));
        uint64_t total = 0;
        uint32_t worst = 0;
        for (uint32_t f : b->wave_fill) { total += f; worst = std::max(worst, f); }
        b->n_matches = total;
        if (worst <= b->region_cap) { b->synced = true; return TS_OK; }
        // a wave's region overflowed: size the regions for the fullest wave and rescan
        b->region_cap = (uint32_t)(((uint64_t)worst + worst / 8 + 64 + 3) & ~3ull);
        b->match_cap = (uint64_t)b->region_cap * b->total_waves;
        HIP_TRY(c, b->d_matches.ensure(b->match_cap * 4));
        int rc = ts_batch_scan(b, b->last_input, b->last_stream);
        if (rc != TS_OK) return rc;
    }
    return c->fail(TS_ERR_STATE, "match buffer kept overflowing");
}

int ts_batch_get_info(const ts_batch *b, ts_batch_info *info) {
    if (!b || !info) return TS_ERR_INVALID_ARG;
    info->n_segments = b->segs.size();
    info->total_bases = b->total_bases;
    info->input_bytes = b->input_bytes;
    info->n_windows = b->n_windows;
    info->n_tiles = b->tiles.size();
    info->match_capacity = b->match_cap;
    info->n_matches = b->synced ? b->n_matches : 0;
    info->algorithmic_bytes = b->total_bases + 32ull * b->n_windows + 4ull * info->n_matches;
    info->last_kernel_ms = b->last_ms;
    info->avg_kernel_ms = b->avg_ms;
    info->kernel_launches = b->avg_n;
    return TS_OK;
}

const void *ts_batch_windows_ptr(const ts_batch *b) { return b ? b->d_windows.p : nullptr; }
const void *ts_batch_matches_ptr(const ts_batch *b) { return b ? b->d_matches.p : nullptr; }

int ts_batch_segment_summary(ts_batch *b, void *d_out, void *stream) {
    if (!b || !d_out) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    const size_t ns = b->segs.size();
    if (!b->d_segtab.p) {
        std::vector<uint32_t> first(ns + 1);
        std::vector<uint64_t> nwin(ns);
        for (size_t i = 0; i < ns; ++i) { first[i] = b->segs[i].first_tile; nwin[i] = b->segs[i].n_windows; }
        first[ns] = (uint32_t)b->tiles.size();
        const size_t bytes_first = ((ns + 1) * 4 + 15) & ~15ull;
        HIP_TRY(c, b->d_segtab.ensure(bytes_first + ns * 8 + 16));
        HIP_TRY(c, hipMemcpy(b->d_segtab.p, first.data(), (ns + 1) * 4, hipMemcpyHostToDevice));
        if (ns) HIP_TRY(c, hipMemcpy((char *)b->d_segtab.p + bytes_first, nwin.data(), ns * 8, hipMemcpyHostToDevice));
    }
    const size_t bytes_first = ((ns + 1) * 4 + 15) & ~15ull;
    int e = ts_k_launch_summary((const uint32_t *)b->d_stats.p, (const uint32_t *)b->d_segtab.p,
                                (const uint64_t *)((char *)b->d_segtab.p + bytes_first), (uint32_t)ns,
                                (unsigned long long *)d_out, stream);
    if (e != 0) return c->fail(TS_ERR_HIP, "summary kernel launch failed");
    return TS_OK;
}

// --------------------------------------------------------------- shared host post-processing
// Turns one segment's raw results into SegmentData: window records (float metrics evaluated on
// the host from the integer counts, as the reference does), terminal flags, block calling
// (src/teloscope.cpp:642-657).  `matches` arrive with absolute positions and FORWARD/CANONICAL set.
// `matches` (malloc'd by the caller, position-ordered, terminal flags not yet set; may be null when nm == 0)
// becomes the segment's match array.
static int finalize_segment(ts_ctx *c, bool tips, uint64_t seg_len, uint64_t abs_pos,
                            const uint32_t *win_raw, uint64_t n_windows, ts_match *matches, uint64_t nm,
                            ts_segment_out &o, unsigned spare_threads) {
    const ts_params &P = c->params;
    std::memset(&o, 0, sizeof o);
    if (!tips && n_windows) {
        o.windows = (ts_window *)std::calloc(n_windows, sizeof(ts_window));
        if (!o.windows) return c->fail(TS_ERR_ALLOC, "out of host memory");
        o.n_windows = n_windows;
        const bool nuc = P.out_gc || P.out_entropy;
        // windows are independent: a segment with very many of them (a multi-gigabase contig) is split over the
        // host threads its job can spare
        auto convert = [&](uint64_t k0, uint64_t k1) {
        for (uint64_t kwin = k0; kwin < k1; ++kwin) {
            const uint32_t *r = &win_raw[kwin * 8];
            ts_window &w = o.windows[kwin];
            const uint64_t ws = kwin * P.step;
            w.window_start = abs_pos + ws;
            w.current_window_size = (uint32_t)std::min<uint64_t>(P.window_size, seg_len - ws);
            if (nuc) for (int i = 0; i < 4; ++i) w.nucleotide_counts[i] = r[i];
            if (P.out_gc) w.gc_content = ts::gc_content(w.nucleotide_counts, w.current_window_size);
            if (P.out_entropy) w.shannon_entropy = ts::shannon_entropy(w.nucleotide_counts, w.current_window_size);
            w.canonical_covered = r[4];
            w.non_canonical_covered = r[5];
            w.fwd_covered = r[6];
            w.rev_covered = r[7];
        }
        };
        const unsigned nth = n_windows >= (1u << 18) ? std::min<unsigned>(spare_threads, (unsigned)(n_windows >> 16)) : 1u;
        if (nth <= 1u) {
            convert(0, n_windows);
        } else {
            std::vector<std::thread> pool;
            const uint64_t share = (n_windows + nth - 1) / nth;
            for (unsigned t = 0; t < nth; ++t)
                pool.emplace_back(convert, std::min<uint64_t>(n_windows, t * share), std::min<uint64_t>(n_windows, (t + 1) * share));
            for (std::thread &th : pool) th.join();
        }
    }
    if (nm > 0xFFFFFFFFull) { std::free(matches); return c->fail(TS_ERR_UNSUPPORTED, "more than 2^32 matches in one segment"); }
    o.matches = nm ? matches : nullptr;
    o.n_matches = nm;
    if (!nm) std::free(matches);
    const uint64_t term_end = seg_len > P.terminal_limit ? seg_len - P.terminal_limit : 0;
    uint64_t nfwd = 0;
    for (uint64_t i = 0; i < nm; ++i) {                             // isTerminal, src/teloscope.cpp:451-459
        ts_match &m = o.matches[i];
        const uint64_t rel = m.position - abs_pos;
        if (rel <= P.terminal_limit || rel >= term_end) m.flags |= TS_MATCH_TERMINAL;
        nfwd += (m.flags & TS_MATCH_FORWARD) ? 1u : 0u;
    }
    // the two walks take their orientation's records out of the one position-ordered array (they leave the
    // terminal zone after a few thousand records: no per-orientation index lists of the whole segment)
    std::vector<ts_block> term, its;
    uint64_t fwd_boundary = abs_pos, rev_boundary = abs_pos + seg_len;
    if (nfwd >= 2)
        fwd_boundary = ts::terminal_blocks(c->bp, o.matches, nullptr, nm, term, seg_len, abs_pos, true, 1);
    if (nm - nfwd >= 2)
        rev_boundary = ts::terminal_blocks(c->bp, o.matches, nullptr, nm, term, seg_len, abs_pos, false, 0);
    if (!tips && fwd_boundary < rev_boundary && nm >= 2)
        ts::interstitial_blocks(c->bp, o.matches, nm, its, fwd_boundary, rev_boundary);
    auto copy_blocks = [&](const std::vector<ts_block> &v, ts_block *&dst, uint64_t &n) -> bool {
        n = v.size();
        dst = nullptr;
        if (v.empty()) return true;
        dst = (ts_block *)std::malloc(v.size() * sizeof(ts_block));
        if (!dst) return false;
        std::memcpy(dst, v.data(), v.size() * sizeof(ts_block));
        return true;
    };
    if (!copy_blocks(term, o.terminal_blocks, o.n_terminal_blocks) ||
        !copy_blocks(its, o.interstitial_blocks, o.n_interstitial_blocks))
        return c->fail(TS_ERR_ALLOC, "out of host memory");
    return TS_OK;
}

// --------------------------------------------------------------- download + host post-processing
int ts_batch_download(ts_batch *b, const char *const *host_seqs, ts_segment_out *out) {
    (void)host_seqs;
    if (!b || !out) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    if (!b->synced) { int rc = ts_batch_sync(b); if (rc != TS_OK) return rc; }
    const size_t nt = b->tiles.size(), ns = b->segs.size();

    // (plain arrays: a vector would zero-fill hundreds of MB that the copies below overwrite)
    const std::unique_ptr<uint32_t[]> wins_buf(new uint32_t[b->n_windows * 8 + 1]), recs_buf(new uint32_t[b->n_matches + 1]);
    uint32_t *const wins = wins_buf.get(), *const recs = recs_buf.get();
    const uint64_t nrecs = b->n_matches;
    // tile directory: records of tile t are recs[tile_off[t] .. +tile_stats[4t]), in position order
    std::vector<unsigned long long> tile_off(nt + 1);
    std::vector<uint32_t> tile_stats(4 * (nt + 1));
    if (b->n_windows) HIP_TRY(c, hipMemcpy(wins, b->d_windows.p, b->n_windows * 32, hipMemcpyDeviceToHost));
    // pack the per-wave regions into one dense stream on the device, then one D2H copy
    std::vector<unsigned long long> dense_base(b->total_waves + 1, 0);
    for (uint32_t w = 0; w < b->total_waves; ++w) dense_base[w + 1] = dense_base[w] + b->wave_fill[w];
    if (b->n_matches) {
        HIP_TRY(c, b->d_dense.ensure(b->n_matches * 4));
        HIP_TRY(c, b->d_dense_base.ensure((size_t)(b->total_waves + 1) * 8));
        HIP_TRY(c, hipMemcpy(b->d_dense_base.p, dense_base.data(), (size_t)(b->total_waves + 1) * 8, hipMemcpyHostToDevice));
        int e = ts_k_launch_compact((const uint32_t *)b->d_matches.p, (const uint32_t *)b->d_fill.p,
                                    (const unsigned long long *)b->d_dense_base.p, b->region_cap,
                                    b->total_waves, (uint32_t *)b->d_dense.p, nullptr);
        if (e != 0) return c->fail(TS_ERR_HIP, "compaction kernel launch failed");
        HIP_TRY(c, hipMemcpy(recs, b->d_dense.p, b->n_matches * 4, hipMemcpyDeviceToHost));
    }
    if (nt) HIP_TRY(c, hipMemcpy(tile_off.data(), b->d_tile_off.p, nt * 8, hipMemcpyDeviceToHost));
    if (nt) HIP_TRY(c, hipMemcpy(tile_stats.data(), b->d_stats.p, nt * 16, hipMemcpyDeviceToHost));
    // region offset -> dense offset (tile t was scanned by wave t mod total_waves)
    for (size_t t = 0; t < nt; ++t) {
        const uint32_t w = (uint32_t)(t % b->total_waves);
        tile_off[t] = tile_off[t] - (unsigned long long)w * b->region_cap + dense_base[w];
    }

    // Host post-processing (absolute positions, terminal flags, block calling, float window metrics)
    // is O(matches) per segment and independent between segments: one job per segment, largest
    // first, on up to 16 host threads — the reference's own decomposition (one job per path).
    const uint16_t klen = (uint16_t)c->k;
    std::vector<size_t> order(ns);
    for (size_t i = 0; i < ns; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return b->segs[x].len > b->segs[y].len; });
    std::atomic<size_t> next{0};
    std::atomic<int> first_err{TS_OK};
    const unsigned hw_threads = std::max(1u, std::thread::hardware_concurrency());
    const unsigned spare = std::max(1u, std::min(16u, hw_threads) / (unsigned)std::max<size_t>(1, std::min<size_t>(ns, 16)));
    auto worker = [&]() {
        for (;;) {
            const size_t oi = next.fetch_add(1);
            if (oi >= ns || first_err.load() != TS_OK) return;
            const size_t si = order[oi];
            const SegPlan &sp = b->segs[si];
            // matches: tile-relative packed records -> absolute MatchInfo, tiles in position order
            uint64_t nm = 0;
            for (uint32_t t = 0; t < sp.n_tiles; ++t) nm += tile_stats[4ull * (sp.first_tile + t)];
            ts_match *arr = nm ? (ts_match *)std::malloc(nm * sizeof(ts_match)) : nullptr;
            if (nm && !arr) { int expected = TS_OK; first_err.compare_exchange_strong(expected, c->fail(TS_ERR_ALLOC, "out of host memory")); return; }
            uint64_t at = 0;
            int rc = TS_OK;
            for (const Region &rg : sp.regions) {
                for (uint32_t t = 0; t < rg.n_tiles && rc == TS_OK; ++t) {
                    const uint32_t ti = rg.first_tile + t;
                    const uint64_t tile_rel = rg.start + (uint64_t)t * rg.tile_bases;   // segment-relative
                    const uint64_t r0 = tile_off[ti], r1 = r0 + tile_stats[4ull * ti];
                    if (r1 > nrecs || at + (r1 - r0) > nm) { rc = c->fail(TS_ERR_STATE, "tile directory out of range"); break; }
                    const uint64_t pos0 = sp.abs_pos + tile_rel;
                    for (uint64_t ri = r0; ri < r1; ++ri) {
                        const uint32_t rec = recs[ri];
                        ts_match &m = arr[at++];
                        std::memset(&m, 0, sizeof m);
                        m.position = pos0 + (rec >> 2);
                        m.match_size = klen;
                        m.flags = (uint8_t)(((rec & 2u) ? TS_MATCH_FORWARD : 0u) | ((rec & 1u) ? TS_MATCH_CANONICAL : 0u));
                    }
                }
            }
            if (rc != TS_OK) std::free(arr);
            if (rc == TS_OK)
                rc = finalize_segment(c, b->tips, sp.len, sp.abs_pos,
                                      sp.n_windows ? &wins[sp.win_base * 8] : nullptr, b->tips ? 0 : sp.n_windows,
                                      arr, nm, out[si], spare);
            if (rc != TS_OK) { int expected = TS_OK; first_err.compare_exchange_strong(expected, rc); return; }
        }
    };
    for (size_t i = 0; i < ns; ++i) std::memset(&out[i], 0, sizeof out[i]);
    const unsigned nthreads = (unsigned)std::min<size_t>({(size_t)16, ns, (size_t)std::max(1u, std::thread::hardware_concurrency())});
    if (nthreads <= 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (unsigned i = 0; i < nthreads; ++i) pool.emplace_back(worker);
        for (std::thread &th : pool) th.join();
    }
    return first_err.load();
}

// --------------------------------------------------------------- device block calling (row f1)
int ts_batch_download_blocks(ts_batch *b, ts_segment_out *out) {
    if (!b || !out) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    if (!b->synced) { int rc = ts_batch_sync(b); if (rc != TS_OK) return rc; }
    const ts_params &P = c->params;
    const size_t ns = b->segs.size(), nt = b->tiles.size();
    for (size_t i = 0; i < ns; ++i) std::memset(&out[i], 0, sizeof out[i]);
    if (ns == 0) return TS_OK;

    std::vector<uint32_t> first(ns + 1);
    std::vector<unsigned long long> inoff(ns), slen(ns), sabs(ns);
    for (size_t i = 0; i < ns; ++i) {
        first[i] = b->segs[i].first_tile; inoff[i] = b->segs[i].in_off;
        slen[i] = b->segs[i].len; sabs[i] = b->segs[i].abs_pos;
    }
    first[ns] = (uint32_t)nt;
    DevBuf d_first, d_inoff, d_len, d_abs, d_bounds, d_blocks, d_count;
    auto release = [&]() { d_first.release(); d_inoff.release(); d_len.release(); d_abs.release();
                           d_bounds.release(); d_blocks.release(); d_count.release(); };
    std::vector<TsDevBlock> blocks;
    uint32_t cap = (uint32_t)std::min<uint64_t>(64ull * ns + 4096 + b->n_matches / 256, 1u << 26);
    for (int attempt = 0; attempt < 2; ++attempt) {
        if (d_first.ensure((ns + 1) * 4) != hipSuccess || d_inoff.ensure(ns * 8) != hipSuccess ||
            d_len.ensure(ns * 8) != hipSuccess || d_abs.ensure(ns * 8) != hipSuccess ||
            d_bounds.ensure(ns * 16) != hipSuccess || d_blocks.ensure((size_t)cap * sizeof(TsDevBlock)) != hipSuccess ||
            d_count.ensure(16) != hipSuccess) { release(); return c->fail(TS_ERR_ALLOC, "device allocation failed"); }
        if (hipMemcpy(d_first.p, first.data(), (ns + 1) * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_inoff.p, inoff.data(), ns * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_len.p, slen.data(), ns * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(d_abs.p, sabs.data(), ns * 8, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemset(d_count.p, 0, 16) != hipSuccess) { release(); return c->fail(TS_ERR_HIP, "H2D copy failed"); }
        TsBlockCallParams Q{};
        Q.tiles = (const TsTile *)b->d_tiles.p;
        Q.tile_off = (const unsigned long long *)b->d_tile_off.p;
        Q.tile_stats = (const uint32_t *)b->d_stats.p;
        Q.matches = (const uint32_t *)b->d_matches.p;
        Q.blocks = (TsDevBlock *)d_blocks.p;
        Q.n_blocks = (uint32_t *)d_count.p;
        Q.block_cap = cap;
        Q.terminal_limit = P.terminal_limit; Q.max_match_dist = P.max_match_dist;
        Q.min_block_len = P.min_block_len; Q.max_block_dist = P.max_block_dist;
        Q.min_block_counts = P.min_block_counts; Q.min_block_density = P.min_block_density;
        Q.k = c->k; Q.its_min_len = (uint32_t)(uint16_t)(2 * c->bp.first_pattern_len);
        if (ts_k_launch_block_call(&Q, (const uint32_t *)d_first.p, (const unsigned long long *)d_inoff.p,
                                   (const unsigned long long *)d_len.p, (const unsigned long long *)d_abs.p,
                                   (uint32_t)ns, (uint32_t)nt, (unsigned long long *)d_bounds.p, b->tips ? 0 : 1,
                                   nullptr) != 0) { release(); return c->fail(TS_ERR_HIP, "block-calling kernel launch failed"); }
        uint32_t nb = 0;
        if (hipMemcpy(&nb, d_count.p, 4, hipMemcpyDeviceToHost) != hipSuccess) { release(); return c->fail(TS_ERR_HIP, "D2H copy failed"); }
        if (nb > cap) { cap = nb + 1024; continue; }            // rare: more blocks than provisioned, rerun
        blocks.resize(nb);
        if (nb && hipMemcpy(blocks.data(), d_blocks.p, (size_t)nb * sizeof(TsDevBlock), hipMemcpyDeviceToHost) != hipSuccess) {
            release(); return c->fail(TS_ERR_HIP, "D2H copy failed");
        }
        break;
    }
    release();

    // order: terminal blocks in push order (forward walk, then reverse walk); interstitial by start
    std::sort(blocks.begin(), blocks.end(), [](const TsDevBlock &x, const TsDevBlock &y) {
        if (x.seg != y.seg) return x.seg < y.seg;
        const uint32_t kx = x.kind == 2 ? 1 : 0, ky = y.kind == 2 ? 1 : 0;
        if (kx != ky) return kx < ky;
        if (kx) return x.start < y.start;
        return x.kind != y.kind ? x.kind < y.kind : x.seq < y.seq;
    });
    const std::unique_ptr<uint32_t[]> wins_buf(new uint32_t[(b->tips ? 0 : b->n_windows * 8) + 1]);   // (not zero-filled)
    uint32_t *const wins = wins_buf.get();
    if (!b->tips && b->n_windows)
        HIP_TRY(c, hipMemcpy(wins, b->d_windows.p, b->n_windows * 32, hipMemcpyDeviceToHost));
    // per segment: window records (float metrics on the host) + its slice of the sorted block list;
    // independent between segments, so they are finalised on up to 16 host threads
    std::vector<size_t> blk_begin(ns + 1, blocks.size());
    {
        size_t bi = 0;
        for (size_t si = 0; si < ns; ++si) {
            blk_begin[si] = bi;
            while (bi < blocks.size() && blocks[bi].seg == si) ++bi;
        }
        blk_begin[ns] = bi;
    }
    std::atomic<size_t> next{0};
    std::atomic<int> first_err{TS_OK};
    const unsigned hw_threads = std::max(1u, std::thread::hardware_concurrency());
    const unsigned spare = std::max(1u, std::min(16u, hw_threads) / (unsigned)std::max<size_t>(1, std::min<size_t>(ns, 16)));
    auto worker = [&]() {
        for (size_t si; (si = next.fetch_add(1)) < ns && first_err.load() == TS_OK;) {
            const SegPlan &sp = b->segs[si];
            int rc = finalize_segment(c, b->tips, sp.len, sp.abs_pos, sp.n_windows ? &wins[sp.win_base * 8] : nullptr,
                                      b->tips ? 0 : sp.n_windows, nullptr, 0, out[si], spare);       // windows only
            std::vector<ts_block> term, its;
            for (size_t bi = blk_begin[si]; rc == TS_OK && bi < blk_begin[si + 1]; ++bi) {
                ts_block t{};
                std::memcpy(&t, &blocks[bi], sizeof(ts_block));
                (blocks[bi].kind == 2 ? its : term).push_back(t);
            }
            auto put = [&](const std::vector<ts_block> &v, ts_block *&dst, uint64_t &n) -> bool {
                n = v.size(); dst = nullptr;
                if (v.empty()) return true;
                dst = (ts_block *)std::malloc(v.size() * sizeof(ts_block));
                if (!dst) return false;
                std::memcpy(dst, v.data(), v.size() * sizeof(ts_block));
                return true;
            };
            if (rc == TS_OK && (!put(term, out[si].terminal_blocks, out[si].n_terminal_blocks) ||
                                !put(its, out[si].interstitial_blocks, out[si].n_interstitial_blocks)))
                rc = c->fail(TS_ERR_ALLOC, "out of host memory");
            if (rc != TS_OK) { int expected = TS_OK; first_err.compare_exchange_strong(expected, rc); return; }
        }
    };
    const unsigned nthreads = (unsigned)std::min<size_t>({(size_t)16, ns, (size_t)std::max(1u, std::thread::hardware_concurrency())});
    if (nthreads <= 1) {
        worker();
    } else {
        std::vector<std::thread> pool;
        for (unsigned i = 0; i < nthreads; ++i) pool.emplace_back(worker);
        for (std::thread &th : pool) th.join();
    }
    return first_err.load();
}

void ts_free_segments(ts_segment_out *out, size_t n_segs) {
    if (!out) return;
    for (size_t i = 0; i < n_segs; ++i) {
        std::free(out[i].windows);
        std::free(out[i].matches);
        std::free(out[i].terminal_blocks);
        std::free(out[i].interstitial_blocks);
        std::memset(&out[i], 0, sizeof out[i]);
    }
}

// =========================================================================== general path
// For parameter sets outside the tiled kernel's closed form (mixed-length pattern sets, pattern
// lengths > 9, or a longest pattern exceeding min(step, window-step) where the reference's
// uint32 start index wraps): ts_generic_match + ts_generic_windows (generic.hip) on the device,
// then only ordering work on the host.  One segment at a time; this is the slow exact path.
static int scan_group_generic(ts_ctx *c, const ts_segment_in *segs, const std::vector<size_t> &which,
                              bool tips, ts_segment_out *out) {
    if (which.empty()) return TS_OK;
    if (!c->generic_ok)
        return c->fail(TS_ERR_UNSUPPORTED, "unsupported parameter set: more than 8 pattern lengths, a pattern longer "
                                           "than 32 or a non-ACGT pattern");
    std::lock_guard<std::mutex> lk(c->mtx);
    HIP_TRY(c, hipSetDevice(c->device));
    const ts_params &P = c->params;
    const uint32_t s = P.step, w = P.window_size, ov = w - s, L = c->longest;
    DevBuf d_seq, d_mask, d_win;
    std::vector<uint32_t> mask, wins;
    std::vector<ts_match> matches;
    struct Hit { uint64_t k, p; uint16_t len; uint8_t flags; };
    std::vector<Hit> hits;
    int rc = TS_OK;
    for (size_t wi = 0; wi < which.size() && rc == TS_OK; ++wi) {
        const ts_segment_in &sg = segs[which[wi]];
        const uint64_t N = sg.len;
        matches.clear();
        uint64_t nwin = 0;
        // regions exactly as scanSegment picks them (src/teloscope.cpp:576-583; uint32 product)
        std::vector<std::pair<uint64_t, uint64_t>> regions;
        if (tips) {
            const uint32_t twice = 2u * P.terminal_limit;
            if (N > twice) { regions.emplace_back(0, P.terminal_limit); regions.emplace_back(N - P.terminal_limit, P.terminal_limit); }
            else if (N) regions.emplace_back(0, N);
        } else if (N) {
            regions.emplace_back(0, N);
            nwin = ceil_div(N, s);
        }
        for (const auto &rg : regions) {
            const uint64_t r0 = rg.first, n = rg.second;
            if (d_seq.ensure(n + 16) != hipSuccess || d_mask.ensure(n * 4 + 16) != hipSuccess) { rc = c->fail(TS_ERR_ALLOC, "device allocation failed"); break; }
            if (hipMemcpy(d_seq.p, sg.seq + r0, n, hipMemcpyHostToDevice) != hipSuccess) { rc = c->fail(TS_ERR_HIP, "H2D copy failed"); break; }
            if (ts_k_launch_generic_match((const unsigned char *)d_seq.p, n, &c->gpat, P.fold_case, (uint32_t *)d_mask.p, nullptr) != 0) { rc = c->fail(TS_ERR_HIP, "generic match kernel launch failed"); break; }
            if (!tips) {
                TsGenericGeom Q{};
                Q.n = N; Q.s = s; Q.w = w; Q.longest = L; Q.nuc_on = (P.out_gc || P.out_entropy) ? 1u : 0u; Q.fold = P.fold_case;
                if (d_win.ensure(nwin * 32 + 16) != hipSuccess) { rc = c->fail(TS_ERR_ALLOC, "device allocation failed"); break; }
                if (ts_k_launch_generic_windows((const unsigned char *)d_seq.p, (const uint32_t *)d_mask.p, &c->gpat, &Q, nwin, (uint32_t *)d_win.p, nullptr) != 0) { rc = c->fail(TS_ERR_HIP, "generic window kernel launch failed"); break; }
                wins.resize(nwin * 8);
                if (hipMemcpy(wins.data(), d_win.p, nwin * 32, hipMemcpyDeviceToHost) != hipSuccess) { rc = c->fail(TS_ERR_HIP, "D2H copy failed"); break; }
            }
            mask.resize(n);
            if (hipMemcpy(mask.data(), d_mask.p, n * 4, hipMemcpyDeviceToHost) != hipSuccess) { rc = c->fail(TS_ERR_HIP, "D2H copy failed"); break; }

            // enumerate matches in (position, length) order; in full-scan mode keep those some
            // window's own scan pushes (src/teloscope.cpp:485) and order them by that window
            hits.clear();
            const uint32_t t1 = s - L, t2 = ov - L;                     // uint32 wrap, src/teloscope.cpp:413-415
            const uint32_t start_index = t1 < t2 ? t1 : t2;
            for (uint64_t p = 0; p < n; ++p) {
                const uint32_t m = mask[p];
                if (!m) continue;
                for (uint32_t li = 0; li < c->gpat.nlen; ++li) {
                    const uint32_t b = (m >> (3 * li)) & 7u;
                    if (!(b & 1u)) continue;
                    const uint32_t len = c->gpat.len[li];
                    const uint8_t fl = (uint8_t)(((b & 2u) ? TS_MATCH_FORWARD : 0u) | ((b & 4u) ? TS_MATCH_CANONICAL : 0u));
                    uint64_t k = 0;
                    if (!tips) {
                        const uint64_t e = p + len - 1;
                        if (ov == 0) {
                            k = p / s;
                            const uint64_t cws = std::min<uint64_t>(w, N - k * s);
                            if ((p - k * s) + len > cws) continue;      // crosses its only window's end
                        } else if (e < std::min<uint64_t>(w, N)) {
                            k = 0;                                      // window 0 scans everything it holds
                        } else {
                            k = (e - ov) / s;                           // the one window with j >= overlap
                            if (p < k * s || (p - k * s) < start_index) continue;
                        }
                    }
                    hits.push_back(Hit{k, r0 + p, (uint16_t)len, fl});
                }
            }
            if (!tips)
                std::stable_sort(hits.begin(), hits.end(), [](const Hit &a, const Hit &b) { return a.k < b.k; });
            for (const Hit &h : hits) {
                ts_match m{};
                m.position = sg.abs_pos + h.p;
                m.match_size = h.len;
                m.flags = h.flags;
                matches.push_back(m);
            }
        }
        if (rc != TS_OK) break;
        ts_match *arr = matches.empty() ? nullptr : (ts_match *)std::malloc(matches.size() * sizeof(ts_match));
        if (!matches.empty() && !arr) { rc = c->fail(TS_ERR_ALLOC, "out of host memory"); break; }
        if (arr) std::memcpy(arr, matches.data(), matches.size() * sizeof(ts_match));
        rc = finalize_segment(c, tips, N, sg.abs_pos, wins.data(), nwin, arr, matches.size(), out[which[wi]], 16u);   // (general path: one segment at a time)
    }
    d_seq.release(); d_mask.release(); d_win.release();
    return rc;
}

// Uploads all segments of a batch: the bases are gathered into a ring of two pinned chunks laid out
// like the device buffer, each chunk leaving by DMA (hipMemcpyAsync on a private stream) while the
// next one is being filled — one copy per read would cost ~10 us each, one pageable 3 GB copy ~0.5 s.
// Bytes between segments are never read as bases (the kernel masks everything past a segment's end).
static int batch_upload_all(ts_batch *b, const std::vector<const char *> &seqs) {
    ts_ctx *c = b->ctx;
    if (!batch_input(b, false)) return c->fail(TS_ERR_ALLOC, "cannot allocate device input buffer");   // every byte is uploaded below
    constexpr size_t kChunk = 32u << 20;
    if (!c->up_stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(c, hipHostMalloc(&c->pin[i], kChunk, hipHostMallocDefault));
            HIP_TRY(c, hipEventCreateWithFlags(&c->pin_ev[i], hipEventDisableTiming));
        }
    }
    const size_t nseg = b->segs.size();
    size_t seg = 0;                 // first segment that may still have bytes at or beyond the chunk start
    int slot = 0;
    bool used[2] = {false, false};
    // A chunk is staged by several threads (one memcpy stream fills pinned memory at ~10 GB/s, a fraction of
    // what the link moves) while the previous chunk's DMA is in flight.
    struct Piece { char *dst; const char *src; size_t len; };
    std::vector<Piece> pieces;
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nthr = std::min(8u, std::max(1u, hw / 2u));
    for (uint64_t c0 = 0; c0 < b->input_bytes; c0 += kChunk) {
        const uint64_t c1 = std::min<uint64_t>(c0 + kChunk, b->input_bytes);
        if (used[slot]) HIP_TRY(c, hipEventSynchronize(c->pin_ev[slot]));
        char *dst = (char *)c->pin[slot];
        while (seg < nseg && b->segs[seg].in_off + b->segs[seg].len <= c0) ++seg;
        pieces.clear();
        size_t bytes = 0;
        for (size_t i = seg; i < nseg && b->segs[i].in_off < c1; ++i) {
            const uint64_t s0 = std::max<uint64_t>(b->segs[i].in_off, c0);
            const uint64_t s1 = std::min<uint64_t>(b->segs[i].in_off + b->segs[i].len, c1);
            if (s1 > s0) { pieces.push_back({dst + (s0 - c0), seqs[i] + (s0 - b->segs[i].in_off), (size_t)(s1 - s0)}); bytes += s1 - s0; }
        }
        const unsigned nt = bytes >= (4u << 20) ? nthr : 1u;
        if (nt == 1u) {
            for (const Piece &pc : pieces) std::memcpy(pc.dst, pc.src, pc.len);
        } else {
            // thread t copies the bytes [t, t+1) * share of the concatenated pieces
            const size_t share = (bytes + nt - 1) / nt;
            std::vector<std::thread> pool;
            pool.reserve(nt);
            for (unsigned t = 0; t < nt; ++t)
                pool.emplace_back([&, t] {
                    const size_t lo = (size_t)t * share, hi = std::min(bytes, lo + share);
                    size_t at = 0;
                    for (const Piece &pc : pieces) {
                        const size_t a = std::max(lo, at), z = std::min(hi, at + pc.len);
                        if (z > a) std::memcpy(pc.dst + (a - at), pc.src + (a - at), z - a);
                        at += pc.len;
                        if (at >= hi) break;
                    }
                });
            for (std::thread &th : pool) th.join();
        }
        HIP_TRY(c, hipMemcpyAsync((char *)b->d_in.p + c0, dst, c1 - c0, hipMemcpyHostToDevice, c->up_stream));
        HIP_TRY(c, hipEventRecord(c->pin_ev[slot], c->up_stream));
        used[slot] = true;
        slot ^= 1;
    }
    HIP_TRY(c, hipStreamSynchronize(c->up_stream));
    return TS_OK;
}

// =========================================================================== scanSegment, batched
static int scan_group(ts_ctx *ctx, const ts_segment_in *segs, const std::vector<size_t> &which,
                      bool tips, ts_segment_out *out) {
    if (which.empty()) return TS_OK;
    const auto t_begin = std::chrono::steady_clock::now();
    std::vector<uint64_t> lens(which.size()), abs(which.size());
    for (size_t i = 0; i < which.size(); ++i) { lens[i] = segs[which[i]].len; abs[i] = segs[which[i]].abs_pos; }
    ts_batch *b = ts_batch_create(ctx, lens.data(), abs.data(), which.size(), tips, 0);
    if (!b) return ctx->error.rfind("unsupported", 0) == 0 ? TS_ERR_UNSUPPORTED : TS_ERR_HIP;
    std::vector<const char *> ptrs(which.size());
    for (size_t i = 0; i < which.size(); ++i) ptrs[i] = segs[which[i]].seq;
    const bool timing = getenv("TS_TIMING") != nullptr;          // stage times to stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) {
        return std::chrono::duration<double, std::milli>(y - x).count();
    };
    const auto t0 = now();
    int rc = batch_upload_all(b, ptrs);
    const auto t1 = now();
    if (rc == TS_OK) rc = ts_batch_scan(b, nullptr, nullptr);
    if (rc == TS_OK) rc = ts_batch_sync(b);
    const auto t2 = now();
    std::vector<ts_segment_out> tmp(which.size());
    if (rc == TS_OK) rc = ts_batch_download(b, nullptr, tmp.data());
    if (timing)
        fprintf(stderr, "ts_scan_segments: plan %.1f ms, upload %.1f ms, scan %.1f ms, download + host post-processing %.1f ms\n",
                ms(t_begin, t0), ms(t0, t1), ms(t1, t2), ms(t2, now()));
    if (rc == TS_OK)
        for (size_t i = 0; i < which.size(); ++i) out[which[i]] = tmp[i];
    else
        ts_free_segments(tmp.data(), tmp.size());
    ts_batch_destroy(b);
    return rc;
}

int ts_scan_segments(ts_ctx *ctx, const ts_segment_in *segs, size_t n_segs, ts_segment_out *out) {
    if (!ctx || (n_segs && (!segs || !out))) return TS_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> api(ctx->api_mtx);
    for (size_t i = 0; i < n_segs; ++i) {
        std::memset(&out[i], 0, sizeof out[i]);
        if (segs[i].len && !segs[i].seq) return ctx->fail(TS_ERR_INVALID_ARG, "null sequence pointer");
    }
    std::vector<size_t> full, tips;
    for (size_t i = 0; i < n_segs; ++i) (segs[i].tips_only ? tips : full).push_back(i);
    std::string why;
    int rc = full_scan_supported(ctx, why) ? scan_group(ctx, segs, full, false, out)
                                           : scan_group_generic(ctx, segs, full, false, out);
    if (rc == TS_OK) rc = ctx->fast_ok ? scan_group(ctx, segs, tips, true, out)
                                       : scan_group_generic(ctx, segs, tips, true, out);
    if (rc != TS_OK) ts_free_segments(out, n_segs);
    return rc;
}

// scanSegment for callers that do not read the match vectors: scan, block calling and the per-segment
// counts all stay on the device; windows, blocks and four counters per segment cross PCIe.
static int scan_group_blocks(ts_ctx *ctx, const ts_segment_in *segs, const std::vector<size_t> &which, bool tips,
                             ts_segment_out *out, ts_segment_counts *counts) {
    if (which.empty()) return TS_OK;
    std::vector<uint64_t> lens(which.size()), abs(which.size());
    for (size_t i = 0; i < which.size(); ++i) { lens[i] = segs[which[i]].len; abs[i] = segs[which[i]].abs_pos; }
    ts_batch *b = ts_batch_create(ctx, lens.data(), abs.data(), which.size(), tips, 0);
    if (!b) return ctx->error.rfind("unsupported", 0) == 0 ? TS_ERR_UNSUPPORTED : TS_ERR_HIP;
    std::vector<const char *> ptrs(which.size());
    for (size_t i = 0; i < which.size(); ++i) ptrs[i] = segs[which[i]].seq;
    int rc = batch_upload_all(b, ptrs);
    if (rc == TS_OK) rc = ts_batch_scan(b, nullptr, nullptr);
    if (rc == TS_OK) rc = ts_batch_sync(b);
    std::vector<ts_segment_out> tmp(which.size());
    if (rc == TS_OK) rc = ts_batch_download_blocks(b, tmp.data());
    std::vector<unsigned long long> summary(4 * which.size());
    if (rc == TS_OK && counts) {
        DevBuf d_sum;
        if (d_sum.ensure(summary.size() * 8 + 16) != hipSuccess) rc = ctx->fail(TS_ERR_ALLOC, "out of device memory");
        if (rc == TS_OK) rc = ts_batch_segment_summary(b, d_sum.p, nullptr);
        if (rc == TS_OK && hipMemcpy(summary.data(), d_sum.p, summary.size() * 8, hipMemcpyDeviceToHost) != hipSuccess)
            rc = ctx->fail(TS_ERR_HIP, "summary download failed");
        d_sum.release();
    }
    if (rc == TS_OK) {
        for (size_t i = 0; i < which.size(); ++i) {
            out[which[i]] = tmp[i];
            if (counts)
                counts[which[i]] = ts_segment_counts{tips ? 0 : summary[4 * i], summary[4 * i + 1], summary[4 * i + 2], summary[4 * i + 3]};
        }
    } else {
        ts_free_segments(tmp.data(), tmp.size());
    }
    ts_batch_destroy(b);
    return rc;
}

int ts_scan_segments_blocks(ts_ctx *ctx, const ts_segment_in *segs, size_t n_segs, ts_segment_out *out,
                            ts_segment_counts *counts) {
    if (!ctx || (n_segs && (!segs || !out))) return TS_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> api(ctx->api_mtx);
    for (size_t i = 0; i < n_segs; ++i) {
        std::memset(&out[i], 0, sizeof out[i]);
        if (counts) counts[i] = ts_segment_counts{0, 0, 0, 0};
        if (segs[i].len && !segs[i].seq) return ctx->fail(TS_ERR_INVALID_ARG, "null sequence pointer");
    }
    std::vector<size_t> full, tips;
    for (size_t i = 0; i < n_segs; ++i) (segs[i].tips_only ? tips : full).push_back(i);
    // parameter sets outside the tiled kernel take the general path and drop the match vectors afterwards
    auto via_matches = [&](const std::vector<size_t> &which, bool tips_mode) -> int {
        int rc = scan_group_generic(ctx, segs, which, tips_mode, out);
        if (rc != TS_OK) return rc;
        for (size_t i : which) {
            if (counts) {
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
    int rc = full_scan_supported(ctx, why) ? scan_group_blocks(ctx, segs, full, false, out, counts) : via_matches(full, false);
    if (rc == TS_OK) rc = ctx->fast_ok ? scan_group_blocks(ctx, segs, tips, true, out, counts) : via_matches(tips, true);
    if (rc != TS_OK) ts_free_segments(out, n_segs);
    return rc;
}

// =========================================================================== ReadTelomereFilter
int ts_filter_reads(ts_ctx *ctx, const char *const *seqs, const uint64_t *lens, size_t n_reads,
                    uint8_t *pass) {
    if (!ctx || (n_reads && (!seqs || !lens || !pass))) return TS_ERR_INVALID_ARG;
    if (!ctx->read_filter) return ctx->fail(TS_ERR_STATE, "context was not made by ts_create_read_filter");
    if (n_reads == 0) return TS_OK;
    std::lock_guard<std::mutex> api(ctx->api_mtx);
    std::vector<uint64_t> rl(n_reads);
    for (size_t i = 0; i < n_reads; ++i) {
        uint64_t n = lens[i];
        if (n && seqs[i][n - 1] == '\r') --n;             // src/read-filter.cpp:38-40
        rl[i] = n;
    }
    if (!ctx->fast_ok) {
        // mixed-length pattern sets: general kernels + host block predicate
        std::vector<ts_segment_in> in(n_reads);
        for (size_t i = 0; i < n_reads; ++i) { in[i].seq = seqs[i]; in[i].len = rl[i]; in[i].abs_pos = 0; in[i].tips_only = 1; }
        std::vector<ts_segment_out> out(n_reads);
        int rc = ts_scan_segments(ctx, in.data(), n_reads, out.data());
        if (rc != TS_OK) return rc;
        for (size_t i = 0; i < n_reads; ++i) pass[i] = out[i].n_terminal_blocks != 0;
        ts_free_segments(out.data(), n_reads);
        return TS_OK;
    }
    // tiled path: whole-read tips scan, then the terminal-block predicate on the device; only one
    // byte per read comes back
    const bool timing = getenv("TS_TIMING") != nullptr;          // stage times to stderr
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b2) {
        return std::chrono::duration<double, std::milli>(b2 - a).count();
    };
    const auto t0 = now();
    ts_batch *b = ts_batch_create(ctx, rl.data(), nullptr, n_reads, 1, 0);
    if (!b) return ctx->error.rfind("unsupported", 0) == 0 ? TS_ERR_UNSUPPORTED : TS_ERR_HIP;
    const auto t1 = now();
    int rc = batch_upload_all(b, std::vector<const char *>(seqs, seqs + n_reads));
    const auto t2 = now();
    if (rc == TS_OK) rc = ts_batch_scan(b, nullptr, nullptr);
    if (rc == TS_OK) rc = ts_batch_sync(b);
    const auto t3 = now();
    if (rc == TS_OK) {
        ts_ctx *c = ctx;
        auto run = [&]() -> int {
            const size_t ns = n_reads;
            std::vector<uint32_t> first(ns + 1);
            std::vector<unsigned long long> inoff(ns), slen(ns);
            for (size_t i = 0; i < ns; ++i) { first[i] = b->segs[i].first_tile; inoff[i] = b->segs[i].in_off; slen[i] = b->segs[i].len; }
            first[ns] = (uint32_t)b->tiles.size();
            DevBuf d_first, d_inoff, d_len, d_pass;
            HIP_TRY(c, d_first.ensure((ns + 1) * 4));
            HIP_TRY(c, d_inoff.ensure(ns * 8));
            HIP_TRY(c, d_len.ensure(ns * 8));
            HIP_TRY(c, d_pass.ensure(ns + 16));
            HIP_TRY(c, hipMemcpy(d_first.p, first.data(), (ns + 1) * 4, hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(d_inoff.p, inoff.data(), ns * 8, hipMemcpyHostToDevice));
            HIP_TRY(c, hipMemcpy(d_len.p, slen.data(), ns * 8, hipMemcpyHostToDevice));
            TsPredParams Q{};
            Q.terminal_limit = c->params.terminal_limit;
            Q.max_match_dist = c->params.max_match_dist;
            Q.min_block_len = c->params.min_block_len;
            Q.max_block_dist = c->params.max_block_dist;
            Q.min_block_counts = c->params.min_block_counts;
            Q.min_block_density = c->params.min_block_density;
            Q.k = c->k;
            int e = ts_k_launch_predicate((const TsTile *)b->d_tiles.p, (const unsigned long long *)b->d_tile_off.p,
                                          (const uint32_t *)b->d_stats.p, (const uint32_t *)b->d_matches.p,
                                          (const uint32_t *)d_first.p, (const unsigned long long *)d_inoff.p,
                                          (const unsigned long long *)d_len.p, (uint32_t)ns, &Q,
                                          (unsigned char *)d_pass.p, nullptr);
            if (e != 0) return c->fail(TS_ERR_HIP, "predicate kernel launch failed");
            HIP_TRY(c, hipMemcpy(pass, d_pass.p, ns, hipMemcpyDeviceToHost));
            d_first.release(); d_inoff.release(); d_len.release(); d_pass.release();
            return TS_OK;
        };
        rc = run();
    }
    const auto t4 = now();
    ts_batch_destroy(b);
    if (timing)
        fprintf(stderr, "ts_filter_reads: plan %.1f ms, upload %.1f ms, scan %.1f ms, predicate %.1f ms, free %.1f ms\n",
                ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, now()));
    return rc;
}

}  // extern "C"
