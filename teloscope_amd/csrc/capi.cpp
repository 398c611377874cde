// capi.cpp — the C-ABI of libteloscan (include/teloscan.h): contexts, device-resident
// batches, the batched scanSegment / ReadTelomereFilter entry points and their host
// post-processing.  There is no CPU scan path in this library: without a HIP device
// ts_create() fails.
#include <hip/hip_runtime.h>
#include <immintrin.h>

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
#include <tuple>
#include <vector>

#include <cctype>
#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>

#include "capi_internal.hpp"

namespace {

thread_local std::string g_create_error;

}  // namespace

// ---------------------------------------------------------------------------------------- buffer pool
hipError_t BufferPool::take(size_t need, DevBuf &out) {
    if (need == 0) need = 16;
    {
        std::lock_guard<std::mutex> g(m_);
        size_t best = free_.size();
        for (size_t i = 0; i < free_.size(); ++i)
            if (free_[i].bytes >= need && free_[i].bytes <= 2 * need + (1u << 20) &&
                (best == free_.size() || free_[i].bytes < free_[best].bytes)) best = i;
        if (best != free_.size()) {
            held_ -= free_[best].bytes;
            out = std::move(free_[best]);
            free_.erase(free_.begin() + (long)best);
            return hipSuccess;
        }
    }
    out.release();
    hipError_t e = out.ensure(need);
    if (e != hipSuccess) {                       // out of memory: drop what the pool holds and try once more
        clear();
        (void)hipGetLastError();
        e = out.ensure(need);
    }
    return e;
}

void BufferPool::give(DevBuf &&b) {
    if (!b.p) return;
    std::lock_guard<std::mutex> g(m_);
    if (free_.size() >= kMaxBlocks || held_ + b.bytes > kMaxHeld) { b.release(); return; }
    held_ += b.bytes;
    free_.push_back(std::move(b));
}

void BufferPool::clear() {
    std::lock_guard<std::mutex> g(m_);
    free_.clear();
    held_ = 0;
}

namespace {

constexpr uint32_t kMaxLds = 160u * 1024u;
constexpr uint64_t kEventRing = 64;            // scans whose HIP-event times a batch remembers
constexpr uint32_t kMaxBlocksPerTile = 448;
constexpr uint32_t kMaxChunks = 8;             // tiles up to ~16 k positions per wave
constexpr uint32_t kTipsChunks = 8;            // tips-only / read batches: ~16 k positions per tile (15 kb reads, TS_GEOMETRY sweeps: six chunks until round 4;
                                               // eight since the stage's entries are 16 bits — profiles/r05/reads_tilings.txt)

// The measurement / test knobs of the host entry points (ts_ctx::Knobs), from the environment: at ts_create and ts_refresh_env,
// never per call.
void read_env_knobs(ts_ctx *c) {
    auto is = [](const char *name, char v) { const char *e = getenv(name); return e && e[0] == v; };
    ts_ctx::Knobs k;
    k.timing = getenv("TS_TIMING") != nullptr;
    k.gen_host_blocks = is("TS_GEN_HOST_BLOCKS", '1');
    k.gen_compact_always = is("TS_GEN_COMPACT", '1');     // (A/B: the dense stream made even where nothing reads it)
    k.gen_prefetch = !is("TS_GEN_PREFETCH", '0');
    k.gen_list = !is("TS_GEN_LIST", '0');
    if (const char *e = getenv("TS_GEN_ABL")) k.gen_abl = (uint32_t)atoi(e);
    k.packed_upload = !is("TS_PACKED_UPLOAD", '0');
    if (const char *e = getenv("TS_PACKED_MIN_BYTES")) k.packed_min_bytes = strtoull(e, nullptr, 10);
    if (const char *e = getenv("TS_STAGE_THREADS")) { const int n = atoi(e); if (n > 0) k.stage_threads = (uint32_t)std::min(n, 64); }
    if (const char *e = getenv("TS_SIDE_PRIORITY")) k.side_priority = atoi(e);
    k.side_probe = !is("TS_SIDE_PROBE", '0');
    k.rec16 = !is("TS_REC32", '1');
    if (const char *e = getenv("TS_SCAN_EVENTS")) k.scan_events = atoi(e);
    c->knobs = k;
}

// diagnostics: TS_DEALT_TILES=1 keeps every scan on the round-robin tile assignment (A/B against on-demand tiles)
bool ts_env_flag(const char *name) {
    const char *v = getenv(name);
    return v && *v && *v != '0';
}

// CPUs of the NUMA node of HIP device `dev`, from sysfs (Linux): /sys/bus/pci/devices/<bus id>/numa_node and
// /sys/devices/system/node/node<N>/cpulist.  Empty when anything is missing.
std::vector<int> device_node_cpus(int dev) {
    std::vector<int> cpus;
    if (ts_env_flag("TS_NO_NUMA_BIND")) return cpus;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, dev) != hipSuccess) return cpus;
    for (char *q = bus; *q; ++q) *q = (char)tolower((unsigned char)*q);
    int node = -1;
    {
        FILE *f = fopen((std::string("/sys/bus/pci/devices/") + bus + "/numa_node").c_str(), "r");
        if (!f) return cpus;
        if (fscanf(f, "%d", &node) != 1) node = -1;
        fclose(f);
    }
    if (node < 0) return cpus;
    FILE *f = fopen(("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r");
    if (!f) return cpus;
    int a = 0, b = 0;
    for (;;) {                                             // "0-63,128-191"
        if (fscanf(f, "%d", &a) != 1) break;
        b = a;
        int ch = fgetc(f);
        if (ch == '-') { if (fscanf(f, "%d", &b) != 1) break; ch = fgetc(f); }
        for (int k = a; k <= b && k < 4096; ++k) cpus.push_back(k);
        if (ch != ',') break;
    }
    fclose(f);
    return cpus;
}

// groups of workgroups that share a ticket counter (TS_TICKET_GROUPS overrides, for measurements)
uint32_t ticket_groups_wanted() {
    uint32_t g = 32;
    if (const char *e = getenv("TS_TICKET_GROUPS")) { const int n = atoi(e); if (n > 0) g = (uint32_t)n; }
    return std::min<uint32_t>(g, TS_MAX_TICKET_GROUPS);
}

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
    kp.emit = 0;                         // (ts_batch_set_emit)
    kp.kdist = P.max_match_dist;
    if (tips) {
        kp.halo_blocks = 0;
        kp.straddle_fix = 0; kp.windows_on = 0; kp.nuc_on = 0; kp.block_sums = 0; kp.acc_blocks = 0;
    } else {
        const uint32_t s = P.step, w = P.window_size;
        kp.s = s; kp.w = w;
        kp.s_inv = (65536u + s - 1u) / s;
        kp.halo_blocks = (w + s - 1) / s - 1;            // window i reaches into step blocks i .. i + ceil(w/s) - 1
        kp.straddle_fix = (w == s) ? 1u : 0u;
        kp.windows_on = 1;
        kp.nuc_on = (P.out_gc || P.out_entropy) ? 1u : 0u;
        kp.block_sums = (w % s == 0) ? 1u : 0u;
        kp.acc_blocks = (kp.block_sums && !ts_env_flag("TS_ACC_PER_WINDOW")) ? 1u : 0u;      // (TS_ACC_PER_WINDOW=1: A/B)
    }
    // Search (waves per workgroup, chunks per tile) for the best modelled throughput:
    //   owned bases per tile x occupancy factor / instructions per tile.
    // Instructions: ~470 per 2016-position chunk (decode, probes, planes, per-match pass: the weights of round 1's kernel), ~170 per
    // pass of the window loop (64 window fields per pass), ~100 fixed per tile.  Occupancy factors are
    // measured (profiles/r01/geometry_sweep.txt): against 16 waves per CU the kernel loses 12 % at 12
    // waves and 31 % at 8; more than 16 is not reachable (two workgroups per CU did not co-reside).
    // Round 2: built for 80 VGPRs, two workgroups of 10 waves do share a CU (20 waves; each has its own table copy and half of
    // the LDS, so tiles are shorter): against 16 x 8 chunks, 2 x 10 x 6 chunks measures -3.2 % on configs[1], -1.2 % with the
    // default flags (-8 % and 0 on another box: see `sparse` below); with k = 7 (a table of 20 KB per copy) only
    // 5 chunks and a small record stage fit and it loses 37 %.  So: considered only for window scans with byte tables, at least
    // 6 chunks, a record stage of 256 and all accumulator copies; its factor is what those measurements give under the
    // instruction model below.  (Read batches: +4 % on one box, -1 % on another, where the predicate kernel that runs beside
    // the scan finds fewer free registers; they stay on one workgroup.)
    // What the extra waves hide is the latency chains of the per-match work: the gain needs matches.  On random sequence a
    // position matches with probability patterns / 4^k: 3 % for configs[1] (-2.7 % at 1 Gb) and for a k = 5 motif with one
    // mismatch (-3 %); 0.9 % with the default flags and 0.05 % with -x 0 (both +-1 %): those stay on 16 waves
    // (profiles/r02/planner_check.txt).
    const bool sparse = (double)c->patterns.size() < 0.015 * (double)(1ull << (2 * k));
    static const struct { uint32_t waves, wgs; double factor; } kOccupancy[] = {
        {16, 1, 1.0}, {10, 2, 1.06}, {12, 1, 0.88}, {8, 1, 0.69}, {4, 1, 0.43}, {2, 1, 0.22}, {1, 1, 0.11}};
    uint32_t nch_min = 1;
    if (!tips) nch_min = (uint32_t)ceil_div((uint64_t)(1 + kp.halo_blocks) * kp.s + 63, TS_CHUNK);
    // TS_GEOMETRY="waves,chunks" pins the search to one point (0 = any): a test knob for the contract that the
    // output does not depend on the tiling ("independent of GPU count and tile size")
    uint32_t pin_waves = 0, pin_nch = 0;
    uint32_t pin_stage = 0, pin_wgs = 0;               // (4th field: workgroups per CU, for measurements such as 2 x 12 waves)
    if (const char *g = getenv("TS_GEOMETRY")) sscanf(g, "%u,%u,%u,%u", &pin_waves, &pin_nch, &pin_stage, &pin_wgs);
    const uint32_t nch_max = tips ? std::max(kTipsChunks, pin_nch) : std::max(nch_min, std::max(kMaxChunks, pin_nch));
    double best = 0.0;
    TsScanParams best_kp{};
    uint32_t best_wpt = 0;
    for (const auto &occ : kOccupancy) {
        if (pin_waves && occ.waves != pin_waves) continue;
        const uint32_t wgs = (pin_waves && pin_wgs >= 1 && pin_wgs <= 4) ? pin_wgs : occ.wgs;
        const uint32_t kMaxLds = ::kMaxLds / wgs;
        for (uint32_t nch = nch_min; nch <= nch_max; ++nch) {
            if (nch * TS_CHUNK + 64u > 65535u) break;        // the match queue holds 16-bit plane coordinates
            if (pin_nch && nch != pin_nch) continue;
            TsScanParams cand = kp;
            cand.waves_per_wg = occ.waves;
            cand.wgs_per_cu = wgs;
            cand.nch = nch;
            cand.stage_u16 = ((uint64_t)nch * TS_CHUNK + 64u <= (1u << 14) && !ts_env_flag("TS_STAGE_U32")) ? 1u : 0u;      // (position << 2 | flags) in 16 bits: the stage's entries (TS_STAGE_U32=1: A/B)
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
                const uint32_t spare = (kMaxLds - (uint32_t)ts_k_lds_bytes(&cand)) / occ.waves / (cand.stage_u16 ? 2u : 4u);
                cand.stage_cap = std::min<uint32_t>(pin_stage ? pin_stage : 1024u, 128u + (spare & ~15u));
                while ((uint32_t)ts_k_lds_bytes(&cand) > kMaxLds) cand.stage_cap -= 16;
            }
            // (the kernels for the 2-bit tables of k >= 7 need 97 VGPRs: built for 80 they spill)
            if (wgs > 1 && !(kp.pair_byte_table && kp.fc_byte_table)) continue;      // not built: it would spill (kernels.hip: scan_variant)
            if (wgs > 1 && !pin_waves && (tips || sparse || !(kp.pair_byte_table && kp.fc_byte_table) || nch < 6 || cand.stage_cap < 256 || cand.acc_copies < 4)) continue;
            const double passes = tips ? 0.0 : (double)ceil_div((uint64_t)cwpt * 4, 64);
            // fewer accumulator copies serialise the window adds of a pass: 8 chunks with 2 copies measured 2.5 %
            // slower than 7 chunks with 4 on the headline configuration, where the model alone says 1 % faster
            const double copies = cand.acc_copies >= 4 ? 1.0 : cand.acc_copies == 2 ? 0.96 : 0.92;
            // a stage that holds a tile of random sequence whole (its matches at the pattern set's density, + 15 % + most of a pass) lets
            // the records leave once, at the tile's end; a smaller one flushes from inside the chunk loop, where a store sits in
            // the memory queue ahead of the chunk loads that are waited for next
            const double tile_records = (double)c->patterns.size() / (double)(1ull << (2 * std::min<uint32_t>(k, 16))) * (double)cwpt * cand.s;
            const double whole = (double)cand.stage_cap >= 1.15 * tile_records + 48.0 ? 1.0 : 0.93;
            const double score = (double)cwpt * cand.s * occ.factor * copies * whole / (470.0 * nch + 170.0 * passes + 100.0);
            if (score > best) { best = score; best_kp = cand; best_wpt = cwpt; }
        }
    }
    if (best > 0.0) {
        kp = best_kp; wpt = best_wpt;
        {   // the exact 22-bit magic of the step, where it exists for every position of a tile (see kernels.hip: the per-match pass)
            const uint64_t s64 = std::max<uint32_t>(kp.s, 1), magic = ((1ull << 22) + s64 - 1) / s64, err = magic * s64 - (1ull << 22);
            const uint64_t max_u = (uint64_t)kp.nch * TS_CHUNK + 64u;
            kp.s_magic22 = (uint32_t)magic;
            kp.div_exact = (!tips && magic < (1u << 24) && max_u < (1u << 24) && max_u * magic < (1ull << 32) && max_u * err < (1ull << 22)) ? 1u : 0u;
        }
        kp.vis_wide = ((uint64_t)kp.nch * TS_CHUNK + 64u <= (1u << 14)) ? 0u : 1u;      // (position << 2 | flags) in 16 bits?
        return true;
    }
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

// the first tile at or after `t` whose segment differs from tile t's is found by the callers; here: the range
// [lo, hi) of tiles -> what it covers (window records, bytes of the input layout, owned bases)
void range_cover(const ts_batch *b, uint64_t lo, uint64_t hi, ts_range_info &r) {
    r = ts_range_info{};
    r.tile_begin = lo; r.tile_end = hi;
    if (lo >= hi) return;
    const TsTile &first = b->tiles[lo];
    r.window_begin = b->tips ? 0 : first.win_out;
    r.window_end = r.window_begin;
    r.input_begin = first.in_off & ~15ull;
    uint64_t in_end = 0;
    for (uint64_t t = lo; t < hi; ++t) {
        const TsTile &T = b->tiles[t];
        r.bases += T.own_len;
        if (!b->tips) r.window_end = std::max<uint64_t>(r.window_end, T.win_out + T.nwin);
        // what ts_scan_tiles loads for this tile: nch chunks of TS_CHUNK positions from the 16-byte-aligned
        // address at or below its first base, each lane 32 bytes (the last lane reaches 32 bytes past a chunk)
        const uint32_t sh = (uint32_t)(T.in_off & 15ull);
        const uint64_t span = (uint64_t)(T.nwin + b->kp.halo_blocks) * b->kp.s;
        const uint64_t need = sh + std::min<uint64_t>(T.nrel, span + 16u);
        uint64_t nch = (need + 16u + TS_CHUNK - 1u) / TS_CHUNK;
        nch = std::min<uint64_t>(std::max<uint64_t>(nch, 1), b->kp.nch);
        in_end = std::max<uint64_t>(in_end, (T.in_off - sh) + nch * TS_CHUNK + 64u);
    }
    r.input_end = std::min<uint64_t>(in_end, b->input_bytes);
}

}  // namespace

void ts_ctx::bind_this_thread() const {
    if (node_cpus.empty()) return;
    cpu_set_t cur, want;
    CPU_ZERO(&cur); CPU_ZERO(&want);
    if (pthread_getaffinity_np(pthread_self(), sizeof cur, &cur) != 0) return;
    int n = 0;
    for (int k : node_cpus)
        if (k < CPU_SETSIZE && CPU_ISSET(k, &cur)) { CPU_SET(k, &want); ++n; }
    if (n > 0) (void)pthread_setaffinity_np(pthread_self(), sizeof want, &want);       // (threads started from here inherit it)
}

// The buffers of a batch that emits (ts_batch_set_emit): per-wave regions of visible records, the per-tile chain summaries
// and, planned here, every tile's terminal-zone word.  Idempotent.
static int ensure_emit_buffers(ts_batch *b) {
    ts_ctx *c = b->ctx;
    if (!b->kp.emit || b->d_chain.p) return TS_OK;
    const size_t nt = (size_t)b->range_tiles();
    {
        HIP_TRY(c, c->pool.take(std::max<uint64_t>((uint64_t)b->vis_cap * b->total_waves, 4) * (b->kp.vis_wide ? 4 : 2) + 16, b->d_vis));
        HIP_TRY(c, c->pool.take((nt + 1) * 16, b->d_chain));
        if (b->kp.emit == 2u) return TS_OK;                 // (a read batch: no terminal-zone words, every record is in the zone)
        HIP_TRY(c, c->pool.take((nt + 1) * 4, b->d_zone));
        // terminal zone of a segment (isTerminal, src/teloscope.cpp:451-459): rel <= t || rel >= N - t, the whole of a
        // segment no longer than t; per tile as two 16-bit thresholds on the tile-relative position
        std::vector<uint32_t> zone(nt + 1, TS_ZONE_NONE);
        const uint64_t tl = c->params.terminal_limit;
        for (size_t i = 0; i < nt; ++i) {
            const TsTile &T = b->tiles[b->tile_lo + i];
            const SegPlan &sp = b->segs[T.seg];
            const uint64_t rel0 = T.in_off - sp.in_off;
            uint64_t zlo = rel0 <= tl ? tl + 1 - rel0 : 0, zhi = 0;
            if (sp.len > tl) zhi = sp.len - tl > rel0 ? sp.len - tl - rel0 : 0;
            zone[i] = (uint32_t)std::min<uint64_t>(zlo, 0xFFFF) | ((uint32_t)std::min<uint64_t>(zhi, 0xFFFF) << 16);
        }
        HIP_TRY(c, hipMemcpy(b->d_zone.p, zone.data(), (nt + 1) * 4, hipMemcpyHostToDevice));
    }
    return TS_OK;
}

// Allocates (from the context's pool) the device state of the batch's tile range; idempotent.
int ts_batch_ensure_device(ts_batch *b) {
    ts_ctx *c = b->ctx;
    if (b->allocated) return TS_OK;
    DEVICE_TRY(c);
    const size_t nt = (size_t)b->range_tiles();
    if (b->evs.empty()) {
        b->evs.assign(2 * kEventRing, nullptr);
        for (hipEvent_t &e : b->evs) HIP_TRY(c, hipEventCreate(&e));
    }
    HIP_TRY(c, c->pool.take(std::max<size_t>(nt, 1) * sizeof(TsTile), b->d_tiles));
    if (!b->ext_windows) HIP_TRY(c, c->pool.take(std::max<uint64_t>(b->win_hi - b->win_lo, 1) * 32, b->d_windows));
    HIP_TRY(c, c->pool.take(std::max<uint64_t>((uint64_t)b->region_cap * b->total_waves, 4) * 4 + 16, b->d_matches));   // + 16: the read predicate fetches whole aligned 16-byte blocks
    HIP_TRY(c, c->pool.take((nt + 1) * 8, b->d_tile_off));
    if (!b->ext_stats) HIP_TRY(c, c->pool.take((nt + 1) * 16, b->d_stats));
    HIP_TRY(c, c->pool.take((size_t)b->total_waves * 8 + 16, b->d_fill));       // records needed per wave, then visible records needed per wave
    HIP_TRY(c, c->pool.take(2 * TS_MAX_TICKET_GROUPS * TS_TICKET_STRIDE * 4, b->d_tickets));
    HIP_TRY(c, hipMemset(b->d_tickets.p, 0, 2 * TS_MAX_TICKET_GROUPS * TS_TICKET_STRIDE * 4));
    if (nt) HIP_TRY(c, hipMemcpy(b->d_tiles.p, b->tiles.data() + b->tile_lo, nt * sizeof(TsTile), hipMemcpyHostToDevice));
    if (ts_k_prepare(b->lds_bytes) != 0) return c->fail(TS_ERR_HIP, "cannot raise dynamic LDS limit");
    b->allocated = true;
    return TS_OK;
}

void ts_batch_release_input(ts_batch *b) {
    if (b && b->d_in.p) b->ctx->pool.give(std::move(b->d_in));
}

namespace {

// Sizes the launch (one or two persistent workgroups per CU, whose waves take tiles or are dealt them) and the per-wave
// record regions for the batch's tile range.  Every wave appends to its own region of the match buffer;
// small ranges get the worst case (every base a match), large ones bases/4 spread evenly, grown on overflow
// by ts_batch_sync (worst case for wave w = the owned bases of the tiles it is dealt: t = w, w + waves, ...).
void size_launch(ts_batch *b) {
    const uint32_t wpw = b->kp.waves_per_wg;
    const uint64_t nt = b->range_tiles();
    const int ncu = b->ctx->num_cu > 0 ? b->ctx->num_cu : 256;
    b->grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(ceil_div(nt, wpw), 1), (uint64_t)ncu * std::max<uint32_t>(b->kp.wgs_per_cu, 1));
    b->total_waves = b->grid * wpw;
    uint64_t worst = 256;
    {
        std::vector<uint64_t> per_wave(b->total_waves, 0);
        for (uint64_t t = 0; t < nt; ++t) per_wave[t % b->total_waves] += b->tiles[b->tile_lo + t].own_len;
        for (uint64_t v : per_wave) worst = std::max(worst, v);
    }
    const uint64_t want = b->match_cap_request ? b->match_cap_request : b->range_bases / 4 + 4096;
    uint64_t cap = ceil_div(want, b->total_waves);
    // Tiles are taken on demand by the waves (see ts_scan_tiles); dealt round-robin to small ranges, whose regions are
    // sized for the worst case of the tiles a wave is dealt, and to the rescans after an overflow.  (TS_DEALT_TIPS=1
    // deals the tiles of tips / read batches: the read predicate's sensitivity to the record placement was measured with it,
    // profiles/r02/reads_taken_vs_dealt.txt.)
    b->dealt_tiles = b->tips && ts_env_flag("TS_DEALT_TIPS");
    if (b->range_bases <= (64ull << 20) && !b->match_cap_request) { cap = worst; b->dealt_tiles = true; }
    cap = std::min<uint64_t>(std::max<uint64_t>(cap, 256), worst);
    b->region_cap = (uint32_t)((cap + 3) & ~3ull);
    b->match_cap = (uint64_t)b->region_cap * b->total_waves;
    // visible records (kp.emit): canonical matches at their density on random sequence, every match of a terminal zone
    // (a few per cent of its bases; every k-th base inside a telomere), twice the even share per wave, grown on overflow
    b->vis_cap = 0;
    if (b->tips && b->kp.emit == 2u) {
        // a read batch: the canonical records' indices — their density on random sequence, twice the even share per wave, and
        // room for a few telomeric reads (a canonical record every k bases over thousands of bases); grown on overflow
        const ts_ctx *c = b->ctx;
        uint64_t ncanon = 0;
        for (const ts::Pattern &p : c->patterns) ncanon += p.is_canonical ? 1 : 0;
        const double d_canon = (double)std::max<uint64_t>(ncanon, 1) / (double)(1ull << (2 * std::min<uint32_t>(c->k, 16)));
        const uint64_t vwant = (uint64_t)((double)b->range_bases * d_canon * 1.5);
        uint64_t vcap = 2 * ceil_div(vwant, b->total_waves) + 8192;
        if (b->dealt_tiles && !b->match_cap_request) vcap = worst;
        vcap = std::min<uint64_t>(vcap, std::max<uint64_t>(worst, 256));
        if (const char *e = getenv("TS_VIS_CAP")) { const long v = atol(e); if (v > 0) vcap = (uint64_t)v; }
        b->vis_cap = (uint32_t)((vcap + 7) & ~7ull);
    } else if (!b->tips) {
        const ts_ctx *c = b->ctx;
        uint64_t zone_bases = 0;
        const uint64_t tl = c->params.terminal_limit;
        for (uint64_t t = 0; t < nt; ++t) {
            const TsTile &T = b->tiles[b->tile_lo + t];
            const SegPlan &sp = b->segs[T.seg];
            const uint64_t rel0 = T.in_off - sp.in_off, rel1 = rel0 + T.own_len;
            if (rel0 <= tl || sp.len <= tl || rel1 > sp.len - tl) zone_bases += T.own_len;
        }
        uint64_t ncanon = 0;
        for (const ts::Pattern &p : c->patterns) ncanon += p.is_canonical ? 1 : 0;
        const double d_canon = (double)std::max<uint64_t>(ncanon, 1) / (double)(1ull << (2 * std::min<uint32_t>(c->k, 16)));
        const uint64_t vwant = (uint64_t)((double)b->range_bases * d_canon * 1.5) + zone_bases / 8;
        // (which wave scans which tile is decided at run time: a wave may take several of the few tiles that lie in a telomere,
        // where every k-th base is a visible record — room for a dozen of those per wave, so that an overflow, which costs a
        // rescan with dealt tiles, stays an event for inputs that are telomere through and through)
        const uint64_t tile_bases = (uint64_t)b->wpt * std::max<uint32_t>(b->kp.s, 1);
        uint64_t vcap = 2 * ceil_div(vwant, b->total_waves) + 12 * ceil_div(tile_bases, std::max<uint32_t>(c->k, 1)) + 2048;
        if (b->dealt_tiles && !b->match_cap_request) vcap = worst;          // small ranges: the worst case, like the record regions
        vcap = std::min<uint64_t>(vcap, std::max<uint64_t>(worst, 256));
        if (const char *e = getenv("TS_VIS_CAP")) { const long v = atol(e); if (v > 0) vcap = (uint64_t)v; }   // (tests: force the overflow -> regrow -> rescan path)
        b->vis_cap = (uint32_t)((vcap + 7) & ~7ull);
    }
}

void set_range(ts_batch *b, uint64_t lo, uint64_t hi) {
    ts_range_info r;
    range_cover(b, lo, hi, r);
    b->tile_lo = lo; b->tile_hi = hi;
    b->win_lo = r.window_begin; b->win_hi = r.window_end;
    b->in_lo = r.input_begin; b->in_hi = r.input_end;
    b->range_bases = r.bases;
    if (lo == 0 && hi == b->tiles.size()) {          // the whole plan: every byte of the layout, every window
        b->in_lo = 0; b->in_hi = b->input_bytes;
        b->win_lo = 0; b->win_hi = b->n_windows;
    }
    size_launch(b);
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
    if (!canonical_in || !fwd_out || !rev_out || std::strlen(canonical_in) > 63) return TS_ERR_INVALID_ARG;
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
    for (const char *p = raw_csv; *p;) {                        // a seed longer than a ts_pattern holds: refused, never dropped or truncated
        const char *q = std::strchr(p, ',');
        const size_t len = q ? (size_t)(q - p) : std::strlen(p);
        if (len > 63) return TS_ERR_UNSUPPORTED;
        p += len + (q ? 1 : 0);
    }
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
    for (size_t i = 0; i < n_patterns; ++i)
        if (patterns[i].len > 63) {
            g_create_error = "ts_create: unsupported pattern: longer than the 63 bases a ts_pattern holds";
            return nullptr;
        }
    const bool plan_only = params->device == kNoDevice;
    int ndev = 0;
    if (!plan_only && (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)) {
        g_create_error = "ts_create: no usable HIP device (libteloscan has no CPU fallback)";
        return nullptr;
    }
    ts_ctx *c = new ts_ctx();
    c->params = *params;
    c->read_filter = read_filter;
    read_env_knobs(c);
    if (read_filter) {                                   // makeReadFilterInput, src/read-filter.cpp:10-30
        if (!min_block_len_set) c->params.min_block_len = 42;
        c->params.terminal_limit = std::numeric_limits<uint32_t>::max() / 2;
        c->params.out_gc = c->params.out_entropy = c->params.out_matches = c->params.out_its = 0;
        c->params.fold_case = 1;
    }
    if (plan_only) {
        c->device = kNoDevice;
        c->num_cu = 256;                                 // MI355X: the plan of a batch does not depend on it (the launch grid does)
    } else {
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
        c->node_cpus = device_node_cpus(dev);
    }

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
    if (c->params.out_entropy && c->params.window_size && c->params.window_size <= (1u << 22)) ts::entropy_terms(c->params.window_size, c->entropy_term);
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
        // beyond 8 lengths or 32 bases: the wide form's tables (whatever a ts_pattern[] can hold: 63 lengths of up to 63 bases)
        if (!ok && lens.size() <= 63 && lens.back() <= 63 && lens.front() >= 1) {
            bool acgt = true;
            std::vector<unsigned long long> lo, hi;
            std::vector<uint8_t> wflags;
            std::vector<uint32_t> first;
            for (size_t li = 0; li < lens.size() && acgt; ++li) {
                first.push_back((uint32_t)lo.size());
                std::vector<std::tuple<unsigned long long, unsigned long long, uint8_t>> v;
                for (const ts::Pattern &p : c->patterns) {
                    if (p.seq.size() != lens[li]) continue;
                    unsigned long long a = 0, b = 0;
                    for (size_t i = 0; i < p.seq.size(); ++i) {
                        const int code = ts::base_code(p.seq[i]);
                        if (code < 0) { acgt = false; break; }
                        if (i < 32) a |= (unsigned long long)code << (2 * i); else b |= (unsigned long long)code << (2 * (i - 32));
                    }
                    v.emplace_back(a, b, (uint8_t)((p.is_forward ? 1 : 0) | (p.is_canonical ? 2 : 0)));
                }
                std::sort(v.begin(), v.end());
                for (const auto &e : v) { lo.push_back(std::get<0>(e)); hi.push_back(std::get<1>(e)); wflags.push_back(std::get<2>(e)); }
            }
            first.push_back((uint32_t)lo.size());
            if (acgt && plan_only) {
                c->generic_ok = true; c->gen_wide = true; c->wide_lens = lens;
            } else if (acgt && c->d_wlo.ensure(lo.size() * 8 + 16) == hipSuccess && c->d_whi.ensure(hi.size() * 8 + 16) == hipSuccess &&
                       c->d_wflags.ensure(wflags.size() + 16) == hipSuccess && c->d_wlen.ensure(std::max<size_t>(lens.size(), 64) * 4 + 16) == hipSuccess &&       // (read as a table of 64: blockcall.hip)
                       hipMemset(c->d_wlen.p, 0, std::max<size_t>(lens.size(), 64) * 4) == hipSuccess &&
                       c->d_wfirst.ensure(first.size() * 4 + 16) == hipSuccess &&
                       hipMemcpy(c->d_wlo.p, lo.data(), lo.size() * 8, hipMemcpyHostToDevice) == hipSuccess &&
                       hipMemcpy(c->d_whi.p, hi.data(), hi.size() * 8, hipMemcpyHostToDevice) == hipSuccess &&
                       hipMemcpy(c->d_wflags.p, wflags.data(), wflags.size(), hipMemcpyHostToDevice) == hipSuccess &&
                       hipMemcpy(c->d_wlen.p, lens.data(), lens.size() * 4, hipMemcpyHostToDevice) == hipSuccess &&
                       hipMemcpy(c->d_wfirst.p, first.data(), first.size() * 4, hipMemcpyHostToDevice) == hipSuccess) {
                c->wpat.lo = (const unsigned long long *)c->d_wlo.p; c->wpat.hi = (const unsigned long long *)c->d_whi.p;
                c->wpat.flags = (const uint8_t *)c->d_wflags.p; c->wpat.len = (const uint32_t *)c->d_wlen.p;
                c->wpat.first = (const uint32_t *)c->d_wfirst.p;
                c->wpat.nlen = (uint32_t)lens.size(); c->wpat.npat = (uint32_t)lo.size();
                c->wide_lens = lens;
                c->generic_ok = true; c->gen_wide = true;
            }
        }
        if (ok && plan_only) {
            c->generic_ok = true;                        // (tables stay on the host: nothing is ever launched)
        } else if (ok && c->d_gcodes.ensure(codes.size() * 8) == hipSuccess && c->d_gflags.ensure(flags.size() + 16) == hipSuccess &&
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
            if (!plan_only && (c->d_table.ensure(table.size() * 4) != hipSuccess ||
                hipMemcpy(c->d_table.p, table.data(), table.size() * 4, hipMemcpyHostToDevice) != hipSuccess)) {
                g_create_error = "ts_create: cannot upload match table";
                delete c;
                return nullptr;
            }
            // TS_FORCE_GENERAL=1 (read when the context is made): the general kernels take every scan of this context although the
            // tiled kernel could — two independent device implementations of one parameter set, compared at full size by the tests
            const char *fg = getenv("TS_FORCE_GENERAL");
            if (fg && fg[0] == '1' && c->generic_ok) c->why_not = "TS_FORCE_GENERAL: the general kernels take every scan of this context";
            else c->fast_ok = true;
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
    if (ctx->device != kNoDevice) {
        DeviceGuard g(ctx->device);
        ctx->d_table.release();
        ctx->d_gcodes.release();
        ctx->d_gflags.release();
        ctx->d_wlo.release(); ctx->d_whi.release(); ctx->d_wflags.release(); ctx->d_wlen.release(); ctx->d_wfirst.release();
        ctx->pool.clear();
        for (int i = 0; i < ts_ctx::kUpSlots; ++i) {
            ctx->pin_up[i].release();
            if (ctx->pin_up_ev[i]) (void)hipEventDestroy(ctx->pin_up_ev[i]);
            ctx->d_pack[i].release(); ctx->d_runs[i].release(); ctx->pin_runs[i].release();
        }
        for (hipEvent_t e : ctx->gen_ev) if (e) (void)hipEventDestroy(e);
        for (PinBuf &pb : ctx->pin_down) pb.release();
        for (hipStream_t st : {ctx->up_stream, ctx->scan_stream, ctx->down_stream, ctx->side_stream})
            if (st) (void)hipStreamDestroy(st);
        delete ctx;
        return;
    }
    delete ctx;
}

}  // extern "C"

// The tiled kernel implements the closed form of the reference's carry loop, valid when
// w == s or longest <= min(s, w - s); outside it the reference's uint32
// arithmetic wraps and only a literal emulation reproduces it.
bool ts_full_scan_supported(const ts_ctx *c, std::string &why) {
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

extern "C" {

int ts_uses_fast_path(const ts_ctx *ctx) {
    std::string why;
    return ctx && ts_full_scan_supported(ctx, why) ? 1 : 0;
}

int ts_pack_bases(const char *src, uint64_t n, int fold_case, uint8_t *dst, ts_invalid_run *runs, uint64_t run_cap, uint64_t *n_runs) {
    if ((!src || !dst) && n) return TS_ERR_INVALID_ARG;
    if (n > 0xFFFFFFFFull) return TS_ERR_INVALID_ARG;            // (run positions are 32 bits: pack in pieces)
    ts::PackRuns R;
    ts::pack_bases((const unsigned char *)src, (size_t)n, dst, fold_case != 0, 0u, R);
    R.finish();
    if (n_runs) *n_runs = R.runs.size();
    if (R.runs.size() > run_cap) return TS_ERR_INVALID_ARG;
    for (size_t i = 0; i < R.runs.size(); ++i) { runs[i].start = R.runs[i].start; runs[i].len = R.runs[i].len; }
    return TS_OK;
}

int ts_bind_thread_to_device(const ts_ctx *ctx) {
    if (!ctx || ctx->node_cpus.empty()) return 0;
    cpu_set_t before, after;
    CPU_ZERO(&before); CPU_ZERO(&after);
    if (pthread_getaffinity_np(pthread_self(), sizeof before, &before) != 0) return 0;
    ctx->bind_this_thread();
    if (pthread_getaffinity_np(pthread_self(), sizeof after, &after) != 0) return 0;
    for (int k : ctx->node_cpus) if (k < CPU_SETSIZE && CPU_ISSET(k, &after)) {
        // bound iff nothing outside the node is left in the mask
        for (int j = 0; j < CPU_SETSIZE; ++j)
            if (CPU_ISSET(j, &after) && !std::binary_search(ctx->node_cpus.begin(), ctx->node_cpus.end(), j)) return 0;
        return 1;
    }
    return 0;
}

// Box calibration for measurements (bench.py's roofline.box): what this device issues (wave-instructions per ns, hand-written
// independent v_and_b32 at four waves per SIMD) and streams (read + write bytes per ns of a 16-byte grid-strided copy of 1 GiB).
int ts_box_probe(ts_ctx *ctx, double *valu_wave_instr_per_ns, double *copy_bytes_per_ns) {
    if (!ctx || !valu_wave_instr_per_ns || !copy_bytes_per_ns) return TS_ERR_INVALID_ARG;
    DEVICE_TRY(ctx);
    const unsigned long long bytes = 1ull << 30;
    // (its own allocation, freed on return: handed to the context's pool the 2 GiB would stay held for the rest of the process)
    struct Scratch { void *p = nullptr; ~Scratch() { if (p) (void)hipFree(p); } } scratch;
    HIP_TRY(ctx, hipMalloc(&scratch.p, (size_t)(2 * bytes)));
    HIP_TRY(ctx, hipMemset(scratch.p, 0x41, (size_t)bytes));
    const int e = ts_k_box_probe(scratch.p, bytes, ctx->num_cu, valu_wave_instr_per_ns, copy_bytes_per_ns, nullptr);
    if (e != 0) return ctx->fail(TS_ERR_HIP, "box probe failed");
    return TS_OK;
}

int ts_streams_concurrent(ts_ctx *ctx, void *stream_a, void *stream_b) {
    if (!ctx) return TS_ERR_INVALID_ARG;
    DEVICE_TRY(ctx);
    int ok = 0;
    if (ts_k_streams_concurrent(stream_a, stream_b, &ok) != 0) return ctx->fail(TS_ERR_HIP, "stream probe failed");
    return ok;
}

int ts_refresh_env(ts_ctx *ctx) {
    if (!ctx) return TS_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(ctx->api_mtx);          // (not under a running call)
    read_env_knobs(ctx);
    return TS_OK;
}

int ts_takes_text_input(const ts_ctx *ctx, int tips_only) {
    // (round 4: the general path stages its groups through the same upload as the tiled path — every format, every set)
    std::string why;
    return ctx && ((tips_only ? ctx->fast_ok : ts_full_scan_supported(ctx, why)) || ctx->generic_ok) ? 1 : 0;
}

// =========================================================================== batches
// Planning is host work only (it also runs on a planning-only context); device state is allocated from the
// context's pool when the batch first needs it.
ts_batch *ts_batch_create(ts_ctx *ctx, const uint64_t *seg_lens, const uint64_t *abs_pos,
                          size_t n_segs, int tips_only, uint64_t match_capacity) {
    if (!ctx) return nullptr;
    if (n_segs && !seg_lens) { ctx->fail(TS_ERR_INVALID_ARG, "ts_batch_create: null seg_lens"); return nullptr; }
    std::string why;
    if (tips_only ? !ctx->fast_ok : !ts_full_scan_supported(ctx, why)) {
        ctx->fail(TS_ERR_UNSUPPORTED, "unsupported parameter set: " + (tips_only ? ctx->why_not : why));
        return nullptr;
    }

    ts_batch *b = new ts_batch();
    b->ctx = ctx;
    b->tips = tips_only != 0;
    uint32_t wpt = 1;
    if (!plan_geometry(ctx, b->tips, b->kp, wpt, why)) {
        ctx->fail(TS_ERR_UNSUPPORTED, "unsupported parameter set: " + why);
        delete b;
        return nullptr;
    }
    b->wpt = wpt;
    if (getenv("TS_TIMING"))
        fprintf(stderr, "ts_batch_create: %s scan, k=%u (%s pair table), %u x %u waves per CU, %u chunks and %u windows per tile, LDS %d B per workgroup "
                        "(match queue %u, record stage %u, %u accumulator copies)\n", b->tips ? "tips-only" : "window", ctx->k,
                ctx->pair_byte_table ? "byte" : "2-bit",
                b->kp.wgs_per_cu, b->kp.waves_per_wg, b->kp.nch, wpt, ts_k_lds_bytes(&b->kp), (unsigned)TS_LIST, b->kp.stage_cap, b->kp.acc_copies);
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
    b->match_cap_request = match_capacity;
    if (b->tiles.size() >= 0x7FFFFFFFull) {
        ctx->fail(TS_ERR_UNSUPPORTED, "too many tiles in one batch");
        delete b;
        return nullptr;
    }
    b->lds_bytes = (uint32_t)ts_k_lds_bytes(&b->kp);
    // TS_READ_EMIT=1 (an experiment of round 5, off by default): a read filter's batches (ReadTelomereFilter::matches,
    // /root/reference/src/read-filter.cpp:10-45: tips-only, every read terminal zone as a whole) have their scans leave the indices of
    // the canonical records (kp.emit = 2), and the predicate visits only the chains that hold one (predicate.hip:
    // ts_read_predicate_canon) — a twentieth of the records, but each visit is a scattered 32-byte look at the match stream, a
    // 128-byte line of HBM traffic: 0.35 ms per 5e5 reads against the streaming walk's 0.31, and the scan pays 5 % for the indices
    // (profiles/r05/reads_canonical_index_experiment.txt).  The same pass bytes on the read-filter fuzz either way.
    if (b->tips && ctx->read_filter && b->kp.stage_u16) {
        const char *e = getenv("TS_READ_EMIT");
        if (e && e[0] == '1' && ts_k_read_index_built()) b->kp.emit = 2u;     // (only a library built with -DTS_READ_INDEX_BUILD=1 has it)
    }
    // records leave as the stage holds them: 16 bits each where tile positions fit 14 bits (TS_REC32=1 / ts_batch_set_record_bits: 32)
    b->kp.rec16 = (b->kp.stage_u16 && ctx->knobs.rec16 && b->kp.emit != 2u) ? 1u : 0u;
    set_range(b, 0, b->tiles.size());
    return b;
}

void ts_batch_destroy(ts_batch *b) {
    if (!b) return;
    ts_ctx *c = b->ctx;
    if (c->device != kNoDevice) {
        DeviceGuard g(c->device);
        for (DevBuf *d : {&b->d_in, &b->d_tiles, &b->d_windows, &b->d_matches, &b->d_tile_off, &b->d_stats, &b->d_fill, &b->d_tickets,
                          &b->d_segtab, &b->d_dense, &b->d_dense_base, &b->d_scan_tmp, &b->d_readtab, &b->d_shard_segs,
                          &b->d_shard_bounds, &b->d_shard_tmp, &b->d_shard_cand, &b->d_vis, &b->d_chain, &b->d_zone})
            c->pool.give(std::move(*d));
        for (hipEvent_t e : b->evs)
            if (e) (void)hipEventDestroy(e);
        if (b->ev_fork) (void)hipEventDestroy(b->ev_fork);
        if (b->ev_join) (void)hipEventDestroy(b->ev_join);
    }
    delete b;
}

uint64_t ts_batch_segment_offset(const ts_batch *b, size_t i) {
    return (b && i < b->segs.size()) ? b->segs[i].in_off : 0;
}

int ts_batch_get_tiles(const ts_batch *b, uint64_t first, uint64_t n, ts_tile_info *out) {
    if (!b || (n && !out) || first > b->tiles.size() || n > b->tiles.size() - first) return TS_ERR_INVALID_ARG;
    for (uint64_t i = 0; i < n; ++i) {
        const TsTile &T = b->tiles[first + i];
        out[i].seg_index = T.seg;
        out[i].seg_offset = T.in_off - b->segs[T.seg].in_off;
        out[i].first_window = b->tips ? 0 : T.win_out;
        out[i].n_windows = b->tips ? 0 : T.nwin;
        out[i].owned_bases = T.own_len;
    }
    return TS_OK;
}

int ts_batch_range_info(const ts_batch *b, uint64_t tile_begin, uint64_t tile_end, ts_range_info *out) {
    if (!b || !out || tile_begin > tile_end || tile_end > b->tiles.size()) return TS_ERR_INVALID_ARG;
    range_cover(b, tile_begin, tile_end, *out);
    return TS_OK;
}

int ts_batch_partition(const ts_batch *b, uint32_t n_parts, uint32_t part, uint64_t *tile_begin, uint64_t *tile_end) {
    if (!b || !n_parts || part >= n_parts || !tile_begin || !tile_end) return TS_ERR_INVALID_ARG;
    // consecutive ranges of equal owned bases (tiles are near-equal work), the boundaries moved — by at most the
    // terminal zone + context of a segment — so that none falls near a segment's end (shard.cpp)
    std::vector<uint64_t> cut;
    ts_shard_boundaries(b, n_parts, cut);
    *tile_begin = cut[part];
    *tile_end = cut[part + 1];
    return TS_OK;
}

int ts_batch_restrict(ts_batch *b, uint64_t tile_begin, uint64_t tile_end) {
    if (!b || tile_begin > tile_end || tile_end > b->tiles.size()) return TS_ERR_INVALID_ARG;
    if (b->allocated || b->scanned) return b->ctx->fail(TS_ERR_STATE, "ts_batch_restrict after the batch was used on the device");
    set_range(b, tile_begin, tile_end);
    return TS_OK;
}

// TS_EMIT=0 keeps every batch from emitting (A/B measurements: the interstitial search then reads every record again, as it
// does for results adopted from elsewhere)
static bool emit_allowed() { const char *e = getenv("TS_EMIT"); return !(e && e[0] == '0'); }

int ts_batch_set_emit(ts_batch *b, int on) {
    if (!b) return TS_ERR_INVALID_ARG;
    if (b->dense) return b->ctx->fail(TS_ERR_STATE, "ts_batch_set_emit on a batch that adopted results");
    if (b->tips) return TS_OK;                           // (a tips-only batch ignores it: a read batch keeps its own form, see ts_batch_create)
    b->kp.emit = (on && emit_allowed()) ? 1u : 0u;
    return TS_OK;
}

int ts_batch_bind_results(ts_batch *b, void *d_windows, void *d_tile_stats) {
    if (!b) return TS_ERR_INVALID_ARG;
    if (b->allocated || b->scanned) return b->ctx->fail(TS_ERR_STATE, "ts_batch_bind_results after the batch was used on the device");
    b->ext_windows = (uint32_t *)d_windows;
    b->ext_stats = (uint32_t *)d_tile_stats;
    return TS_OK;
}

// The kernels never look past a segment's end (the padding between segments is masked by position), so
// zero-filling only serves callers that fill the buffer piecewise and later inspect it.
static void *batch_input(ts_batch *b, bool zero_fill) {
    if (!b || b->ctx->device == kNoDevice) return nullptr;
    if (!b->d_in.p) {
        DeviceGuard g(b->ctx->device);
        const size_t bytes = (size_t)(b->in_hi - b->in_lo) + 64;
        if (b->ctx->pool.take(bytes, b->d_in) != hipSuccess) return nullptr;
        if (zero_fill) (void)hipMemset(b->d_in.p, 0, bytes);
    }
    return b->d_in.p;
}

void *ts_batch_input_ptr(ts_batch *b) { return batch_input(b, true); }
}  // extern "C"
void *ts_batch_input_ptr_nozero(ts_batch *b) { return batch_input(b, false); }
extern "C" {

int ts_batch_upload(ts_batch *b, size_t i, const char *seq) {
    if (!b || i >= b->segs.size() || (!seq && b->segs[i].len)) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    DEVICE_TRY(c);
    if (!ts_batch_input_ptr(b)) return c->fail(TS_ERR_ALLOC, "cannot allocate device input buffer");
    // the part of the segment the batch's range reads
    const uint64_t s0 = std::max<uint64_t>(b->segs[i].in_off, b->in_lo);
    const uint64_t s1 = std::min<uint64_t>(b->segs[i].in_off + b->segs[i].len, b->in_hi);
    if (s1 > s0)
        HIP_TRY(c, hipMemcpy((char *)b->d_in.p + (s0 - b->in_lo), seq + (s0 - b->segs[i].in_off), s1 - s0, hipMemcpyHostToDevice));
    return TS_OK;
}

int ts_batch_scan(ts_batch *b, const void *d_input, void *stream) {
    if (!b) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    DEVICE_TRY(c);
    if (b->dense) return c->fail(TS_ERR_STATE, "ts_batch_scan on a batch that adopted results");
    { int rc = ts_batch_ensure_device(b); if (rc != TS_OK) return rc; }
    { int rc = ensure_emit_buffers(b); if (rc != TS_OK) return rc; }
    b->emitted = b->kp.emit != 0u;
    if (!d_input) d_input = ts_batch_input_ptr(b);
    if (!d_input) return c->fail(TS_ERR_ALLOC, "no device input buffer");
    hipStream_t st = (hipStream_t)stream;
    b->last_input = d_input;
    b->last_stream = stream;
    b->scanned = true;
    b->synced = false;

    // The device sees the range's slice of every array; tile descriptors carry offsets of the WHOLE plan
    // (in_off into the input layout, win_out into the window array), so those two base pointers are shifted
    // back by what precedes the range (never dereferenced there).
    TsScanParams &kp = b->kp;
    kp.in = (const uint8_t *)d_input - b->in_lo;
    kp.tiles = (const TsTile *)b->d_tiles.p;
    kp.table = (const uint32_t *)c->d_table.p;
    kp.windows_out = b->windows_ptr() - b->win_lo * 8ull;
    kp.matches_out = (uint32_t *)b->d_matches.p;
    kp.tile_off = (unsigned long long *)b->d_tile_off.p;
    kp.tile_stats = b->stats_ptr();
    kp.wave_fill = (uint32_t *)b->d_fill.p;
    // Tiles are handed out on demand (see ts_scan_tiles) unless the per-wave record counts must be reproducible:
    // small ranges, whose regions are sized for the worst case of the tiles a wave is DEALT, and the rescans after
    // an overflow, which size the regions from the counts of the scan before.
    kp.tile_tickets = (uint32_t *)b->d_tickets.p;
    kp.ticket_groups = std::min<uint32_t>(b->grid, ticket_groups_wanted());
    kp.dynamic_tiles = (b->dealt_tiles || ts_env_flag("TS_DEALT_TILES")) ? 0u : 1u;
    kp.ticket_slot = (uint32_t)(b->ticket_seq & 1u);
    kp.region_cap = b->region_cap;
    kp.ntiles = (uint32_t)b->range_tiles();
    kp.vis_out = b->d_vis.p;
    kp.vis_cap = b->vis_cap;
    kp.tile_zone = (const uint32_t *)b->d_zone.p;
    kp.tile_chain = (uint32_t *)b->d_chain.p;
    // a shard whose message buffer is known (ts_batch_bind_shard_message): the emitting scan packs the owned windows' records
    // into the message itself (the base is shifted back by what precedes the owned windows, never dereferenced there)
    kp.win_packed = nullptr;
    b->msg_windows = nullptr;
    if (kp.emit && b->bound_msg && b->shard_parts && !b->tips && b->shard_r.own_hi > b->shard_r.own_lo && b->shard_L.window_bytes) {
        const uint64_t w0 = b->tiles[b->shard_r.own_lo].win_out;
        const TsTile &lastT = b->tiles[b->shard_r.own_hi - 1];
        kp.win_pack_lo = w0;
        kp.win_pack_hi = lastT.win_out + lastT.nwin;
        kp.win_pack_bytes = b->shard_L.window_bytes;
        kp.win_field_bits = b->shard_L.field_bits;
        kp.win_packed = (uint8_t *)b->bound_msg + b->shard_L.off_windows - w0 * (uint64_t)b->shard_L.window_bytes;
        b->msg_windows = b->bound_msg;
    }

    const size_t slot = (size_t)(b->scan_seq % kEventRing);
    static_assert(kEventRing == 64, "timed_mask holds one bit per ring slot");
    const bool timed = c->knobs.scan_events >= 2 && b->time_every && (b->scan_seq % b->time_every) == 0;
    if (timed) { HIP_TRY(c, hipEventRecord(b->evs[2 * slot], st)); b->timed_mask |= 1ull << slot; }
    else b->timed_mask &= ~(1ull << slot);
    {
        std::lock_guard<std::mutex> lk(c->mtx);
        int e = ts_k_launch_scan(&kp, b->grid, b->lds_bytes, stream);
        if (e != 0) return c->fail(TS_ERR_HIP, std::string("scan kernel launch: ") + hipGetErrorString((hipError_t)e));
    }
    if (kp.dynamic_tiles) ++b->ticket_seq;        // the launch is enqueued: it zeroes the other counter for the next one
    if (c->knobs.scan_events >= 1) HIP_TRY(c, hipEventRecord(b->evs[2 * slot + 1], st));
    ++b->scan_seq;
    return TS_OK;
}

int ts_batch_set_timing(ts_batch *b, uint32_t every) {
    if (!b) return TS_ERR_INVALID_ARG;
    b->time_every = every;
    return TS_OK;
}

int ts_batch_set_record_bits(ts_batch *b, int bits) {
    if (!b || (bits != 16 && bits != 32)) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    if (b->scanned) return c->fail(TS_ERR_STATE, "ts_batch_set_record_bits: before the first scan");
    if (bits == 32) { b->kp.rec16 = 0u; return TS_OK; }
    // 16-bit records are the stage's own entries: tile positions below 2^14 (every geometry the planner picks for w <= 8192)
    if (!b->kp.stage_u16 || b->kp.emit == 2u || b->dense)
        return c->fail(TS_ERR_UNSUPPORTED, "16-bit records need 16-bit stage entries (tile positions below 2^14)");
    b->kp.rec16 = 1u;
    return TS_OK;
}

int ts_batch_wait_scan(ts_batch *b, void *stream) {
    if (!b) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    if (!b->scanned || b->dense || b->scan_seq == 0) return c->fail(TS_ERR_STATE, "ts_batch_wait_scan needs a scanned batch");
    if (stream == b->last_stream) return TS_OK;                 // (the same stream: in order already)
    if (c->knobs.scan_events < 1) return c->fail(TS_ERR_STATE, "ts_batch_wait_scan: the scan's events are switched off (TS_SCAN_EVENTS=0)");
    DEVICE_TRY(c);
    const size_t last = (size_t)((b->scan_seq - 1) % kEventRing);
    HIP_TRY(c, hipStreamWaitEvent((hipStream_t)stream, b->evs[2 * last + 1], 0));
    return TS_OK;
}

int ts_batch_sync(ts_batch *b) {
    if (!b) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    if (b->dense && b->synced) return TS_OK;
    if (!b->scanned) return c->fail(TS_ERR_STATE, "ts_batch_sync before ts_batch_scan");
    DEVICE_TRY(c);
    for (int attempt = 0; attempt < 4; ++attempt) {
        {   // kernel times of the scans since the previous sync (the ring keeps the latest kEventRing)
            const size_t last = (size_t)((b->scan_seq - 1) % kEventRing);
            if (c->knobs.scan_events >= 1) HIP_TRY(c, hipEventSynchronize(b->evs[2 * last + 1]));
            const uint64_t from = std::max(b->harvested, b->scan_seq > kEventRing ? b->scan_seq - kEventRing : 0);
            double sum = 0.0;
            float ms = 0.f;
            uint64_t ntimed = 0;
            for (uint64_t q = from; q < b->scan_seq; ++q) {
                const size_t s2 = (size_t)(q % kEventRing);
                if (!((b->timed_mask >> s2) & 1ull)) continue;              // (ts_batch_set_timing: not every scan has a start event)
                HIP_TRY(c, hipEventElapsedTime(&ms, b->evs[2 * s2], b->evs[2 * s2 + 1]));
                sum += ms;
                ++ntimed;
            }
            if (ntimed) {
                b->avg_n = ntimed;
                b->avg_ms = sum / (double)ntimed;
                b->last_ms = ms;
            }
            b->harvested = b->scan_seq;
        }
        const uint32_t nfill = b->total_waves * (b->emitted ? 2u : 1u);          // with emit: the visible records needed follow
        b->wave_fill.assign(nfill, 0u);
        HIP_TRY(c, hipMemcpyAsync(b->wave_fill.data(), b->d_fill.p, (size_t)nfill * 4, hipMemcpyDeviceToHost, (hipStream_t)b->last_stream));
        HIP_TRY(c, hipStreamSynchronize((hipStream_t)b->last_stream));
        uint64_t total = 0;
        uint32_t worst = 0, worst_vis = 0;
        for (uint32_t w = 0; w < b->total_waves; ++w) { total += b->wave_fill[w]; worst = std::max(worst, b->wave_fill[w]); }
        for (uint32_t w = b->total_waves; w < nfill; ++w) worst_vis = std::max(worst_vis, b->wave_fill[w]);
        b->n_matches = total;
        if (worst <= b->region_cap && worst_vis <= b->vis_cap) { b->synced = true; return TS_OK; }
        // a wave's region overflowed: size the regions for the fullest wave and rescan, from now on with the tiles
        // dealt round-robin (the counts of one such scan are those of the next)
        b->dealt_tiles = true;
        if (worst > b->region_cap) {
            b->region_cap = (uint32_t)(((uint64_t)worst + worst / 8 + 64 + 3) & ~3ull);
            b->match_cap = (uint64_t)b->region_cap * b->total_waves;
            if (b->d_matches.bytes < b->match_cap * 4 + 16) {
                c->pool.give(std::move(b->d_matches));
                HIP_TRY(c, c->pool.take(b->match_cap * 4 + 16, b->d_matches));
            }
        }
        if (worst_vis > b->vis_cap) {
            b->vis_cap = (uint32_t)(((uint64_t)worst_vis + worst_vis / 8 + 64 + 7) & ~7ull);
            const size_t need = (size_t)b->vis_cap * b->total_waves * (b->kp.vis_wide ? 4 : 2) + 16;
            if (b->d_vis.bytes < need) {
                c->pool.give(std::move(b->d_vis));
                HIP_TRY(c, c->pool.take(need, b->d_vis));
            }
        }
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
    // of the range the batch executes (the whole plan unless restricted), by SURVEY 8(d): a window scan reads 1 B per base and
    // writes 32 B per window and 4 B per match; a tips-only / read batch reads 1 B per scanned base and what leaves it is one
    // bit per segment (the pass bit of a read, the handful of blocks of a contig end) — its match stream is an intermediate
    info->algorithmic_bytes = b->tips ? b->range_bases + (b->segs.size() + 7) / 8
                                      : b->range_bases + 32ull * (b->win_hi - b->win_lo) + 4ull * info->n_matches;
    info->last_kernel_ms = b->last_ms;
    info->avg_kernel_ms = b->avg_ms;
    info->kernel_launches = b->avg_n;
    return TS_OK;
}

const void *ts_batch_windows_ptr(const ts_batch *b) { return b ? b->windows_ptr() : nullptr; }
const void *ts_batch_matches_ptr(const ts_batch *b) { return b && !b->kp.rec16 ? b->records_ptr() : nullptr; }   // (16-bit records: no raw view)
const void *ts_batch_tile_stats_ptr(const ts_batch *b) { return b ? b->stats_ptr() : nullptr; }

int ts_batch_export(ts_batch *b, void *d_dense, uint64_t dense_capacity, void *d_total, void *stream) {
    if (!b || !d_dense || !d_total) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    DEVICE_TRY(c);
    if (!b->scanned || b->dense) return c->fail(TS_ERR_STATE, "ts_batch_export needs a scanned batch");
    const uint32_t nt = (uint32_t)b->range_tiles();
    HIP_TRY(c, b->d_dense_base.p && b->d_dense_base.bytes >= ((size_t)nt + 1) * 8 ? hipSuccess : c->pool.take(((size_t)nt + 1) * 8, b->d_dense_base));
    const size_t tmp_bytes = (size_t)ts_k_scan_tmp_bytes(nt);
    HIP_TRY(c, b->d_scan_tmp.p && b->d_scan_tmp.bytes >= tmp_bytes ? hipSuccess : c->pool.take(tmp_bytes, b->d_scan_tmp));
    int e = ts_k_launch_tile_order_export(b->stats_ptr(), (const unsigned long long *)b->d_tile_off.p, (const uint32_t *)b->d_matches.p,
                                          (const uint32_t *)b->d_fill.p, b->region_cap, b->total_waves, nt,
                                          (unsigned long long *)b->d_dense_base.p, b->d_scan_tmp.p, (uint32_t *)d_dense,
                                          dense_capacity, (unsigned long long *)d_total, b->records16() ? 1 : 0, stream);
    if (e != 0) return c->fail(TS_ERR_HIP, std::string("export kernel launch: ") + hipGetErrorString((hipError_t)e));
    return TS_OK;
}

int ts_batch_wire16_ok(const ts_batch *b) {
    if (!b) return 0;
    const ts_ctx *c = b->ctx;
    // records: (position << 2 | flags) with position < nch * TS_CHUNK + 64 <= 2^14; tile counts <= a tile's bases;
    // window fields: nucleotide counts <= w, covered bases = k x matches <= k x w
    const bool records = (uint64_t)b->kp.nch * TS_CHUNK + 64u <= (1u << 14);
    const bool windows = b->tips || (uint64_t)c->k * c->params.window_size <= 65535u;
    return records && windows ? 1 : 0;
}

int ts_wire_widen_u16(ts_ctx *ctx, const void *d_src_u16, void *d_dst_u32, uint64_t n, void *stream) {
    if (!ctx || (n && (!d_src_u16 || !d_dst_u32))) return TS_ERR_INVALID_ARG;
    DEVICE_TRY(ctx);
    const int e = ts_k_launch_widen_u16((const uint16_t *)d_src_u16, (uint32_t *)d_dst_u32, n, stream);
    if (e != 0) return ctx->fail(TS_ERR_HIP, std::string("widen kernel launch: ") + hipGetErrorString((hipError_t)e));
    return TS_OK;
}

int ts_batch_adopt(ts_batch *b, void *d_windows, void *d_tile_stats, const void *d_dense, uint64_t n_matches, void *stream) {
    if (!b || !d_tile_stats || (n_matches && !d_dense) || (!b->tips && b->n_windows && !d_windows)) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    DEVICE_TRY(c);
    if (!b->whole()) return c->fail(TS_ERR_STATE, "ts_batch_adopt needs an unrestricted batch");
    if (b->scanned && !b->dense) return c->fail(TS_ERR_STATE, "ts_batch_adopt on a batch that was scanned locally");
    const size_t nt = b->tiles.size();
    if (!b->d_tiles.p) {
        HIP_TRY(c, c->pool.take(std::max<size_t>(nt, 1) * sizeof(TsTile), b->d_tiles));
        if (nt) HIP_TRY(c, hipMemcpy(b->d_tiles.p, b->tiles.data(), nt * sizeof(TsTile), hipMemcpyHostToDevice));
        HIP_TRY(c, c->pool.take((nt + 1) * 8, b->d_tile_off));
        HIP_TRY(c, c->pool.take((size_t)ts_k_scan_tmp_bytes((uint32_t)nt), b->d_scan_tmp));
    }
    b->ext_windows = (uint32_t *)d_windows;
    b->ext_stats = (uint32_t *)d_tile_stats;
    b->ext_dense = (const uint32_t *)d_dense;
    // tile directory of a dense tile-ordered stream: first record of tile t = records of the tiles before it
    int e = ts_k_launch_tile_offsets(b->ext_stats, (uint32_t)nt, (unsigned long long *)b->d_tile_off.p, b->d_scan_tmp.p, stream);
    if (e != 0) return c->fail(TS_ERR_HIP, std::string("tile-offset kernel launch: ") + hipGetErrorString((hipError_t)e));
    HIP_TRY(c, hipStreamSynchronize((hipStream_t)stream));
    b->dense = true;
    b->scanned = true;
    b->synced = true;
    b->allocated = true;
    b->n_matches = n_matches;
    b->last_stream = stream;
    return TS_OK;
}

int ts_batch_segment_summary(ts_batch *b, void *d_out, void *stream) {
    if (!b || !d_out) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    DEVICE_TRY(c);
    if (!b->whole()) return c->fail(TS_ERR_STATE, "ts_batch_segment_summary needs an unrestricted batch");
    const size_t ns = b->segs.size();
    if (!b->d_segtab.p) {
        std::vector<uint32_t> first(ns + 1);
        std::vector<uint64_t> nwin(ns);
        for (size_t i = 0; i < ns; ++i) { first[i] = b->segs[i].first_tile; nwin[i] = b->segs[i].n_windows; }
        first[ns] = (uint32_t)b->tiles.size();
        const size_t bytes_first = ((ns + 1) * 4 + 15) & ~15ull;
        HIP_TRY(c, c->pool.take(bytes_first + ns * 8 + 16, b->d_segtab));
        HIP_TRY(c, hipMemcpy(b->d_segtab.p, first.data(), (ns + 1) * 4, hipMemcpyHostToDevice));
        if (ns) HIP_TRY(c, hipMemcpy((char *)b->d_segtab.p + bytes_first, nwin.data(), ns * 8, hipMemcpyHostToDevice));
    }
    const size_t bytes_first = ((ns + 1) * 4 + 15) & ~15ull;
    int e = ts_k_launch_summary(b->stats_ptr(), (const uint32_t *)b->d_segtab.p,
                                (const uint64_t *)((char *)b->d_segtab.p + bytes_first), (uint32_t)ns,
                                (unsigned long long *)d_out, stream);
    if (e != 0) return c->fail(TS_ERR_HIP, "summary kernel launch failed");
    return TS_OK;
}

}  // extern "C"

// --------------------------------------------------------------- shared host post-processing
// Turns one segment's raw results into SegmentData: window records (float metrics evaluated on
// the host from the integer counts, as the reference does), terminal flags, block calling
// (src/teloscope.cpp:642-657).  `matches` arrive with absolute positions and FORWARD/CANONICAL set.
// `matches` (malloc'd by the caller, position-ordered, terminal flags not yet set; may be null when nm == 0)
// becomes the segment's match array.
int ts_finalize_segment(ts_ctx *c, bool tips, uint64_t seg_len, uint64_t abs_pos,
                        const uint32_t *win_raw, uint64_t n_windows, ts_match *matches, uint64_t nm,
                        ts_segment_out &o, unsigned spare_threads, const TsDevBlock *pre_blocks, size_t n_pre, bool have_pre) {
    const ts_params &P = c->params;
    std::memset(&o, 0, sizeof o);
    if (!tips && n_windows) {
        o.windows = (ts_window *)std::malloc(n_windows * sizeof(ts_window));
        if (!o.windows) return c->fail(TS_ERR_ALLOC, "out of host memory");
        o.n_windows = n_windows;
        const bool nuc = P.out_gc || P.out_entropy;
        // windows are independent: a segment with very many of them (a multi-gigabase contig) is split over the
        // host threads its job can spare
        auto convert = [&](uint64_t k0, uint64_t k1) {
        for (uint64_t kwin = k0; kwin < k1; ++kwin) {
            const uint32_t *r = &win_raw[kwin * 8];
            ts_window &w = o.windows[kwin];
            std::memset(&w, 0, sizeof w);
            const uint64_t ws = kwin * P.step;
            w.window_start = abs_pos + ws;
            w.current_window_size = (uint32_t)std::min<uint64_t>(P.window_size, seg_len - ws);
            if (nuc) for (int i = 0; i < 4; ++i) w.nucleotide_counts[i] = r[i];
            if (P.out_gc) w.gc_content = ts::gc_content(w.nucleotide_counts, w.current_window_size);
            if (P.out_entropy) w.shannon_entropy = ts::shannon_entropy_memo(w.nucleotide_counts, w.current_window_size, c->entropy_term);
            w.canonical_covered = r[4];
            w.non_canonical_covered = r[5];
            w.fwd_covered = r[6];
            w.rev_covered = r[7];
        }
        };
        const unsigned nth = n_windows >= (1u << 16) ? std::min<unsigned>(spare_threads, (unsigned)(n_windows >> 14)) : 1u;
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
    {
        auto flag = [&](uint64_t i0, uint64_t i1, uint64_t *fwd_out) {
            uint64_t f = 0;
            for (uint64_t i = i0; i < i1; ++i) {                    // isTerminal, src/teloscope.cpp:451-459
                ts_match &m = o.matches[i];
                const uint64_t rel = m.position - abs_pos;
                if (rel <= P.terminal_limit || rel >= term_end) m.flags |= TS_MATCH_TERMINAL;
                f += (m.flags & TS_MATCH_FORWARD) ? 1u : 0u;
            }
            *fwd_out = f;
        };
        const unsigned nth = nm >= (1u << 20) ? std::max(1u, std::min<unsigned>(spare_threads, (unsigned)(nm >> 18))) : 1u;
        std::vector<uint64_t> f(nth, 0);
        if (nth <= 1u) flag(0, nm, &f[0]);
        else {
            std::vector<std::thread> pool;
            const uint64_t share = (nm + nth - 1) / nth;
            for (unsigned t = 0; t < nth; ++t)
                pool.emplace_back(flag, std::min<uint64_t>(nm, t * share), std::min<uint64_t>(nm, (t + 1) * share), &f[t]);
            for (std::thread &th : pool) th.join();
        }
        for (uint64_t v : f) nfwd += v;
    }
    // the two walks take their orientation's records out of the one position-ordered array (they leave the
    // terminal zone after a few thousand records: no per-orientation index lists of the whole segment)
    std::vector<ts_block> term, its;
    if (have_pre) {
        // the blocks were called on the device (sorted: terminal blocks in push order, then interstitial blocks by start)
        for (size_t q = 0; q < n_pre; ++q) {
            ts_block b;
            std::memcpy(&b, &pre_blocks[q], sizeof b);
            (pre_blocks[q].kind == 2 ? its : term).push_back(b);
        }
    }
    uint64_t fwd_boundary = abs_pos, rev_boundary = abs_pos + seg_len;
    if (have_pre) {
    } else if (nfwd >= 2)
        fwd_boundary = ts::terminal_blocks(c->bp, o.matches, nullptr, nm, term, seg_len, abs_pos, true, 1);
    if (!have_pre && nm - nfwd >= 2)
        rev_boundary = ts::terminal_blocks(c->bp, o.matches, nullptr, nm, term, seg_len, abs_pos, false, 0);
    if (!have_pre && !tips && fwd_boundary < rev_boundary && nm >= 2)
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

namespace {

// Host landing area of a download: pinned memory of the context when the download fits it (a group of the
// host entry points' pipeline: DMA at link rate, no runtime staging), plain heap memory otherwise.
struct HostLanding {
    std::unique_ptr<char[]> heap;
    char *base = nullptr;
    size_t used = 0, cap = 0;
    static constexpr size_t kPinnedMax = 768ull << 20;
    HostLanding(PinBuf &pin, size_t bytes) {
        bytes += 256;
        if (bytes <= kPinnedMax && pin.ensure(std::max<size_t>(bytes + bytes / 4, 32u << 20)) == hipSuccess) {
            base = (char *)pin.p;
        } else {
            (void)hipGetLastError();
            heap.reset(new char[bytes]);            // (not zero-filled)
            base = heap.get();
        }
        cap = bytes;
    }
    void *carve(size_t bytes) {
        void *p = base + used;
        used += (bytes + 63) & ~(size_t)63;
        return used <= cap ? p : nullptr;
    }
};

}  // namespace

// A tile's packed records -> ts_match: one 16-byte store per record {position; match_size | flags << 16}, past the caches
// (the 1.5 GB of a 3 Gb scan's records are read by the caller much later; an ordinary store would first fetch every line
// it fills).  m is 16-byte aligned.  rec = (tile-relative position << 2) | forward << 1 | canonical.
static_assert(sizeof(ts_match) == 16 && TS_MATCH_FORWARD == 1 && TS_MATCH_CANONICAL == 2 && TS_MATCH_TERMINAL == 4, "record layout");

__attribute__((target("avx2")))
static void expand_records_avx2(const uint32_t *recs, uint32_t cnt, ts_match *m, uint64_t abs_pos, uint64_t rel0,
                                uint64_t terminal_limit, uint64_t term_end, uint16_t klen) {
    const __m256i vrel0 = _mm256_set1_epi64x((long long)rel0), vabs = _mm256_set1_epi64x((long long)abs_pos);
    const __m256i vtl = _mm256_set1_epi64x((long long)terminal_limit), vte = _mm256_set1_epi64x((long long)term_end - 1);
    const __m256i one = _mm256_set1_epi64x(1), vk = _mm256_set1_epi64x((long long)klen), four16 = _mm256_set1_epi64x(4ll << 16);
    uint32_t i = 0;
    for (; i + 4 <= cnt; i += 4) {
        const __m256i r = _mm256_cvtepu32_epi64(_mm_loadu_si128((const __m128i *)(recs + i)));
        const __m256i rel = _mm256_add_epi64(vrel0, _mm256_srli_epi64(r, 2));
        // terminal: rel <= limit or rel >= term_end  (positions are far below 2^63: signed compares)
        const __m256i interior = _mm256_and_si256(_mm256_cmpgt_epi64(rel, vtl), _mm256_cmpgt_epi64(vte, _mm256_sub_epi64(rel, one)));
        // flags: forward (record bit 1) -> bit 0, canonical (record bit 0) -> bit 1, terminal -> bit 2; all shifted to bit 16
        const __m256i fwd = _mm256_slli_epi64(_mm256_and_si256(_mm256_srli_epi64(r, 1), one), 16);
        const __m256i can = _mm256_slli_epi64(_mm256_and_si256(r, one), 17);
        const __m256i ter = _mm256_andnot_si256(interior, four16);
        const __m256i w1 = _mm256_or_si256(_mm256_or_si256(vk, fwd), _mm256_or_si256(can, ter));
        const __m256i pos = _mm256_add_epi64(vabs, rel);
        const __m256i lo = _mm256_unpacklo_epi64(pos, w1), hi = _mm256_unpackhi_epi64(pos, w1);   // {p0 w0 | p2 w2}, {p1 w1 | p3 w3}
        _mm256_stream_si256((__m256i *)(m + i), _mm256_permute2x128_si256(lo, hi, 0x20));
        _mm256_stream_si256((__m256i *)(m + i + 2), _mm256_permute2x128_si256(lo, hi, 0x31));
    }
    for (; i < cnt; ++i) {
        const uint32_t rec = recs[i];
        const uint64_t rel = rel0 + (rec >> 2);
        const uint64_t fl = ((rec & 2u) ? TS_MATCH_FORWARD : 0u) | ((rec & 1u) ? TS_MATCH_CANONICAL : 0u) |
                            ((rel <= terminal_limit || rel >= term_end) ? TS_MATCH_TERMINAL : 0u);
        _mm_stream_si128((__m128i *)&m[i], _mm_set_epi64x((long long)((uint64_t)klen | (fl << 16)), (long long)(abs_pos + rel)));
    }
}

static void expand_records(const uint32_t *recs, uint32_t cnt, ts_match *m, uint64_t abs_pos, uint64_t rel0,
                           uint64_t terminal_limit, uint64_t term_end, uint16_t klen) {
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    auto one = [&](uint32_t i) {
        const uint32_t rec = recs[i];
        const uint64_t rel = rel0 + (rec >> 2);
        const uint64_t fl = ((rec & 2u) ? TS_MATCH_FORWARD : 0u) | ((rec & 1u) ? TS_MATCH_CANONICAL : 0u) |
                            ((rel <= terminal_limit || rel >= term_end) ? TS_MATCH_TERMINAL : 0u);
        _mm_stream_si128((__m128i *)&m[i], _mm_set_epi64x((long long)((uint64_t)klen | (fl << 16)), (long long)(abs_pos + rel)));
    };
    uint32_t i = 0;
    if (have_avx2 && term_end >= 1 && term_end < (1ull << 62) && terminal_limit < (1ull << 62)) {
        if (cnt && (((uintptr_t)m) & 31u)) { one(0); i = 1; }              // (to the 32-byte boundary the 4-record rounds store at)
        if (i < cnt) expand_records_avx2(recs + i, cnt - i, m + i, abs_pos, rel0, terminal_limit, term_end, klen);
        return;
    }
    for (; i < cnt; ++i) one(i);
}

// Result arrays of many megabytes are first touched by the threads that fill them; on 2 MB pages (when the
// kernel grants them) that is 512 times fewer page faults.  free() releases them like any malloc'd block.
void *ts_alloc_large(size_t bytes) {
    constexpr size_t kHuge = size_t(2) << 20;
    if (bytes < 4 * kHuge) return std::malloc(bytes);
    void *p = nullptr;
    if (posix_memalign(&p, kHuge, (bytes + kHuge - 1) & ~(kHuge - 1)) != 0) return std::malloc(bytes);
    (void)madvise(p, (bytes + kHuge - 1) & ~(kHuge - 1), MADV_HUGEPAGE);
    return p;
}

namespace {

// f(i) for i in [0, n) on up to max_threads host threads (dynamic: an atomic counter hands out the indices)
template <typename F>
void parallel_for(size_t n, unsigned max_threads, F &&f) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const unsigned nt = (unsigned)std::min<size_t>({(size_t)max_threads, (size_t)hw, n});
    if (nt <= 1) { for (size_t i = 0; i < n; ++i) f(i); return; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t)
        pool.emplace_back([&] { for (size_t i; (i = next.fetch_add(1)) < n;) f(i); });
    for (std::thread &th : pool) th.join();
}

// Block calling ON THE DEVICE (getTerminalBlocks / getInterstitialBlocks, src/teloscope.cpp:29-256) over the
// batch's resident match stream: the blocks of all segments, sorted by (segment; terminal blocks in push order:
// forward walk then reverse walk; interstitial blocks by start).
}  // namespace

// Block calling on the device over ANY resident match stream addressed by a tile directory: the tiled kernel's (a batch) or the
// general kernels' dense stream (gen_lens != 0: their record format and pattern lengths, ts_internal.h).  tab: the kernels'
// per-segment table; sums_out (nullable): the per-segment sums {matches, forward (seen), matches, canonical, forward (owned)}.
int ts_device_block_call_raw(ts_ctx *c, const TsTile *d_tiles, const unsigned long long *d_tile_off, const uint32_t *d_stats,
                             const uint32_t *d_matches, uint64_t n_matches_hint, const std::vector<TsShardSegIn> &tab, size_t nt,
                             bool tips, unsigned long long gen_lens, const uint32_t *d_chain, uint32_t *d_work, hipStream_t st,
                             std::vector<TsDevBlock> &blocks, std::vector<unsigned long long> *sums_out, int rec16,
                             const uint32_t *d_wide_len, bool unordered) {
    const ts_params &P = c->params;
    const size_t ns = tab.size();
    blocks.clear();
    if (sums_out) sums_out->assign(ns * 5, 0ull);
    if (!ns) return TS_OK;
    const size_t off_bounds = ns * sizeof(TsShardSegIn), off_count = off_bounds + ns * 16, off_sums = off_count + 16, off_range = off_sums + ns * 40, tab_bytes = off_range + ns * 16;
    DevBuf d_tab, d_blocks;
    struct Return { ts_ctx *c; DevBuf &a, &b2; ~Return() { c->pool.give(std::move(a)); c->pool.give(std::move(b2)); } } give_back{c, d_tab, d_blocks};
    HIP_TRY(c, c->pool.take(tab_bytes, d_tab));
    char *const dt = (char *)d_tab.p;
    uint32_t cap = (uint32_t)std::min<uint64_t>(64ull * ns + 4096 + n_matches_hint / 256, 1u << 26);
    bool done = false;
    for (int attempt = 0; attempt < 6 && !done; ++attempt) {
        // (behind the blocks: the list of chains the interstitial screening hands to its evaluation kernel)
        const uint32_t cand_cap = 2u * cap + 256u;
        const size_t off_cand = (size_t)cap * sizeof(TsDevBlock);
        HIP_TRY(c, c->pool.take(off_cand + (size_t)cand_cap * 8 + 16, d_blocks));
        HIP_TRY(c, hipMemcpyAsync(dt, tab.data(), off_bounds, hipMemcpyHostToDevice, st));
        HIP_TRY(c, hipMemsetAsync(dt + off_count, 0, 16, st));
        TsBlockCallParams Q{};
        Q.tiles = d_tiles;
        Q.tile_off = d_tile_off;
        Q.tile_stats = d_stats;
        Q.matches = d_matches;
        Q.blocks = (TsDevBlock *)d_blocks.p;
        Q.n_blocks = (uint32_t *)(dt + off_count);
        Q.block_cap = cap;
        Q.cand = (uint32_t *)((char *)d_blocks.p + off_cand);
        Q.n_cand = (uint32_t *)(dt + off_count) + 1;
        Q.cand_cap = cand_cap;
        Q.terminal_limit = P.terminal_limit; Q.max_match_dist = P.max_match_dist;
        Q.min_block_len = P.min_block_len; Q.max_block_dist = P.max_block_dist;
        Q.min_block_counts = P.min_block_counts; Q.min_block_density = P.min_block_density;
        Q.k = c->k; Q.its_min_len = (uint32_t)(uint16_t)(2 * c->bp.first_pattern_len);
        Q.gen_lens = gen_lens;
        Q.rec16 = rec16 ? 1u : 0u;
        // (the general path's wide records and push-ordered streams: blockcall.hip, MODE 1)
        Q.wide = d_wide_len ? 1u : 0u; Q.wide_len = d_wide_len;
        Q.unordered = unordered ? 1u : 0u; Q.its_range = (unsigned long long *)(dt + off_range);
        if (ts_k_launch_block_call(&Q, (const TsShardSegIn *)dt, (uint32_t)ns, 0u, (uint32_t)nt, (unsigned long long *)(dt + off_bounds),
                                   nullptr, tips ? 0 : 1, nullptr, (unsigned long long *)(dt + off_sums), d_chain, d_work, st) != 0)
            return c->fail(TS_ERR_HIP, "block-calling kernel launch failed");
        uint32_t nb = 0;
        HIP_TRY(c, hipMemcpyAsync(&nb, dt + off_count, 4, hipMemcpyDeviceToHost, st));
        if (sums_out) HIP_TRY(c, hipMemcpyAsync(sums_out->data(), dt + off_sums, ns * 40, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if (nb > cap) { cap = nb + 1024; c->pool.give(std::move(d_blocks)); continue; }   // rare: more blocks than provisioned, rerun
        blocks.resize(nb);
        if (nb) {
            HIP_TRY(c, hipMemcpyAsync(blocks.data(), d_blocks.p, (size_t)nb * sizeof(TsDevBlock), hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipStreamSynchronize(st));
        }
        done = true;
    }
    // (block counts are deterministic: an attempt is sized by the count of the one before)
    if (!done) return c->fail(TS_ERR_STATE, "device block calling kept overflowing its block buffer");
    std::sort(blocks.begin(), blocks.end(), [](const TsDevBlock &x, const TsDevBlock &y) {
        if (x.seg != y.seg) return x.seg < y.seg;
        const uint32_t kx = x.kind == 2 ? 1 : 0, ky = y.kind == 2 ? 1 : 0;
        if (kx != ky) return kx < ky;
        if (kx) {
            // interstitial blocks: in the order the reference's walk emits them — ascending start over a stream in position order; the
            // general path's other streams carry the stream index of the chain's first record (blockcall.hip, MODE 1)
            const unsigned long long ox = ((unsigned long long)x.pad << 32) | x.seq, oy = ((unsigned long long)y.pad << 32) | y.seq;
            if (ox != oy) return ox < oy;
            return x.start < y.start;
        }
        return x.kind != y.kind ? x.kind < y.kind : x.seq < y.seq;
    });
    return TS_OK;
}

namespace {

int device_block_call(ts_batch *b, hipStream_t st, std::vector<TsDevBlock> &blocks) {
    ts_ctx *c = b->ctx;
    const size_t ns = b->segs.size(), nt = b->tiles.size();
    // segment table of the kernels (a whole batch: every segment with all its tiles, both ends its own)
    std::vector<TsShardSegIn> tab(ns);
    for (size_t i = 0; i < ns; ++i) {
        TsShardSegIn &S = tab[i];
        S.in_off = b->segs[i].in_off; S.len = b->segs[i].len; S.abs_pos = b->segs[i].abs_pos;
        S.t0 = S.o0 = b->segs[i].first_tile;
        S.t1 = S.o1 = b->segs[i].first_tile + b->segs[i].n_tiles;
        S.flags = TS_SEG_F_HAS_START | TS_SEG_F_HAS_END;
        S.lo_rel = 0; S.hi_rel = b->segs[i].len;
        S.seg = (uint32_t)i;
    }
    // (with the scan's chain summaries the interstitial search screens the tiles and walks only the listed ones)
    const bool from_scan = b->chain_valid() && !b->tips;
    if (from_scan && b->d_scan_tmp.bytes < (nt + 2) * 4) {
        c->pool.give(std::move(b->d_scan_tmp));
        HIP_TRY(c, c->pool.take((nt + 2) * 4, b->d_scan_tmp));
    }
    return ts_device_block_call_raw(c, (const TsTile *)b->d_tiles.p, (const unsigned long long *)b->d_tile_off.p, b->stats_ptr(),
                                    b->records_ptr(), b->n_matches, tab, nt, b->tips, 0ull,
                                    from_scan ? (const uint32_t *)b->d_chain.p : nullptr, from_scan ? (uint32_t *)b->d_scan_tmp.p : nullptr, st, blocks, nullptr,
                                    b->records16() ? 1 : 0);
}

unsigned finalize_threads() {
    static const unsigned v = [] {
        if (const char *e = getenv("TS_HOST_THREADS")) { const int n = atoi(e); if (n > 0) return (unsigned)n; }
        return 16u;
    }();
    return v;
}

// What a download leaves in host memory before post-processing.
struct Fetched {
    std::unique_ptr<HostLanding> land;
    std::vector<TsDevBlock> blocks;
    const uint32_t *wins = nullptr, *recs = nullptr, *tile_stats = nullptr;
    unsigned long long *tile_off = nullptr;
    uint64_t nrecs = 0;
    bool with_matches = false;
};

// Device work + D2H of a synced whole batch (on its stream), into pinned landing area `pin`.
int batch_fetch(ts_batch *b, bool with_matches, PinBuf &pin, Fetched &F) {
    ts_ctx *c = b->ctx;
    const size_t nt = b->tiles.size();
    hipStream_t st = (hipStream_t)b->last_stream;
    F.with_matches = with_matches;
    if (b->msg_windows) return c->fail(TS_ERR_STATE, "the batch's scan packed its window records into a shard message (ts_batch_bind_shard_message): there are no 32-byte records to download");
    { int rc = device_block_call(b, st, F.blocks); if (rc != TS_OK) return rc; }
    const uint64_t nwin_dl = b->tips ? 0 : b->n_windows;
    const uint64_t nrecs = with_matches ? b->n_matches : 0;
    F.nrecs = nrecs;
    F.land.reset(new HostLanding(pin, nwin_dl * 32 + (with_matches ? nrecs * 4 + (nt + 1) * 24 : 0) + 1024));
    HostLanding &land = *F.land;
    uint32_t *const wins = (uint32_t *)land.carve(nwin_dl * 32 + 4);
    if (!wins) return c->fail(TS_ERR_ALLOC, "out of host memory");
    F.wins = wins;
    if (nwin_dl) HIP_TRY(c, hipMemcpyAsync(wins, b->windows_ptr(), nwin_dl * 32, hipMemcpyDeviceToHost, st));
    if (with_matches) {
        uint32_t *const recs = (uint32_t *)land.carve(nrecs * 4 + 4);
        // tile directory: records of tile t are recs[tile_off[t] .. +tile_stats[4t]), in position order
        unsigned long long *const tile_off = (unsigned long long *)land.carve((nt + 1) * 8);
        uint32_t *const tile_stats = (uint32_t *)land.carve((nt + 1) * 16);
        if (!recs || !tile_off || !tile_stats) return c->fail(TS_ERR_ALLOC, "out of host memory");
        F.recs = recs; F.tile_off = tile_off; F.tile_stats = tile_stats;
        std::vector<unsigned long long> dense_base;
        if (b->dense) {
            if (nrecs) HIP_TRY(c, hipMemcpyAsync(recs, b->records_ptr(), nrecs * 4, hipMemcpyDeviceToHost, st));
        } else if (nrecs) {
            // pack the per-wave regions into one dense stream on the device, then one D2H copy
            dense_base.assign(b->total_waves + 1, 0);
            for (uint32_t w = 0; w < b->total_waves; ++w) dense_base[w + 1] = dense_base[w] + b->wave_fill[w];
            if (b->d_dense.bytes < nrecs * 4) { c->pool.give(std::move(b->d_dense)); HIP_TRY(c, c->pool.take(nrecs * 4, b->d_dense)); }
            if (b->d_dense_base.bytes < (size_t)(b->total_waves + 1) * 8) {
                c->pool.give(std::move(b->d_dense_base));
                HIP_TRY(c, c->pool.take((size_t)(b->total_waves + 1) * 8, b->d_dense_base));
            }
            HIP_TRY(c, hipMemcpyAsync(b->d_dense_base.p, dense_base.data(), (size_t)(b->total_waves + 1) * 8, hipMemcpyHostToDevice, st));
            int e = ts_k_launch_compact((const uint32_t *)b->d_matches.p, (const uint32_t *)b->d_fill.p,
                                        (const unsigned long long *)b->d_dense_base.p, b->region_cap,
                                        b->total_waves, (uint32_t *)b->d_dense.p, b->records16() ? 1 : 0, st);
            if (e != 0) return c->fail(TS_ERR_HIP, "compaction kernel launch failed");
            HIP_TRY(c, hipMemcpyAsync(recs, b->d_dense.p, nrecs * 4, hipMemcpyDeviceToHost, st));
        }
        if (nt) HIP_TRY(c, hipMemcpyAsync(tile_off, b->d_tile_off.p, nt * 8, hipMemcpyDeviceToHost, st));
        if (nt) HIP_TRY(c, hipMemcpyAsync(tile_stats, b->stats_ptr(), nt * 16, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        if (!b->dense && nrecs) {
            // region offset -> dense offset.  The wave that scanned a tile is read off its region offset (a tile
            // without records at the very end of a full region lands on the next wave's base: the same place).
            for (size_t t = 0; t < nt; ++t) {
                const uint32_t w = (uint32_t)(tile_off[t] / b->region_cap);
                tile_off[t] = tile_off[t] - (unsigned long long)w * b->region_cap + dense_base[w];
            }
        }
    } else {
        HIP_TRY(c, hipStreamSynchronize(st));
    }
    return TS_OK;
}

// Host post-processing of a fetched batch: SegmentData per segment.  Everything is independent per window, per
// tile and per block, so the work is cut into pieces of a few ten thousand records and spread over the host
// threads whatever the segment sizes are (one 250 Mb contig keeps all threads busy): window records with their
// float metrics (evaluated on the host from the integer counts, as the reference does), packed records ->
// MatchInfo with absolute position and the terminal flag (isTerminal, src/teloscope.cpp:451-459), and the
// device-called blocks.
int batch_finalize(ts_batch *b, const Fetched &F, ts_segment_out *out) {
    ts_ctx *c = b->ctx;
    const ts_params &P = c->params;
    const size_t ns = b->segs.size();
    for (size_t i = 0; i < ns; ++i) std::memset(&out[i], 0, sizeof out[i]);
    const bool nuc = P.out_gc || P.out_entropy;
    // per-segment arrays
    std::vector<uint64_t> seg_nm(ns, 0);
    std::vector<uint64_t> tile_out;                              // output index (within its segment's array) of a tile's first record
    if (F.with_matches) {
        tile_out.resize(b->tiles.size());
        for (size_t si = 0; si < ns; ++si) {
            const SegPlan &sp = b->segs[si];
            uint64_t nm = 0;
            for (uint32_t t = 0; t < sp.n_tiles; ++t) { tile_out[sp.first_tile + t] = nm; nm += F.tile_stats[4ull * (sp.first_tile + t)]; }
            if (nm > 0xFFFFFFFFull) return c->fail(TS_ERR_UNSUPPORTED, "more than 2^32 matches in one segment");
            seg_nm[si] = nm;
        }
    }
    int rc = TS_OK;
    for (size_t si = 0; si < ns && rc == TS_OK; ++si) {
        const SegPlan &sp = b->segs[si];
        if (!b->tips && sp.n_windows) {
            out[si].windows = (ts_window *)std::malloc(sp.n_windows * sizeof(ts_window));
            if (!out[si].windows) rc = c->fail(TS_ERR_ALLOC, "out of host memory");
            out[si].n_windows = sp.n_windows;
        }
        if (rc == TS_OK && seg_nm[si]) {
            out[si].matches = (ts_match *)ts_alloc_large(seg_nm[si] * sizeof(ts_match));
            if (!out[si].matches) rc = c->fail(TS_ERR_ALLOC, "out of host memory");
            out[si].n_matches = seg_nm[si];
        }
    }
    if (rc != TS_OK) { ts_free_segments(out, ns); return rc; }

    struct Piece { uint32_t seg; bool windows; uint64_t a, z; };     // windows [a, z) of a segment, or its tiles [a, z)
    std::vector<Piece> pieces;
    constexpr uint64_t kWinPiece = 1u << 13, kTilePiece = 64;
    for (size_t si = 0; si < ns; ++si) {
        const SegPlan &sp = b->segs[si];
        if (!b->tips)
            for (uint64_t a = 0; a < sp.n_windows; a += kWinPiece) pieces.push_back({(uint32_t)si, true, a, std::min<uint64_t>(sp.n_windows, a + kWinPiece)});
        if (seg_nm[si])
            for (uint64_t a = 0; a < sp.n_tiles; a += kTilePiece) pieces.push_back({(uint32_t)si, false, a, std::min<uint64_t>(sp.n_tiles, a + kTilePiece)});
    }
    const uint16_t klen = (uint16_t)c->k;
    std::atomic<int> bad{0};
    parallel_for(pieces.size(), finalize_threads(), [&](size_t pi) {
        const Piece &pc = pieces[pi];
        const SegPlan &sp = b->segs[pc.seg];
        ts_segment_out &o = out[pc.seg];
        if (pc.windows) {
            for (uint64_t kwin = pc.a; kwin < pc.z; ++kwin) {
                const uint32_t *r = &F.wins[(sp.win_base + kwin) * 8];
                ts_window &w = o.windows[kwin];
                std::memset(&w, 0, sizeof w);
                const uint64_t ws = kwin * P.step;
                w.window_start = sp.abs_pos + ws;
                w.current_window_size = (uint32_t)std::min<uint64_t>(P.window_size, sp.len - ws);
                if (nuc) for (int i = 0; i < 4; ++i) w.nucleotide_counts[i] = r[i];
                if (P.out_gc) w.gc_content = ts::gc_content(w.nucleotide_counts, w.current_window_size);
                if (P.out_entropy) w.shannon_entropy = ts::shannon_entropy_memo(w.nucleotide_counts, w.current_window_size, c->entropy_term);
                w.canonical_covered = r[4];
                w.non_canonical_covered = r[5];
                w.fwd_covered = r[6];
                w.rev_covered = r[7];
            }
            return;
        }
        const uint64_t term_end = sp.len > P.terminal_limit ? sp.len - P.terminal_limit : 0;
        for (uint64_t t = pc.a; t < pc.z; ++t) {
            const size_t ti = sp.first_tile + t;
            const uint32_t cnt = F.tile_stats[4ull * ti];
            if (!cnt) continue;
            const uint64_t r0 = F.tile_off[ti];
            if (r0 + cnt > F.nrecs) { bad.store(1); return; }
            const uint64_t rel0 = b->tiles[ti].in_off - sp.in_off;          // segment-relative position of the tile
            expand_records(F.recs + r0, cnt, o.matches + tile_out[ti], sp.abs_pos, rel0, P.terminal_limit, term_end, klen);
        }
        _mm_sfence();
    });
    if (bad.load()) { ts_free_segments(out, ns); return c->fail(TS_ERR_STATE, "tile directory out of range"); }
    // blocks: the sorted list's slice of every segment
    size_t bi = 0;
    for (size_t si = 0; si < ns; ++si) {
        size_t nterm = 0, nits = 0;
        const size_t b0 = bi;
        while (bi < F.blocks.size() && F.blocks[bi].seg == si) { (F.blocks[bi].kind == 2 ? nits : nterm)++; ++bi; }
        auto fill = [&](ts_block *&dst, uint64_t &n, size_t count, bool its) -> bool {
            n = count; dst = nullptr;
            if (!count) return true;
            dst = (ts_block *)std::malloc(count * sizeof(ts_block));
            if (!dst) return false;
            size_t at = 0;
            for (size_t q = b0; q < bi; ++q)
                if ((F.blocks[q].kind == 2) == its) std::memcpy(&dst[at++], &F.blocks[q], sizeof(ts_block));
            return true;
        };
        if (!fill(out[si].terminal_blocks, out[si].n_terminal_blocks, nterm, false) ||
            !fill(out[si].interstitial_blocks, out[si].n_interstitial_blocks, nits, true)) {
            ts_free_segments(out, ns);
            return c->fail(TS_ERR_ALLOC, "out of host memory");
        }
    }
    return TS_OK;
}

int download_impl(ts_batch *b, ts_segment_out *out, bool with_matches) {
    if (!b || !out) return TS_ERR_INVALID_ARG;
    ts_ctx *c = b->ctx;
    DEVICE_TRY(c);
    if (!b->whole()) return c->fail(TS_ERR_STATE, "a download needs an unrestricted batch (adopt the shards' results first)");
    if (!b->synced) { int rc = ts_batch_sync(b); if (rc != TS_OK) return rc; }
    std::lock_guard<std::mutex> dl(c->down_mtx);
    Fetched F;
    int rc = batch_fetch(b, with_matches, c->pin_down[0], F);
    if (rc == TS_OK) rc = batch_finalize(b, F, out);
    return rc;
}

}  // namespace

// the two phases of a download, for the host entry points' pipeline (pipeline.cpp): device work + D2H into the
// pinned landing area `slot` of the context, then host post-processing (which may run while the next group's
// fetch uses the other slot)
struct ts_fetched { Fetched F; };
ts_fetched *ts_batch_fetch(ts_batch *b, bool with_matches, int slot, int *rc_out) {
    ts_fetched *f = new ts_fetched();
    *rc_out = batch_fetch(b, with_matches, b->ctx->pin_down[slot & 1], f->F);
    if (*rc_out != TS_OK) { delete f; return nullptr; }
    return f;
}
int ts_batch_finalize(ts_batch *b, ts_fetched *f, ts_segment_out *out) {
    const int rc = batch_finalize(b, f->F, out);
    delete f;
    return rc;
}

extern "C" {

// --------------------------------------------------------------- download + host post-processing
// SegmentData of every segment of a synced whole batch: windows, all match records, and the blocks — called on the
// device, like ts_batch_download_blocks — so the only host work per record is its expansion to MatchInfo.
int ts_batch_download(ts_batch *b, const char *const *host_seqs, ts_segment_out *out) {
    (void)host_seqs;
    return download_impl(b, out, true);
}

// --------------------------------------------------------------- device block calling (row f1)
int ts_batch_download_blocks(ts_batch *b, ts_segment_out *out) { return download_impl(b, out, false); }

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

}  // extern "C"
