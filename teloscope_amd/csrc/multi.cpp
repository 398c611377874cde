// multi.cpp — Teloscope::scanSegment over several devices in ONE process: ts_scan_segments_multi.
//
// The reference queues one thread-pool job per path and merges the jobs' PathData under a mutex, then sorts by
// seqPos (src/input.cpp:719-733, :1036-1037, include/teloscope.h:262-266).  Here the batch's ONE plan is cut into
// one shard per context (consecutive tile ranges of equal bases, shard.cpp): one host thread per context uploads the
// bases its shard reads over that device's own PCIe link, scans, calls blocks on the device, packs the shard's
// message and brings it back over the same link; the host merges the messages into SegmentData in input order.
// What crosses PCIe on the way back is what the reference's writers read (windows, blocks, canonicalMatches,
// terminal nonCanonicalMatches), not the match stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "capi_internal.hpp"

namespace {

// out[i].matches of a single-context ts_scan_segments -> the visible records only (what the multi-device form returns),
// counts[i] = the sizes the vectors had
void keep_visible(const ts_segment_in *segs, size_t n, ts_segment_out *out, ts_segment_counts *counts) {
    for (size_t i = 0; i < n; ++i) {
        ts_segment_out &o = out[i];
        ts_segment_counts cnt{segs[i].tips_only ? 0 : o.n_windows, o.n_matches, 0, 0};
        uint64_t keep = 0;
        for (uint64_t m = 0; m < o.n_matches; ++m) {
            const ts_match &g = o.matches[m];
            cnt.n_canonical += (g.flags & TS_MATCH_CANONICAL) ? 1 : 0;
            cnt.n_forward += (g.flags & TS_MATCH_FORWARD) ? 1 : 0;
            if (!segs[i].tips_only && (g.flags & (TS_MATCH_CANONICAL | TS_MATCH_TERMINAL))) o.matches[keep++] = g;
        }
        if (segs[i].tips_only) cnt.n_canonical = 0;           // (a tips-only SegmentData has no canonicalMatches)
        if (counts) counts[i] = cnt;
        o.n_matches = keep;
        if (!keep) { std::free(o.matches); o.matches = nullptr; }
    }
}

struct PartJob {
    ts_ctx *ctx = nullptr;
    int rc = TS_OK;
    std::string err;
    const unsigned char *msg = nullptr;      // the shard's message, in the context's pinned landing area
    uint64_t msg_bytes = 0;
    std::unique_lock<std::mutex> landing;    // ... which stays this call's until the messages are merged
};

// One shard on one context: plan, upload, scan, pack, download.
void run_part(PartJob &job, uint32_t n_parts, uint32_t part, const std::vector<uint64_t> &lens, const std::vector<uint64_t> &abs,
              const ts_segment_in *segs, bool tips) {
    ts_ctx *c = job.ctx;
    auto fail = [&](int rc) { job.rc = rc; job.err = ts_last_error(c); };
    c->bind_this_thread();                                   // (the caller holds the context's call lock: its streams and pinned ring are this call's)
    DeviceGuard g(c->device);
    if (g.error() != hipSuccess) { job.rc = TS_ERR_HIP; job.err = "hipSetDevice failed"; return; }
    if (int rc = ts_pipeline_ensure_streams(c); rc != TS_OK) return fail(rc);
    ts_batch *b = ts_batch_create(c, lens.data(), abs.data(), lens.size(), tips ? 1 : 0, 0);
    if (!b) return fail(TS_ERR_UNSUPPORTED);
    struct Destroy { ts_batch *b; ~Destroy() { ts_batch_destroy(b); } } destroy{b};
    uint32_t scale = 1;
    if (int rc = ts_batch_restrict_shard(b, n_parts, part, scale); rc != TS_OK) return fail(rc);
    ts_shard_info si{};
    ts_batch_shard_info(b, n_parts, part, scale, &si);
    if (int rc = ts_batch_ensure_device(b); rc != TS_OK) return fail(rc);
    int slot = 0;
    bool used[ts_ctx::kUpSlots] = {false, false, false};
    if (int rc = ts_pipeline_upload_batch(b, segs, &slot, used); rc != TS_OK) return fail(rc);      // (an empty shard uploads nothing)
    hipEvent_t uploaded = nullptr;
    if (hipEventCreateWithFlags(&uploaded, hipEventDisableTiming) != hipSuccess) return fail(c->fail(TS_ERR_HIP, "hipEventCreate failed"));
    struct DestroyEv { hipEvent_t e; ~DestroyEv() { (void)hipEventDestroy(e); } } destroy_ev{uploaded};
    if (hipEventRecord(uploaded, c->up_stream) != hipSuccess || hipStreamWaitEvent(c->scan_stream, uploaded, 0) != hipSuccess)
        return fail(c->fail(TS_ERR_HIP, "stream ordering failed"));
    if (int rc = ts_batch_scan(b, nullptr, c->scan_stream); rc != TS_OK) return fail(rc);
    if (int rc = ts_batch_sync(b); rc != TS_OK) return fail(rc);          // waits; regrows + rescans on overflow
    (void)hipStreamSynchronize(c->up_stream);                              // the pinned ring is free again
    ts_batch_release_input(b);
    DevBuf d_msg;
    struct Return { ts_ctx *c; DevBuf &a; ~Return() { c->pool.give(std::move(a)); } } give_back{c, d_msg};
    for (int attempt = 0; attempt < 8; ++attempt) {
        ts_batch_shard_info(b, n_parts, part, scale, &si);
        if (d_msg.bytes < si.msg_bytes) {
            c->pool.give(std::move(d_msg));
            if (c->pool.take(si.msg_bytes, d_msg) != hipSuccess) return fail(c->fail(TS_ERR_ALLOC, "cannot allocate the shard's message buffer"));
        }
        if (int rc = ts_batch_pack_shard(b, d_msg.p, si.msg_bytes, c->scan_stream); rc != TS_OK) return fail(rc);
        TsShardHeader H{};
        if (hipMemcpyAsync(&H, d_msg.p, sizeof H, hipMemcpyDeviceToHost, c->scan_stream) != hipSuccess ||
            hipStreamSynchronize(c->scan_stream) != hipSuccess) return fail(c->fail(TS_ERR_HIP, "shard header read-back failed"));
        if (H.flags & (TS_SHARD_F_VISIBLE_OVERFLOW | TS_SHARD_F_BLOCK_OVERFLOW)) {      // rare: pack again into a larger message
            ts_shard_status st{};
            ts_shard_peek(&H, sizeof H, &st);
            scale *= std::max<uint32_t>(2, st.scale_factor_needed);
            ts_batch_set_shard_scale(b, scale);
            continue;
        }
        // the whole message: what the header says is used of each section would do, but the sections are small
        // next to the window records, and one copy keeps the layout the receiver computes
        job.landing = std::unique_lock<std::mutex>(c->down_mtx);
        if (c->pin_down[0].ensure(std::max<size_t>(si.msg_bytes + si.msg_bytes / 4, 1u << 20)) != hipSuccess)
            return fail(c->fail(TS_ERR_ALLOC, "cannot allocate the pinned landing area"));
        if (hipMemcpyAsync(c->pin_down[0].p, d_msg.p, si.msg_bytes, hipMemcpyDeviceToHost, c->scan_stream) != hipSuccess ||
            hipStreamSynchronize(c->scan_stream) != hipSuccess) return fail(c->fail(TS_ERR_HIP, "shard message download failed"));
        job.msg = (const unsigned char *)c->pin_down[0].p;
        job.msg_bytes = si.msg_bytes;
        return;
    }
    fail(c->fail(TS_ERR_STATE, "a shard's message kept overflowing"));
}

}  // namespace

extern "C" int ts_scan_segments_multi(ts_ctx *const *ctxs, size_t n_ctx, const ts_segment_in *segs, size_t n_segs,
                                      ts_segment_out *out, ts_segment_counts *counts) {
    if (!ctxs || !n_ctx || (n_segs && (!segs || !out))) return TS_ERR_INVALID_ARG;
    for (size_t i = 0; i < n_ctx; ++i) if (!ctxs[i]) return TS_ERR_INVALID_ARG;
    ts_ctx *c0 = ctxs[0];
    for (size_t i = 0; i < n_segs; ++i) {
        std::memset(&out[i], 0, sizeof out[i]);
        if (counts) counts[i] = ts_segment_counts{0, 0, 0, 0};
        if (segs[i].len && !segs[i].seq) return c0->fail(TS_ERR_INVALID_ARG, "null sequence pointer");
        if (segs[i].input_format > TS_INPUT_PACKED2) return c0->fail(TS_ERR_INVALID_ARG, "unknown input_format");
    }
    if (!n_segs) return TS_OK;
    for (size_t i = 0; i < n_ctx; ++i)
        for (size_t j = 0; j < i; ++j)
            if (ctxs[i] == ctxs[j]) return c0->fail(TS_ERR_INVALID_ARG, "ts_scan_segments_multi: a context is listed twice (contexts may share a device, not be the same object)");
    for (size_t i = 1; i < n_ctx; ++i)
        if (std::memcmp(&ctxs[i]->params, &c0->params, offsetof(ts_params, device)) != 0 || ctxs[i]->patterns.size() != c0->patterns.size())
            return c0->fail(TS_ERR_INVALID_ARG, "ts_scan_segments_multi: the contexts were created with different parameters");

    // every context's call lock, in address order (two concurrent calls over the same contexts cannot deadlock)
    std::vector<ts_ctx *> order(ctxs, ctxs + n_ctx);
    std::sort(order.begin(), order.end());
    std::vector<std::unique_lock<std::mutex>> locks;
    for (ts_ctx *c : order) locks.emplace_back(c->api_mtx);

    // segments [a, z) of `which` on one context alone: the single-device pipeline, reduced to the same view
    auto on_ctx = [&](ts_ctx *c, const std::vector<size_t> &which, size_t a, size_t z) -> int {
        if (z <= a) return TS_OK;
        std::vector<ts_segment_in> in(z - a);
        std::vector<ts_segment_out> tmp(z - a);
        std::vector<ts_segment_counts> cnt(z - a);
        for (size_t i = a; i < z; ++i) in[i - a] = segs[which[i]];
        int rc = ts_scan_segments_unlocked(c, in.data(), in.size(), tmp.data());
        if (rc != TS_OK) return rc;
        keep_visible(in.data(), in.size(), tmp.data(), cnt.data());
        for (size_t i = a; i < z; ++i) { out[which[i]] = tmp[i - a]; if (counts) counts[which[i]] = cnt[i - a]; }
        return TS_OK;
    };
    // everything on the first context alone
    auto single = [&](const std::vector<size_t> &which) -> int { return on_ctx(c0, which, 0, which.size()); };
    // Parameter sets outside the tiled kernel (the general kernels have no shard results): whole segments dealt to the contexts
    // in consecutive runs of equal bases — the reference's own decomposition, one job per path (src/input.cpp:719-724) —
    // one host thread per context, results in input order.
    auto whole_segments = [&](const std::vector<size_t> &which) -> int {
        if (n_ctx == 1 || which.size() < 2) return single(which);
        uint64_t total = 0;
        for (size_t i : which) total += segs[i].len;
        std::vector<size_t> cut(n_ctx + 1, which.size());
        cut[0] = 0;
        {
            uint64_t acc = 0;
            size_t d = 1;
            for (size_t i = 0; i < which.size() && d < n_ctx; ++i) {
                acc += segs[which[i]].len;
                while (d < n_ctx && (unsigned __int128)acc * n_ctx >= (unsigned __int128)total * d) cut[d++] = i + 1;
            }
        }
        std::vector<int> rcs(n_ctx, TS_OK);
        std::vector<std::thread> pool;
        for (size_t d = 0; d < n_ctx; ++d)
            pool.emplace_back([&, d] {
                // (on_ctx allocates: an exception must not leave the thread — std::terminate across the C boundary)
                try { rcs[d] = on_ctx(ctxs[d], which, cut[d], cut[d + 1]); }
                catch (const std::bad_alloc &) { rcs[d] = ctxs[d]->fail(TS_ERR_ALLOC, "out of host memory"); }
                catch (...) { rcs[d] = ctxs[d]->fail(TS_ERR_ALLOC, "exception in a context's scan thread"); }
            });
        for (std::thread &th : pool) th.join();
        for (size_t d = 0; d < n_ctx; ++d)
            if (rcs[d] != TS_OK) return d == 0 ? rcs[d] : c0->fail(rcs[d], "context " + std::to_string(d) + ": " + ts_last_error(ctxs[d]));
        return TS_OK;
    };

    std::vector<size_t> full, tips;
    for (size_t i = 0; i < n_segs; ++i) (segs[i].tips_only ? tips : full).push_back(i);
    int rc = TS_OK;
    for (int mode = 0; mode < 2 && rc == TS_OK; ++mode) {
        const std::vector<size_t> &which = mode ? tips : full;
        if (which.empty()) continue;
        std::string why;
        const bool tiled = mode ? c0->fast_ok : ts_full_scan_supported(c0, why);
        if (!tiled) { rc = whole_segments(which); continue; }
        std::vector<uint64_t> lens(which.size()), abs(which.size());
        std::vector<ts_segment_in> in(which.size());
        for (size_t i = 0; i < which.size(); ++i) { in[i] = segs[which[i]]; lens[i] = in[i].len; abs[i] = in[i].abs_pos; }
        ts_batch *plan = ts_batch_create(c0, lens.data(), abs.data(), lens.size(), mode, 0);
        if (!plan) { rc = TS_ERR_UNSUPPORTED; break; }
        struct Destroy { ts_batch *b; ~Destroy() { ts_batch_destroy(b); } } destroy{plan};
        const uint32_t n_parts = (uint32_t)n_ctx;
        std::vector<PartJob> jobs(n_parts);
        std::vector<std::thread> pool;
        for (uint32_t p = 0; p < n_parts; ++p) {
            jobs[p].ctx = ctxs[p];
            pool.emplace_back(run_part, std::ref(jobs[p]), n_parts, p, std::cref(lens), std::cref(abs), in.data(), mode != 0);
        }
        for (std::thread &th : pool) th.join();
        for (uint32_t p = 0; p < n_parts && rc == TS_OK; ++p)
            if (jobs[p].rc != TS_OK) rc = c0->fail(jobs[p].rc, "shard " + std::to_string(p) + ": " + jobs[p].err);
        if (rc != TS_OK) break;
        std::vector<const void *> msgs(n_parts);
        std::vector<uint64_t> bytes(n_parts);
        for (uint32_t p = 0; p < n_parts; ++p) { msgs[p] = jobs[p].msg; bytes[p] = jobs[p].msg_bytes; }
        std::vector<ts_segment_out> tmp(which.size());
        std::vector<ts_segment_counts> cnt(which.size());
        const int frc = ts_shards_finalize(plan, msgs.data(), bytes.data(), n_parts, tmp.data(), cnt.data());
        for (PartJob &j : jobs) if (j.landing.owns_lock()) j.landing.unlock();
        if (frc == TS_SHARD_NEED_FULL || frc == TS_SHARD_RETRY_SYNC || frc == TS_SHARD_RETRY_GROW) {
            // (the parts synced and regrew by themselves; what is left is an input the shards' assumptions do not hold for)
            rc = single(which);
            continue;
        }
        if (frc != TS_OK) { rc = frc; break; }
        for (size_t i = 0; i < which.size(); ++i) { out[which[i]] = tmp[i]; if (counts) counts[which[i]] = cnt[i]; }
    }
    if (rc != TS_OK) ts_free_segments(out, n_segs);
    return rc;
}
