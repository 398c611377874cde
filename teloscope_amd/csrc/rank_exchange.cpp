// rank_exchange.cpp — the rank form's ONE exchange for a C++ host: every rank's shard message to one rank, as a single
// grouped send / recv over RCCL (xGMI between the GPUs of a node).  What teloscope_amd/distributed.py's ShardExchange does
// through torch.distributed, for a host that has no Python in it (north_star: "host code stays C++17 ... single RCCL
// gather").  The reference's counterpart is its in-process merge of per-path results (src/input.cpp:719-733,
// include/teloscope.h:262-266); nothing of RCCL exists there.
//
// librccl is opened at run time (dlopen), not linked: libteloscan.so loads on a host without RCCL, and only
// ts_exchange_* fail there (TS_ERR_UNSUPPORTED).  Sizes are known to both sides from the plan (ts_batch_shard_info), so a
// step posts its sends and receives without a count exchange and without a host synchronisation; everything is
// asynchronous on the caller's stream.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>

#include "capi_internal.hpp"

namespace {

// the few RCCL entry points used, by their C signatures (rccl.h: ncclResult_t is an int-sized enum, 0 = success;
// ncclUniqueId is 128 opaque bytes passed BY VALUE; ncclUint8 = 1)
struct UniqueId { char internal[128]; };
typedef void *Comm;
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*Send)(const void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, Comm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string why;
};
constexpr int kUint8 = 1;

Rccl &rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, [] {
        // TS_RCCL_LIB names the library instead of the usual places (a site's own build; the tests point it at a file that
        // does not exist to reach the path below without uninstalling RCCL)
        const char *forced = getenv("TS_RCCL_LIB");
        std::string last;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            R.lib = dlopen(forced && forced[0] ? forced : name, RTLD_NOW | RTLD_LOCAL);
            if (R.lib) break;
            const char *e = dlerror();               // (read ONCE: the call clears the error, a second one returns NULL)
            last = e ? e : "dlopen failed";
            if (forced && forced[0]) break;
        }
        if (!R.lib) { R.why = "librccl not found: " + last; return; }
        auto sym = [&](const char *n) -> void * {
            void *p = dlsym(R.lib, n);
            if (!p && R.why.empty()) R.why = std::string("librccl lacks ") + n;
            return p;
        };
        R.GetUniqueId = (int (*)(UniqueId *))sym("ncclGetUniqueId");
        R.CommInitRank = (int (*)(Comm *, int, UniqueId, int))sym("ncclCommInitRank");
        R.CommDestroy = (int (*)(Comm))sym("ncclCommDestroy");
        R.Send = (int (*)(const void *, size_t, int, int, Comm, hipStream_t))sym("ncclSend");
        R.Recv = (int (*)(void *, size_t, int, int, Comm, hipStream_t))sym("ncclRecv");
        R.GroupStart = (int (*)())sym("ncclGroupStart");
        R.GroupEnd = (int (*)())sym("ncclGroupEnd");
        R.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
    });
    return R;
}

std::string g_exchange_error;           // failures before a context is at hand (ts_exchange_unique_id)
std::mutex g_exchange_error_mtx;

}  // namespace

struct ts_exchange {
    ts_ctx *ctx = nullptr;
    Comm comm = nullptr;
    int rank = 0, n_ranks = 1;
};

#define RCCL_TRY(ctx, R, expr)                                                                         \
    do {                                                                                               \
        const int _e = (expr);                                                                         \
        if (_e != 0)                                                                                   \
            return (ctx)->fail(TS_ERR_HIP, std::string(#expr) + ": " + ((R).GetErrorString ? (R).GetErrorString(_e) : "RCCL error")); \
    } while (0)

extern "C" {

int ts_exchange_unique_id(void *id_out) {
    if (!id_out) return TS_ERR_INVALID_ARG;
    Rccl &R = rccl();
    if (!R.why.empty()) { std::lock_guard<std::mutex> g(g_exchange_error_mtx); g_exchange_error = R.why; return TS_ERR_UNSUPPORTED; }
    UniqueId id;
    const int e = R.GetUniqueId(&id);
    if (e != 0) { std::lock_guard<std::mutex> g(g_exchange_error_mtx); g_exchange_error = std::string("ncclGetUniqueId: ") + R.GetErrorString(e); return TS_ERR_HIP; }
    std::memcpy(id_out, id.internal, sizeof id.internal);
    return TS_OK;
}

ts_exchange *ts_exchange_create(ts_ctx *ctx, const void *id, int rank, int n_ranks) {
    if (!ctx) return nullptr;
    if (!id || n_ranks < 1 || rank < 0 || rank >= n_ranks) { ctx->fail(TS_ERR_INVALID_ARG, "ts_exchange_create: bad rank, size or id"); return nullptr; }
    if (ctx->device == kNoDevice) { ctx->fail(TS_ERR_NO_DEVICE, "planning-only context: no HIP device behind it"); return nullptr; }
    Rccl &R = rccl();
    if (!R.why.empty()) { ctx->fail(TS_ERR_UNSUPPORTED, R.why); return nullptr; }
    DeviceGuard g(ctx->device);                    // the communicator belongs to the context's device
    if (g.error() != hipSuccess) { ctx->fail(TS_ERR_HIP, "hipSetDevice failed"); return nullptr; }
    UniqueId uid;
    std::memcpy(uid.internal, id, sizeof uid.internal);
    Comm comm = nullptr;
    const int e = R.CommInitRank(&comm, n_ranks, uid, rank);
    if (e != 0) { ctx->fail(TS_ERR_HIP, std::string("ncclCommInitRank: ") + R.GetErrorString(e)); return nullptr; }
    ts_exchange *x = new ts_exchange;
    x->ctx = ctx; x->comm = comm; x->rank = rank; x->n_ranks = n_ranks;
    return x;
}

void ts_exchange_destroy(ts_exchange *x) {
    if (!x) return;
    if (x->comm) { DeviceGuard g(x->ctx->device); (void)rccl().CommDestroy(x->comm); }
    delete x;
}

// One grouped round: rank != dst sends its message; dst posts a receive per other rank (and, when d_recv[dst] is given
// and is not d_msg itself, sends to and receives from itself: the loop-back a one-GPU box rehearses the call pattern
// with).  Nothing is read back; the caller waits on `stream` before it reuses the buffers.
int ts_exchange_gather(ts_exchange *x, int dst, const void *d_msg, uint64_t my_bytes, void *const *d_recv,
                       const uint64_t *msg_bytes, void *stream) {
    if (!x) return TS_ERR_INVALID_ARG;
    ts_ctx *c = x->ctx;
    if (dst < 0 || dst >= x->n_ranks) return c->fail(TS_ERR_INVALID_ARG, "ts_exchange_gather: dst is not a rank");
    if (my_bytes && !d_msg) return c->fail(TS_ERR_INVALID_ARG, "ts_exchange_gather: null message");
    if (x->rank == dst && (!d_recv || !msg_bytes)) return c->fail(TS_ERR_INVALID_ARG, "ts_exchange_gather: the receiving rank passes d_recv and msg_bytes");
    DEVICE_TRY(c);
    Rccl &R = rccl();
    hipStream_t st = (hipStream_t)stream;
    RCCL_TRY(c, R, R.GroupStart());
    int rc = TS_OK;
    auto post = [&](int e, const char *what) { if (e != 0 && rc == TS_OK) rc = c->fail(TS_ERR_HIP, std::string(what) + ": " + R.GetErrorString(e)); };
    if (x->rank != dst) {
        post(R.Send(d_msg, (size_t)my_bytes, kUint8, dst, x->comm, st), "ncclSend");
    } else {
        for (int p = 0; p < x->n_ranks && rc == TS_OK; ++p) {
            if (p == dst) {
                if (d_recv[p] && d_recv[p] != d_msg) {
                    if (msg_bytes[p] != my_bytes) { rc = c->fail(TS_ERR_INVALID_ARG, "ts_exchange_gather: msg_bytes[dst] differs from this rank's message"); break; }
                    post(R.Send(d_msg, (size_t)my_bytes, kUint8, dst, x->comm, st), "ncclSend (self)");
                    post(R.Recv(d_recv[p], (size_t)msg_bytes[p], kUint8, dst, x->comm, st), "ncclRecv (self)");
                }
                continue;
            }
            if (!d_recv[p]) { rc = c->fail(TS_ERR_INVALID_ARG, "ts_exchange_gather: no receive buffer for a rank"); break; }
            post(R.Recv(d_recv[p], (size_t)msg_bytes[p], kUint8, p, x->comm, st), "ncclRecv");
        }
    }
    const int e = R.GroupEnd();                       // (always closed: a group left open would swallow the next call)
    if (e != 0 && rc == TS_OK) rc = c->fail(TS_ERR_HIP, std::string("ncclGroupEnd: ") + R.GetErrorString(e));
    return rc;
}

const char *ts_exchange_last_error(void) {
    std::lock_guard<std::mutex> g(g_exchange_error_mtx);
    static thread_local std::string copy;
    copy = g_exchange_error;
    return copy.c_str();
}

}  // extern "C"
