// api.hip -- context lifecycle of libgpscal_hip.so and the RCCL pose-chain
// all-gather.  gfx950 only; there is no CPU fallback anywhere in this library.
#include "common.hpp"

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>

using namespace gpscal;

extern "C" const char *gpscal_strerror(int code)
{
    switch (code) {
    case GPSCAL_OK: return "ok";
    case GPSCAL_EINVAL: return "invalid argument";
    case GPSCAL_ENODEV: return "no usable gfx950 device";
    case GPSCAL_EHIP: return "HIP runtime error";
    case GPSCAL_ENOMEM: return "out of memory";
    case GPSCAL_ESIZE: return "track sizes differ / empty segment";
    case GPSCAL_ERANGE: return "capacity too small";
    case GPSCAL_ECOMM: return "RCCL error";
    default: return "unknown error";
    }
}

extern "C" int gpscal_create(gpscal_ctx **out, int device_id, unsigned flags)
{
    (void)flags;
    if (!out) return GPSCAL_EINVAL;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) {
        (void)hipGetLastError();
        return GPSCAL_ENODEV;
    }
    if (device_id < 0 || device_id >= count) return GPSCAL_ENODEV;
    auto *ctx = new gpscal_ctx;
    ctx->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess || hipGetDeviceProperties(&ctx->prop, device_id) != hipSuccess) {
        delete ctx;
        return GPSCAL_ENODEV;
    }
    // The code objects in this library are gfx950 only.
    if (strncmp(ctx->prop.gcnArchName, "gfx950", 6) != 0) {
        delete ctx;
        return GPSCAL_ENODEV;
    }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return GPSCAL_ENODEV;
    }
    cache_revive(ctx->stream);
    *out = ctx;
    return GPSCAL_OK;
}

extern "C" int gpscal_comm_destroy(gpscal_ctx *ctx);

extern "C" int gpscal_destroy(gpscal_ctx *ctx)
{
    if (!ctx) return GPSCAL_EINVAL;
    (void)hipSetDevice(ctx->device);
    if (ctx->comm) (void)gpscal_comm_destroy(ctx);
    if (ctx->order_event) (void)hipEventDestroy(ctx->order_event);
    for (int k = 0; k < gpscal_ctx::MAX_SIDE; ++k) {
        if (ctx->side_stream[k]) {
            (void)hipStreamSynchronize(ctx->side_stream[k]);
            cache_retire(ctx->side_stream[k]);  // (a build's source grouping leaves cached blocks here)
            (void)hipStreamDestroy(ctx->side_stream[k]);
        }
        if (ctx->side_event[k]) (void)hipEventDestroy(ctx->side_event[k]);
    }
    if (ctx->worker_stream) {
        (void)hipStreamSynchronize(ctx->worker_stream);
        cache_retire(ctx->worker_stream);
        (void)hipStreamDestroy(ctx->worker_stream);
    }
    if (ctx->stream) {
        (void)hipStreamSynchronize(ctx->stream);
        cache_retire(ctx->stream);  // the stream's cached temporaries go back to the driver; a batch or index that
                                    // outlives the context frees its blocks plainly (cache_free)
        (void)hipStreamDestroy(ctx->stream);
    }
    delete ctx;
    return GPSCAL_OK;
}

extern "C" int gpscal_sync(gpscal_ctx *ctx)
{
    if (!ctx) return GPSCAL_EINVAL;
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}

extern "C" void *gpscal_stream(gpscal_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

static int order_streams(gpscal_ctx *ctx, hipStream_t first, hipStream_t then)
{
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->order_event) GPSCAL_HIP(ctx, hipEventCreateWithFlags(&ctx->order_event, hipEventDisableTiming));
    // the wait captures the event's state when it is enqueued: one event serves every call
    GPSCAL_HIP(ctx, hipEventRecord(ctx->order_event, first));
    GPSCAL_HIP(ctx, hipStreamWaitEvent(then, ctx->order_event, 0));
    return GPSCAL_OK;
}

extern "C" int gpscal_wait_for_stream(gpscal_ctx *ctx, void *producer_stream)
{
    if (!ctx) return GPSCAL_EINVAL;
    if ((hipStream_t)producer_stream == ctx->stream) return GPSCAL_OK;
    return order_streams(ctx, (hipStream_t)producer_stream, ctx->stream);
}

extern "C" int gpscal_make_stream_wait(gpscal_ctx *ctx, void *consumer_stream)
{
    if (!ctx) return GPSCAL_EINVAL;
    if ((hipStream_t)consumer_stream == ctx->stream) return GPSCAL_OK;
    return order_streams(ctx, ctx->stream, (hipStream_t)consumer_stream);
}

extern "C" const char *gpscal_last_error(gpscal_ctx *ctx) { return ctx ? ctx->last_error.c_str() : "no context"; }

extern "C" int gpscal_device_info(gpscal_ctx *ctx, char *buf, size_t cap)
{
    if (!ctx || !buf || cap == 0) return GPSCAL_EINVAL;
    snprintf(buf, cap, "%s %s CUs=%d HBM=%.0fGiB libgpscal_hip 0.1", ctx->prop.name, ctx->prop.gcnArchName,
             ctx->prop.multiProcessorCount, (double)ctx->prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
    return GPSCAL_OK;
}

// ------------------------------------------------------------------- RCCL
// librccl is loaded on first use so that single-GPU hosts never touch it.
namespace {
struct NcclId {
    char internal[GPSCAL_COMM_ID_BYTES];
};
typedef int (*fn_getid)(NcclId *);
typedef int (*fn_init)(void **, int, NcclId, int);
typedef int (*fn_destroy)(void *);
typedef int (*fn_allgather)(const void *, void *, size_t, int, void *, hipStream_t);
typedef int (*fn_bcast)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_group)(void);
typedef const char *(*fn_err)(int);
struct Rccl {
    void *h = nullptr;
    fn_getid get_id = nullptr;
    fn_init init = nullptr;
    fn_destroy destroy = nullptr;
    fn_allgather all_gather = nullptr;
    fn_bcast broadcast = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
    fn_err err = nullptr;
};
Rccl *rccl()
{
    static Rccl r;
    if (r.h) return &r;
    void *h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return nullptr;
    r.get_id = (fn_getid)dlsym(h, "ncclGetUniqueId");
    r.init = (fn_init)dlsym(h, "ncclCommInitRank");
    r.destroy = (fn_destroy)dlsym(h, "ncclCommDestroy");
    r.all_gather = (fn_allgather)dlsym(h, "ncclAllGather");
    r.broadcast = (fn_bcast)dlsym(h, "ncclBroadcast");
    r.group_start = (fn_group)dlsym(h, "ncclGroupStart");
    r.group_end = (fn_group)dlsym(h, "ncclGroupEnd");
    r.err = (fn_err)dlsym(h, "ncclGetErrorString");
    if (!r.get_id || !r.init || !r.destroy || !r.all_gather || !r.broadcast || !r.group_start || !r.group_end)
        return nullptr;
    r.h = h;
    return &r;
}
constexpr int kNcclDouble = 8;  // ncclFloat64 / ncclDouble
}  // namespace

extern "C" int gpscal_comm_unique_id(void *id_bytes)
{
    Rccl *r = rccl();
    if (!r || !id_bytes) return GPSCAL_ECOMM;
    NcclId id;
    memset(&id, 0, sizeof id);
    if (r->get_id(&id) != 0) return GPSCAL_ECOMM;
    memcpy(id_bytes, &id, sizeof id);
    return GPSCAL_OK;
}

extern "C" int gpscal_comm_init(gpscal_ctx *ctx, const void *id_bytes, int rank, int world)
{
    if (!ctx || !id_bytes || world < 1 || rank < 0 || rank >= world) return fail(ctx, GPSCAL_EINVAL, "gpscal_comm_init: bad argument");
    Rccl *r = rccl();
    if (!r) return fail(ctx, GPSCAL_ECOMM, "librccl.so could not be loaded");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    NcclId id;
    memcpy(&id, id_bytes, sizeof id);
    void *comm = nullptr;
    int rc = r->init(&comm, world, id, rank);
    if (rc != 0) return fail(ctx, GPSCAL_ECOMM, r->err ? r->err(rc) : "ncclCommInitRank failed");
    ctx->comm = comm;
    ctx->rank = rank;
    ctx->world = world;
    return GPSCAL_OK;
}

extern "C" int gpscal_comm_destroy(gpscal_ctx *ctx)
{
    if (!ctx) return GPSCAL_EINVAL;
    Rccl *r = rccl();
    if (ctx->comm && r) (void)r->destroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->world = 1;
    ctx->rank = 0;
    return GPSCAL_OK;
}

// Ragged all-gather of float64 pose chains: rank k contributes counts[k]
// doubles.  Equal counts use one ncclAllGather; ragged counts use one grouped
// ncclBroadcast per rank (payloads are KBs..MBs: latency-bound either way,
// SURVEY section 5 / 8e).
extern "C" int gpscal_allgather_chains(gpscal_ctx *ctx, const double *local, const int *counts, double *all)
{
    if (!ctx || !counts || !all) return fail(ctx, GPSCAL_EINVAL, "gpscal_allgather_chains: bad argument");
    if (!ctx->comm) return fail(ctx, GPSCAL_ECOMM, "gpscal_comm_init has not been called");
    Rccl *r = rccl();
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    const int W = ctx->world;
    size_t total = 0;
    bool equal = true;
    std::vector<size_t> offs(W + 1, 0);
    for (int k = 0; k < W; ++k) {
        if (counts[k] < 0) return fail(ctx, GPSCAL_EINVAL, "negative count");
        equal = equal && counts[k] == counts[0];
        offs[k + 1] = offs[k] + (size_t)counts[k];
    }
    total = offs[W];
    if (!local && counts[ctx->rank] > 0) return fail(ctx, GPSCAL_EINVAL, "gpscal_allgather_chains: local is NULL");
    // test hook: take the grouped-broadcast (ragged) path even when the counts are equal, so that a
    // one-rank box can run it
    if (getenv("GPSCAL_COMM_FORCE_RAGGED")) equal = false;
    InArg<double> in;
    OutArg<double> out;
    GPSCAL_HIP(ctx, in.bind(ctx, local, (size_t)counts[ctx->rank]));
    GPSCAL_HIP(ctx, out.bind(ctx, all, total));
    int rc = 0;
    if (equal) {
        if (counts[0] > 0) rc = r->all_gather(in.dev, out.dev, (size_t)counts[0], kNcclDouble, ctx->comm, ctx->stream);
    } else {
        // own slice in place, then every rank broadcasts its slice
        if (counts[ctx->rank] > 0)
            GPSCAL_HIP(ctx, hipMemcpyAsync(out.dev + offs[ctx->rank], in.dev, sizeof(double) * (size_t)counts[ctx->rank],
                                           hipMemcpyDeviceToDevice, ctx->stream));
        rc = r->group_start();
        for (int k = 0; k < W && rc == 0; ++k)
            if (counts[k] > 0)
                rc = r->broadcast(out.dev + offs[k], out.dev + offs[k], (size_t)counts[k], kNcclDouble, k, ctx->comm,
                                  ctx->stream);
        int rc2 = r->group_end();
        if (rc == 0) rc = rc2;
    }
    if (rc != 0) return fail(ctx, GPSCAL_ECOMM, r->err ? r->err(rc) : "RCCL collective failed");
    // device pointers in and out: the call returns after enqueue on the context's stream, like every other entry point
    // (gpscal_wait_for_stream / gpscal_make_stream_wait order it against the caller's streams).  A staged input must
    // outlive the collective and a host output must be complete on return: only then does the host wait.
    bool sync = in.tmp.p != nullptr;
    GPSCAL_HIP(ctx, out.commit(ctx, &sync));
    if (sync) GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}
