// common.hpp -- context, error plumbing and host<->HBM argument staging shared by
// every translation unit of libgpscal_hip.so.  gfx950 only; no CPU fallback.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/gpscal.h"

struct gpscal_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t order_event = nullptr;  // gpscal_wait_for_stream / gpscal_make_stream_wait
    // side streams of the ICP graph's independent chains (created on first use, shared by every scan batch)
    static constexpr int MAX_SIDE = 8;
    hipStream_t side_stream[MAX_SIDE] = {};
    hipEvent_t side_event[MAX_SIDE] = {};
    // worker stream of the LOAM chain's second node thread (created on first use, kept: its block cache then serves
    // every later run; destroyed and trimmed by gpscal_destroy)
    hipStream_t worker_stream = nullptr;
    hipDeviceProp_t prop{};
    std::string last_error;
    void *comm = nullptr;  // ncclComm_t, owned by comm.hip
    int rank = 0, world = 1;
};

namespace gpscal {

inline int fail(gpscal_ctx *ctx, int code, const char *what, hipError_t e = hipSuccess)
{
    if (ctx) {
        char buf[512];
        if (e != hipSuccess)
            snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
        else
            snprintf(buf, sizeof buf, "%s", what);
        ctx->last_error = buf;
    }
    return code;
}

#define GPSCAL_HIP(ctx, expr)                                                    \
    do {                                                                         \
        hipError_t e__ = (expr);                                                 \
        if (e__ != hipSuccess) return ::gpscal::fail((ctx), GPSCAL_EHIP, #expr, e__); \
    } while (0)

// True when ptr addresses device (HBM) memory the kernels can use in place.
inline bool is_device_ptr(const void *ptr)
{
    if (!ptr) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, ptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // plain malloc memory: not an error for us
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

// GPSCAL_POISON=<byte> fills every fresh device allocation with that byte (1 = 0xAB): a result that
// changes under it, or between two byte values, reads memory nobody wrote (debug aid).  0x41 makes stale
// floats 12.08 -- a plausible coordinate -- and 0xAB -1.2e-12.
inline int poison_byte()
{
    static const int v = [] {
        const char *e = getenv("GPSCAL_POISON");
        if (!e) return -1;
        const long b = strtol(e, nullptr, 0);
        return b == 1 ? 0xAB : (int)(b & 0xff);
    }();
    return v;
}
inline bool poison() { return poison_byte() >= 0; }

// Per-call grid sets of the LOAM chain come from the stream's block cache (no device-wide sync on release).
// GPSCAL_POOL_GRIDS=0 goes back to hipMalloc / hipFree (debugging aid).
inline bool pool_grids()
{
    static const bool on = [] {
        const char *e = getenv("GPSCAL_POOL_GRIDS");
        return e ? atoi(e) != 0 : true;
    }();
    return on;
}


// ---------------------------------------------------------------- per-stream block cache
// Per-call temporaries come from a cache of hipMalloc'd blocks owned by the library, one cache per stream:
// a block goes back to the cache of the stream its last user enqueued work on, and whoever takes it next
// enqueues on that same stream -- in-order execution is all the ordering reuse needs, so neither taking nor
// returning a block synchronises the device (hipMalloc / hipFree do).  This replaces hipMallocAsync /
// hipFreeAsync: with ROCm 7.2's own runtime the stream-ordered pool made this library abort inside
// gpscal_knn_build and return stale output buffers (the same binary is correct with plain allocations and
// under the older runtime PyTorch bundles); round 1's "pooled grid sets are not reproducible" was the same thing.
struct BlockCache {
    std::mutex mu;
    std::multimap<size_t, void *> free_blocks;    // cached, by size
    std::unordered_map<void *, size_t> handed_out;  // live blocks -> size
    size_t cached_bytes = 0;
    static size_t round_size(size_t bytes)
    {
        size_t s = 512;
        if (bytes <= (1u << 20)) {
            while (s < bytes) s <<= 1;
            return s;
        }
        const size_t step = bytes < (64ull << 20) ? (1ull << 20) : (16ull << 20);
        return (bytes + step - 1) / step * step;
    }
};
inline std::mutex &cache_registry_mutex()
{
    static std::mutex *m = new std::mutex;  // leaked on purpose: no static-destruction order to worry about
    return *m;
}
inline std::unordered_map<hipStream_t, BlockCache *> &cache_registry()
{
    static auto *reg = new std::unordered_map<hipStream_t, BlockCache *>;
    return *reg;
}
// streams whose cache has been retired (the stream was destroyed): a block that comes back for one of them
// -- a scan batch destroyed after its context -- goes straight to hipFree
inline std::unordered_set<hipStream_t> &cache_retired()
{
    static auto *dead = new std::unordered_set<hipStream_t>;
    return *dead;
}
inline BlockCache &cache_of(hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(cache_registry_mutex());
    BlockCache *&c = cache_registry()[stream];
    if (!c) c = new BlockCache;
    return *c;
}
// Frees every cached block of the stream (the caller has synchronised the stream).
inline void cache_trim(hipStream_t stream, size_t keep_bytes = 0)
{
    BlockCache &C = cache_of(stream);
    std::lock_guard<std::mutex> lk(C.mu);
    while (C.cached_bytes > keep_bytes && !C.free_blocks.empty()) {
        auto it = std::prev(C.free_blocks.end());  // largest first
        (void)hipFree(it->second);
        C.cached_bytes -= it->first;
        C.free_blocks.erase(it);
    }
}
// A new stream starts with a live cache (the runtime may hand out the handle of a destroyed stream again).
inline void cache_revive(hipStream_t stream)
{
    std::lock_guard<std::mutex> lk(cache_registry_mutex());
    cache_retired().erase(stream);
}
// The stream is about to be destroyed (the caller has synchronised it): its cached blocks go back to the driver, its
// registry entry goes, and blocks still handed out are freed plainly when they come back.
inline void cache_retire(hipStream_t stream)
{
    cache_trim(stream);
    std::lock_guard<std::mutex> lk(cache_registry_mutex());
    auto it = cache_registry().find(stream);
    if (it != cache_registry().end()) {
        delete it->second;
        cache_registry().erase(it);
    }
    cache_retired().insert(stream);
}
// The context's k-th side stream and event, created on first use (the chains of a scan batch's graph, the two side
// chains of its build).  A side stream may own cached blocks (the source grouping of a build runs on one), so it is
// registered with the block cache like the context's own stream and retired in gpscal_destroy.
inline hipError_t side_stream_of(gpscal_ctx *ctx, int k, hipStream_t *st, hipEvent_t *ev)
{
    if (!ctx->side_stream[k]) {
        hipError_t e = hipStreamCreateWithFlags(&ctx->side_stream[k], hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        cache_revive(ctx->side_stream[k]);
    }
    if (!ctx->side_event[k]) {
        hipError_t e = hipEventCreateWithFlags(&ctx->side_event[k], hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    if (st) *st = ctx->side_stream[k];
    if (ev) *ev = ctx->side_event[k];
    return hipSuccess;
}
// Out of memory: what idles in ANY stream's cache is given back (up to an eighth of the device per stream may sit there).
inline void cache_trim_all()
{
    std::vector<hipStream_t> streams;
    {
        std::lock_guard<std::mutex> lk(cache_registry_mutex());
        for (auto &kv : cache_registry()) streams.push_back(kv.first);
    }
    for (hipStream_t st : streams) {
        (void)hipStreamSynchronize(st);  // work that uses a cached block may still be queued
        cache_trim(st);
    }
}
inline hipError_t cache_alloc(void **out, size_t bytes, hipStream_t stream)
{
    BlockCache &C = cache_of(stream);
    const size_t need = BlockCache::round_size(bytes);
    {
        std::lock_guard<std::mutex> lk(C.mu);
        auto it = C.free_blocks.lower_bound(need);
        if (it != C.free_blocks.end() && it->first <= need + need / 2) {  // close enough in size
            *out = it->second;
            C.cached_bytes -= it->first;
            C.handed_out[*out] = it->first;
            C.free_blocks.erase(it);
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(out, need);
    if (e != hipSuccess) {  // make room: drop what is cached, here and in every other stream's cache
        (void)hipGetLastError();
        cache_trim_all();
        e = hipMalloc(out, need);
        if (e != hipSuccess) return e;
    }
    std::lock_guard<std::mutex> lk(C.mu);
    C.handed_out[*out] = need;
    return hipSuccess;
}
// Cached (idle) bytes per stream before blocks go back to the driver: an eighth of the device's memory (36 GB of
// an MI355X's 288 GB), at least 8 GiB.  A fixed 8 GiB was less than the pools of one LOAM run over 48 segments
// (9.6 GB): runs that alternated between two shapes trimmed and re-allocated gigabytes every call (0.205 s
// against 0.053 s per run).
inline size_t cache_cap()
{
    static const size_t cap = [] {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
            (void)hipGetLastError();
            total_b = 0;
        }
        return std::max<size_t>(8ull << 30, total_b / 8);
    }();
    return cap;
}
inline void cache_free(void *p, hipStream_t stream)
{
    {
        std::lock_guard<std::mutex> lk(cache_registry_mutex());
        if (cache_retired().count(stream)) {  // the stream is gone: nothing to order the reuse against
            (void)hipFree(p);
            return;
        }
    }
    const size_t CACHE_CAP = cache_cap();
    BlockCache &C = cache_of(stream);
    bool trim = false;
    {
        std::lock_guard<std::mutex> lk(C.mu);
        auto it = C.handed_out.find(p);
        if (it == C.handed_out.end()) {  // not ours (should not happen): give it back the plain way
            (void)hipFree(p);
            return;
        }
        C.free_blocks.emplace(it->second, p);
        C.cached_bytes += it->second;
        C.handed_out.erase(it);
        trim = C.cached_bytes > CACHE_CAP;
    }
    if (trim) {
        (void)hipStreamSynchronize(stream);
        cache_trim(stream, CACHE_CAP / 4 * 3);
    }
}

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipStream_t pool_stream = nullptr;  // set: block of this stream's cache (see BlockCache)
    bool pooled = false;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) {
            if (pooled) cache_free(p, pool_stream);
            else (void)hipFree(p);
        }
        p = nullptr;
        n = 0;
    }
    // Plain hipMalloc: for buffers that outlive the call (indices, batch state).
    hipError_t alloc(size_t count)
    {
        release();
        if (count == 0) count = 1;
        pooled = false;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        if (e == hipSuccess) {
            n = count;
            if (poison()) {
                // the library's streams do not wait for the null stream: finish the fill before anyone writes
                (void)hipMemset(p, poison_byte(), count * sizeof(T));
                (void)hipDeviceSynchronize();
            }
        }
        return e;
    }
    // A block of the stream's cache: for per-call temporaries.  No device-wide synchronisation on either
    // end (hipMalloc / hipFree have one); the buffer may only be used by work enqueued on `stream`.
    hipError_t alloc_async(size_t count, hipStream_t stream)
    {
        release();
        if (count == 0) count = 1;
        hipError_t e = cache_alloc(reinterpret_cast<void **>(&p), count * sizeof(T), stream);
        if (e == hipSuccess) {
            pooled = true;
            pool_stream = stream;
            n = count;
            if (poison()) (void)hipMemsetAsync(p, poison_byte(), count * sizeof(T), stream);
        } else {
            p = nullptr;
            pooled = false;
        }
        return e;
    }
};

// Read-only argument: device pointers pass through, host pointers are staged.
template <class T>
struct InArg {
    const T *dev = nullptr;
    DevBuf<T> tmp;
    hipError_t bind(gpscal_ctx *ctx, const T *ptr, size_t count)
    {
        if (!ptr || count == 0) {
            dev = nullptr;
            return hipSuccess;
        }
        if (is_device_ptr(ptr)) {
            dev = ptr;
            return hipSuccess;
        }
        hipError_t e = tmp.alloc_async(count, ctx->stream);
        if (e != hipSuccess) return e;
        e = hipMemcpyAsync(tmp.p, ptr, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream);
        dev = tmp.p;
        return e;
    }
};

// Output argument: device pointers are written in place; host pointers get a
// device scratch buffer that commit() copies back (caller then syncs).
template <class T>
struct OutArg {
    T *dev = nullptr;
    T *host = nullptr;
    size_t count = 0;
    DevBuf<T> tmp;
    hipError_t bind(gpscal_ctx *ctx, T *ptr, size_t cnt)
    {
        count = cnt;
        if (!ptr || cnt == 0) {
            dev = nullptr;
            host = nullptr;
            return hipSuccess;
        }
        if (is_device_ptr(ptr)) {
            dev = ptr;
            return hipSuccess;
        }
        host = ptr;
        hipError_t e = tmp.alloc_async(cnt, ctx->stream);
        dev = tmp.p;
        return e;
    }
    // returns true when a host copy was enqueued (caller must sync the stream)
    hipError_t commit(gpscal_ctx *ctx, bool *needs_sync, size_t cnt_override = (size_t)-1)
    {
        if (!host) return hipSuccess;
        size_t c = cnt_override == (size_t)-1 ? count : cnt_override;
        *needs_sync = true;
        if (c == 0) return hipSuccess;
        return hipMemcpyAsync(host, dev, c * sizeof(T), hipMemcpyDeviceToHost, ctx->stream);
    }
};

inline int div_up(long long a, long long b) { return (int)((a + b - 1) / b); }

}  // namespace gpscal
