// common.hpp -- context, error plumbing and host<->HBM argument staging shared by
// every translation unit of libgpscal_hip.so.  gfx950 only; no CPU fallback.
#pragma once

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/gpscal.h"

struct gpscal_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
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

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipStream_t pool_stream = nullptr;  // set: stream-ordered allocation (hipMallocAsync pool)
    bool pooled = false;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release()
    {
        if (p) {
            if (pooled) (void)hipFreeAsync(p, pool_stream);
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
    // Stream-ordered allocation from the device pool: for per-call temporaries.  Freed in
    // stream order too, so no device-wide synchronisation (hipFree would add one).
    hipError_t alloc_async(size_t count, hipStream_t stream)
    {
        release();
        if (count == 0) count = 1;
        pooled = true;
        pool_stream = stream;
        hipError_t e = hipMallocAsync(reinterpret_cast<void **>(&p), count * sizeof(T), stream);
        if (e == hipSuccess) {
            n = count;
            if (poison()) (void)hipMemsetAsync(p, poison_byte(), count * sizeof(T), stream);
        } else
            pooled = false;
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
