// Rate of per-lane gathers by access width on gfx950 (4, 8, 16 bytes per lane), random and in runs of 4 lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
template <class T> __device__ float first(const T &v);
template <> __device__ float first<float>(const float &v) { return v; }
template <> __device__ float first<float2>(const float2 &v) { return v.x + v.y; }
template <> __device__ float first<float4>(const float4 &v) { return v.x + v.w; }
template <class T, int SPAN>
__global__ __launch_bounds__(256) void gather(const T *__restrict__ tab, unsigned mask, int iters, float *out)
{
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned grp = tid / SPAN, sub = tid % SPAN;
    unsigned x = grp * 2654435761u + 12345u;
    float acc = 0.f;
    for (int i = 0; i < iters; i += 8) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x = x * 1664525u + 1013904223u;
            const unsigned idx = (((x >> 8) & mask) & ~(unsigned)(SPAN - 1)) + sub;
            v[u] = tab[idx];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += first(v[u]);
    }
    if (acc == 123.456f) out[tid] = acc;
}
template <class T, int SPAN>
static double run(unsigned bytes, float *out)
{
    const int nblk = 256 * 16, iters = 256;
    const unsigned n = bytes / sizeof(T);
    T *tab;
    (void)hipMalloc(&tab, bytes);
    (void)hipMemset(tab, 0, bytes);
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(a);
        hipLaunchKernelGGL((gather<T, SPAN>), dim3(nblk), dim3(256), 0, 0, tab, n - 1, iters, out);
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        (void)hipEventElapsedTime(&ms, a, b);
    }
    (void)hipFree(tab);
    return (double)nblk * 256 * iters / ms * 1e-6;
}
int main()
{
    float *out;
    (void)hipMalloc(&out, (size_t)256 * 16 * 256 * 4);
    for (unsigned kb : {16u, 1024u, 16384u}) {
        const unsigned bytes = kb * 1024u;
        printf("table %6u KB  G lane-loads/s:  4B random %.0f run4 %.0f run16 %.0f | 8B random %.0f run4 %.0f run16 %.0f | 16B random %.0f run4 %.0f run16 %.0f\n", kb,
               run<float, 1>(bytes, out), run<float, 4>(bytes, out), run<float, 16>(bytes, out),
               run<float2, 1>(bytes, out), run<float2, 4>(bytes, out), run<float2, 16>(bytes, out),
               run<float4, 1>(bytes, out), run<float4, 4>(bytes, out), run<float4, 16>(bytes, out));
    }
    return 0;
}
