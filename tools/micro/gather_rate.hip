// Rate of 16-byte gathers on gfx950: one lane = one random float4 (A), against four adjacent lanes = four
// consecutive float4 of one random 64-byte segment (B), against 16 adjacent lanes = 256 B (C).  Table sizes from
// L1-resident to L2-resident.  hipcc --offload-arch=gfx950 -O3 gather_rate.hip -o gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int SPAN>
__global__ __launch_bounds__(256) void gather(const float4 *__restrict__ tab, unsigned mask, int iters, float *out)
{
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned grp = tid / SPAN, sub = tid % SPAN;
    unsigned x = grp * 2654435761u + 12345u;
    float acc = 0.f;
    for (int i = 0; i < iters; i += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            x = x * 1664525u + 1013904223u;
            const unsigned idx = (((x >> 8) & mask) & ~(unsigned)(SPAN - 1)) + sub;
            v[u] = tab[idx];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].w;
    }
    if (acc == 123.456f) out[tid] = acc;
}
int main()
{
    const int nblk = 256 * 24, iters = 256;
    float *out;
    hipMalloc(&out, (size_t)nblk * 256 * 4);
    for (unsigned logn : {10u, 13u, 16u, 20u, 22u}) {  // entries: 16 KB, 128 KB, 1 MB, 16 MB, 64 MB
        const unsigned n = 1u << logn;
        float4 *tab;
        hipMalloc(&tab, (size_t)n * 16);
        hipMemset(tab, 0, (size_t)n * 16);
        hipEvent_t a, b;
        hipEventCreate(&a);
        hipEventCreate(&b);
        float ms[3];
        for (int k = 0; k < 3; ++k) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(a);
                if (k == 0) hipLaunchKernelGGL(gather<1>, dim3(nblk), dim3(256), 0, 0, tab, n - 1, iters, out);
                if (k == 1) hipLaunchKernelGGL(gather<4>, dim3(nblk), dim3(256), 0, 0, tab, n - 1, iters, out);
                if (k == 2) hipLaunchKernelGGL(gather<16>, dim3(nblk), dim3(256), 0, 0, tab, n - 1, iters, out);
                hipEventRecord(b);
                hipEventSynchronize(b);
                hipEventElapsedTime(&ms[k], a, b);
            }
        }
        const double loads = (double)nblk * 256 * iters;
        printf("table %8u KB: span1 %.1f G lane-loads/s (%.2f per CU-cycle @2.4GHz)  span4 %.1f (%.2f)  span16 %.1f (%.2f)\n",
               n / 64, loads / ms[0] * 1e-6, loads / ms[0] * 1e-6 / 256 / 2.4, loads / ms[1] * 1e-6,
               loads / ms[1] * 1e-6 / 256 / 2.4, loads / ms[2] * 1e-6, loads / ms[2] * 1e-6 / 256 / 2.4);
        hipFree(tab);
    }
    return 0;
}
