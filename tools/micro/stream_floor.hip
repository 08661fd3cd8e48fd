// Floor of the converged ICP iteration: how long does it take just to READ the three per-query streams of the step
// kernel (float4 + float4 + uint per query, 4 194 304 queries = 151 MB) with a trivial per-block reduction, in
// workgroups of 128 threads -- once from HBM (buffers larger than the MALL in between), and repeatedly (MALL-resident).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(128) void rd(const float4 *__restrict__ a, const float4 *__restrict__ b,
                                          const unsigned *__restrict__ c, float *__restrict__ out, int qpt)
{
    float acc = 0.f;
    for (int q = 0; q < qpt; ++q) {
        const size_t i = ((size_t)blockIdx.x * qpt + q) * 128 + threadIdx.x;
        const float4 x = a[i], y = b[i];
        const unsigned z = c[i];
        acc += x.x * y.x + x.y * y.y + x.z * y.z + x.w + y.w + (float)(z & 0xffff);
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    __shared__ float s[2];
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = s[0] + s[1];
}
int main()
{
    const size_t n = 64ull * 65536;
    float4 *a, *b;
    unsigned *c;
    float *out;
    char *flush;
    (void)hipMalloc(&a, n * 16);
    (void)hipMalloc(&b, n * 16);
    (void)hipMalloc(&c, n * 4);
    (void)hipMalloc(&out, n / 128 * 4);
    (void)hipMalloc(&flush, 1ull << 30);
    (void)hipMemset(a, 0, n * 16);
    (void)hipMemset(b, 0, n * 16);
    (void)hipMemset(c, 0, n * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int qpt : {1, 2, 4}) {
        const int nblk = (int)(n / 128 / qpt);
        float best_hot = 1e9f, best_cold = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            float ms;
            (void)hipMemsetAsync(flush, rep, 1ull << 30, 0);  // evicts the MALL
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(rd, dim3(nblk), dim3(128), 0, 0, a, b, c, out, qpt);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (rep) best_cold = ms < best_cold ? ms : best_cold;
            for (int k = 0; k < 3; ++k) {
                (void)hipEventRecord(e0);
                hipLaunchKernelGGL(rd, dim3(nblk), dim3(128), 0, 0, a, b, c, out, qpt);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (rep) best_hot = ms < best_hot ? ms : best_hot;
            }
        }
        printf("queries per thread %d (%d workgroups): after a 1 GiB fill %.1f us (%.2f TB/s), repeated %.1f us (%.2f TB/s)\n", qpt, nblk,
               1e3 * best_cold, n * 36.0 / (best_cold * 1e-3) / 1e12, 1e3 * best_hot, n * 36.0 / (best_hot * 1e-3) / 1e12);
    }
    return 0;
}
