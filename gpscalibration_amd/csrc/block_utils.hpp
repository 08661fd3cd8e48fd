// block_utils.hpp -- workgroup-wide helpers shared by sr.hip and loam_pipeline.hip: ordered
// compaction ranks, a bitonic sort over LDS or HBM keys, LDS bitsets, and the VoxelGrid filter
// (pcl::VoxelGrid<PointXYZI> restated) run by one 512-thread workgroup.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

namespace gpscal {

constexpr int SBLOCK = 512;
constexpr int SWAVES = SBLOCK / 64;
constexpr int SR_MAX_POINTS = 60000;  // POINTSNUM, common.h:15: the size of the reference's work arrays
constexpr int SR_BIT_WORDS = (SR_MAX_POINTS + 31) / 32 + 1;
constexpr int LDS_KEYS = 4096;  // sectors / voxel clouds up to this size sort in LDS
constexpr int N_RINGS = 16;
constexpr double PI_D = 3.14159265358979323846;

// ---------------------------------------------------------------- block helpers
struct BlockShared {
    int wcnt[SWAVES][N_RINGS];
    int base[N_RINGS + 1];
    int ring_cnt[N_RINGS];
    int bnd[N_RINGS];
    int scan_start[N_RINGS], scan_end[N_RINGS];
    int first_fin, last_fin, hp_idx, cloud_size;
    float start_ori, end_ori;
    int counts[5];
    int overflow;
    // voxel grid
    float red[SWAVES][6];
    int vg_minb[3], vg_mul[3], vg_copy, vg_m, vg_div2;
};

// rank of this thread among the threads with flag set (thread order), and the block total
__device__ __forceinline__ int block_rank(BlockShared &S, bool flag, int &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    if (lane == 0) S.wcnt[wave][0] = __popcll(m);
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SWAVES; ++w) {
        const int c = S.wcnt[w][0];
        before += w < wave ? c : 0;
        tot += c;
    }
    __syncthreads();
    total = tot;
    return before + __popcll(m & ((1ull << lane) - 1ull));
}

__device__ __forceinline__ int next_pow2(int n)
{
    int p = 1;
    while (p < n) p <<= 1;
    return p;
}

// ascending bitonic sort of K[0, np2) by the whole workgroup
__device__ inline void block_bitonic_sort(unsigned long long *K, int np2)
{
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < (np2 >> 1); t += SBLOCK) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int p = i | j;
                const bool up = (i & k) == 0;
                const unsigned long long a = K[i], b = K[p];
                if ((a > b) == up) {
                    K[i] = b;
                    K[p] = a;
                }
            }
            __syncthreads();
        }
}

// Stable LSD radix sort of the 64-bit keys K[0, n) by bits [32, 32 + bits) -- the voxel number of block_voxel_grid's
// keys (cell << 32 | input index; the keys come in input order, so a stable sort by the cell IS the sort by the whole
// key) -- by the whole workgroup, 4 bits per pass, T[0, n) as the second buffer; the result ends in K.  Thread t owns
// the keys [t c, (t + 1) c): it counts its digits into its own column of cnt[16][SBLOCK] (32 KiB of LDS), an
// exclusive scan over the 8 192 counters in (digit, thread) order gives every (digit, thread) its first slot, and
// the thread moves its keys in order.  ~2 (bits / 4) sweeps over the keys against log2(n)^2 / 2 of the bitonic
// network: clouds above the LDS sort size no longer pay ~100 passes of one workgroup over global memory.
__device__ inline void block_radix_sort_cells(BlockShared &S, unsigned long long *K, unsigned long long *T, int n,
                                              int bits, unsigned *cnt)
{
    const int c = (n + SBLOCK - 1) / SBLOCK;
    const int i0 = min(n, (int)threadIdx.x * c), i1 = min(n, i0 + c);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long *src = K, *dst = T;
    for (int shift = 32; shift < 32 + bits; shift += 4) {
#pragma unroll
        for (int d = 0; d < 16; ++d) cnt[d * SBLOCK + threadIdx.x] = 0u;
        for (int i = i0; i < i1; ++i) cnt[(int)((src[i] >> shift) & 15ull) * SBLOCK + threadIdx.x] += 1u;
        __syncthreads();
        // exclusive scan of cnt[0, 16 * SBLOCK): 16 consecutive counters per thread, then the threads' sums
        unsigned loc[16], sum = 0u;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            loc[k] = cnt[threadIdx.x * 16 + k];
            sum += loc[k];
        }
        unsigned inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_up(inc, o);
            if (lane >= o) inc += v;
        }
        if (lane == 63) S.wcnt[wave][0] = (int)inc;
        __syncthreads();
        unsigned before = inc - sum;
#pragma unroll
        for (int w = 0; w < SWAVES; ++w) before += w < wave ? (unsigned)S.wcnt[w][0] : 0u;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            cnt[threadIdx.x * 16 + k] = before;
            before += loc[k];
        }
        __syncthreads();
        for (int i = i0; i < i1; ++i) {
            const unsigned long long k = src[i];
            const int slot = (int)((k >> shift) & 15ull) * SBLOCK + threadIdx.x;
            dst[cnt[slot]] = k;
            cnt[slot] += 1u;
        }
        __syncthreads();
        unsigned long long *t = src;
        src = dst;
        dst = t;
    }
    if (src != K) {
        for (int i = threadIdx.x; i < n; i += SBLOCK) K[i] = src[i];
        __syncthreads();
    }
}

__device__ __forceinline__ bool get_bit(unsigned *bits, int i)
{
    return (__hip_atomic_load(&bits[i >> 5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> (i & 31)) & 1u;
}
__device__ __forceinline__ void set_bit(unsigned *bits, int i) { atomicOr(&bits[i >> 5], 1u << (i & 31)); }

__device__ __forceinline__ float atan2_f(float y, float x) { return (float)atan2((double)y, (double)x); }
__device__ __forceinline__ float atan_f(float x) { return (float)atan((double)x); }

__device__ __forceinline__ int ring_of(int a)
{
    // SR:307-325
    switch (a) {
    case -15: return 0;
    case -13: return 1;
    case -11: return 2;
    case -9: return 3;
    case -7: return 4;
    case -5: return 5;
    case -4: return 6;
    case -3: return 7;
    case -2: return 8;
    case -1: return 9;
    case 0: return 10;
    case 1: return 11;
    case 3: return 12;
    case 5: return 13;
    case 7: return 14;
    case 9: return 15;
    default: return -1;
    }
}

__device__ __forceinline__ bool fin3(float x, float y, float z) { return isfinite(x) && isfinite(y) && isfinite(z); }

// ---------------------------------------------------------------- voxel grid
// pcl::VoxelGrid (PCL 1.8.0 voxel_grid.hpp applyFilter) of src[0, n) with a cubic leaf:
// centroids of all four fields, output ordered by cell id, points of a cell summed in input
// order.  Appends at dst[*count ...] (bounded by cap; S.overflow is raised beyond it).
// Called by every thread of the workgroup; *count is a workgroup-shared counter.
__device__ inline void block_voxel_grid(BlockShared &S, const float4 *__restrict__ src, int n, float leaf,
                                 float4 *__restrict__ dst, int cap, int *count, unsigned long long *lds_keys,
                                 unsigned long long *g_keys, int g_cap, int lds_cap = LDS_KEYS)
{
    if (n <= 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float inv = 1.0f / leaf;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = threadIdx.x; i < n; i += SBLOCK) {
        const float4 p = src[i];
        if (!fin3(p.x, p.y, p.z)) continue;
        mn[0] = fminf(mn[0], p.x); mx[0] = fmaxf(mx[0], p.x);
        mn[1] = fminf(mn[1], p.y); mx[1] = fmaxf(mx[1], p.y);
        mn[2] = fminf(mn[2], p.z); mx[2] = fmaxf(mx[2], p.z);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int o = 32; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o));
        }
    if (lane == 0)
        for (int a = 0; a < 3; ++a) {
            S.red[wave][a] = mn[a];
            S.red[wave][3 + a] = mx[a];
        }
    __syncthreads();
    if (threadIdx.x == 0) {
        float lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
            lo[a] = INFINITY;
            hi[a] = -INFINITY;
            for (int w = 0; w < SWAVES; ++w) {
                lo[a] = fminf(lo[a], S.red[w][a]);
                hi[a] = fmaxf(hi[a], S.red[w][3 + a]);
            }
        }
        S.vg_copy = 0;
        S.vg_m = lo[0] <= hi[0] ? 1 : 0;  // any finite point
        if (S.vg_m) {
            long long d[3];
            int minb[3], divb[3];
            for (int a = 0; a < 3; ++a) {
                d[a] = (long long)((hi[a] - lo[a]) * inv) + 1;
                minb[a] = (int)floorf(lo[a] * inv);
                divb[a] = (int)floorf(hi[a] * inv) - minb[a] + 1;
                S.vg_minb[a] = minb[a];
            }
            if (d[0] * d[1] * d[2] > 2147483647LL) S.vg_copy = 1;  // PCL warns and returns the input
            S.vg_mul[0] = 1;
            S.vg_mul[1] = divb[0];
            S.vg_mul[2] = divb[0] * divb[1];
            S.vg_div2 = divb[2];
        }
    }
    __syncthreads();
    if (!S.vg_m) return;
    if (S.vg_copy) {
        const int base = *count;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += SBLOCK) {
            if (base + i < cap) dst[base + i] = src[i];
            else S.overflow = 1;
        }
        if (threadIdx.x == 0) *count = base + n;
        __syncthreads();
        return;
    }
    int np2 = next_pow2(n);
    // `lds_cap`: keys the caller's LDS buffer holds (a kernel launched with more dynamic LDS sorts larger clouds there)
    if (np2 > lds_cap && np2 > g_cap) {  // host sized the scratch from the input counts; cannot happen
        S.overflow = 1;
        return;
    }
    // above the LDS size, with room for a second key buffer: radix sort by the voxel number instead of the bitonic
    // network in global memory (no padding to a power of two then)
    // (also when the keys would fit the LDS buffer but leave no room for the counters: 105 network passes otherwise)
    const bool radix_g = (np2 > lds_cap || (n > 1024 && n + 4096 > lds_cap && lds_cap > LDS_KEYS)) && g_cap >= 2 * n;
    // in LDS as well once the caller's buffer also holds the 32 KiB of counters behind the keys (second buffer: the
    // global scratch): ~6 sweeps over the keys against 91 / 105 passes of the network at 8 192 / 16 384 keys
    const bool radix_l = !radix_g && np2 <= lds_cap && n > 1024 && n + 4096 <= lds_cap && g_cap >= n;
    const bool radix = radix_g || radix_l;
    unsigned long long *K = np2 <= lds_cap && !radix_g ? lds_keys : g_keys;
    if (radix) np2 = n;
    for (int i = threadIdx.x; i < np2; i += SBLOCK) {
        unsigned long long key = ~0ull;
        if (i < n) {
            const float4 p = src[i];
            if (fin3(p.x, p.y, p.z)) {
                const int i0 = (int)(floorf(p.x * inv) - (float)S.vg_minb[0]);
                const int i1 = (int)(floorf(p.y * inv) - (float)S.vg_minb[1]);
                const int i2 = (int)(floorf(p.z * inv) - (float)S.vg_minb[2]);
                const int idx = i0 * S.vg_mul[0] + i1 * S.vg_mul[1] + i2 * S.vg_mul[2];
                key = ((unsigned long long)(unsigned)idx << 32) | (unsigned)i;
            }
        }
        K[i] = key;
    }
    __syncthreads();
    if (radix) {
        // bits: 2^bits > number of voxels, so a non-finite point's all-ones key sorts behind every voxel
        const long long cells = (long long)S.vg_mul[2] * S.vg_div2;
        int bits = 1;
        while (bits < 32 && (1ll << bits) <= cells) ++bits;
        if (radix_g) block_radix_sort_cells(S, K, g_keys + n, n, bits, reinterpret_cast<unsigned *>(lds_keys));
        else block_radix_sort_cells(S, K, g_keys, n, bits, reinterpret_cast<unsigned *>(lds_keys + n));
    } else {
        block_bitonic_sort(K, np2);
    }
    // one thread per run of equal cell ids
    const int base = *count;
    __syncthreads();
    int produced = 0;
    for (int j0 = 0; j0 < n; j0 += SBLOCK) {
        const int j = j0 + threadIdx.x;
        unsigned long long kj = ~0ull;
        bool start = false;
        if (j < n) {
            kj = K[j];
            start = kj != ~0ull && (j == 0 || (unsigned)(K[j - 1] >> 32) != (unsigned)(kj >> 32));
        }
        int tot;
        const int r = block_rank(S, start, tot);
        if (start) {
            float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
            int e = j;
            const unsigned cell = (unsigned)(kj >> 32);
            while (e < n) {
                const unsigned long long ke = K[e];
                if (ke == ~0ull || (unsigned)(ke >> 32) != cell) break;
                const float4 p = src[(unsigned)ke];
                sx += p.x; sy += p.y; sz += p.z; sw += p.w;
                ++e;
            }
            const float c = (float)(e - j);
            const int o = base + produced + r;
            if (o < cap) dst[o] = make_float4(sx / c, sy / c, sz / c, sw / c);
            else S.overflow = 1;
        }
        produced += tot;
    }
    __syncthreads();
    if (threadIdx.x == 0) *count = base + produced;
    __syncthreads();
}


}  // namespace gpscal
