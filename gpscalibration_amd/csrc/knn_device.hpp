// knn_device.hpp -- device side of the multi-level uniform-grid k-NN: descriptors,
// the exact query (knn_query) and its helpers.  Shared by knn_icp.hip (ICP, k-NN ABI)
// and loam.hip (LOAM odometry correspondences).  Layout and algorithm: knn_icp.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "wave_reduce.hpp"

namespace gpscal {

constexpr int MAX_LEVELS = 8;
constexpr int BLOCK = 256;
constexpr int NACC_PLAIN = 17;   // n, sum p(3), sum q(3), sum p q^T(9), sum dist
constexpr int NACC_WEIGHTED = 25;  // + sw, sw2, sum w2 p(3), sum w2 q(3) (w-sums replace n)

struct GridDesc {
    float ox, oy, oz, inv_h;
    float h, margin;
    int nx, ny, nz;
    int tile;  // 1: cells numbered in 8x8 xy tiles (source grouping only, never searched)
    long long cell_base;
};

struct PairDesc {
    long long tgt_off;  // into tgt4
    long long src_off;  // into src4 / nn arrays / weights
    int m, n;
    int nlevels;
    int pblk_off;  // first partial slot of this pair
    int pblk_cnt;
    int coarse_from;  // first level whose cell counters are aggregated in LDS at build time
    GridDesc lv[MAX_LEVELS];
};

// Search statistics (tools/search_stats.py; -DGPSCAL_STATS builds only): counters per ICP iteration,
// NSTAT per iteration, read back through gpscal_debug_stats.
#ifdef GPSCAL_STATS
constexpr int NSTAT = 24, STAT_ITERS = 64;
static __device__ unsigned long long g_stats[NSTAT * STAT_ITERS];
static __device__ int g_stat_iter;
#define STAT_ADD(k, v) atomicAdd(&g_stats[g_stat_iter * NSTAT + (k)], (unsigned long long)(v))
#define STAT_WAVE(k, v)                                      \
    do {                                                     \
        const unsigned long long v_ = (v);                   \
        if ((threadIdx.x & 63) == 0) STAT_ADD(k, v_);        \
    } while (0)
__device__ __forceinline__ unsigned stat_wave_max(unsigned v)
{
    for (int o = 32; o; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o));
    return v;
}
#else
#define STAT_ADD(k, v) ((void)0)
#define STAT_WAVE(k, v) ((void)0)
#endif

// ------------------------------------------------------------------ helpers

__device__ __forceinline__ int f2ord(float f)
{
    int b = __float_as_int(f);
    return b >= 0 ? b : b ^ 0x7fffffff;
}
__host__ __device__ __forceinline__ float ord2f(int k)
{
    int b = k >= 0 ? k : k ^ 0x7fffffff;
#ifdef __HIP_DEVICE_COMPILE__
    return __int_as_float(b);
#else
    float f;
    memcpy(&f, &b, 4);
    return f;
#endif
}
__device__ __forceinline__ bool finite3(float x, float y, float z)
{
    return isfinite(x) && isfinite(y) && isfinite(z);
}

// The squared distance every implementation shares (oracle: orc_sqdist).
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sqdist(float ax, float ay, float az, float bx, float by, float bz)
{
    // x and y are subtracted as one packed operation (v_pk_add_f32: the same IEEE subtraction, half the issue slots)
    const float2v a = {ax, ay}, b = {bx, by};
    const float2v d = a - b;
    const float dz = az - bz;
    return __fmaf_rn(dz, dz, __fmaf_rn(d.y, d.y, __fmul_rn(d.x, d.x)));
}

// XCD-aware block remap (cdna_hip_programming.md T1, bijective form): logical
// blocks that are adjacent in memory land on the same XCD, i.e. the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk)
{
    int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

__device__ __forceinline__ int cell_coord(float p, float o, float inv_h, int n)
{
    int c = (int)floorf((p - o) * inv_h);
    return min(max(c, 0), n - 1);
}

// Cells of one level, in the numbering cell_of uses (a tiled level rounds x and y up to whole 8x8 tiles).
__host__ __device__ __forceinline__ long long grid_cells(const GridDesc &G)
{
    return G.tile ? (long long)G.nz * ((G.ny + 7) / 8) * ((G.nx + 7) / 8) * 64 : (long long)G.nx * G.ny * G.nz;
}

__device__ __forceinline__ long long cell_of(const GridDesc &G, float x, float y, float z)
{
    int cx = cell_coord(x, G.ox, G.inv_h, G.nx);
    int cy = cell_coord(y, G.oy, G.inv_h, G.ny);
    int cz = cell_coord(z, G.oz, G.inv_h, G.nz);
    if (G.tile) {
        // 8x8 tiles in xy keep 256 consecutive points a compact patch, so the
        // fused kernel's LDS box stays small
        const int ntx = (G.nx + 7) >> 3, nty = (G.ny + 7) >> 3;
        return G.cell_base + ((((long long)cz * nty + (cy >> 3)) * ntx + (cx >> 3)) << 6) + ((cy & 7) << 3) + (cx & 7);
    }
    return G.cell_base + ((long long)cz * G.ny + cy) * G.nx + cx;
}

template <int K>
struct Best {
    static constexpr bool COOP = false;  // k > 1: per-lane scanning only
    static constexpr bool WARM_START = true;
    static constexpr bool BALL = false;
    float d[K];
    int i[K];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            d[k] = INFINITY;
            i[k] = 0x7fffffff;
        }
    }
    // only candidates closer than sqrt(r2) count: a query with fewer than K of them stops climbing at the level
    // that covers the radius instead of the coarsest one (the caller discards results at or beyond r2 anyway)
    __device__ __forceinline__ void init_radius(float r2)
    {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            d[k] = r2;
            i[k] = 0x7fffffff;
        }
    }
    __device__ __forceinline__ float worst() const { return d[K - 1]; }
    __device__ __forceinline__ bool seeded() const { return i[K - 1] != 0x7fffffff; }
    __device__ __forceinline__ void consider(float d2, const float4 &c, unsigned)
    {
        const int idx = __float_as_int(c.w);
        if (d2 < d[K - 1] || (d2 == d[K - 1] && idx < i[K - 1])) {
            if (K > 1) {
                // a coarser level re-visits the points of the finer ones
                bool dup = false;
#pragma unroll
                for (int s = 0; s < K - 1; ++s) dup |= i[s] == idx;
                if (dup) return;
            }
            d[K - 1] = d2;
            i[K - 1] = idx;
#pragma unroll
            for (int s = K - 1; s > 0; --s) {
                bool sw = d[s] < d[s - 1] || (d[s] == d[s - 1] && i[s] < i[s - 1]);
                if (sw) {
                    float td = d[s]; d[s] = d[s - 1]; d[s - 1] = td;
                    int ti = i[s]; i[s] = i[s - 1]; i[s - 1] = ti;
                }
            }
        }
    }
};

// The k best as K ascending 64-bit keys (d2 bits << 32 | index: d2 >= 0, so a key orders like (d2, index)), updated by
// a branch-free insertion network: c_k = x < key_k for every k, then key_k <- c_k ? (c_{k-1} ? key_{k-1} : x) : key_k --
// K compares and ~4 K selects, no masked block, no bubble loop.  With 64 queries in a wave some lane accepts nearly
// every candidate, so the wave pays the insertion per candidate STEP; Best<K>'s guarded bubble insert is ~45 vector +
// ~15 scalar instructions there (self_nn_kernel, K = 6: 6 020 + 3 560 per wave by the SQ counters), this one 34 + 10.
// Same acceptance rule, order and duplicate rule as Best<K> (a coarser level re-visits the points of the finer ones:
// a key that is already in the list is not inserted again), hence the same results.
template <int K>
struct BestKeys {
    static constexpr bool COOP = false;
    static constexpr bool WARM_START = true;
    static constexpr bool BALL = false;
    unsigned long long key[K];
    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int k = 0; k < K; ++k) key[k] = ((unsigned long long)0x7f800000u << 32) | 0x7fffffffu;  // (+inf, INT_MAX)
    }
    __device__ __forceinline__ void init_radius(float r2)
    {
#pragma unroll
        for (int k = 0; k < K; ++k) key[k] = ((unsigned long long)__float_as_uint(r2) << 32) | 0x7fffffffu;
    }
    __device__ __forceinline__ float dist2(int k) const { return __uint_as_float((unsigned)(key[k] >> 32)); }
    __device__ __forceinline__ int index(int k) const { return (int)(unsigned)key[k]; }
    __device__ __forceinline__ float worst() const { return dist2(K - 1); }
    __device__ __forceinline__ bool seeded() const { return index(K - 1) != 0x7fffffff; }
    __device__ __forceinline__ void consider(float d2, const float4 &c, unsigned)
    {
        const unsigned long long x = ((unsigned long long)__float_as_uint(d2) << 32) | __float_as_uint(c.w);
        bool lt[K];
        bool dup = false;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            lt[k] = x < key[k];
            if (K > 1) dup |= x == key[k];
        }
#pragma unroll
        for (int k = K - 1; k > 0; --k) {
            const unsigned long long in = lt[k - 1] ? key[k - 1] : x;
            key[k] = (lt[k] && !dup) ? in : key[k];
        }
        key[0] = (lt[0] && !dup) ? x : key[0];
    }
};

// 1-NN record of the ICP kernel.  (d2, index) live in one 64-bit key -- d2 >= 0, so
// its float bits order like the value and the key orders exactly like (d2, index):
// accepting a candidate is one v_cmp_lt_u64 and three v_cndmask.  `pos` is the
// winner's position in `sorted` (WARM = the warm-start candidate is still best).
struct BestQ {
    static constexpr bool COOP = true;  // long runs are scanned by the whole wave
    static constexpr bool WARM_START = true;  // worst() of a fresh record is a real candidate's distance
    static constexpr bool BALL = true;        // worst() is the squared distance of the 1-NN so far: a ball radius
    static constexpr unsigned WARM = 0xffffffffu;
    static constexpr unsigned LIST = 0xfffffff0u;  // LIST + j: j-th entry of q0's neighbour list
    unsigned long long key;
    unsigned pos;
    __device__ __forceinline__ void init()
    {
        key = ((unsigned long long)0x7f800000u << 32) | 0x7fffffffu;  // (+inf, INT_MAX)
        pos = WARM;
    }
    __device__ __forceinline__ float dist2() const { return __uint_as_float((unsigned)(key >> 32)); }
    __device__ __forceinline__ int index() const { return (int)(unsigned)key; }
    __device__ __forceinline__ float worst() const { return dist2(); }
    __device__ __forceinline__ bool seeded() const { return index() != 0x7fffffff; }
    __device__ __forceinline__ void consider(float d2, const float4 &c, unsigned p)
    {
        const unsigned long long k2 = ((unsigned long long)__float_as_uint(d2) << 32) | __float_as_uint(c.w);
        if (k2 < key) {
            key = k2;
            pos = p;
        }
    }
};

// Nearest point whose index lies in [a0,a1) or [b0,b1), other than `closest`, within an initial
// squared radius: laserOdometry's walks over the adjacent rings of the last sweep (LO:613-677,
// 769-844) as one filtered search.  Ties go to the candidate the sequential walk meets first:
// indices above `closest` ascending, then indices below it descending.
struct BestRing {
    static constexpr bool COOP = false;
    static constexpr bool WARM_START = false;  // the bound is a radius, not a candidate
    static constexpr bool BALL = false;
    float d;
    int i;
    unsigned ord;
    int a0, a1, b0, b1, closest;
    int base;  // added to the index a candidate carries (a per-ring grid numbers its points from 0)
    __device__ __forceinline__ void init(float bound2, int closest_, int a0_, int a1_, int b0_, int b1_)
    {
        base = 0;
        d = bound2;
        i = -1;
        ord = 0u;
        closest = closest_;
        a0 = a0_; a1 = a1_; b0 = b0_; b1 = b1_;
    }
    __device__ __forceinline__ float worst() const { return d; }
    __device__ __forceinline__ bool seeded() const { return i >= 0; }  // holds a candidate, not just the radius
    __device__ __forceinline__ void consider(float d2, const float4 &c, unsigned)
    {
        const int idx = __float_as_int(c.w) + base;
        const bool in = ((idx >= a0 && idx < a1) || (idx >= b0 && idx < b1)) && idx != closest;
        if (!in) return;
        const unsigned o = idx > closest ? (unsigned)(idx - closest) : 0x40000000u + (unsigned)(closest - idx);
        if (d2 < d || (d2 == d && i >= 0 && o < ord)) {
            d = d2;
            i = idx;
            ord = o;
        }
    }
};

// Four consecutive cell_start entries fetched as ONE 16-byte load (dword aligned).
struct __attribute__((packed, aligned(4))) CellQuad {
    unsigned a, b, c, d;
};

// Candidates [s, e) of `sorted`, scanned by the lane itself with four 16-byte
// gathers in flight.
template <class BT, bool SAME_CLOUD = false>
__device__ __forceinline__ void scan_short(BT &B, bool act, const float4 *__restrict__ sorted, unsigned s, unsigned e,
                                           float px, float py, float pz)
{
#ifdef GPSCAL_STATS
    {
        const unsigned len = act ? e - s : 0u, trips = (len + 3) / 4;
        const unsigned wt = stat_wave_max(trips);
        STAT_WAVE(7, wt);             // wave-level groups of four
        STAT_ADD(8, len);             // lane-level candidates
        STAT_ADD(16, trips);          // lane-level groups of four
    }
#endif
    if (!act) return;
#ifdef GPSCAL_SCAN_DIAG  // ablation (wrong results): 1 = no candidate is read, 2 = only the first of a run
    if (GPSCAL_SCAN_DIAG == 1) return;
    if (GPSCAL_SCAN_DIAG == 2) e = min(e, s + 1u);
#endif
    // Addresses.  SAME_CLOUD (the caller's lanes all search one level of one cloud, < 2^25 points): one SIGNED 32-bit
    // byte offset per group, relative to the run of the wave's first active lane (other lanes' runs lie on either side
    // of it, at most 2^25 points away whatever the batch's size); the four loads of a group share one 64-bit address
    // (+ immediates) instead of four.  Otherwise (lanes in different clouds of a set) positions are addressed in full.
    const unsigned p0 = SAME_CLOUD ? __builtin_amdgcn_readfirstlane(s) : 0u;
    const char *base = reinterpret_cast<const char *>(sorted + p0);
    auto ld = [&](unsigned jj, unsigned k) -> float4 {
        if constexpr (SAME_CLOUD)
            return *reinterpret_cast<const float4 *>(base + (ptrdiff_t)((int)(jj - p0) * 16) + 16 * (int)k);
        else
            return sorted[jj + k];
    };
    unsigned j = s;
    // full groups of four: no clamps, no per-candidate guards
    for (; j + 4 <= e; j += 4) {
        const unsigned off = j;
        const float4 c0 = ld(off, 0), c1 = ld(off, 1), c2 = ld(off, 2), c3 = ld(off, 3);
        B.consider(sqdist(px, py, pz, c0.x, c0.y, c0.z), c0, j);
        B.consider(sqdist(px, py, pz, c1.x, c1.y, c1.z), c1, j + 1);
        B.consider(sqdist(px, py, pz, c2.x, c2.y, c2.z), c2, j + 2);
        B.consider(sqdist(px, py, pz, c3.x, c3.y, c3.z), c3, j + 3);
    }
    if (j < e) {  // the last one to three, still fetched together
        const unsigned last = e - 1;
        // (SAME_CLOUD: the two entries behind a run's end are read unclamped -- build_grids pads the array -- and not considered)
        const float4 c0 = ld(j, 0);
        const float4 c1 = SAME_CLOUD ? ld(j, 1) : ld(min(j + 1, last), 0);
        const float4 c2 = SAME_CLOUD ? ld(j, 2) : ld(min(j + 2, last), 0);
        B.consider(sqdist(px, py, pz, c0.x, c0.y, c0.z), c0, j);
        if (j + 1 < e) B.consider(sqdist(px, py, pz, c1.x, c1.y, c1.z), c1, j + 1);
        if (j + 2 < e) B.consider(sqdist(px, py, pz, c2.x, c2.y, c2.z), c2, j + 2);
    }
}

constexpr unsigned COOP_MIN = 12;   // runs longer than this are worth the whole wave ...
constexpr int COOP_MAX_OWNERS = 6;  // ... but only while few lanes have one (owners * L/64 < L/4)

__device__ __forceinline__ float readlane_f(float v, int l)
{
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// Must be reached by all 64 lanes of the wave (wave-uniform control flow).  Short
// runs are scanned per lane.  A long run (a dense or coarse cell) is scanned by the
// whole wave for its owner: 64 consecutive candidates per step, one coalesced 1 KiB
// load instead of 64 dependent gathers by one lane, then a DPP arg-min.
template <class BT, bool SAME_CLOUD = false>
__device__ __forceinline__ void scan_runs(BT &B, bool act, const float4 *__restrict__ sorted, unsigned s, unsigned e,
                                          float px, float py, float pz)
{
    if constexpr (!BT::COOP) {
        scan_short<BT, SAME_CLOUD>(B, act, sorted, s, e, px, py, pz);
    } else {
        bool lng = act && (e - s) > COOP_MIN;
        unsigned long long m = __ballot(lng);
        if (__popcll(m) > COOP_MAX_OWNERS) {  // everybody has work: lanes scan in parallel
            m = 0ull;
            lng = false;
        }
        scan_short<BT, SAME_CLOUD>(B, act && !lng, sorted, s, e, px, py, pz);
        const int lane = threadIdx.x & 63;
        while (m) {  // wave-uniform
            const int owner = __builtin_ctzll(m);
            m &= m - 1;
            const float qx = readlane_f(px, owner), qy = readlane_f(py, owner), qz = readlane_f(pz, owner);
            const unsigned ss = __builtin_amdgcn_readlane(s, owner), ee = __builtin_amdgcn_readlane(e, owner);
            STAT_WAVE(9, 1);
            STAT_WAVE(10, ee - ss);
            float bd = INFINITY;
            int bi = 0x7fffffff;
            unsigned bp = 0;
            float bx = 0.f, by = 0.f, bz = 0.f;
            for (unsigned j0 = ss; j0 < ee; j0 += 64) {
                const unsigned j = j0 + lane;
                if (j < ee) {
                    const float4 c = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sorted + ss) +
                                                                       (size_t)((j - ss) << 4));
                    const float d2 = sqdist(qx, qy, qz, c.x, c.y, c.z);
                    const int ci = __float_as_int(c.w);
                    if (d2 < bd || (d2 == bd && ci < bi)) {
                        bd = d2; bi = ci; bp = j; bx = c.x; by = c.y; bz = c.z;
                    }
                }
            }
            // (d2 >= 0, index >= 0): the 64-bit key orders exactly like (d2, index)
            const unsigned long long key = ((unsigned long long)__float_as_uint(bd) << 32) | (unsigned)bi;
            const unsigned long long kmin = wave_min_u64(key);
            const int win = __builtin_ctzll(__ballot(key == kmin));
            const float wd = readlane_f(bd, win);
            const float4 wc = make_float4(readlane_f(bx, win), readlane_f(by, win), readlane_f(bz, win),
                                          __int_as_float(__builtin_amdgcn_readlane(bi, win)));
            const unsigned wpos = __builtin_amdgcn_readlane(bp, win);
            if (lane == owner && wd < INFINITY) B.consider(wd, wc, wpos);
        }
    }
}

// Per-query geometry at one grid level.
struct CellGeo {
    int cx, cy, cz;
    float fx0, fx1, fy0, fy1, fz0, fz1;  // distances to the faces of the own cell
    __device__ __forceinline__ void set(const GridDesc &G, float px, float py, float pz)
    {
        cx = cell_coord(px, G.ox, G.inv_h, G.nx);
        cy = cell_coord(py, G.oy, G.inv_h, G.ny);
        cz = cell_coord(pz, G.oz, G.inv_h, G.nz);
        const float h = G.h;
        fx0 = px - (G.ox + cx * h); fx1 = (G.ox + (cx + 1) * h) - px;
        fy0 = py - (G.oy + cy * h); fy1 = (G.oy + (cy + 1) * h) - py;
        fz0 = pz - (G.oz + cz * h); fz1 = (G.oz + (cz + 1) * h) - pz;
    }
    // squared radius inside which this level's 3x3x3 block is exhaustive
    __device__ __forceinline__ float settled_r2(const GridDesc &G) const
    {
        const float h = G.h;
        float g = INFINITY;
        if (cx - 1 > 0) g = fminf(g, fx0 + h);
        if (cx + 1 < G.nx - 1) g = fminf(g, fx1 + h);
        if (cy - 1 > 0) g = fminf(g, fy0 + h);
        if (cy + 1 < G.ny - 1) g = fminf(g, fy1 + h);
        if (cz - 1 > 0) g = fminf(g, fz0 + h);
        if (cz + 1 < G.nz - 1) g = fminf(g, fz1 + h);
        g = fmaxf(g - G.margin, 0.f);
        return g == INFINITY ? INFINITY : g * g * 0.99999f;
    }
};

// One level of the fine -> coarse search: the 3x3x3 block of cells around the query.  Rows
// (fixed y,z; x-1..x+1 contiguous in memory) and then single cells are skipped when their
// nearest face is already farther than the k-th best.  Wave-uniform.
template <class BT, bool SAME_CLOUD = false>
__device__ __forceinline__ void block3_level(const GridDesc &G, const CellGeo &C, const float4 *__restrict__ sorted,
                                             const unsigned *__restrict__ cell_start, bool act, float px, float py,
                                             float pz, BT &B)
{
    const float mg = G.margin;
    const float bxl = fmaxf(C.fx0 - mg, 0.f), bxr = fmaxf(C.fx1 - mg, 0.f);
    const bool has_l = C.cx > 0, has_r = C.cx + 1 < G.nx;
    // face distances of the 3 x 3 rows: index 0 = own, 1 = lower, 2 = upper neighbour
    const float by2[3] = {0.f, fmaxf(C.fy0 - mg, 0.f) * fmaxf(C.fy0 - mg, 0.f),
                          fmaxf(C.fy1 - mg, 0.f) * fmaxf(C.fy1 - mg, 0.f)};
    const float bz2[3] = {0.f, fmaxf(C.fz0 - mg, 0.f) * fmaxf(C.fz0 - mg, 0.f),
                          fmaxf(C.fz1 - mg, 0.f) * fmaxf(C.fz1 - mg, 0.f)};
    const bool yok[3] = {true, C.cy > 0, C.cy + 1 < G.ny};
    const bool zok[3] = {true, C.cz > 0, C.cz + 1 < G.nz};
    // rows this lane (lanemask) / some lane of the wave (wavemask) still has to look
    // at; bit r = 3*kz + ky
    unsigned wavemask = 0, lanemask = 0;
    {
        const float w0 = B.worst();
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const bool p = act && yok[r % 3] && zok[r / 3] && (by2[r % 3] + bz2[r / 3]) * 0.99999f <= w0;
            if (p) lanemask |= 1u << r;
            if (__ballot(p) != 0ull) wavemask |= 1u << r;
        }
    }
    // cell_start[row-1 .. row+2] of row r: left | own | right cell boundaries.  The
    // next row's quad is fetched while the current row's candidates are scanned.
    auto load_quad = [&](int r) -> CellQuad {
        CellQuad q = {0u, 0u, 0u, 0u};
        if ((lanemask >> r) & 1u) {
            const int kz = r / 3, ky = r - 3 * kz;
            const int zz = C.cz + (kz == 0 ? 0 : (kz == 1 ? -1 : 1)), yy = C.cy + (ky == 0 ? 0 : (ky == 1 ? -1 : 1));
            // a level has < 2^25 cells: the row's offset inside the level in 32 bits, behind a scalar base
            const unsigned rel = (unsigned)((zz * G.ny + yy) * G.nx + C.cx) * 4u;
            q = *reinterpret_cast<const CellQuad *>(reinterpret_cast<const char *>(cell_start + G.cell_base - 1) + (size_t)rel);
        }
        return q;
    };
    CellQuad nxt = {0u, 0u, 0u, 0u};
    if (wavemask) nxt = load_quad(__builtin_ctz(wavemask));
    while (wavemask) {  // wave-uniform
        const int r = __builtin_ctz(wavemask);
        wavemask &= wavemask - 1;
        const CellQuad q = nxt;
        if (wavemask) nxt = load_quad(__builtin_ctz(wavemask));
        const int kz = r / 3, ky = r - 3 * kz;
        const float rb2 = (ky == 0 ? 0.f : (ky == 1 ? by2[1] : by2[2])) + (kz == 0 ? 0.f : (kz == 1 ? bz2[1] : bz2[2]));
        const bool pass = ((lanemask >> r) & 1u) && rb2 * 0.99999f <= B.worst();
        if (__ballot(pass) == 0ull) continue;  // an earlier row tightened the bound
        STAT_WAVE(5, 1);
        STAT_WAVE(6, __popcll(__ballot(pass)));
        const unsigned c0 = has_l ? q.a : q.b, c1 = q.b, c2 = q.c, c3 = has_r ? q.d : q.c;
        // left | own | right cells are one contiguous run: cells whose face is already
        // within reach are scanned together with the own cell (one pass of loads instead
        // of three dependent ones); a neighbour ruled out now stays ruled out, the bound
        // only tightens.
        const bool pl0 = pass && c0 < c1 && (rb2 + bxl * bxl) * 0.99999f <= B.worst();
        const bool pr0 = pass && c2 < c3 && (rb2 + bxr * bxr) * 0.99999f <= B.worst();
        scan_runs<BT, SAME_CLOUD>(B, pass, sorted, pl0 ? c0 : c1, pr0 ? c3 : c2, px, py, pz);
    }
}

// The same level search for kernels that run at under one wave per SIMD (the LOAM searches), where a search is a
// CHAIN of dependent round trips and the chain's length is what counts: a lane needs ~3 of the 9 rows and which
// ones differs from lane to lane, so visiting the rows one after the other costs the wave the sum over rows of the
// longest run in each (laserOdometry: 119 groups of four per wave against 29 per lane).  Here the own row is scanned
// first (it tightens the bound), then the cell boundaries of the other rows are fetched FLAT_BATCH rows at a time,
// every lane lists the candidate runs of its rows in a wave-private LDS slab (8 x 64 uint2 = 4 KiB per wave) and
// walks its own list in groups of four: the wave pays the longest LIST.  Candidates may be evaluated twice (a
// clamped group): records order by (d2, index), which makes that idempotent.  (Measured on icp_step_kernel, which is
// bound by the rate of gathers and not by their latency: no gain, commit 41733d4.)
constexpr int FLAT_BATCH = 4;
template <class BT>
__device__ __forceinline__ void block3_level_flat(const GridDesc &G, const CellGeo &C,
                                                  const float4 *__restrict__ sorted,
                                                  const unsigned *__restrict__ cell_start, bool act, float px,
                                                  float py, float pz, BT &B, uint2 *__restrict__ slab,
                                                  unsigned rows = 0x1ffu)
{
    // rows: bit r = visit row r (0 = the own row): several waves may share one query's block, each with its rows
    const float mg = G.margin;
    const float bxl = fmaxf(C.fx0 - mg, 0.f), bxr = fmaxf(C.fx1 - mg, 0.f);
    const float bxl2 = bxl * bxl, bxr2 = bxr * bxr;
    const bool has_l = C.cx > 0, has_r = C.cx + 1 < G.nx;
    const float by2[3] = {0.f, fmaxf(C.fy0 - mg, 0.f) * fmaxf(C.fy0 - mg, 0.f),
                          fmaxf(C.fy1 - mg, 0.f) * fmaxf(C.fy1 - mg, 0.f)};
    const float bz2[3] = {0.f, fmaxf(C.fz0 - mg, 0.f) * fmaxf(C.fz0 - mg, 0.f),
                          fmaxf(C.fz1 - mg, 0.f) * fmaxf(C.fz1 - mg, 0.f)};
    const bool yok[3] = {true, C.cy > 0, C.cy + 1 < G.ny};
    const bool zok[3] = {true, C.cz > 0, C.cz + 1 < G.nz};
    const int lane = threadIdx.x & 63;
    const long long own = G.cell_base + ((long long)C.cz * G.ny + C.cy) * G.nx + C.cx;
    const long long dyo = G.nx, dzo = (long long)G.ny * G.nx;
    if (rows & 1u) {  // the own row
        CellQuad q = {0u, 0u, 0u, 0u};
        if (act) q = *reinterpret_cast<const CellQuad *>(cell_start + own - 1);
        const unsigned c0 = has_l ? q.a : q.b, c1 = q.b, c2 = q.c, c3 = has_r ? q.d : q.c;
        const bool pl0 = act && c0 < c1 && bxl2 * 0.99999f <= B.worst();
        const bool pr0 = act && c2 < c3 && bxr2 * 0.99999f <= B.worst();
        STAT_WAVE(5, 1);
        scan_short(B, act, sorted, pl0 ? c0 : c1, pr0 ? c3 : c2, px, py, pz);
    }
    unsigned cnt = 0;
#pragma unroll
    for (int half = 0; half < 8 / FLAT_BATCH; ++half) {
        CellQuad q[FLAT_BATCH];
        bool p[FLAT_BATCH];
#pragma unroll
        for (int t = 0; t < FLAT_BATCH; ++t) {
            const int r = 1 + half * FLAT_BATCH + t, kz = r / 3, ky = r - 3 * kz;
            p[t] = act && ((rows >> r) & 1u) && yok[ky] && zok[kz] && (by2[ky] + bz2[kz]) * 0.99999f <= B.worst();
            q[t] = CellQuad{0u, 0u, 0u, 0u};
            if (p[t]) {
                const long long row = own + (ky == 0 ? 0 : (ky == 1 ? -dyo : dyo)) + (kz == 0 ? 0 : (kz == 1 ? -dzo : dzo));
                q[t] = *reinterpret_cast<const CellQuad *>(cell_start + row - 1);
            }
        }
#pragma unroll
        for (int t = 0; t < FLAT_BATCH; ++t) {
            const int r = 1 + half * FLAT_BATCH + t, kz = r / 3, ky = r - 3 * kz;
            const float rb2 = by2[ky] + bz2[kz];
            const unsigned c0 = has_l ? q[t].a : q[t].b, c1 = q[t].b, c2 = q[t].c, c3 = has_r ? q[t].d : q[t].c;
            const bool pl0 = p[t] && c0 < c1 && (rb2 + bxl2) * 0.99999f <= B.worst();
            const bool pr0 = p[t] && c2 < c3 && (rb2 + bxr2) * 0.99999f <= B.worst();
            const unsigned s = pl0 ? c0 : c1, e = pr0 ? c3 : c2;
            if (p[t] && e > s) {
                slab[cnt * 64 + lane] = make_uint2(s, e);
                ++cnt;
            }
        }
    }
    // every lane walks its own list; the next entry is fetched while the current run is scanned
    unsigned j = 0u, e = 0u, k = 2u;
    uint2 nx = make_uint2(0u, 0u);
    if (cnt > 0u) {
        const uint2 r0 = slab[lane];
        j = r0.x;
        e = r0.y;
    }
    if (cnt > 1u) nx = slab[64 + lane];
    while (__ballot(j < e) != 0ull) {
        STAT_WAVE(18, 1);
        if (j < e) {
            const unsigned last = e - 1u;
            const unsigned j1 = min(j + 1u, last), j2 = min(j + 2u, last), j3 = min(j + 3u, last);
            const float4 c0 = sorted[j], c1 = sorted[j1], c2 = sorted[j2], c3 = sorted[j3];
            B.consider(sqdist(px, py, pz, c0.x, c0.y, c0.z), c0, j);
            B.consider(sqdist(px, py, pz, c1.x, c1.y, c1.z), c1, j1);
            B.consider(sqdist(px, py, pz, c2.x, c2.y, c2.z), c2, j2);
            B.consider(sqdist(px, py, pz, c3.x, c3.y, c3.z), c3, j3);
            j += 4u;
            if (j >= e) {
                j = nx.x;
                e = nx.y;
                nx = make_uint2(0u, 0u);
                if (k < cnt) nx = slab[k * 64 + lane];
                ++k;
            }
        }
    }
}

// One level of the ball search: every cell of a (2R+1)^3 block that the ball around the query
// with the current k-th best distance as radius still reaches.  Slabs (fixed z) and rows (fixed
// y,z) are visited nearest first and dropped when their face is out of reach; of a row only the
// cells under the ball's x-extent are read -- they are one contiguous run of candidates between two
// cell_start entries.  The caller guarantees R*h covers the radius.  Wave-uniform.
template <class BT>
__device__ __forceinline__ void ball_level(const GridDesc &G, const CellGeo &C, const float4 *__restrict__ sorted,
                                           const unsigned *__restrict__ cell_start, bool act, float px, float py,
                                           float pz, BT &B, int R)
{
    const float h = G.h, mg = G.margin;
    for (int iz = 0; iz <= 2 * R; ++iz) {  // 0, -1, +1, -2, +2, ...
        const int dz = (iz & 1) ? -((iz + 1) >> 1) : (iz >> 1);
        const int zz = C.cz + dz;
        float bz = 0.f;
        if (dz != 0) bz = fmaxf((dz < 0 ? C.fz0 : C.fz1) + (float)(abs(dz) - 1) * h - mg, 0.f);
        const float bz2 = bz * bz;
        const bool inz = act && zz >= 0 && zz < G.nz;
        if (__ballot(inz && bz2 * 0.99999f <= B.worst()) == 0ull) continue;
        for (int iy = 0; iy <= 2 * R; ++iy) {
            const int dy = (iy & 1) ? -((iy + 1) >> 1) : (iy >> 1);
            const int yy = C.cy + dy;
            float by = 0.f;
            if (dy != 0) by = fmaxf((dy < 0 ? C.fy0 : C.fy1) + (float)(abs(dy) - 1) * h - mg, 0.f);
            const float rb2 = (by * by + bz2) * 0.99999f;
            const float w = B.worst();
            const bool pass = inz && yy >= 0 && yy < G.ny && rb2 <= w;
            if (__ballot(pass) == 0ull) continue;
            unsigned s = 0u, e = 0u;
            if (pass) {
                // floor((x - o) / h) is monotone in x, and it is what sorted the points into cells:
                // the cells of px -+ hw bracket every point within hw of px
                const float hw = sqrtf(w - rb2) * 1.00001f + mg;
                const int x0 = cell_coord(px - hw, G.ox, G.inv_h, G.nx), x1 = cell_coord(px + hw, G.ox, G.inv_h, G.nx);
                const char *lv = reinterpret_cast<const char *>(cell_start + G.cell_base);  // scalar base, 32-bit offsets
                const unsigned row = (unsigned)((zz * G.ny + yy) * G.nx);
                s = *reinterpret_cast<const unsigned *>(lv + (size_t)((row + (unsigned)x0) * 4u));
                e = *reinterpret_cast<const unsigned *>(lv + (size_t)((row + (unsigned)x1 + 1u) * 4u));
            }
            scan_runs<BT, true>(B, pass, sorted, s, e, px, py, pz);
        }
    }
}

// Exact k-NN of (px,py,pz) in pair P.  MUST be called by all 64 lanes of a wave (act = false
// for lanes without a query): control flow is wave-uniform so that long runs can be scanned
// cooperatively.
//
// Fine -> coarse search (ball_r = 0): a level's 3x3x3 block of cells settles the query when the
// k-th best distance is within the distance to the nearest face of the block that still has
// cells behind it.  The coarsest level has <= 2 cells per axis, so it always settles.  A good
// starting candidate (the previous iteration's neighbour) removes most of the memory traffic.
//
// Ball search (ball_r = R > 0, record types with BT::BALL): a query that holds a candidate at
// distance r searches only the finest level whose (2R+1)^3 block covers the ball of radius r --
// cells 2.5x .. R x 2.5x smaller than the level the 3x3x3 rule needs, i.e. fewer candidates under
// the ball when the query is far from the surface the points sample.  Queries without a
// candidate first climb the 3x3x3 blocks until they hold one.
template <class BT, bool ALLOW_BALL = false, bool FLAT = false>
__device__ __forceinline__ void knn_query(const PairDesc &P, const float4 *__restrict__ sorted,
                                          const unsigned *__restrict__ cell_start, bool act, float px, float py,
                                          float pz, BT &B, int ball_r = 0, uint2 *__restrict__ slab = nullptr)
{
    if (!act) px = py = pz = 0.f;
    const bool ballmode = ALLOW_BALL && BT::BALL && ball_r > 0;
    // Pass 0, fine -> coarse.  A lane that already holds a candidate skips the levels that cannot
    // settle it: level l settles every query whose best is within h_l (the 3x3x3 block reaches at
    // least one cell beyond the query's own), so the first such level is searched alone.  Without a
    // candidate (first iteration) the search starts at level 0.  In ball mode only lanes without a
    // candidate take this pass, and they leave it as soon as they hold one.
    // Pass 1 (ball mode): the one level chosen from the candidate's distance.
    int lvl = 0;  // pass 0: first level to search; pass 1: the level to search
    bool todo = act;
    {
        const float w0 = B.worst();
        // (a record that starts with a bare radius -- no candidate yet -- still climbs from level 0: the fine
        // levels usually settle it with far fewer candidates than the level of the radius holds)
        if (BT::WARM_START && w0 < INFINITY && B.seeded()) {
            if (ballmode) todo = false;
            lvl = P.nlevels - 1;
            for (int l = P.nlevels - 2; l >= 0; --l) {
                const float g = P.lv[l].h * 0.999f - P.lv[l].margin;
                if (g > 0.f && w0 <= g * g) lvl = l;
            }
        }
    }
    bool ball = ballmode && act && !todo;
    for (int pass = 0; pass < (ballmode ? 2 : 1); ++pass) {
        if (pass == 1) {
            if (__ballot(ball) == 0ull) break;
            const float w0 = B.worst();
            lvl = P.nlevels - 1;
            for (int l = P.nlevels - 2; l >= 0; --l) {
                const float g = (float)ball_r * P.lv[l].h * 0.999f - P.lv[l].margin;
                if (g > 0.f && w0 <= g * g) lvl = l;
            }
        }
        for (int l = 0; l < P.nlevels; ++l) {
            const bool a = pass == 0 ? (todo && l >= lvl) : (ball && l == lvl);
            if (pass == 0 && __ballot(todo) == 0ull) break;
            if (__ballot(a) == 0ull) continue;
            const GridDesc &G = P.lv[l];
            CellGeo C;
            C.set(G, px, py, pz);
            // pass 1: a wave whose balls are all inside the 3x3x3 block takes the block search
            // (prefetched rows); the top level has <= 2 cells per axis, so the block is the whole level
            const float g1 = G.h * 0.999f - G.margin;
            const bool far = pass == 1 && l != P.nlevels - 1 && !(g1 > 0.f && B.worst() <= g1 * g1);
            STAT_WAVE(3, 1);
            STAT_WAVE(4, __popcll(__ballot(a)));
            STAT_WAVE(11 + min(l, 4), __popcll(__ballot(a)));
            if (!ALLOW_BALL || __ballot(a && far) == 0ull) {
                if ((FLAT || !BT::COOP) && slab)  // latency-bound callers hand in a wave-private slab (block3_level_flat)
                    block3_level_flat(G, C, sorted, cell_start, a, px, py, pz, B, slab);
                else
                    block3_level<BT, true>(G, C, sorted, cell_start, a, px, py, pz, B);
            } else {
                ball_level(G, C, sorted, cell_start, a, px, py, pz, B, ball_r);
            }
            if (pass == 0 && a) {
                if (B.worst() <= C.settled_r2(G))
                    todo = false;
                else if (ballmode && B.worst() < INFINITY) {
                    todo = false;
                    ball = true;
                }
            }
        }
    }
}

// knn_query with one index PER LANE (`Pl`: the lane's pair descriptor; every lane passes a valid pointer, lanes
// without a query pass act = false): the lanes of a wave search different clouds in one pass over the levels
// instead of one pass per distinct cloud.  laserOdometry's adjacent-ring searches are the user: the 64 features
// of a wave want 4-8 different rings, and a pass per ring made lo_search_kernel a chain of ~20 dependent
// searches.  Fine -> coarse 3x3x3 search only (no ball search); the grid geometry lives in
// vector registers instead of scalar ones, the candidates a lane sees are exactly those of knn_query.
template <class BT>
__device__ __forceinline__ void knn_query_lanes(const PairDesc *__restrict__ Pl, const float4 *__restrict__ sorted,
                                                const unsigned *__restrict__ cell_start, bool act, float px, float py,
                                                float pz, BT &B, uint2 *__restrict__ slab = nullptr)
{
    if (!act) px = py = pz = 0.f;
    const int nl = Pl->nlevels;
    bool todo = act;
    // A seed candidate's distance names the level to search: level l's 3x3x3 block holds every point within h_l
    // (less the margin) of the query, so the first level with h_l >= that distance settles the query and the
    // finer ones are skipped.
    int lvl = 0;
    {
        const float w0 = B.worst();
        if (w0 < INFINITY && B.seeded()) {  // a bare radius (laserOdometry's 5 m) would start at cells the size of it
            lvl = nl - 1;
            for (int l = MAX_LEVELS - 2; l >= 0; --l) {
                if (l >= nl - 1) continue;
                const float g = Pl->lv[l].h * 0.999f - Pl->lv[l].margin;
                if (g > 0.f && w0 <= g * g) lvl = l;
            }
        }
    }
    for (int l = 0; l < MAX_LEVELS; ++l) {
        if (__ballot(todo) == 0ull) break;
        const bool a = todo && l < nl && l >= lvl;
        if (__ballot(a) == 0ull) continue;
        const GridDesc G = Pl->lv[min(l, nl - 1)];
        CellGeo C;
        C.set(G, px, py, pz);
        if (slab)
            block3_level_flat(G, C, sorted, cell_start, a, px, py, pz, B, slab);
        else
            block3_level(G, C, sorted, cell_start, a, px, py, pz, B);
        if (a && (B.worst() <= C.settled_r2(G) || l == nl - 1)) todo = false;
    }
}

}  // namespace gpscal
