// knn_icp.hip -- exact k-NN over a multi-level uniform grid and the fused ICP
// iteration (transform -> 1-NN -> weighted centroid/covariance partials) plus
// the per-pair solve (partials -> 3x3 SVD -> pose update).  gfx950 only.
//
// Replaces pcl::KdTreeFLANN::setInputCloud / nearestKSearch (laserOdometry.cpp:
// 538-539,603,758,1119-1120; laserMapping.cpp:750-751,760,867) and the
// correspondence + BFTWithWeight loop of track_calibration.cc:145-181,366-545
// generalised to 3-D point clouds.
//
// Data layout in HBM (per batch of scan pairs):
//   tgt4        float4[sum m]            xyz + index-in-pair, caller order
//   sorted      float4[sum m * L]        per (pair, level): points grouped by cell,
//                                        cell id = (z*ny + y)*nx + x (x fastest, so
//                                        the 3 x-neighbours of a row are ONE run)
//   cell_start  uint32[total cells + 1]  global exclusive scan of the cell counts
//                                        = absolute position of each cell in `sorted`
//   src3        float[sum n][3]          sources grouped by their own level-0 cell
//                                        (rigid motion keeps that order coherent);
//                                        src_orig int32[sum n]: their original indices
//   warm_q      float4[sum n]            last iteration's neighbour of every source point:
//                                        xyz + its two certified radii in one word;
//                                        warm_i int32[sum n]: its index
//   nn_idx/sqd  int32/float[sum n]       correspondences, in grouped order
//   partials    double[blocks][NACC]     per-block sums, reduced in fixed order
//   pose        double[pairs][16] + float[pairs][12]
// Roofline: HBM; algorithmic bytes per iteration = 20 n + 12 m.
#include "common.hpp"
#include "knn_device.hpp"
#include "knn_host.hpp"
#include "svd3.hpp"
#include "wave_reduce.hpp"

#include <algorithm>
#include <deque>
#include <chrono>
#include <cmath>
#include <cstdlib>

namespace gpscal {

// --------------------------------------------------------------- build path

// `raw` holds the `total` points of pairs offs[0 .. npairs] (offs[0] = the position of its first point in `out`:
// several sources may fill one packed array one after the other).
__global__ void pack_points_kernel(const char *__restrict__ raw, int stride, const long long *__restrict__ offs,
                                   int npairs, long long total, float4 *__restrict__ out)
{
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= total) return;
    const long long i = offs[0] + r;
    // pair of point i: last b with offs[b] <= i
    int lo = 0, hi = npairs;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (offs[mid] <= i) lo = mid; else hi = mid;
    }
    const float *p = reinterpret_cast<const float *>(raw + (size_t)r * stride);
    out[i] = make_float4(p[0], p[1], p[2], __int_as_float((int)(i - offs[lo])));
}

// bbox[pair][6] holds order-preserving int keys: min xyz then max xyz.  One set of six atomics per workgroup
// (the waves meet in LDS first) and BB_PT points per thread: with an atomic set per wave of 4 points per lane,
// 64 x 65 536 points put 256 atomics on every address and the kernel ran at 0.5 TB/s (143 us).
constexpr int BB_PT = 16;
__global__ __launch_bounds__(BLOCK) void bbox_kernel(const float4 *__restrict__ pts, const long long *__restrict__ offs,
                                                      int *__restrict__ bbox)
{
    __shared__ float s_box[BLOCK / 64][6];
    int b = blockIdx.y;
    long long o = offs[b];
    int m = (int)(offs[b + 1] - o);
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i0 = blockIdx.x * BLOCK * 4 + threadIdx.x; i0 < m; i0 += gridDim.x * BLOCK * 4) {
        float4 p[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {  // four loads in flight
            const int i = i0 + t * BLOCK;
            p[t] = i < m ? pts[o + i] : make_float4(NAN, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (!finite3(p[t].x, p[t].y, p[t].z)) continue;
            mn[0] = fminf(mn[0], p[t].x); mx[0] = fmaxf(mx[0], p[t].x);
            mn[1] = fminf(mn[1], p[t].y); mx[1] = fmaxf(mx[1], p[t].y);
            mn[2] = fminf(mn[2], p[t].z); mx[2] = fmaxf(mx[2], p[t].z);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], s));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], s));
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            s_box[wave][a] = mn[a];
            s_box[wave][3 + a] = mx[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float v = s_box[0][threadIdx.x];
        for (int w = 1; w < BLOCK / 64; ++w)
            v = threadIdx.x < 3 ? fminf(v, s_box[w][threadIdx.x]) : fmaxf(v, s_box[w][threadIdx.x]);
        if (threadIdx.x < 3) atomicMin(&bbox[b * 6 + threadIdx.x], f2ord(v));
        else atomicMax(&bbox[b * 6 + threadIdx.x], f2ord(v));
    }
}

// Counting sort of every point into every level's cells, in two kernels.
//
// grid_count_kernel: rank of each point inside its cell = the value its atomicAdd on
// the cell counter returns; the rank is kept so that the scatter pass needs no atomics.
// Coarse levels have few cells (the top one <= 8): tens of thousands of global atomics on
// a handful of addresses serialise, so their increments are aggregated per workgroup in
// LDS (one LDS atomic per point, one global atomic per touched cell and workgroup).
constexpr int CO_MAX = 4096;    // LDS histogram bins = cells of the aggregated levels
constexpr int GC_PT = 4;        // points per thread
constexpr int GC_CHUNK = BLOCK * GC_PT;

__global__ __launch_bounds__(BLOCK) void grid_count_kernel(const PairDesc *__restrict__ pairs,
                                                            const float4 *__restrict__ tgt4,
                                                            unsigned *__restrict__ counts,
                                                            unsigned *__restrict__ ranks, long long total_points)
{
    __shared__ unsigned hist[CO_MAX];
    const int b = blockIdx.y;
    const PairDesc &P = pairs[b];
    const int base_i = blockIdx.x * GC_CHUNK;
    if (base_i >= P.m) return;  // uniform
    const int lc = P.coarse_from;  // levels lc.. are aggregated in LDS
    const long long co_base = lc < P.nlevels ? P.lv[lc].cell_base : 0;
    const GridDesc &Gt = P.lv[P.nlevels - 1];
    // (grid_cells, not nx*ny*nz: a tiled level -- the source grouping -- numbers whole 8x8 tiles; with the
    // plain product a small tiled cloud left LDS bins unzeroed and unflushed: lost points and wild ranks)
    const int nco = lc < P.nlevels ? (int)(Gt.cell_base + grid_cells(Gt) - co_base) : 0;
    for (int t = threadIdx.x; t < nco; t += BLOCK) hist[t] = 0u;
    __syncthreads();
    unsigned rl[GC_PT][MAX_LEVELS];
    float4 pt[GC_PT];
#pragma unroll
    for (int k = 0; k < GC_PT; ++k) {
        const int i = base_i + k * BLOCK + (int)threadIdx.x;
        pt[k] = make_float4(NAN, 0.f, 0.f, 0.f);
        if (i < P.m) pt[k] = tgt4[P.tgt_off + i];
        if (!finite3(pt[k].x, pt[k].y, pt[k].z)) continue;
#pragma unroll
        for (int l = 0; l < MAX_LEVELS; ++l) {
            if (l >= P.nlevels) break;
            const long long c = cell_of(P.lv[l], pt[k].x, pt[k].y, pt[k].z);
            if (l < lc) {
                ranks[(long long)l * total_points + P.tgt_off + i] = atomicAdd(&counts[c], 1u);
            } else {
                rl[k][l] = atomicAdd(&hist[(int)(c - co_base)], 1u);
            }
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < nco; t += BLOCK) {
        const unsigned h = hist[t];
        if (h) hist[t] = atomicAdd(&counts[co_base + t], h);  // base of this workgroup's points in the cell
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GC_PT; ++k) {
        const int i = base_i + k * BLOCK + (int)threadIdx.x;
        if (!finite3(pt[k].x, pt[k].y, pt[k].z)) continue;
#pragma unroll
        for (int l = 0; l < MAX_LEVELS; ++l) {
            if (l >= P.nlevels) break;
            if (l < lc) continue;
            const long long c = cell_of(P.lv[l], pt[k].x, pt[k].y, pt[k].z);
            ranks[(long long)l * total_points + P.tgt_off + i] = hist[(int)(c - co_base)] + rl[k][l];
        }
    }
}

__global__ __launch_bounds__(BLOCK) void grid_scatter_kernel(const PairDesc *__restrict__ pairs,
                                                              const float4 *__restrict__ tgt4,
                                                              const unsigned *__restrict__ ranks,
                                                              const unsigned *__restrict__ cell_start,
                                                              float4 *__restrict__ sorted, long long total_points)
{
    const int b = blockIdx.y;
    const PairDesc &P = pairs[b];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.m; i += gridDim.x * blockDim.x) {
        const float4 p = tgt4[P.tgt_off + i];
        if (!finite3(p.x, p.y, p.z)) continue;
        for (int l = 0; l < P.nlevels; ++l) {
            const long long c = cell_of(P.lv[l], p.x, p.y, p.z);
            sorted[cell_start[c] + ranks[(long long)l * total_points + P.tgt_off + i]] = p;
        }
    }
}

// ---- exclusive scan (uint32), 2048 elements per block
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = BLOCK * SCAN_ITEMS;

__global__ __launch_bounds__(BLOCK) void scan_tile_kernel(const unsigned *__restrict__ in,
                                                           unsigned *__restrict__ out,
                                                           unsigned *__restrict__ tile_sums, long long n)
{
    __shared__ unsigned wave_tot[BLOCK / 64];
    long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
    unsigned v[SCAN_ITEMS];
    unsigned sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        sum += v[k];
    }
    // inclusive wave scan of the per-thread sums
    unsigned inc = sum;
    int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        unsigned t = __shfl_up(inc, s);
        if (lane >= s) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    unsigned woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < BLOCK / 64; ++w) {
        unsigned t = wave_tot[w];
        if (w < wave) woff += t;
        total += t;
    }
    unsigned run = woff + inc - sum;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}

__global__ void scan_add_kernel(unsigned *__restrict__ data, const unsigned *__restrict__ tile_off, long long n)
{
    long long i = (long long)blockIdx.x * SCAN_TILE + threadIdx.x;
    unsigned add = tile_off[blockIdx.x];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k, i += BLOCK)
        if (i < n) data[i] += add;
}

static int exclusive_scan(gpscal_ctx *ctx, const unsigned *in, unsigned *out, long long n)
{
    int tiles = div_up(n, SCAN_TILE);
    DevBuf<unsigned> sums, sums_scanned;
    GPSCAL_HIP(ctx, sums.alloc_async(tiles, ctx->stream));
    hipLaunchKernelGGL(scan_tile_kernel, dim3(tiles), dim3(BLOCK), 0, ctx->stream, in, out, sums.p, n);
    if (tiles > 1) {
        GPSCAL_HIP(ctx, sums_scanned.alloc_async(tiles, ctx->stream));
        int rc = exclusive_scan(ctx, sums.p, sums_scanned.p, tiles);
        if (rc) return rc;
        hipLaunchKernelGGL(scan_add_kernel, dim3(tiles), dim3(BLOCK), 0, ctx->stream, out, sums_scanned.p, n);
    }
    GPSCAL_HIP(ctx, hipGetLastError());
    // temporaries are returned to the pool in stream order
    return GPSCAL_OK;
}

// ------------------------------------------------------------- query path

// Stand-alone batched search (gpscal_knn_search): one lane per query.
template <int K>
__global__ __launch_bounds__(BLOCK) void knn_search_kernel(const PairDesc *__restrict__ pairs,
                                                            const float4 *__restrict__ sorted,
                                                            const unsigned *__restrict__ cell_start,
                                                            const char *__restrict__ qraw, int stride, int n,
                                                            int *__restrict__ idx, float *__restrict__ sqd)
{
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    const PairDesc &P = pairs[0];
    float px = 0.f, py = 0.f, pz = 0.f;
    if (i < n) {
        const float *q = reinterpret_cast<const float *>(qraw + (size_t)i * stride);
        px = q[0]; py = q[1]; pz = q[2];
    }
    BestKeys<K> B;
    B.init();
    knn_query(P, sorted, cell_start, i < n && finite3(px, py, pz), px, py, pz, B);
    if (i >= n) return;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        bool ok = B.index(k) != 0x7fffffff;
        idx[(size_t)i * K + k] = ok ? B.index(k) : -1;
        sqd[(size_t)i * K + k] = ok ? B.dist2(k) : INFINITY;
    }
}

// Self-neighbour pass of the index build (ICP only).  For every target point q0, by
// ORIGINAL index g = tgt_off + idx:
//   nbr[g][0..3]  its 4 nearest other points (xyz + index bits),
//   pt_r2[g].x    r_a^2 = (D1/2)^2: a query closer than r_a to q0 has q0 as its unique
//                 nearest neighbour (|p-q| >= D1 - |p-q0| > |p-q0| for every other q);
//   pt_r2[g].y    r_b^2 = (D5/2)^2: closer than r_b, the nearest neighbour is q0 or one of
//                 the 4 listed points (everything else is at least D5 from q0).
// D1 / D5 = distance from q0 to its nearest / 5th nearest other point.  The 0.99 factor
// absorbs the float rounding of the three distances involved.
#ifndef GPSCAL_SELF_NN_FLAT
#define GPSCAL_SELF_NN_FLAT 1
#endif
#ifndef GPSCAL_SELFNN_MINW
#define GPSCAL_SELFNN_MINW 1
#endif
__global__ __launch_bounds__(BLOCK, GPSCAL_SELFNN_MINW) void self_nn_kernel(const PairDesc *__restrict__ pairs,
                                                         const float4 *__restrict__ sorted,
                                                         const unsigned *__restrict__ cell_start,
                                                         const float4 *__restrict__ pts4, float4 *__restrict__ nbr,
                                                         float2 *__restrict__ pt_r2)
{
#if GPSCAL_SELF_NN_FLAT
    __shared__ uint2 s_slab[BLOCK / 64][8 * 64];  // per wave: the run lists of block3_level_flat
    uint2 *slab = &s_slab[threadIdx.x >> 6][0];
#else
    uint2 *slab = nullptr;
#endif
    const int b = blockIdx.y;
    const PairDesc &P = pairs[b];
    // level-0 block of this pair = positions [first0, end0)
    const unsigned first0 = cell_start[P.lv[0].cell_base];
    const unsigned end0 = cell_start[P.lv[0].cell_base + grid_cells(P.lv[0])];
    for (unsigned j0 = first0 + blockIdx.x * BLOCK; j0 < end0; j0 += gridDim.x * BLOCK) {
        const unsigned j = j0 + threadIdx.x;
        const bool act = j < end0;
        float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act) c = sorted[j];
        BestKeys<6> B;
        B.init();
        knn_query(P, sorted, cell_start, act, c.x, c.y, c.z, B, 0, slab);
        if (!act) continue;
        const int own = __float_as_int(c.w);
        const long long g = P.tgt_off + own;
        // the five nearest OTHER points, in (d2, index) order
        float od[5];
        int oi[5];
        int n = 0;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (B.index(k) != own && n < 5) {
#pragma unroll
                for (int t = 0; t < 5; ++t)
                    if (t == n) {
                        od[t] = B.dist2(k);
                        oi[t] = B.index(k);
                    }
                ++n;
            }
        }
#pragma unroll
        for (int t = 0; t < 5; ++t)
            if (t >= n) {
                od[t] = INFINITY;
                oi[t] = 0x7fffffff;
            }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // a missing entry (cloud of < 5 points) sits at infinity: never the nearest
            float4 v = make_float4(INFINITY, INFINITY, INFINITY, __int_as_float(0x7fffffff));
            if (oi[t] != 0x7fffffff) v = pts4[P.tgt_off + oi[t]];
            nbr[4 * g + t] = v;
        }
        pt_r2[g] = make_float2(isfinite(od[0]) ? 0.25f * 0.99f * od[0] : 3.0e38f,
                               isfinite(od[4]) ? 0.25f * 0.99f * od[4] : 3.0e38f);
    }
}

// The fused ICP correspondence kernel: transform -> exact 1-NN -> weighted
// centroid / covariance partials.  One lane per source point, QPT batches of 256
// consecutive points of the spatially grouped source per workgroup.
//
// Per query, cheapest test first (all exact, see self_nn_kernel for the proofs):
//   1. warm start: last iteration's neighbour q0 and its two radii are read back from
//      per-query streams; closer than r_a, q0 is PROVEN to be the nearest neighbour;
//   2. closer than r_b, the answer is q0 or one of its 4 listed neighbours: one 64-byte
//      gather, five distance evaluations;
//   3. otherwise the multi-level grid search, seeded with the best so far so the row and
//      cell bounds prune most of the 3x3x3 block; long runs of candidates are scanned
//      by the whole wave (scan_runs).
// In a converged alignment ~93 % of the queries stop at 1 and nearly all others at 2.
// An LDS-staged variant of level 0 (workgroup box of cells copied to LDS) was built
// and measured SLOWER than this pruned global path (the box holds ~3.3 points per
// query against ~2.4 the query reads); see DESIGN.md.
// Queries per workgroup of the step kernel.  Smaller workgroups give wave slots and LDS back sooner (a workgroup
// retires with its slowest wave: in the search-heavy iterations the waves of one workgroup differ a lot); with
// single-wave workgroups the kernel itself is fastest (launch times 346 ... 47 us against 385 ... 49 us for 256
// threads) but four concurrent chains of 16 384-workgroup launches then cost more than they hide.
// Both sizes are compiled; the batch picks one (batch_setup_sources): 128 threads for batches that fill the chip many
// times over (measured at 64 x 65 536 points, 4 chains: 64 / 128 / 256 threads = 728 / 775 / 746 k iterations/s), 256
// for small ones, where the launch is latency-bound and the solve kernel's sum over the workgroups' partial sums is
// on the critical path (one 65 536-point pair: 33.4 k iterations/s with 128 threads, 35.8 k with 256).

// ---- the sums of one iteration.  The step kernel issues vector instructions in 80 % of its cycles while it searches
// and in 91 % of them in a converged launch (SQ counters, round 3), so the sums are written
// for instruction count: a lane without a correspondence contributes zeros through zeroed INPUTS (straight-line
// code, no masked block), products enter by one fma, the number of correspondences is counted by ballot on the
// scalar unit instead of a 17th float64 sum, and sqrt((double)d2) is a float32 rsq with one float64 Newton step.
template <bool WEIGHTED>
struct SumLayout {
    static constexpr int NACC = WEIGHTED ? NACC_WEIGHTED : NACC_PLAIN;
    static constexpr int NT = NACC - 1;                  // float64 terms a lane carries
    static constexpr int COUNT = WEIGHTED ? 24 : 0;      // the sum that is the number of correspondences
    static constexpr int FIRST = WEIGHTED ? 0 : 1;       // term j is sum FIRST + j
};

// sqrt((double)d2) for a float32 d2 >= 0 to 2e-14 relative (v_rsq_f32 is good to 1 ulp = delta, the Newton step
// leaves 1.5 delta^2); 0 for zero and denormal d2.  7 instructions against ~22 for the correctly rounded sqrt.
__device__ __forceinline__ double sqrt_of_f32(float d2)
{
    const float y = d2 >= 1.17549435e-38f ? __builtin_amdgcn_rsqf(d2) : 0.f;
    const double D = d2, Y = y;
    const double s = D * Y;
    const double e = __fma_rn(-s, s, D);
    return __fma_rn(e * Y, 0.5, s);
}

// One correspondence into the lane's terms.  INIT: the terms are set, not added to (the first query of a lane).
// A lane without a correspondence passes found = false: its inputs become zeros, so do its terms.
template <bool WEIGHTED, bool INIT>
__device__ __forceinline__ void pair_terms(double *t, bool found, float px, float py, float pz, const float4 &nq,
                                           float bd, double w, bool want_dist)
{
    const double dpx = found ? px : 0.f, dpy = found ? py : 0.f, dpz = found ? pz : 0.f;
    const double qx = found ? nq.x : 0.f, qy = found ? nq.y : 0.f, qz = found ? nq.z : 0.f;
    const double dist = want_dist ? sqrt_of_f32(found ? bd : 0.f) : 0.0;
#define GPSCAL_TERM(j, v) t[j] = INIT ? (v) : t[j] + (v)
#define GPSCAL_PROD(j, a, b) t[j] = INIT ? (a) * (b) : __fma_rn(a, b, t[j])
    if (WEIGHTED) {
        if (!found) w = 0.0;
        const double w2 = w * w;
        GPSCAL_TERM(0, w);
        GPSCAL_PROD(1, w, dpx); GPSCAL_PROD(2, w, dpy); GPSCAL_PROD(3, w, dpz);
        GPSCAL_PROD(4, w, qx);  GPSCAL_PROD(5, w, qy);  GPSCAL_PROD(6, w, qz);
        const double ax = w2 * dpx, ay = w2 * dpy, az = w2 * dpz;
        GPSCAL_PROD(7, ax, qx);  GPSCAL_PROD(8, ax, qy);  GPSCAL_PROD(9, ax, qz);
        GPSCAL_PROD(10, ay, qx); GPSCAL_PROD(11, ay, qy); GPSCAL_PROD(12, ay, qz);
        GPSCAL_PROD(13, az, qx); GPSCAL_PROD(14, az, qy); GPSCAL_PROD(15, az, qz);
        GPSCAL_TERM(16, dist);
        GPSCAL_TERM(17, w2);
        GPSCAL_TERM(18, ax); GPSCAL_TERM(19, ay); GPSCAL_TERM(20, az);
        GPSCAL_PROD(21, w2, qx); GPSCAL_PROD(22, w2, qy); GPSCAL_PROD(23, w2, qz);
    } else {
        GPSCAL_TERM(0, dpx); GPSCAL_TERM(1, dpy); GPSCAL_TERM(2, dpz);
        GPSCAL_TERM(3, qx);  GPSCAL_TERM(4, qy);  GPSCAL_TERM(5, qz);
        GPSCAL_PROD(6, dpx, qx);  GPSCAL_PROD(7, dpx, qy);  GPSCAL_PROD(8, dpx, qz);
        GPSCAL_PROD(9, dpy, qx);  GPSCAL_PROD(10, dpy, qy); GPSCAL_PROD(11, dpy, qz);
        GPSCAL_PROD(12, dpz, qx); GPSCAL_PROD(13, dpz, qy); GPSCAL_PROD(14, dpz, qz);
        GPSCAL_TERM(15, dist);
    }
#undef GPSCAL_TERM
#undef GPSCAL_PROD
}

// v + (v of the lane `shift` below in the row, 0 beyond the row's start): bound_ctrl supplies the zero, so no
// register has to be cleared in front of the move.
template <int CTRL>
__device__ __forceinline__ double dpp_shr_add_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
    return v + __hiloint2double(hi, lo);
}

// Sum of every term over the 64 lanes, in fixed order: 8 terms at a time are transposed through the wave's slab
// (lane (k, c) adds the copies of term k of lanes c, c + 8, ... c + 56, three DPP steps fold the 8 columns); lane 8k+7
// ends up with term g0+k and hands it to `sink(term, value)`.  ~20 vector instructions per 8 terms against ~160 for
// 8 full DPP wave reductions.  The LDS traffic of this transpose (16 KB per wave) is what bounds a converged launch,
// so it is laid out free of bank conflicts: rows of 72 doubles (a row starts 16 banks behind the previous one) and
// column-wise 8-byte reads -- 32 lanes of a read touch 4 rows x 16 banks = every bank once -- where row-wise 16-byte
// reads of 512-byte rows hit four banks sixteen times.
constexpr int SLAB_PITCH = 72;
template <int NT, class Sink>
__device__ __forceinline__ void wave_reduce_terms(const double *t, double *__restrict__ slab, Sink sink)
{
    const int lane = threadIdx.x & 63;
    const int k = lane >> 3, c = lane & 7;
    const double *col = slab + k * SLAB_PITCH + c;
#pragma unroll
    for (int g0 = 0; g0 < NT; g0 += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (g0 + j < NT) slab[j * SLAB_PITCH + lane] = t[g0 + j];
        __builtin_amdgcn_wave_barrier();
        double v = ((col[0] + col[8]) + (col[16] + col[24])) + ((col[32] + col[40]) + (col[48] + col[56]));
        v = dpp_shr_add_f64<0x111>(v);  // row_shr:1
        v = dpp_shr_add_f64<0x112>(v);  // row_shr:2
        v = dpp_shr_add_f64<0x114>(v);  // row_shr:4 -> lane 8k+7 holds term k
        if (c == 7 && g0 + k < NT) sink(g0 + k, v);
        __builtin_amdgcn_wave_barrier();
    }
}

// The work of one workgroup of the step: its slice of the pair's grouped source (QPT batches of STEP_BLOCK points from
// `first`) through transform, certificates and search; leaves every wave's sums in wsum[wave][0 .. NACC) (the caller
// synchronises the workgroup and adds them).  T: the pair's float32 pose (12 values).  Shared by icp_step_kernel
// (one launch per iteration) and icp_persistent_kernel (all iterations of small batches in one launch).
struct StepArgs {
    const float *__restrict__ src3;  // grouped source points, 12 bytes each
    const double *__restrict__ wsrc;
    const float4 *__restrict__ sorted;
    const float4 *__restrict__ nbr;
    const float2 *__restrict__ pt_r2;
    const unsigned *__restrict__ cell_start;
    int *__restrict__ nn_idx;
    float *__restrict__ nn_sqd;
    float4 *__restrict__ warm_q;  // last iteration's neighbour: x, y, z and its two certified radii in one word
    int *__restrict__ warm_i;     // ... and its index (read only by queries the first certificate does not settle)
};
template <int QPT, bool WEIGHTED, bool BALL, int STEP_BLOCK, bool FLAT = false>
__device__ __forceinline__ unsigned step_body(const StepArgs &A, const PairDesc &P, const float *T, int first, int src_n,
                                          long long src_off, long long tgt_off, int diag, int write_nn,
                                          double (&wsum)[STEP_BLOCK / 64][WEIGHTED ? NACC_WEIGHTED : NACC_PLAIN],
                                          double (&tslab)[STEP_BLOCK / 64][8][SLAB_PITCH])
{
    const float *__restrict__ src3 = A.src3;
    const double *__restrict__ wsrc = A.wsrc;
    const float4 *__restrict__ sorted = A.sorted;
    const float4 *__restrict__ nbr = A.nbr;
    const float2 *__restrict__ pt_r2 = A.pt_r2;
    const unsigned *__restrict__ cell_start = A.cell_start;
    int *__restrict__ nn_idx = A.nn_idx;
    float *__restrict__ nn_sqd = A.nn_sqd;
    float4 *__restrict__ warm_q = A.warm_q;
    int *__restrict__ warm_i = A.warm_i;
    const float r00 = T[0], r01 = T[1], r02 = T[2], tx = T[3];
    const float r10 = T[4], r11 = T[5], r12 = T[6], ty = T[7];
    const float r20 = T[8], r21 = T[9], r22 = T[10], tz = T[11];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    using SL = SumLayout<WEIGHTED>;
    double t[SL::NT];
    if (QPT > 1) {
#pragma unroll
        for (int k = 0; k < SL::NT; ++k) t[k] = 0.0;
    }
    unsigned n_found = 0;  // correspondences of this wave
    unsigned n_need = 0;  // queries of this wave that took the grid search

#pragma unroll 1
    for (int q = 0; q < QPT; ++q) {
        const int i = first + q * STEP_BLOCK + (int)threadIdx.x;
        const bool valid = i < src_n;
        float sx = 0.f, sy = 0.f, sz = 0.f;
        float4 wq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (valid) {
            // two coalesced streams, 28 bytes per query: the point, and last iteration's neighbour with its radii
            const float *sp = src3 + 3 * (src_off + i);
            sx = sp[0]; sy = sp[1]; sz = sp[2];
            wq = warm_q[src_off + i];
        }
        // both radii in one word: the upper 16 bits of each float (truncation only shrinks a radius, which keeps the
        // certificates valid); a zero word = no remembered neighbour
        const unsigned pr = __float_as_uint(wq.w);
        const float ra2 = __uint_as_float(pr & 0xffff0000u), rb2 = __uint_as_float(pr << 16);
        bool ok = valid && finite3(sx, sy, sz);
        const float px = __fmaf_rn(r00, sx, __fmaf_rn(r01, sy, __fmaf_rn(r02, sz, tx)));
        const float py = __fmaf_rn(r10, sx, __fmaf_rn(r11, sy, __fmaf_rn(r12, sz, ty)));
        const float pz = __fmaf_rn(r20, sx, __fmaf_rn(r21, sy, __fmaf_rn(r22, sz, tz)));
        const bool warm = ok && pr != 0u;
        const float d0 = sqdist(px, py, pz, wq.x, wq.y, wq.z);
        const bool t1 = warm && d0 < ra2;  // tier 1: inside r_a, the remembered neighbour is proven nearest
        bool need = ok && !t1;             // still needs the grid search
        // the neighbour's index: a third stream only for the queries tier 1 leaves (and for everybody in the launch that
        // writes the correspondences out)
        int wi = 0x7fffffff;
        if (warm && (!t1 || (write_nn & 1))) wi = warm_i[src_off + i];
        float4 nq = make_float4(wq.x, wq.y, wq.z, __int_as_float(wi));  // the neighbour this iteration ends with
        BestQ B;
        B.init();
        if (warm) {
            B.consider(d0, nq, BestQ::WARM);
            if (need && d0 < rb2 && !(diag & 4)) {
                // tier 2: inside r_b the answer is q0 or one of its 4 listed neighbours
                const float4 *nb = nbr + 4 * (tgt_off + wi);
                const float4 n0 = nb[0], n1 = nb[1], n2 = nb[2], n3 = nb[3];
                B.consider(sqdist(px, py, pz, n0.x, n0.y, n0.z), n0, BestQ::LIST + 0);
                B.consider(sqdist(px, py, pz, n1.x, n1.y, n1.z), n1, BestQ::LIST + 1);
                B.consider(sqdist(px, py, pz, n2.x, n2.y, n2.z), n2, BestQ::LIST + 2);
                B.consider(sqdist(px, py, pz, n3.x, n3.y, n3.z), n3, BestQ::LIST + 3);
                need = false;
            }
        }
        if (diag & 1) need = false;
        n_need += (unsigned)__popcll(__ballot(need));
#ifdef GPSCAL_STATS
        STAT_WAVE(0, __popcll(__ballot(valid)));
        STAT_WAVE(1, __popcll(__ballot(need)));
        STAT_WAVE(2, __ballot(need) != 0ull ? 1 : 0);
        STAT_WAVE(17, __popcll(__ballot(ok && !need && B.pos != BestQ::WARM)));  // settled by tier 2 with a new neighbour
#endif
        if (__ballot(need) != 0ull)  // wave-uniform: a settled wave skips the search's set-up as well
            knn_query<BestQ, BALL, FLAT>(P, sorted, cell_start, need, px, py, pz, B, diag >> 8,
                                         FLAT ? reinterpret_cast<uint2 *>(&tslab[wave][0][0]) : nullptr);
        ok = ok && (t1 || B.index() != 0x7fffffff);  // (a query tier 1 settles may not have read its neighbour's index)
        if (ok && B.pos != BestQ::WARM) {
            // the neighbour changed: remember it and its radii for the next iteration
            nq = B.pos < BestQ::LIST ? sorted[B.pos] : nbr[4 * (tgt_off + wi) + (B.pos - BestQ::LIST)];
            const float2 r2 = pt_r2[tgt_off + B.index()];
            const unsigned npr = (__float_as_uint(r2.x) & 0xffff0000u) | (__float_as_uint(r2.y) >> 16);
            warm_q[src_off + i] = make_float4(nq.x, nq.y, nq.z, __uint_as_float(npr));
            warm_i[src_off + i] = __float_as_int(nq.w);
        }
        const float bd = B.dist2();
        if (valid && (write_nn & 1)) {  // the correspondences are an output of the run's last iteration only
            nn_idx[src_off + i] = ok ? B.index() : -1;
            nn_sqd[src_off + i] = ok ? bd : INFINITY;
        }
        n_found += (unsigned)__popcll(__ballot(ok));
        pair_terms<WEIGHTED, QPT == 1>(t, ok, px, py, pz, nq, bd, WEIGHTED ? (valid ? wsrc[src_off + i] : 0.0) : 1.0,
                                       !(write_nn & 2));
    }
    // Wave sums in fixed order (the caller adds the waves' rows)
    wave_reduce_terms<SL::NT>(t, &tslab[wave][0][0], [&](int k, double v) { wsum[wave][SL::FIRST + k] = v; });
    if (lane == 0) wsum[wave][SL::COUNT] = (double)n_found;
    return n_need;
}

#ifndef GPSCAL_STEP_MINW
#define GPSCAL_STEP_MINW 1
#endif
// FLAT: the rows of a level are walked as per-lane lists (block3_level_flat, the wave's transpose slab holds them) --
// the form for batches of one or two scans, where a launch is one wave per SIMD and its duration is the length of
// the longest dependent chain of loads: a wave then pays its longest lane's list instead of the sum over the rows
// of the longest run in each.
template <int QPT, bool WEIGHTED, bool BALL, int STEP_BLOCK, bool FLAT = false>
__global__ __launch_bounds__(STEP_BLOCK, GPSCAL_STEP_MINW) void icp_step_kernel(
    const PairDesc *__restrict__ pairs, const int *__restrict__ blk_pair, const int *__restrict__ blk_first,
    const float *__restrict__ src3, const double *__restrict__ wsrc, const float4 *__restrict__ sorted,
    const float4 *__restrict__ nbr, const float2 *__restrict__ pt_r2, const unsigned *__restrict__ cell_start,
    const float *__restrict__ pose32, int *__restrict__ nn_idx, float *__restrict__ nn_sqd,
    float4 *__restrict__ warm_q, int *__restrict__ warm_i, double *__restrict__ partials, int nblk, int diag,
    int write_nn, int uni_n, int uni_m, int uni_bpp, int uni_pair0)
{
    constexpr int NACC = WEIGHTED ? NACC_WEIGHTED : NACC_PLAIN;
    __shared__ double wsum[STEP_BLOCK / 64][NACC];
    __shared__ double tslab[STEP_BLOCK / 64][8][SLAB_PITCH];  // per-wave transpose slab (4.5 KiB / wave)

    const int lb = xcd_remap(blockIdx.x, nblk);
    // A batch of equal-sized scans (uni_n > 0: every source cloud uni_n points, every target cloud uni_m, stored back
    // to back) needs no table to find a workgroup's slice: two dependent scalar loads less in front of the streams,
    // which is a fifth of a workgroup's life in the converged state.
    int b, first, src_n;
    long long src_off, tgt_off;
    if (uni_n > 0) {
        const int bl = lb / uni_bpp;
        b = uni_pair0 + bl;
        first = (lb - bl * uni_bpp) * (STEP_BLOCK * QPT);
        src_n = uni_n;
        src_off = (long long)b * uni_n;
        tgt_off = (long long)b * uni_m;
    } else {
        b = __builtin_amdgcn_readfirstlane(blk_pair[lb]);
        first = __builtin_amdgcn_readfirstlane(blk_first[lb]);
        src_n = pairs[b].n;
        src_off = pairs[b].src_off;
        tgt_off = pairs[b].tgt_off;
    }
    const StepArgs A = {src3, wsrc, sorted, nbr, pt_r2, cell_start, nn_idx, nn_sqd, warm_q, warm_i};
    step_body<QPT, WEIGHTED, BALL, STEP_BLOCK, FLAT>(A, pairs[b], pose32 + (size_t)b * 12, first, src_n, src_off, tgt_off,
                                                     diag, write_nn, wsum, tslab);
    __syncthreads();
    if (threadIdx.x < NACC) {
        double v = 0.0;
#pragma unroll
        for (int w = 0; w < STEP_BLOCK / 64; ++w) v += wsum[w][threadIdx.x];
        partials[(size_t)lb * NACC + threadIdx.x] = v;
    }
}

// Agent-scope (sc1) loads and stores: data handed from one workgroup to another inside a launch
// (icp_persistent_kernel) must not be served from, nor stay in, a CU's vector L1 (MI355X_MICROARCH.md, "Workgroup
// dispatch, XCD placement & inter-workgroup visibility").
template <class T>
__device__ __forceinline__ T ld_agent(const T *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <class T>
__device__ __forceinline__ void st_agent(T *p, T v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// A pair's sums from the 64 lanes' shares of its workgroups' partial sums, in fixed order: the terms by the same LDS
// transpose as in the step kernel (two passes of ~20 vector instructions instead of sixteen DPP wave reductions of 18),
// the count by one DPP reduction.  Lane 0 ends up with the totals in a[] (the other lanes' a[] are unspecified).
template <bool WEIGHTED>
__device__ __forceinline__ void wave_totals(double (&a)[WEIGHTED ? NACC_WEIGHTED : NACC_PLAIN], double *__restrict__ slab,
                                            double *__restrict__ tot)
{
    using SL = SumLayout<WEIGHTED>;
    constexpr int NACC = SL::NACC;
    wave_reduce_terms<SL::NT>(&a[SL::FIRST], slab, [&](int k, double v) { tot[SL::FIRST + k] = v; });
    const double cnt = wave_sum(a[SL::COUNT]);
    if ((threadIdx.x & 63) == 0) tot[SL::COUNT] = cnt;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int k = 0; k < NACC; ++k) a[k] = tot[k];
}

// From a pair's total sums to its new pose (TC:416-541 on 3-D data): one lane.  AGENT: the pose is read and written
// with agent-scope accesses (the persistent kernel hands it from workgroup to workgroup).
template <bool WEIGHTED, bool AGENT>
__device__ __forceinline__ void solve_pose(const double (&a)[WEIGHTED ? NACC_WEIGHTED : NACC_PLAIN], double *T,
                                           float *pose32_b, double *err_slot)
{
    const double sw = a[0];
    const double cnt = WEIGHTED ? a[24] : a[0];
    if (err_slot) *err_slot = cnt > 0.0 ? a[16] / cnt : 0.0;
    if (!(sw > 0.0)) return;  // no correspondences: pose unchanged
    double cp[3] = {a[1] / sw, a[2] / sw, a[3] / sw};
    double cq[3] = {a[4] / sw, a[5] / sw, a[6] / sw};
    double H[9];
    if (WEIGHTED) {
        // H = sum w^2 (p - cp)(q - cq)^T expanded in raw moments
        const double sw2 = a[17];
        const double sp2[3] = {a[18], a[19], a[20]}, sq2[3] = {a[21], a[22], a[23]};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                H[3 * r + c] = a[7 + 3 * r + c] - cp[r] * sq2[c] - sp2[r] * cq[c] + sw2 * cp[r] * cq[c];
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) H[3 * r + c] = a[7 + 3 * r + c] - sw * cp[r] * cq[c];
    }
    double R[9];
    kabsch_from_H(H, R);
    double t[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) t[r] = cq[r] - (R[3 * r] * cp[0] + R[3 * r + 1] * cp[1] + R[3 * r + 2] * cp[2]);
    // T <- [R|t] * T
    double To[12], Tn[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) To[k] = AGENT ? ld_agent(T + k) : T[k];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            double v = R[3 * r] * To[c] + R[3 * r + 1] * To[4 + c] + R[3 * r + 2] * To[8 + c];
            if (c == 3) v += t[r];
            Tn[4 * r + c] = v;
        }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        if (AGENT) {
            st_agent(T + k, Tn[k]);
            st_agent(pose32_b + k, (float)Tn[k]);
        } else {
            T[k] = Tn[k];
            pose32_b[k] = (float)Tn[k];
        }
    }
}

// One wave per pair: reduce the pair's block partials in fixed order, solve the
// rigid transform, compose the pose.
template <bool WEIGHTED>
__global__ __launch_bounds__(64) void icp_solve_kernel(const PairDesc *__restrict__ pairs,
                                                        const double *__restrict__ partials,
                                                        double *__restrict__ pose64, float *__restrict__ pose32,
                                                        double *__restrict__ err_hist, int it, int iters_cap, int pair0)
{
    constexpr int NACC = WEIGHTED ? NACC_WEIGHTED : NACC_PLAIN;
    const int b = pair0 + blockIdx.x;
    const PairDesc &P = pairs[b];
    const int lane = threadIdx.x;
    double a[NACC];
#pragma unroll
    for (int k = 0; k < NACC; ++k) a[k] = 0.0;
    for (int j = lane; j < P.pblk_cnt; j += 64) {
        const double *pp = partials + (size_t)(P.pblk_off + j) * NACC;
#pragma unroll
        for (int k = 0; k < NACC; ++k) a[k] += pp[k];
    }
    __shared__ double slab[8 * SLAB_PITCH];
    __shared__ double tot[NACC];
    wave_totals<WEIGHTED>(a, slab, tot);
    if (lane != 0) return;
    solve_pose<WEIGHTED, false>(a, pose64 + (size_t)b * 16, pose32 + (size_t)b * 12,
                                err_hist ? err_hist + (size_t)b * iters_cap + it : nullptr);
}

// ------------------------------------------------------------ all iterations of a small batch in one launch
//
// One 65 536-point pair is 2 MiB of compulsory traffic per iteration -- 0.26 us at HBM speed -- so a run of such a
// pair is a chain of launch latencies (step ~20 us + solve ~9 us per iteration through the captured graph).  For
// batches whose step grid fits the chip at once (every workgroup resident), the whole run is ONE launch:
// per iteration a workgroup works its slice (step_body), stores its partial sums with agent-scope stores, drains
// them and takes a ticket of its pair; the workgroup that draws the pair's last ticket of the iteration adds the
// partial sums in the solve kernel's order, solves, stores the new pose (agent scope) and raises the pair's
// generation; every workgroup of the pair polls the generation (one lane, agent-scope loads, s_sleep) and goes on.
// Same sums in the same order as icp_step_kernel + icp_solve_kernel: bit-identical poses.  The hand-off is the
// sc1 form of MI355X_MICROARCH.md's table (stores and loads of the handed-off bytes all agent scope, every storing
// wave drained, one lane signals after the workgroup's barrier, the consumer loads after its own ticket returned /
// its poll matched).  Every wait is bounded: a workgroup that polls longer than ~a second raises the pair's error
// word and every workgroup leaves (the host reports GPSCAL_EHIP); the grid is only launched when all of it is
// resident (2 workgroups per CU at most), so nothing waits for a workgroup that cannot start.
struct IcpCtl {  // one 128-byte line per pair
    unsigned ticket, gen, error, pad[29];
};
constexpr int PERSIST_BLOCK = 256;
constexpr unsigned PERSIST_SPIN_LIMIT = 4000000u;

template <bool WEIGHTED, bool BALL>
__global__ __launch_bounds__(PERSIST_BLOCK) void icp_persistent_kernel(
    const PairDesc *__restrict__ pairs, const int *__restrict__ blk_pair, const int *__restrict__ blk_first,
    StepArgs A, double *__restrict__ partials, double *__restrict__ pose64, float *__restrict__ pose32,
    double *__restrict__ err_hist, IcpCtl *__restrict__ ctl, int nblk, int iters, int iters_cap, int want_err, int ball_r)
{
    constexpr int NACC = WEIGHTED ? NACC_WEIGHTED : NACC_PLAIN;
    constexpr int NW = PERSIST_BLOCK / 64;
    __shared__ double wsum[NW][NACC];
    __shared__ double tslab[NW][8][SLAB_PITCH];
    __shared__ float s_pose[12];
    __shared__ int s_flag;

    const int lb = xcd_remap(blockIdx.x, nblk);
    const int b = __builtin_amdgcn_readfirstlane(blk_pair[lb]);
    const int first = __builtin_amdgcn_readfirstlane(blk_first[lb]);
    const PairDesc &P = pairs[b];
    const int src_n = P.n, nb_pair = P.pblk_cnt, slot0 = P.pblk_off;
    const long long src_off = P.src_off, tgt_off = P.tgt_off;
    IcpCtl *c = ctl + b;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    for (int it = 0; it < iters; ++it) {
        if (threadIdx.x < 12) s_pose[threadIdx.x] = ld_agent(pose32 + (size_t)b * 12 + threadIdx.x);
        __syncthreads();
        step_body<1, WEIGHTED, BALL, PERSIST_BLOCK>(A, P, s_pose, first, src_n, src_off, tgt_off, ball_r << 8,
                                                    (it == iters - 1 ? 1 : 0) | (want_err ? 0 : 2), wsum, tslab);
        __syncthreads();
        if (threadIdx.x < NACC) {
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += wsum[w][threadIdx.x];
            st_agent(partials + (size_t)lb * NACC + threadIdx.x, v);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains before the workgroup signals
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned t = __hip_atomic_fetch_add(&c->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_flag = t == (unsigned)(it + 1) * (unsigned)nb_pair - 1u ? 1 : 0;
        }
        __syncthreads();
        if (s_flag && wave == 0) {  // this workgroup delivered the pair's last partial sums: it solves
            double a[NACC];
#pragma unroll
            for (int k = 0; k < NACC; ++k) a[k] = 0.0;
            for (int j = lane; j < nb_pair; j += 64) {
                const double *pp = partials + (size_t)(slot0 + j) * NACC;
#pragma unroll
                for (int k = 0; k < NACC; ++k) a[k] += ld_agent(pp + k);
            }
            wave_totals<WEIGHTED>(a, &tslab[0][0][0], &wsum[0][0]);  // (wave 0's slab and row are free by now)
            if (lane == 0) {
                solve_pose<WEIGHTED, true>(a, pose64 + (size_t)b * 16, pose32 + (size_t)b * 12,
                                           want_err ? err_hist + (size_t)b * iters_cap + it : nullptr);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                st_agent(&c->gen, (unsigned)(it + 1));
            }
        }
        if (it + 1 == iters) break;  // nobody reads the last pose inside the launch
        if (threadIdx.x == 0) {  // wait for the pair's new pose
            unsigned spins = 0;
            int bad = 0;
            while (ld_agent(&c->gen) < (unsigned)(it + 1)) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > PERSIST_SPIN_LIMIT || ld_agent(&c->error) != 0u) {
                    st_agent(&c->error, 1u);
                    bad = 1;
                    break;
                }
            }
            s_flag = bad;
        }
        __syncthreads();
        if (s_flag) return;  // (uniform) a wait ran out: give up, the host reports it
        __syncthreads();     // s_flag is written again next iteration
    }
}

__global__ void fill_warm_kernel(float4 *__restrict__ warm_q, int *__restrict__ warm_i, long long n)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    warm_q[i] = make_float4(0.f, 0.f, 0.f, 0.f);  // radii word 0 = no warm start
    warm_i[i] = 0x7fffffff;
}

// The grouped source as the step kernel reads it: 12 bytes per point, the original indices aside.
__global__ void split_source_kernel(const float4 *__restrict__ src4, float *__restrict__ src3, int *__restrict__ src_orig,
                                    long long n)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = src4[i];
    src3[3 * i] = p.x;
    src3[3 * i + 1] = p.y;
    src3[3 * i + 2] = p.z;
    src_orig[i] = __float_as_int(p.w);
}

__global__ void pose_to_f32_kernel(const double *__restrict__ pose64, float *__restrict__ pose32, int npairs)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npairs * 12) return;
    int b = i / 12, k = i % 12;
    pose32[i] = (float)pose64[(size_t)b * 16 + k];
}

__global__ void unsort_nn_kernel(const PairDesc *__restrict__ pairs, const int *__restrict__ src_orig,
                                 const int *__restrict__ nn_idx, const float *__restrict__ nn_sqd,
                                 int *__restrict__ idx_out, float *__restrict__ sqd_out)
{
    int b = blockIdx.y;
    const PairDesc &P = pairs[b];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += gridDim.x * blockDim.x) {
        int orig = src_orig[P.src_off + i];
        idx_out[P.src_off + orig] = nn_idx[P.src_off + i];
        sqd_out[P.src_off + orig] = nn_sqd[P.src_off + i];
    }
}

__global__ void gather_weights_kernel(const PairDesc *__restrict__ pairs, const int *__restrict__ src_orig,
                                      const double *__restrict__ w_in, double *__restrict__ w_sorted)
{
    int b = blockIdx.y;
    const PairDesc &P = pairs[b];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += gridDim.x * blockDim.x) {
        int orig = src_orig[P.src_off + i];
        w_sorted[P.src_off + i] = w_in[P.src_off + orig];
    }
}

// Puts the points of every cell of a one-level grouping into original-index order.  The counting sort hands
// out the slots of a cell with integer atomics, so the order inside a cell would otherwise differ from build
// to build -- and with it the order in which the step kernel adds the float64 sums.  One lane per cell
// (rank sort in registers up to 16 points, one wave with an element per lane up to 64); a wave takes the runs of 65..512 together (rank of every element, then one
// scatter); longer runs (more than 512 points in one cell: degenerate input) stay as they are.
__global__ __launch_bounds__(BLOCK) void order_runs_kernel(float4 *__restrict__ sorted,
                                                            const unsigned *__restrict__ cell_start, long long ncells)
{
    const long long c = (long long)blockIdx.x * BLOCK + threadIdx.x;
    unsigned s = 0, n = 0;
    if (c < ncells) {
        s = cell_start[c];
        n = cell_start[c + 1] - s;
    }
    // Runs of up to OR_REG points: all of them loaded at once, ranked in registers (indices are distinct), each
    // written to its place.  The insertion sort below walks a chain of dependent global loads (n^2 / 4 of them)
    // and is left to the few longer runs: with it alone the kernel took 434 us for 64 x 65 536 points.
    constexpr int OR_REG = 16;
    if (n >= 2 && n <= 4) {
        float4 e[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) e[t] = t < (int)n ? sorted[s + t] : make_float4(0.f, 0.f, 0.f, __int_as_float(0x7fffffff));
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            unsigned r = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) r += __float_as_int(e[u].w) < __float_as_int(e[t].w) ? 1u : 0u;
            if (t < (int)n && r != (unsigned)t) sorted[s + r] = e[t];
        }
    } else if (n > 4 && n <= OR_REG) {
        float4 e[OR_REG];
#pragma unroll
        for (int t = 0; t < OR_REG; ++t)
            e[t] = t < (int)n ? sorted[s + t] : make_float4(0.f, 0.f, 0.f, __int_as_float(0x7fffffff));
#pragma unroll
        for (int t = 0; t < OR_REG; ++t) {
            unsigned r = 0;
#pragma unroll
            for (int u = 0; u < OR_REG; ++u) r += __float_as_int(e[u].w) < __float_as_int(e[t].w) ? 1u : 0u;
            if (t < (int)n && r != (unsigned)t) sorted[s + r] = e[t];
        }
    }
    const int lane = threadIdx.x & 63;
    // runs of OR_REG+1 .. 64 points (dense cells: poles, wall feet): one element per lane of the owner's wave, the
    // other elements' indices come by readlane -- no dependent memory traffic either
    unsigned long long m64 = __ballot(n > OR_REG && n <= 64);
    while (m64) {  // wave-uniform
        const int owner = __builtin_ctzll(m64);
        m64 &= m64 - 1;
        const unsigned ss = __builtin_amdgcn_readlane(s, owner), nn = __builtin_amdgcn_readlane(n, owner);
        float4 el = make_float4(0.f, 0.f, 0.f, __int_as_float(0x7fffffff));
        if ((unsigned)lane < nn) el = sorted[ss + lane];
        const int my = __float_as_int(el.w);
        unsigned rk = 0;
        for (unsigned j = 0; j < nn; ++j) rk += __builtin_amdgcn_readlane(my, j) < my ? 1u : 0u;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        if ((unsigned)lane < nn && rk != (unsigned)lane) sorted[ss + rk] = el;
    }
    unsigned long long m = __ballot(n > 64 && n <= 512);
    while (m) {  // wave-uniform
        const int owner = __builtin_ctzll(m);
        m &= m - 1;
        const unsigned ss = __builtin_amdgcn_readlane(s, owner), nn = __builtin_amdgcn_readlane(n, owner);
        float4 el[8];
        unsigned rk[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const unsigned i = lane + 64 * t;
            el[t] = make_float4(0.f, 0.f, 0.f, 0.f);
            rk[t] = 0;
            if (i < nn) el[t] = sorted[ss + i];
        }
        for (unsigned j = 0; j < nn; ++j) {
            const int oj = __float_as_int(sorted[ss + j].w);
#pragma unroll
            for (int t = 0; t < 8; ++t) rk[t] += oj < __float_as_int(el[t].w) ? 1u : 0u;
        }
        // every rank is known before the first element moves (one wave, program order)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const unsigned i = lane + 64 * t;
            if (i < nn) sorted[ss + rk[t]] = el[t];
        }
    }
}

// ------------------------------------------------------------- host side


// Chooses the level ladder for one cloud from its bounding box.
static void plan_levels(const float mn[3], const float mx[3], int m, float cell, int max_levels, PairDesc &P)
{
    float ext[3], emax = 0.f;
    for (int a = 0; a < 3; ++a) {
        ext[a] = mx[a] - mn[a];
        if (!(ext[a] >= 0.f)) ext[a] = 0.f;
        emax = std::max(emax, ext[a]);
    }
    float amax = 0.f;
    for (int a = 0; a < 3; ++a) amax = std::max(amax, std::max(std::fabs(mn[a]), std::fabs(mx[a])));
    float h0 = cell;
    if (!(h0 > 0.f)) {
        // lidar clouds are surfaces: aim at ~3 points per occupied cell of the
        // footprint (the two largest extents); cell < 0: -cell points per cell instead of 3
        float e0 = ext[0], e1 = ext[1], e2 = ext[2];
        float area = std::max(e0 * e1, std::max(e0 * e2, e1 * e2));
        h0 = std::sqrt((cell < 0.f ? -cell : 3.0f) * area / (float)std::max(m, 1));
    }
    if (!(h0 > 0.f) || !std::isfinite(h0)) h0 = 1.0f;
    h0 = std::max(h0, emax / 1024.0f);
    // keep level 0 under 2^25 cells
    for (;;) {
        double nc = 1;
        for (int a = 0; a < 3; ++a) nc *= std::floor(ext[a] / h0) + 1.0;
        if (nc <= (double)(1 << 25)) break;
        h0 *= 1.26f;
    }
    float htop = std::max(emax * 0.5f * 1.0001f, h0);  // <= 2 cells per axis: always conclusive
    int L = 1;
    float ratio = 4.0f;
    if (htop > h0 && max_levels > 1) {
        double step = 2.5;  // nominal cell-size ratio between levels (measured: 2.5 best of 2 / 2.5 / 4)
        if (const char *e = getenv("GPSCAL_LEVEL_RATIO")) step = std::max(1.5, atof(e));
        L = 1 + (int)std::ceil(std::log(htop / h0) / std::log(step));
        if (L > max_levels) L = max_levels;
        if (L < 2) L = 2;
        ratio = std::pow(htop / h0, 1.0f / (float)(L - 1));
    }
    P.nlevels = L;
    float h = h0;
    for (int l = 0; l < L; ++l) {
        GridDesc &G = P.lv[l];
        if (l == L - 1 && L > 1) h = htop;
        G.ox = mn[0]; G.oy = mn[1]; G.oz = mn[2];
        G.h = h;
        G.inv_h = 1.0f / h;
        G.nx = (int)std::floor(ext[0] / h) + 1;
        G.ny = (int)std::floor(ext[1] / h) + 1;
        G.nz = (int)std::floor(ext[2] / h) + 1;
        // slack for the float rounding of cell assignment vs. face positions
        G.margin = 1e-4f * h + 16.0f * 1.1920929e-7f * (amax + emax);
        G.tile = max_levels == 1 ? 1 : 0;
        h *= ratio;
    }
}

// Grid sets for several clouds-of-clouds at once (the LOAM nodes index two to four of them per sweep): the
// bounding boxes of all of them come back in ONE read-back, which is the only point where the host waits --
// the grid dimensions are planned on the host.  Sources that name the SAME GridSet are concatenated into it (source
// k's pairs follow source k-1's; positions and cell numbers are the set's own, so a consumer addresses a source's
// pairs as set.pairs + its first pair and shares sorted / cell_start): four index sets then cost the launches of
// one.  Pooled sets (per-call sets of the LOAM chain) are not synchronised at the end either: their temporaries go
// back to the stream's block cache in stream order.
int build_grids_multi(gpscal_ctx *ctx, int nsrc, const GridSource *src, int stride, float cell, int max_levels)
{
    if (stride < 12) return fail(ctx, GPSCAL_EINVAL, "stride_bytes must be >= 12");
    struct Part {  // one distinct GridSet
        GridSet *gs = nullptr;
        std::vector<int> srcs;
        std::vector<long long> rel;   // npairs + 1 positions in the set's packed array
        std::vector<float> cells;     // per pair: its source's cell size hint
        DevBuf<long long> d_off;
        long long total = 0, bbox_at = 0;
        int npairs = 0, mmax = 0;
    };
    std::deque<Part> parts;  // (DevBuf members: no relocation)
    for (int k = 0; k < nsrc; ++k) {
        Part *W = nullptr;
        for (auto &q : parts)
            if (q.gs == src[k].gs) W = &q;
        if (!W) {
            parts.emplace_back();
            W = &parts.back();
            W->gs = src[k].gs;
            W->rel.push_back(0);
        }
        W->srcs.push_back(k);
        const long long *off = src[k].off;
        for (int b = 0; b < src[k].npairs; ++b) {
            const long long m = off[b + 1] - off[b];
            if (m < 0 || m > 0x7fffffff) return fail(ctx, GPSCAL_EINVAL, "bad offsets");
            W->rel.push_back(W->rel.back() + m);
            W->cells.push_back(src[k].cell != 0.f ? src[k].cell : cell);
            W->mmax = std::max(W->mmax, (int)m);
        }
        W->npairs += src[k].npairs;
    }
    long long nbox = 0;
    for (auto &W : parts) {
        W.total = W.rel.back();
        W.bbox_at = nbox;
        nbox += W.npairs;
    }
    DevBuf<int> d_bbox;
    GPSCAL_HIP(ctx, d_bbox.alloc_async((size_t)std::max<long long>(nbox, 1) * 6, ctx->stream));
    std::vector<int> hb((size_t)std::max<long long>(nbox, 1) * 6);
    for (long long b = 0; b < nbox; ++b)
        for (int a = 0; a < 3; ++a) {
            hb[b * 6 + a] = 0x7fffffff;
            hb[b * 6 + 3 + a] = (int)0x80000000;
        }
    GPSCAL_HIP(ctx, hipMemcpyAsync(d_bbox.p, hb.data(), sizeof(int) * hb.size(), hipMemcpyHostToDevice, ctx->stream));
    std::vector<InArg<char> > raws(nsrc);
    for (auto &W : parts) {
        GridSet &gs = *W.gs;
        gs.ctx = ctx;
        gs.npairs = W.npairs;
        gs.off = W.rel;
        GPSCAL_HIP(ctx, gs.pooled ? gs.pts4.alloc_async((size_t)W.total, ctx->stream) : gs.pts4.alloc((size_t)W.total));
        GPSCAL_HIP(ctx, W.d_off.alloc_async(W.npairs + 1, ctx->stream));
        GPSCAL_HIP(ctx, hipMemcpyAsync(W.d_off.p, W.rel.data(), sizeof(long long) * (W.npairs + 1), hipMemcpyHostToDevice,
                                       ctx->stream));
        int pb = 0;
        for (int k : W.srcs) {
            const long long *off = src[k].off;
            const long long tk = off[src[k].npairs] - off[0];
            GPSCAL_HIP(ctx, raws[k].bind(ctx, static_cast<const char *>(src[k].xyz) + (size_t)off[0] * stride,
                                         (size_t)tk * stride));
            if (tk > 0)
                hipLaunchKernelGGL(pack_points_kernel, dim3(div_up(tk, BLOCK)), dim3(BLOCK), 0, ctx->stream, raws[k].dev,
                                   stride, W.d_off.p + pb, src[k].npairs, tk, gs.pts4.p);
            pb += src[k].npairs;
        }
        // bounding boxes
        const int gx = std::max(1, std::min(div_up(W.mmax, BLOCK * BB_PT), 256));
        if (W.npairs > 0 && W.mmax > 0)
            hipLaunchKernelGGL(bbox_kernel, dim3(gx, W.npairs), dim3(BLOCK), 0, ctx->stream, gs.pts4.p, W.d_off.p,
                               d_bbox.p + W.bbox_at * 6);
    }
    GPSCAL_HIP(ctx, hipMemcpyAsync(hb.data(), d_bbox.p, sizeof(int) * hb.size(), hipMemcpyDeviceToHost, ctx->stream));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));

    bool all_pooled = true;
    for (auto &W : parts) {
        GridSet &gs = *W.gs;
        const int npairs = W.npairs;
        const std::vector<long long> &rel = W.rel;
        const int *hbk = hb.data() + W.bbox_at * 6;
        all_pooled = all_pooled && gs.pooled;
        gs.hpairs.assign(npairs, PairDesc{});
        long long cells = 0, sorted_total = 0;
        for (int b = 0; b < npairs; ++b) {
            PairDesc &P = gs.hpairs[b];
            P.tgt_off = rel[b];
            P.m = (int)(rel[b + 1] - rel[b]);
            if (P.m >= (1 << 27))  // scan_short addresses a level's points by signed 32-bit byte offsets
                return fail(ctx, GPSCAL_ERANGE, "cloud too large (2^27 points or more)");
            float mn[3], mx[3];
            for (int a = 0; a < 3; ++a) {
                mn[a] = ord2f(hbk[b * 6 + a]);
                mx[a] = ord2f(hbk[b * 6 + 3 + a]);
                if (!(mn[a] <= mx[a])) mn[a] = mx[a] = 0.f;  // empty / all-NaN cloud
            }
            plan_levels(mn, mx, P.m, W.cells[b], max_levels, P);
            // levels whose cells (counted from the top) fit the LDS histogram are aggregated there
            {
                long long acc = 0;
                P.coarse_from = P.nlevels;
                for (int l = P.nlevels - 1; l >= 0; --l) {
                    const GridDesc &G = P.lv[l];
                    acc += grid_cells(G);
                    if (acc > CO_MAX) break;
                    P.coarse_from = l;
                }
            }
            for (int l = 0; l < P.nlevels; ++l) {
                P.lv[l].cell_base = cells;
                const GridDesc &G = P.lv[l];
                cells += grid_cells(G);
                sorted_total += P.m;
            }
        }
        if (sorted_total >= (1ll << 32) - 1) return fail(ctx, GPSCAL_ERANGE, "batch too large for 32-bit cell offsets");
        gs.total_cells = cells;
        gs.total_sorted = sorted_total;
        GPSCAL_HIP(ctx, gs.pooled ? gs.pairs.alloc_async(npairs, ctx->stream) : gs.pairs.alloc(npairs));
        GPSCAL_HIP(ctx, hipMemcpyAsync(gs.pairs.p, gs.hpairs.data(), sizeof(PairDesc) * npairs, hipMemcpyHostToDevice,
                                       ctx->stream));
        DevBuf<unsigned> counts;
        GPSCAL_HIP(ctx, counts.alloc_async((size_t)cells + 1, ctx->stream));
        GPSCAL_HIP(ctx, gs.pooled ? gs.cell_start_buf.alloc_async((size_t)cells + 1 + 8, ctx->stream)
                                  : gs.cell_start_buf.alloc((size_t)cells + 1 + 8));
        GPSCAL_HIP(ctx, hipMemsetAsync(gs.cell_start_buf.p, 0, sizeof(unsigned) * ((size_t)cells + 9), ctx->stream));
        gs.cell_start = gs.cell_start_buf.p + 4;
        // (+4: scan_short reads up to two entries behind a run's end without looking at them)
        GPSCAL_HIP(ctx, gs.pooled ? gs.sorted.alloc_async((size_t)sorted_total + 4, ctx->stream) : gs.sorted.alloc((size_t)sorted_total + 4));
        GPSCAL_HIP(ctx, hipMemsetAsync(counts.p, 0, sizeof(unsigned) * ((size_t)cells + 1), ctx->stream));
        if (npairs > 0 && W.mmax > 0) {
            int maxlev = 1;
            for (auto &P : gs.hpairs) maxlev = std::max(maxlev, P.nlevels);
            DevBuf<unsigned> ranks;
            GPSCAL_HIP(ctx, ranks.alloc_async((size_t)W.total * maxlev, ctx->stream));
            hipLaunchKernelGGL(grid_count_kernel, dim3(div_up(W.mmax, GC_CHUNK), npairs), dim3(BLOCK), 0, ctx->stream,
                               gs.pairs.p, gs.pts4.p, counts.p, ranks.p, W.total);
            int rc = exclusive_scan(ctx, counts.p, gs.cell_start, cells + 1);
            if (rc) return rc;
            int gxf = std::max(1, std::min(div_up(W.mmax, BLOCK), 1024));
            hipLaunchKernelGGL(grid_scatter_kernel, dim3(gxf, npairs), dim3(BLOCK), 0, ctx->stream, gs.pairs.p,
                               gs.pts4.p, ranks.p, gs.cell_start, gs.sorted.p, W.total);
            GPSCAL_HIP(ctx, hipGetLastError());
        } else {
            GPSCAL_HIP(ctx, hipMemsetAsync(gs.cell_start, 0, sizeof(unsigned) * ((size_t)cells + 1), ctx->stream));
        }
        GPSCAL_HIP(ctx, hipGetLastError());
    }
    // a set with plain allocations is handed to callers that may use it from another stream
    if (!all_pooled) GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}

int build_grids(gpscal_ctx *ctx, const void *xyz, int stride, const long long *off, int npairs, float cell,
                int max_levels, GridSet &gs)
{
    const GridSource one = {xyz, off, npairs, &gs, 0.f};
    return build_grids_multi(ctx, 1, &one, stride, cell, max_levels);
}

// Neighbour lists and certified radii of every target point (ICP only; first use).
int ensure_safe_radius(gpscal_ctx *ctx, GridSet &gs, hipStream_t side, hipEvent_t done)
{
    if (gs.pt_r2.p) return GPSCAL_OK;
    const long long total = gs.off[gs.npairs] - gs.off[0];
    const size_t slots = (size_t)std::max<long long>(total, 1);
    GPSCAL_HIP(ctx, gs.pooled ? gs.nbr.alloc_async(slots * 4, ctx->stream) : gs.nbr.alloc(slots * 4));
    GPSCAL_HIP(ctx, gs.pooled ? gs.pt_r2.alloc_async(slots, ctx->stream) : gs.pt_r2.alloc(slots));
    // (non-finite points are never indexed nor returned, so nobody reads their slots: no clearing pass -- it was 50 us
    // of a 64 x 65 536 build for 300 MB that the kernel below overwrites)
    int mmax = 0;
    for (auto &P : gs.hpairs) mmax = std::max(mmax, P.m);
    hipStream_t st = ctx->stream;
    if (side && done) {  // beside the caller's next work: the side stream picks up behind the grids
        GPSCAL_HIP(ctx, hipEventRecord(done, ctx->stream));
        GPSCAL_HIP(ctx, hipStreamWaitEvent(side, done, 0));
        st = side;
    }
    if (mmax > 0) {
        int gx = std::max(1, std::min(div_up(mmax, BLOCK), 4096));
        hipLaunchKernelGGL(self_nn_kernel, dim3(gx, gs.npairs), dim3(BLOCK), 0, st, gs.pairs.p, gs.sorted.p,
                           gs.cell_start, gs.pts4.p, gs.nbr.p, gs.pt_r2.p);
    }
    if (side && done) GPSCAL_HIP(ctx, hipEventRecord(done, side));
    GPSCAL_HIP(ctx, hipGetLastError());
    // a set with plain allocations may be used from another stream next; a pooled one lives on this stream
    if (!gs.pooled) GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}

}  // namespace gpscal

using namespace gpscal;


struct gpscal_scan_batch {
    gpscal_ctx *ctx = nullptr;
    GridSet *tgt = nullptr;  // owned unless borrowed
    bool borrowed = false;
    int npairs = 0;
    long long total_n = 0;
    bool weighted = false;
    int qpt = 1, nblk = 0, diag = 0;
    int step_block = 128;  // threads per workgroup of icp_step_kernel (128, 256 or 512, by batch size)
    bool flat = false;     // small batch: the step's grid search walks per-lane row lists (latency-bound launches)
    bool persistent = false;  // small batch: all iterations of a run in ONE launch (icp_persistent_kernel)
    DevBuf<IcpCtl> ctl;
    int uni_n = 0, uni_m = 0, uni_bpp = 0;  // equal-sized scans stored back to back: workgroup slices by arithmetic
    int ball_r = 0;  // block radius of the ball search (0 = fine -> coarse 3x3x3 search)
    DevBuf<PairDesc> pairs;  // target descs + source fields
    std::vector<PairDesc> hpairs;
    DevBuf<float4> src4;  // the grouped source as the grouping leaves it (released once split)
    DevBuf<float> src3;   // ... as the step kernel reads it: xyz, 12 bytes per point
    DevBuf<int> src_orig;  // ... and every grouped point's original index
    DevBuf<double> wsorted;
    DevBuf<int> blk_pair, blk_first;
    DevBuf<int> nn_idx;
    DevBuf<float> nn_sqd;
    DevBuf<float4> warm_q;  // warm start: last iteration's neighbour (xyz + radii word) per source point
    DevBuf<int> warm_i;  // ... its index; warm_q.w holds the certified radii (r_a^2, r_b^2), 16 bits each (truncated floats)
    DevBuf<double> partials, pose64, err_hist;
    DevBuf<float> pose32;
    int err_cap = 0;
    // captured graphs by iteration count (callers that alternate between two counts keep both)
    static constexpr int NGRAPH = 4;
    hipGraphExec_t graphs[NGRAPH] = {};
    long long graph_iters[NGRAPH] = {};  // key: iterations, error history wanted
    unsigned graph_used[NGRAPH] = {}, graph_clock = 0;
    void drop_graphs()
    {
        for (int k = 0; k < NGRAPH; ++k)
            if (graphs[k]) {
                (void)hipGraphExecDestroy(graphs[k]);
                graphs[k] = nullptr;
            }
    }
    double build_seconds = 0.0;
    // independent step -> solve chains of the captured graph (launch_step)
    static constexpr int MAX_CHAINS = 8;
    int nchains = 1;
    int chain_pair[MAX_CHAINS + 1] = {}, chain_blk[MAX_CHAINS + 1] = {};
    hipStream_t chain_stream[MAX_CHAINS] = {};  // the context's side streams (not owned)
    hipEvent_t chain_ev[MAX_CHAINS] = {};
    ~gpscal_scan_batch()
    {
        drop_graphs();
        if (tgt && !borrowed) delete tgt;
    }
};

// ------------------------------------------------------------ k-NN C ABI

extern "C" int gpscal_knn_build(gpscal_ctx *ctx, const float *xyz, int m, int stride_bytes, float cell,
                                gpscal_knn_index **index)
{
    if (!ctx || !index || m < 0 || (!xyz && m > 0)) return fail(ctx, GPSCAL_EINVAL, "gpscal_knn_build: bad argument");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    auto *ix = new gpscal_knn_index;
    ix->ctx = ctx;
    long long off[2] = {0, m};
    int rc = build_grids(ctx, xyz, stride_bytes, off, 1, cell, MAX_LEVELS, ix->gs);
    if (rc) {
        delete ix;
        return rc;
    }
    *index = ix;
    return GPSCAL_OK;
}

extern "C" int gpscal_knn_free(gpscal_knn_index *index)
{
    if (!index) return GPSCAL_EINVAL;
    (void)hipSetDevice(index->ctx->device);
    delete index;
    return GPSCAL_OK;
}

template <int K>
static void launch_search(gpscal_ctx *ctx, GridSet &gs, const char *q, int stride, int n, int *idx, float *sqd)
{
    hipLaunchKernelGGL(knn_search_kernel<K>, dim3(div_up(n, BLOCK)), dim3(BLOCK), 0, ctx->stream, gs.pairs.p,
                       gs.sorted.p, gs.cell_start, q, stride, n, idx, sqd);
}

extern "C" int gpscal_knn_search(gpscal_knn_index *index, const float *query, int n, int stride_bytes, int k,
                                 int32_t *idx, float *sqd)
{
    if (!index) return GPSCAL_EINVAL;
    gpscal_ctx *ctx = index->ctx;
    if (n < 0 || k < 1 || k > 8 || stride_bytes < 12 || (n > 0 && (!query || !idx || !sqd)))
        return fail(ctx, GPSCAL_EINVAL, "gpscal_knn_search: bad argument");
    if (n == 0) return GPSCAL_OK;
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    InArg<char> q;
    OutArg<int> oi;
    OutArg<float> od;
    GPSCAL_HIP(ctx, q.bind(ctx, reinterpret_cast<const char *>(query), (size_t)n * stride_bytes));
    GPSCAL_HIP(ctx, oi.bind(ctx, idx, (size_t)n * k));
    GPSCAL_HIP(ctx, od.bind(ctx, sqd, (size_t)n * k));
    switch (k) {
    case 1: launch_search<1>(ctx, index->gs, q.dev, stride_bytes, n, oi.dev, od.dev); break;
    case 2: launch_search<2>(ctx, index->gs, q.dev, stride_bytes, n, oi.dev, od.dev); break;
    case 3: launch_search<3>(ctx, index->gs, q.dev, stride_bytes, n, oi.dev, od.dev); break;
    case 4: launch_search<4>(ctx, index->gs, q.dev, stride_bytes, n, oi.dev, od.dev); break;
    case 5: launch_search<5>(ctx, index->gs, q.dev, stride_bytes, n, oi.dev, od.dev); break;
    case 6: launch_search<6>(ctx, index->gs, q.dev, stride_bytes, n, oi.dev, od.dev); break;
    case 7: launch_search<7>(ctx, index->gs, q.dev, stride_bytes, n, oi.dev, od.dev); break;
    default: launch_search<8>(ctx, index->gs, q.dev, stride_bytes, n, oi.dev, od.dev); break;
    }
    GPSCAL_HIP(ctx, hipGetLastError());
    bool sync = q.tmp.p != nullptr;  // staged input must outlive the kernel
    GPSCAL_HIP(ctx, oi.commit(ctx, &sync));
    GPSCAL_HIP(ctx, od.commit(ctx, &sync));
    if (sync) GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}

// ------------------------------------------------------- scan batch C ABI

static int batch_setup_sources(gpscal_scan_batch *B, const float *src_xyz, int stride, const long long *src_off,
                               const double *w)
{
    gpscal_ctx *ctx = B->ctx;
    const int np = B->npairs;
    // spatially group each source cloud by its own level-0 cells
    GridSet sg;
    sg.pooled = true;  // per-call: blocks of the stream's cache (src4 below takes one of them over)
    float src_cell = 0.f;  // (0: three points per cell of the footprint; GPSCAL_SRC_CELL=-12: twelve)
    if (const char *e = getenv("GPSCAL_SRC_CELL")) src_cell = (float)atof(e);
    int rc = build_grids(ctx, src_xyz, stride, src_off, np, src_cell, 1, sg);
    if (rc) return rc;
    B->total_n = sg.total_sorted;
    // steal the grouped array: with one level `sorted` is exactly src4, except
    // that non-finite points were dropped -- re-pack those at the tail.
    B->hpairs = B->tgt->hpairs;
    long long maxn = 0;
    for (int b = 0; b < np; ++b) {
        B->hpairs[b].src_off = sg.hpairs[b].tgt_off;
        B->hpairs[b].n = sg.hpairs[b].m;
        maxn = std::max<long long>(maxn, sg.hpairs[b].m);
    }
    // original-index order inside every cell: the grouping, and so the order of the sums, is the same in
    // every build
    if (sg.total_cells > 0)
        hipLaunchKernelGGL(order_runs_kernel, dim3(div_up(sg.total_cells, BLOCK)), dim3(BLOCK), 0, ctx->stream,
                           sg.sorted.p, sg.cell_start, sg.total_cells);
    // every source point must be finite: the grouped array then holds exactly sum(n) points (a pair's
    // share can only shrink, so one total decides it)
    unsigned grouped = 0;
    GPSCAL_HIP(ctx, hipMemcpyAsync(&grouped, sg.cell_start + sg.total_cells, sizeof(unsigned), hipMemcpyDeviceToHost,
                                   ctx->stream));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if ((long long)grouped != B->total_n)
        return fail(ctx, GPSCAL_EINVAL, "source cloud contains non-finite points (remove NaNs first, cf. scanRegistration.cpp:260-263)");
    // the grouped array IS src4 (one level): take the buffer over instead of copying it
    std::swap(B->src4.p, sg.sorted.p);
    std::swap(B->src4.n, sg.sorted.n);
    std::swap(B->src4.pooled, sg.sorted.pooled);
    std::swap(B->src4.pool_stream, sg.sorted.pool_stream);
    // ... and is split into what the iterations read (12 bytes per point) and the original indices
    GPSCAL_HIP(ctx, B->src3.alloc_async((size_t)std::max<long long>(3 * B->total_n, 1), ctx->stream));
    GPSCAL_HIP(ctx, B->src_orig.alloc_async((size_t)std::max<long long>(B->total_n, 1), ctx->stream));
    if (B->total_n > 0)
        hipLaunchKernelGGL(split_source_kernel, dim3(div_up(B->total_n, BLOCK)), dim3(BLOCK), 0, ctx->stream, B->src4.p,
                           B->src3.p, B->src_orig.p, B->total_n);
    B->src4.release();
    // block table
    const long long per_blk_target = (long long)ctx->prop.multiProcessorCount * 8 * BLOCK;
    (void)per_blk_target;
    B->qpt = 1;  // measured: QPT 4 costs 117 VGPRs (4 waves/SIMD) and loses in the search-heavy early iterations
    if (const char *e = getenv("GPSCAL_QPT")) B->qpt = atoi(e) == 4 ? 4 : 1;  // tuning knobs
    if (const char *e = getenv("GPSCAL_DIAG")) B->diag = atoi(e);  // ablation (wrong results!)
    {
        // Dense clouds (level-0 cells under 0.25 m, i.e. ~1M points on a 100 m scene): a query that is still
        // decimetres from the surface lies many cells away from it, and the 3x3x3 rule sends it to a level whose
        // cells hold hundreds of points.  Measured (4 pairs x 1M points, 50 iterations): 942 -> 655 us per
        // iteration with R = 4; neutral at 256k points (h0 0.34 m), 5 % slower at 64k (h0 0.68 m).
        double hs = 0.0;
        int cnt = 0;
        for (int b = 0; b < np; ++b)
            if (B->hpairs[b].m > 0) {
                hs += B->hpairs[b].lv[0].h;
                ++cnt;
            }
        if (cnt > 0 && hs / cnt < 0.25) B->ball_r = 4;
    }
    if (const char *e = getenv("GPSCAL_BALL_R")) B->ball_r = std::min(std::max(atoi(e), 0), 8);
    // Workgroup size and row walk of the step kernel by batch size (tools/small_batch_probe.py, k iterations/s with
    // 128 / 256 / 512 threads; "flat" = the per-lane row lists of block3_level_flat):
    //   1 scan pair of 65 536 points    37.0 / 39.8 / 40.0, flat 40.1 / 43.2 / 43.4
    //   2 pairs                          73.3 / 78.0 / 78.7, flat 78.4 / 84.4 / 86.3
    //   8 pairs                          241 / 255 / 256,    flat 246 / 259 / 263
    //   16 pairs                         432 / 450 / 418,    flat 415 / 415 / 401
    //   32 pairs                         645 / 703 / 680,    flat 643 / 485 / 597
    //   64 pairs                         128 and 256 threads within the replayed graph's run-to-run spread
    // Up to 8 scans a launch is about one wave per SIMD and as long as its longest chain of dependent loads: large
    // workgroups (fewer partial sums on the solve kernel's path) and flat row lists (a wave pays its longest lane's
    // list, not the sum over the rows of the longest run in each); from 16 scans on the launches are throughput-bound
    // and the pruned row-by-row walk wins.
    B->step_block = B->total_n <= 524288 ? 512 : (B->total_n < 3ll * (1 << 20) ? 256 : 128);
    B->flat = B->total_n <= 524288;
    if (const char *e = getenv("GPSCAL_ICP_PERSISTENT"))
        if (atoi(e) != 0) B->step_block = std::min(B->step_block, PERSIST_BLOCK);  // (that kernel is built for 256 threads)
    if (const char *e = getenv("GPSCAL_STEP_BLOCK")) B->step_block = atoi(e) == 512 ? 512 : (atoi(e) == 256 ? 256 : 128);
    if (const char *e = getenv("GPSCAL_STEP_FLAT")) B->flat = atoi(e) != 0;
    std::vector<int> bp, bf;
    for (int b = 0; b < np; ++b) {
        PairDesc &P = B->hpairs[b];
        P.pblk_off = (int)bp.size();
        int per = B->step_block * B->qpt;
        for (int f = 0; f < P.n; f += per) {
            bp.push_back(b);
            bf.push_back(f);
        }
        P.pblk_cnt = (int)bp.size() - P.pblk_off;
    }
    B->nblk = (int)bp.size();
    // GPSCAL_ICP_PERSISTENT=1: a batch whose step grid is resident all at once runs the iterations of a run in ONE
    // launch (icp_persistent_kernel) instead of the captured graph.  Off by default: measured on one 65 536-point
    // pair it is no faster (33.3 k against 35.1 k iterations/s) -- the run is the sum of the step's own dependent
    // search chains (1.03 ms of the 1.5 ms), and the in-kernel hand-off + solve (9.4 us per iteration) costs what
    // the two launches cost.
    B->persistent = false;
    if (const char *e = getenv("GPSCAL_ICP_PERSISTENT"))
        B->persistent = atoi(e) != 0 && B->qpt == 1 && B->step_block == PERSIST_BLOCK && B->nblk >= 1 &&
                        B->nblk <= 2 * ctx->prop.multiProcessorCount;
    // equal-sized scans stored back to back (the usual batch): the step kernel finds a workgroup's slice by arithmetic
    {
        bool uni = np > 0 && B->hpairs[0].n > 0;
        for (int b = 0; b < np && uni; ++b) {
            const PairDesc &P = B->hpairs[b];
            uni = P.n == B->hpairs[0].n && P.m == B->hpairs[0].m && P.src_off == (long long)b * B->hpairs[0].n &&
                  P.tgt_off == (long long)b * B->hpairs[0].m;
        }
        if (const char *e = getenv("GPSCAL_ICP_UNIFORM")) uni = uni && atoi(e) != 0;
        B->uni_n = uni ? B->hpairs[0].n : 0;
        B->uni_m = uni ? B->hpairs[0].m : 0;
        B->uni_bpp = uni ? B->hpairs[0].pblk_cnt : 0;
    }
    // chains: contiguous groups of pairs with (nearly) equal block counts; small batches keep one
    {
        int want = np >= 32 ? 4 : (np >= 8 ? 2 : 1);  // measured at 64 pairs x 65 536 points: 1 / 2 / 4 chains = 671 / 729 / 748 k iterations/s
        if (const char *e = getenv("GPSCAL_ICP_CHAINS")) want = std::min(std::max(atoi(e), 1), (int)gpscal_scan_batch::MAX_CHAINS);
        want = std::min(want, std::max(np, 1));
        B->nchains = want;
        B->chain_pair[0] = 0;
        B->chain_blk[0] = 0;
        int p = 0;
        for (int c = 1; c <= want; ++c) {
            const long long target = (long long)B->nblk * c / want;
            while (p < np && (c == want || B->hpairs[p].pblk_off + B->hpairs[p].pblk_cnt <= target)) ++p;
            B->chain_pair[c] = c == want ? np : p;
            B->chain_blk[c] = c == want ? B->nblk : (p < np ? B->hpairs[p].pblk_off : B->nblk);
        }
        static_assert(gpscal_scan_batch::MAX_CHAINS <= gpscal_ctx::MAX_SIDE, "side streams");
        for (int c = 1; c < want; ++c) {  // the context's side streams: created once, on first use
            GPSCAL_HIP(ctx, side_stream_of(ctx, c, &B->chain_stream[c], &B->chain_ev[c]));
        }
    }
    GPSCAL_HIP(ctx, B->blk_pair.alloc_async(bp.size(), ctx->stream));
    GPSCAL_HIP(ctx, B->blk_first.alloc_async(bf.size(), ctx->stream));
    if (!bp.empty()) {
        GPSCAL_HIP(ctx, hipMemcpyAsync(B->blk_pair.p, bp.data(), sizeof(int) * bp.size(), hipMemcpyHostToDevice, ctx->stream));
        GPSCAL_HIP(ctx, hipMemcpyAsync(B->blk_first.p, bf.data(), sizeof(int) * bf.size(), hipMemcpyHostToDevice, ctx->stream));
    }
    GPSCAL_HIP(ctx, B->pairs.alloc_async(np, ctx->stream));
    GPSCAL_HIP(ctx, hipMemcpyAsync(B->pairs.p, B->hpairs.data(), sizeof(PairDesc) * np, hipMemcpyHostToDevice, ctx->stream));
    B->weighted = w != nullptr;
    if (w) {
        InArg<double> win;
        GPSCAL_HIP(ctx, win.bind(ctx, w, (size_t)B->total_n));
        GPSCAL_HIP(ctx, B->wsorted.alloc_async((size_t)B->total_n, ctx->stream));
        int gx = std::max(1, std::min(div_up(maxn, BLOCK), 1024));
        hipLaunchKernelGGL(gather_weights_kernel, dim3(gx, np), dim3(BLOCK), 0, ctx->stream, B->pairs.p, B->src_orig.p,
                           win.dev, B->wsorted.p);
        GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    GPSCAL_HIP(ctx, B->nn_idx.alloc_async((size_t)std::max<long long>(B->total_n, 1), ctx->stream));
    GPSCAL_HIP(ctx, B->nn_sqd.alloc_async((size_t)std::max<long long>(B->total_n, 1), ctx->stream));
    GPSCAL_HIP(ctx, B->warm_q.alloc_async((size_t)std::max<long long>(B->total_n, 1), ctx->stream));
    GPSCAL_HIP(ctx, B->warm_i.alloc_async((size_t)std::max<long long>(B->total_n, 1), ctx->stream));
    // (filled by gpscal_scan_batch_set_pose, which gpscal_scan_batch_create calls last)
    GPSCAL_HIP(ctx, B->partials.alloc_async((size_t)std::max(B->nblk, 1) * NACC_WEIGHTED, ctx->stream));
    if (B->persistent) GPSCAL_HIP(ctx, B->ctl.alloc_async((size_t)np, ctx->stream));
    GPSCAL_HIP(ctx, B->pose64.alloc_async((size_t)np * 16, ctx->stream));
    GPSCAL_HIP(ctx, B->pose32.alloc_async((size_t)np * 12, ctx->stream));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return gpscal_scan_batch_set_pose(B, nullptr);
}

extern "C" int gpscal_scan_batch_create(gpscal_ctx *ctx, int npairs, const float *tgt_xyz, const int64_t *tgt_off,
                                        const float *src_xyz, const int64_t *src_off, const double *w, float cell,
                                        gpscal_scan_batch **batch)
{
    if (!ctx || !batch || npairs < 1 || !tgt_off || !src_off || !tgt_xyz || !src_xyz)
        return fail(ctx, GPSCAL_EINVAL, "gpscal_scan_batch_create: bad argument");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    auto t0 = std::chrono::steady_clock::now();
    auto *B = new gpscal_scan_batch;
    B->ctx = ctx;
    B->npairs = npairs;
    B->tgt = new GridSet;
    // a batch's own index and state are blocks of the stream's cache: a stream of batches of one shape (one per
    // sweep) then allocates nothing after the first (a hipMalloc of these sizes costs ~0.2 ms, sixteen of them
    // were a third of the build); the batch is used on the context's streams only and destroyed after a sync
    B->tgt->pooled = true;
    std::vector<long long> to(tgt_off, tgt_off + npairs + 1), so(src_off, src_off + npairs + 1);
    auto lap = [&](const char *what, std::chrono::steady_clock::time_point &t) {
        if (getenv("GPSCAL_BUILD_TIMING")) {
            auto n = std::chrono::steady_clock::now();
            fprintf(stderr, "scan batch build: %s %.3f ms\n", what, 1e3 * std::chrono::duration<double>(n - t).count());
            t = n;
        }
    };
    // The neighbour lists of the target (self_nn_kernel, a third of the build) and the grouping of the sources do not
    // depend on each other: the kernel runs on a side stream beside the sources' counting sort, whose short kernels and
    // host round trips it hides (2.64 -> 2.45 ms at 64 x 65 536); the context's stream waits for it at the end, inside
    // the measured build time.  (Measured as well: the source grouping on a third stream from the start -- no further
    // gain, the large kernels of the three chains share the machine and each runs as much slower.)
    hipStream_t side = nullptr;
    hipEvent_t side_done = nullptr;
    if (!getenv("GPSCAL_BUILD_SERIAL")) GPSCAL_HIP(ctx, side_stream_of(ctx, 1, &side, &side_done));
    auto tl = t0;
    int rc = build_grids(ctx, tgt_xyz, 12, to.data(), npairs, cell, MAX_LEVELS, *B->tgt);
    lap("target grids", tl);
    if (!rc) rc = ensure_safe_radius(ctx, *B->tgt, side, side_done);
    lap("neighbour lists + radii", tl);
    if (!rc) rc = batch_setup_sources(B, src_xyz, 12, so.data(), w);
    lap("source grouping + state", tl);
    if (side) {
        (void)hipStreamWaitEvent(ctx->stream, side_done, 0);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (rc) {
        if (side) (void)hipStreamSynchronize(side);  // the kernel may still read the set that is about to go
        delete B;
        return rc;
    }
    B->build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *batch = B;
    return GPSCAL_OK;
}

extern "C" int gpscal_scan_batch_set_pose(gpscal_scan_batch *B, const double *T0)
{
    if (!B) return GPSCAL_EINVAL;
    gpscal_ctx *ctx = B->ctx;
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    const int np = B->npairs;
    if (T0 && is_device_ptr(T0)) {
        GPSCAL_HIP(ctx, hipMemcpyAsync(B->pose64.p, T0, sizeof(double) * 16 * np, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        std::vector<double> h((size_t)np * 16, 0.0);
        for (int b = 0; b < np; ++b)
            for (int k = 0; k < 16; ++k) h[(size_t)b * 16 + k] = T0 ? T0[(size_t)b * 16 + k] : (k % 5 == 0 ? 1.0 : 0.0);
        GPSCAL_HIP(ctx, hipMemcpyAsync(B->pose64.p, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice, ctx->stream));
        GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    hipLaunchKernelGGL(pose_to_f32_kernel, dim3(div_up(np * 12, BLOCK)), dim3(BLOCK), 0, ctx->stream, B->pose64.p,
                       B->pose32.p, np);
    // a new pose starts a new problem: the neighbours remembered from the previous run are forgotten, so
    // that a run never profits from correspondences an earlier run computed
    hipLaunchKernelGGL(fill_warm_kernel, dim3(div_up(std::max<long long>(B->total_n, 1), BLOCK)), dim3(BLOCK), 0,
                       ctx->stream, B->warm_q.p, B->warm_i.p, B->total_n);
    GPSCAL_HIP(ctx, hipGetLastError());
    return GPSCAL_OK;
}

#ifdef GPSCAL_STATS
__global__ void stat_set_iter_kernel(int it) { g_stat_iter = it; }
// instrumented builds only (tools/search_stats.py): the counters of every iteration, then cleared
extern "C" int gpscal_debug_stats(unsigned long long *out, int n)
{
    std::vector<unsigned long long> h(NSTAT * STAT_ITERS, 0ull);
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stats), h.size() * 8) != hipSuccess) return GPSCAL_EHIP;
    for (int i = 0; i < n && i < (int)h.size(); ++i) out[i] = h[i];
    std::fill(h.begin(), h.end(), 0ull);
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stats), h.data(), h.size() * 8) != hipSuccess) return GPSCAL_EHIP;
    return GPSCAL_OK;
}
#endif

// The pairs of a batch form `nchains` groups (contiguous halves ...), each with its own dependent chain of
// step -> solve -> step ...: while one group's solve kernel (one wave per pair, ~9 us of dependent float64
// arithmetic) runs, the other groups' step kernels keep the chip busy.  Chain c = pairs [chain_pair[c],
// chain_pair[c+1]) = blocks [chain_blk[c], chain_blk[c+1]); partial-sum slots stay global.
static void launch_step(gpscal_scan_batch *B, bool last, int c, hipStream_t st, bool want_err = true)
{
    GridSet &G = *B->tgt;
    // c < 0: the whole batch in one launch (profiling mode: the launch the roofline is quoted for)
    const int b0 = c < 0 ? 0 : B->chain_blk[c], nb = (c < 0 ? B->nblk : B->chain_blk[c + 1]) - b0;
    if (nb <= 0) return;
#define STEP(QPT, W, BALL)                                      \
    do {                                                        \
        if (B->step_block == 512) STEP_BS(QPT, W, BALL, 512);   \
        else if (B->step_block == 256) STEP_BS(QPT, W, BALL, 256); \
        else STEP_BS(QPT, W, BALL, 128);                        \
    } while (0)
#define STEP_BS(QPT, W, BALL, BS)                                         \
    do {                                                                  \
        if (QPT == 1 && B->flat) STEP_K(1, W, BALL, BS, true);            \
        else STEP_K(QPT, W, BALL, BS, false);                             \
    } while (0)
#define STEP_K(QPT, W, BALL, BS, FLAT)                                                                                   \
    hipLaunchKernelGGL((icp_step_kernel<QPT, W, BALL, BS, FLAT>), dim3(nb), dim3(BS), 0, st, B->pairs.p, B->blk_pair.p + b0, \
                       B->blk_first.p + b0, B->src3.p, B->wsorted.p, G.sorted.p, G.nbr.p, G.pt_r2.p, G.cell_start,   \
                       B->pose32.p, B->nn_idx.p, B->nn_sqd.p, B->warm_q.p, B->warm_i.p,                               \
                       B->partials.p + (size_t)b0 * (B->weighted ? NACC_WEIGHTED : NACC_PLAIN), nb,                  \
                       (B->diag & 0xff) | (B->ball_r << 8), (last ? 1 : 0) | (want_err ? 0 : 2), B->uni_n, B->uni_m,   \
                       B->uni_bpp, c < 0 ? 0 : B->chain_pair[c])
    // the ball search costs the kernel a wave of occupancy: its own instantiation, chosen per batch
    const bool ball = B->ball_r > 0;
    if (B->weighted) {
        if (B->qpt == 4) STEP(4, true, false); else if (ball) STEP(1, true, true); else STEP(1, true, false);
    } else {
        if (B->qpt == 4) STEP(4, false, false); else if (ball) STEP(1, false, true); else STEP(1, false, false);
    }
#undef STEP
#undef STEP_BS
#undef STEP_K
}

static void launch_solve(gpscal_scan_batch *B, int it, int c, hipStream_t st)
{
    const int p0 = c < 0 ? 0 : B->chain_pair[c], np = (c < 0 ? B->npairs : B->chain_pair[c + 1]) - p0;
    if (np <= 0) return;
    if (B->weighted)
        hipLaunchKernelGGL(icp_solve_kernel<true>, dim3(np), dim3(64), 0, st, B->pairs.p, B->partials.p, B->pose64.p,
                           B->pose32.p, B->err_hist.p, it, B->err_cap, p0);
    else
        hipLaunchKernelGGL(icp_solve_kernel<false>, dim3(np), dim3(64), 0, st, B->pairs.p, B->partials.p, B->pose64.p,
                           B->pose32.p, B->err_hist.p, it, B->err_cap, p0);
}

extern "C" int gpscal_scan_batch_icp(gpscal_scan_batch *B, int iters, double *T_out, double *mean_err,
                                     float *step_ms)
{
    if (!B || iters < 0) return GPSCAL_EINVAL;
    gpscal_ctx *ctx = B->ctx;
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    const int np = B->npairs;
    const bool want_err = mean_err != nullptr;  // without it the step kernel skips the distance sum (a float64 sqrt per query)
    if (iters > B->err_cap) {
        B->drop_graphs();  // they hold the old error-history pointer
        GPSCAL_HIP(ctx, B->err_hist.alloc_async((size_t)np * iters, ctx->stream));
        B->err_cap = iters;
    }
    if (step_ms) {
        // profiling mode: every correspondence launch bracketed by HIP events
        std::vector<hipEvent_t> ev((size_t)iters * 2);
        for (auto &e : ev) GPSCAL_HIP(ctx, hipEventCreate(&e));
        for (int it = 0; it < iters; ++it) {
#ifdef GPSCAL_STATS
            hipLaunchKernelGGL(stat_set_iter_kernel, dim3(1), dim3(1), 0, ctx->stream, std::min(it, STAT_ITERS - 1));
#endif
            GPSCAL_HIP(ctx, hipEventRecord(ev[2 * it], ctx->stream));
            launch_step(B, it == iters - 1, -1, ctx->stream, want_err);  // whole batch (the launch the roofline is quoted for)
            GPSCAL_HIP(ctx, hipEventRecord(ev[2 * it + 1], ctx->stream));
            launch_solve(B, it, -1, ctx->stream);
        }
        GPSCAL_HIP(ctx, hipGetLastError());
        GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        for (int it = 0; it < iters; ++it) GPSCAL_HIP(ctx, hipEventElapsedTime(&step_ms[it], ev[2 * it], ev[2 * it + 1]));
        for (auto &e : ev) (void)hipEventDestroy(e);
    } else if (iters > 0 && B->persistent) {
        GridSet &G = *B->tgt;
        GPSCAL_HIP(ctx, hipMemsetAsync(B->ctl.p, 0, sizeof(IcpCtl) * np, ctx->stream));
        if (want_err)  // (a pair without source points has no workgroup: its history reads zero, as the solve kernel leaves it)
            GPSCAL_HIP(ctx, hipMemsetAsync(B->err_hist.p, 0, sizeof(double) * (size_t)np * B->err_cap, ctx->stream));
        const StepArgs A = {B->src3.p, B->wsorted.p, G.sorted.p, G.nbr.p, G.pt_r2.p, G.cell_start,
                            B->nn_idx.p, B->nn_sqd.p, B->warm_q.p, B->warm_i.p};
#define PERSIST(W, BALL)                                                                                            \
    hipLaunchKernelGGL((icp_persistent_kernel<W, BALL>), dim3(B->nblk), dim3(PERSIST_BLOCK), 0, ctx->stream,         \
                       B->pairs.p, B->blk_pair.p, B->blk_first.p, A, B->partials.p, B->pose64.p, B->pose32.p,        \
                       B->err_hist.p, B->ctl.p, B->nblk, iters, B->err_cap, want_err ? 1 : 0, B->ball_r)
        const bool ball = B->ball_r > 0;
        if (B->weighted) {
            if (ball) PERSIST(true, true); else PERSIST(true, false);
        } else {
            if (ball) PERSIST(false, true); else PERSIST(false, false);
        }
#undef PERSIST
        GPSCAL_HIP(ctx, hipGetLastError());
    } else if (iters > 0) {
        // a graph per (iteration count, error history wanted)
        const long long gkey = (long long)iters * 2 + (want_err ? 1 : 0);
        int slot = -1;
        for (int k = 0; k < gpscal_scan_batch::NGRAPH; ++k)
            if (B->graphs[k] && B->graph_iters[k] == gkey) slot = k;
        if (slot < 0) {
            slot = 0;  // an empty slot, else the least recently used one
            for (int k = 0; k < gpscal_scan_batch::NGRAPH; ++k) {
                if (!B->graphs[k]) {
                    slot = k;
                    break;
                }
                if (B->graph_used[k] < B->graph_used[slot]) slot = k;
            }
            if (B->graphs[slot]) {
                (void)hipGraphExecDestroy(B->graphs[slot]);
                B->graphs[slot] = nullptr;
            }
            hipGraph_t g = nullptr;
            // chain 0 on the context's stream, the others on side streams forked from / joined to it
            GPSCAL_HIP(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
            for (int c = 1; c < B->nchains; ++c) {
                GPSCAL_HIP(ctx, hipEventRecord(B->chain_ev[c], ctx->stream));
                GPSCAL_HIP(ctx, hipStreamWaitEvent(B->chain_stream[c], B->chain_ev[c], 0));
            }
            for (int it = 0; it < iters; ++it)
                for (int c = 0; c < B->nchains; ++c) {
                    hipStream_t st = c == 0 ? ctx->stream : B->chain_stream[c];
                    launch_step(B, it == iters - 1, c, st, want_err);
                    launch_solve(B, it, c, st);
                }
            for (int c = 1; c < B->nchains; ++c) {
                GPSCAL_HIP(ctx, hipEventRecord(B->chain_ev[c], B->chain_stream[c]));
                GPSCAL_HIP(ctx, hipStreamWaitEvent(ctx->stream, B->chain_ev[c], 0));
            }
            GPSCAL_HIP(ctx, hipStreamEndCapture(ctx->stream, &g));
            GPSCAL_HIP(ctx, hipGraphInstantiate(&B->graphs[slot], g, nullptr, nullptr, 0));
            (void)hipGraphDestroy(g);
            B->graph_iters[slot] = gkey;
        }
        B->graph_used[slot] = ++B->graph_clock;
        GPSCAL_HIP(ctx, hipGraphLaunch(B->graphs[slot], ctx->stream));
    }
    bool sync = false;
    if (T_out) {
        if (is_device_ptr(T_out)) {
            GPSCAL_HIP(ctx, hipMemcpyAsync(T_out, B->pose64.p, sizeof(double) * 16 * np, hipMemcpyDeviceToDevice, ctx->stream));
        } else {
            GPSCAL_HIP(ctx, hipMemcpyAsync(T_out, B->pose64.p, sizeof(double) * 16 * np, hipMemcpyDeviceToHost, ctx->stream));
            sync = true;
        }
    }
    if (mean_err && iters > 0) {
        // err_hist rows have stride err_cap; copy row by row when it differs
        hipMemcpyKind kind = is_device_ptr(mean_err) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        GPSCAL_HIP(ctx, hipMemcpy2DAsync(mean_err, sizeof(double) * iters, B->err_hist.p, sizeof(double) * B->err_cap,
                                         sizeof(double) * iters, np, kind, ctx->stream));
        sync = sync || kind == hipMemcpyDeviceToHost;
    }
    if (sync) GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (sync && B->persistent && iters > 0 && !step_ms) {  // a bounded wait of the persistent kernel ran out?
        std::vector<IcpCtl> hc((size_t)np);
        GPSCAL_HIP(ctx, hipMemcpy(hc.data(), B->ctl.p, sizeof(IcpCtl) * np, hipMemcpyDeviceToHost));
        for (const IcpCtl &c : hc)
            if (c.error) return fail(ctx, GPSCAL_EHIP, "icp_persistent_kernel: a workgroup waited too long for its pair's pose");
    }
    return GPSCAL_OK;
}

extern "C" int gpscal_scan_batch_correspondences(gpscal_scan_batch *B, int32_t *idx, float *sqd)
{
    if (!B || !idx || !sqd) return GPSCAL_EINVAL;
    gpscal_ctx *ctx = B->ctx;
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    OutArg<int> oi;
    OutArg<float> od;
    GPSCAL_HIP(ctx, oi.bind(ctx, idx, (size_t)B->total_n));
    GPSCAL_HIP(ctx, od.bind(ctx, sqd, (size_t)B->total_n));
    long long maxn = 0;
    for (auto &P : B->hpairs) maxn = std::max<long long>(maxn, P.n);
    int gx = std::max(1, std::min(div_up(maxn, BLOCK), 1024));
    hipLaunchKernelGGL(unsort_nn_kernel, dim3(gx, B->npairs), dim3(BLOCK), 0, ctx->stream, B->pairs.p, B->src_orig.p,
                       B->nn_idx.p, B->nn_sqd.p, oi.dev, od.dev);
    GPSCAL_HIP(ctx, hipGetLastError());
    bool sync = false;
    GPSCAL_HIP(ctx, oi.commit(ctx, &sync));
    GPSCAL_HIP(ctx, od.commit(ctx, &sync));
    if (sync) GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}

extern "C" double gpscal_scan_batch_build_seconds(gpscal_scan_batch *B) { return B ? B->build_seconds : 0.0; }

extern "C" int gpscal_scan_batch_destroy(gpscal_scan_batch *B)
{
    if (!B) return GPSCAL_EINVAL;
    (void)hipSetDevice(B->ctx->device);
    (void)hipStreamSynchronize(B->ctx->stream);
    delete B;
    return GPSCAL_OK;
}

// --------------------------------------------- single-pair conveniences

static int make_borrowed_batch(gpscal_ctx *ctx, gpscal_knn_index *index, const float *src, int n, int stride,
                               const double *w, gpscal_scan_batch **out)
{
    if (!ctx || !index || !src || n < 1) return fail(ctx, GPSCAL_EINVAL, "gpscal_icp: bad argument");
    auto *B = new gpscal_scan_batch;
    B->ctx = ctx;
    B->npairs = 1;
    B->tgt = &index->gs;
    B->borrowed = true;
    long long so[2] = {0, n};
    int rc = ensure_safe_radius(ctx, index->gs);
    if (!rc) rc = batch_setup_sources(B, src, stride, so, w);
    if (rc) {
        delete B;
        return rc;
    }
    *out = B;
    return GPSCAL_OK;
}

extern "C" int gpscal_icp_run(gpscal_ctx *ctx, gpscal_knn_index *index, const float *src_xyz, int n, int stride,
                              const double *w, int iters, const double *T0, double *T_out, double *mean_err_hist)
{
    gpscal_scan_batch *B = nullptr;
    int rc = make_borrowed_batch(ctx, index, src_xyz, n, stride, w, &B);
    if (rc) return rc;
    rc = gpscal_scan_batch_set_pose(B, T0);
    if (!rc) rc = gpscal_scan_batch_icp(B, iters, T_out, mean_err_hist, nullptr);
    (void)hipStreamSynchronize(ctx->stream);
    delete B;
    return rc;
}

extern "C" int gpscal_icp_iterate(gpscal_ctx *ctx, gpscal_knn_index *index, const float *src_xyz, int n, int stride,
                                  const double *w, const double *T_in, double *T_out, double *mean_err)
{
    return gpscal_icp_run(ctx, index, src_xyz, n, stride, w, 1, T_in, T_out, mean_err);
}
