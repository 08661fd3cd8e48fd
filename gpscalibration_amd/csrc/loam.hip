// loam.hip -- LOAM sweep-to-sweep scan matching (laserOdometry's Gauss-Newton loop) for
// many independent sweeps at once: one workgroup runs ALL <= 25 iterations of one
// sweep -- correspondences, residuals, 6x6 normal equations, solve, degeneracy
// projection, convergence test -- with workgroup barriers only.  Sweeps of different
// SLAM segments are independent (LOAM is reset per segment), so a launch carries one
// sweep per segment.  gfx950 only.
//
// Replaces, in /root/reference/src/gpsCalibration/src/lidar_slam/loam/laserOdometry.cpp:
//   TransformToStart / TransformToEnd      :123-150, :156-227 (IMU terms are zero under run.sh)
//   correspondence search                  :592-677, :752-844 (kd-tree k=1 + adjacent-ring scans)
//   point-to-line / point-to-plane terms   :680-746, :847-901
//   Jacobian, normal equations, solve      :909-975 (cv::solve DECOMP_QR -> Householder QR, float64)
//   degeneracy projection                  :977-1004 (cv::eigen -> cyclic Jacobi, float64)
//   update + convergence                   :1005-1028
//   pose accumulation                      :1035-1064
// Points are float4 {x, y, z, intensity}; intensity = ring id + 0.1 * relative time.
// Per-point arithmetic is float32 as in the reference; the 27 sums of the normal
// equations are accumulated in float64 (OpenCV's float32 gemm order is not pinned).
#include "common.hpp"
#include "knn_device.hpp"
#include "knn_host.hpp"
#include "loam_internal.hpp"
#include "wave_reduce.hpp"

#include <algorithm>

namespace gpscal {

#ifndef GPSCAL_LOAM_FLAT
#define GPSCAL_LOAM_FLAT 1  // the searches of lo_search_kernel / lm_point_kernel walk per-lane run lists (block3_level_flat)
#endif
#ifndef GPSCAL_LO_DIAG
#define GPSCAL_LO_DIAG 0  // ablations of lo_search_kernel (wrong results): 1 = no ring searches, 2 = no nearest search
#endif
constexpr int LBLOCK = 512;  // 1024 threads: 2 340 against 2 900 sweeps/s (the register cap of a 16-wave workgroup costs more than the shorter point loop gives)
constexpr int LWAVES = LBLOCK / 64;
constexpr int LSUMS = 28;  // 21 upper-triangle AtA + 6 AtB + 1 row count
constexpr int RS = 8;       // ring-walk candidates fetched per step


__device__ __forceinline__ float4 lo_to_start(const float *tr, float4 p)
{
    // LO:123-150
    const float s = 10 * (p.w - (int)p.w);
    const float rx = s * tr[0], ry = s * tr[1], rz = s * tr[2];
    const float tx = s * tr[3], ty = s * tr[4], tz = s * tr[5];
    float sz, cz, sx, cx, sy, cy;  // per point: the angles scale with the point's relative time
    sincosf(rz, &sz, &cz);
    sincosf(rx, &sx, &cx);
    sincosf(ry, &sy, &cy);
    const float x1 = cz * (p.x - tx) + sz * (p.y - ty);
    const float y1 = -sz * (p.x - tx) + cz * (p.y - ty);
    const float z1 = (p.z - tz);
    const float x2 = x1;
    const float y2 = cx * y1 + sx * z1;
    const float z2 = -sx * y1 + cx * z1;
    return make_float4(cy * x2 - sy * z2, y2, sy * x2 + cy * z2, p.w);
}

__device__ __forceinline__ float4 lo_to_end(const float *tr, float4 p)
{
    // LO:156-227 with the IMU stages (identities at zero IMU) left out
    const float4 p3 = lo_to_start(tr, p);
    const float rx = tr[0], ry = tr[1], rz = tr[2], tx = tr[3], ty = tr[4], tz = tr[5];
    const float x4 = cosf(ry) * p3.x + sinf(ry) * p3.z;
    const float y4 = p3.y;
    const float z4 = -sinf(ry) * p3.x + cosf(ry) * p3.z;
    const float x5 = x4;
    const float y5 = cosf(rx) * y4 - sinf(rx) * z4;
    const float z5 = sinf(rx) * y4 + cosf(rx) * z4;
    return make_float4(cosf(rz) * x5 - sinf(rz) * y5 + tx, sinf(rz) * x5 + cosf(rz) * y5 + ty, z5 + tz,
                       (float)(int)p.w);
}

__device__ __forceinline__ float sq3(float4 a, float4 b)
{
    // the plain expression of LO:627-632: products and sums, no fused multiply-add
    const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    return __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
}

// One row of the linearised system (LO:916-971, s = 1): Jacobian a[6] and b = -0.05 d2.
// sines / cosines of the current transform: the same for every row of an iteration (computing them per row, as
// the expression reads, was a third of lo_iter_kernel's instructions)
struct LoTrig {
    float srx, crx, sry, cry, srz, crz, tx, ty, tz;
    __device__ __forceinline__ void set(const float *tr)
    {
        srx = sinf(tr[0]); crx = cosf(tr[0]);
        sry = sinf(tr[1]); cry = cosf(tr[1]);
        srz = sinf(tr[2]); crz = cosf(tr[2]);
        tx = tr[3]; ty = tr[4]; tz = tr[5];
    }
};

__device__ __forceinline__ void lo_row(const LoTrig &g, float4 pt, float4 cf, double *sum)
{
    const float srx = g.srx, crx = g.crx, sry = g.sry, cry = g.cry;
    const float srz = g.srz, crz = g.crz, tx = g.tx, ty = g.ty, tz = g.tz;
    const float px = pt.x, py = pt.y, pz = pt.z, cx = cf.x, cy = cf.y, cz = cf.z;
    float a[6];
    a[0] = (-crx * sry * srz * px + crx * crz * sry * py + srx * sry * pz + tx * crx * sry * srz - ty * crx * crz * sry -
            tz * srx * sry) * cx +
           (srx * srz * px - crz * srx * py + crx * pz + ty * crz * srx - tz * crx - tx * srx * srz) * cy +
           (crx * cry * srz * px - crx * cry * crz * py - cry * srx * pz + tz * cry * srx + ty * crx * cry * crz -
            tx * crx * cry * srz) * cz;
    a[1] = ((-crz * sry - cry * srx * srz) * px + (cry * crz * srx - sry * srz) * py - crx * cry * pz +
            tx * (crz * sry + cry * srx * srz) + ty * (sry * srz - cry * crz * srx) + tz * crx * cry) * cx +
           ((cry * crz - srx * sry * srz) * px + (cry * srz + crz * srx * sry) * py - crx * sry * pz + tz * crx * sry -
            ty * (cry * srz + crz * srx * sry) - tx * (cry * crz - srx * sry * srz)) * cz;
    a[2] = ((-cry * srz - crz * srx * sry) * px + (cry * crz - srx * sry * srz) * py +
            tx * (cry * srz + crz * srx * sry) - ty * (cry * crz - srx * sry * srz)) * cx +
           (-crx * crz * px - crx * srz * py + ty * crx * srz + tx * crx * crz) * cy +
           ((cry * crz * srx - sry * srz) * px + (crz * sry + cry * srx * srz) * py +
            tx * (sry * srz - cry * crz * srx) - ty * (crz * sry + cry * srx * srz)) * cz;
    a[3] = -(cry * crz - srx * sry * srz) * cx + crx * srz * cy - (crz * sry + cry * srx * srz) * cz;
    a[4] = -(cry * srz + crz * srx * sry) * cx - crx * crz * cy - (sry * srz - cry * crz * srx) * cz;
    a[5] = crx * sry * cx - srx * cy - crx * cry * cz;
    const float b = (float)(-0.05 * (double)cf.w);  // LO:970
    int k = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) sum[k++] += (double)a[r] * (double)a[c];
#pragma unroll
    for (int r = 0; r < 6; ++r) sum[21 + r] += (double)a[r] * (double)b;
    sum[27] += 1.0;
}

// ---- thread-0 dense helpers on register-resident 6x6 systems (float64).  Every loop is fully
// unrolled so that the arrays live in VGPRs: the same code over LDS pointers cost ~100 cycles per
// access on a single lane and dominated the iteration time.
__device__ __forceinline__ void lo_solve_qr6(double (&A)[36], double (&b)[6], double (&x)[6])
{
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        double nrm = 0;
#pragma unroll
        for (int i = k; i < 6; ++i) nrm += A[6 * i + k] * A[6 * i + k];
        nrm = sqrt(nrm);
        if (nrm == 0.0) continue;
        const double alpha = A[6 * k + k] > 0 ? -nrm : nrm;
        double v[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = i >= k ? A[6 * i + k] : 0.0;
        v[k] -= alpha;
        double vv = 0;
#pragma unroll
        for (int i = k; i < 6; ++i) vv += v[i] * v[i];
        if (vv == 0.0) continue;
#pragma unroll
        for (int j = k; j < 6; ++j) {
            double d = 0;
#pragma unroll
            for (int i = k; i < 6; ++i) d += v[i] * A[6 * i + j];
            d = 2 * d / vv;
#pragma unroll
            for (int i = k; i < 6; ++i) A[6 * i + j] -= d * v[i];
        }
        double d = 0;
#pragma unroll
        for (int i = k; i < 6; ++i) d += v[i] * b[i];
        d = 2 * d / vv;
#pragma unroll
        for (int i = k; i < 6; ++i) b[i] -= d * v[i];
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = 0.0;
#pragma unroll
    for (int i = 5; i >= 0; --i) {
        double acc = b[i];
#pragma unroll
        for (int j = i + 1; j < 6; ++j) acc -= A[6 * i + j] * x[j];
        x[i] = A[6 * i + i] != 0.0 ? acc / A[6 * i + i] : 0.0;
    }
}

// eigen-decomposition of the symmetric A (destroyed); Q columns = eigenvectors
__device__ __forceinline__ void lo_eigen_sym6(double (&A)[36], double (&Q)[36])
{
#pragma unroll
    for (int i = 0; i < 36; ++i) Q[i] = (i % 7 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        // The restatement sweeps until the off-diagonal sum underflows, which rounding residue never lets happen:
        // it always runs its 60 sweeps, the last ~50 of them rotations by angles below one ulp.  Stopping at
        // |a_pq| <= 1e-15 sqrt(a_pp a_qq) for every pair (the scaled criterion for positive semi-definite
        // matrices: small eigenvalues keep their relative accuracy) leaves eigenvalues and vectors within a
        // few ulps of that result at a tenth of the dependent float64 arithmetic.
        double off = 0;
        bool conv = true;
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int q = p + 1; q < 6; ++q) {
                const double a2 = A[6 * p + q] * A[6 * p + q];
                off += a2;
                conv = conv && a2 <= 1e-30 * fabs(A[7 * p] * A[7 * q]);
            }
        if (off < 1e-300 || conv) break;
#pragma unroll
        for (int p = 0; p < 6; ++p)
#pragma unroll
            for (int q = p + 1; q < 6; ++q) {
                const double apq = A[6 * p + q];
                if (fabs(apq) < 1e-300) continue;
                const double tau = (A[6 * q + q] - A[6 * p + p]) / (2 * apq);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1 + tau * tau));
                const double c = 1 / sqrt(1 + t * t), s = t * c;
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const double akp = A[6 * k + p], akq = A[6 * k + q];
                    A[6 * k + p] = c * akp - s * akq;
                    A[6 * k + q] = s * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const double apk = A[6 * p + k], aqk = A[6 * q + k];
                    A[6 * p + k] = c * apk - s * aqk;
                    A[6 * q + k] = s * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const double qkp = Q[6 * k + p], qkq = Q[6 * k + q];
                    Q[6 * k + p] = c * qkp - s * qkq;
                    Q[6 * k + q] = s * qkp + c * qkq;
                }
            }
    }
}

// Thread 0: from the 28 block sums to the update x (LO:909-1004 / LM:922-997).  P (6x6, LDS) is
// the degeneracy projector kept from iteration 0; returns false when fewer than min_rows rows.
__device__ __noinline__ bool solve_update(const double (&tot)[LSUMS], bool first, double thresh, double *sP,
                                             int *degenerate, double (&x)[6])
{
    double A[36], b[6];
    {
        int k = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = r; c < 6; ++c) {
                A[6 * r + c] = tot[k];
                A[6 * c + r] = tot[k];
                ++k;
            }
#pragma unroll
        for (int r = 0; r < 6; ++r) b[r] = tot[21 + r];
    }
    if (first) {
        // eigenvalues below the threshold mark degenerate directions: P = V^-1 V2 with rows of V =
        // eigenvectors = sum over kept eigenvectors q q^T; "kept" = all but the trailing run of
        // eigenvalues < thresh in descending order
        double E[36], Q[36];
#pragma unroll
        for (int i = 0; i < 36; ++i) E[i] = A[i];
        lo_eigen_sym6(E, Q);
        double ev[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) ev[i] = E[7 * i];
        // rank of each eigenvalue in descending order (ties by index), then the kept set
        int keep = 6;
        bool kept[6];
        int rank[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            int r = 0;
#pragma unroll
            for (int j = 0; j < 6; ++j) r += (ev[j] > ev[i] || (ev[j] == ev[i] && j < i)) ? 1 : 0;
            rank[i] = r;
        }
        // walk from the smallest (rank 5) up while below the threshold
#pragma unroll
        for (int r = 5; r >= 0; --r) {
            double e = 0;
#pragma unroll
            for (int i = 0; i < 6; ++i) e = rank[i] == r ? ev[i] : e;
            if (keep == r + 1 && e < thresh) keep = r;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) kept[i] = rank[i] < keep;
        *degenerate = keep < 6;
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                double acc = 0;
                // accumulate in descending-eigenvalue order, as the restatement does
#pragma unroll
                for (int k2 = 0; k2 < 6; ++k2)
#pragma unroll
                    for (int i = 0; i < 6; ++i)
                        if (rank[i] == k2 && kept[i]) acc += Q[6 * r + i] * Q[6 * c + i];
                sP[6 * r + c] = acc;
            }
    }
    lo_solve_qr6(A, b, x);
    if (*degenerate) {
        double x2[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double acc = 0;
#pragma unroll
            for (int c = 0; c < 6; ++c) acc += sP[6 * r + c] * x[c];
            x2[r] = acc;
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) x[r] = x2[r];
    }
    return true;
}

// Exact 1-NN of p in one indexed cloud (pair b of a GridSet); all lanes of the wave call it.
// `seed` >= 0: a point of the cloud to start from (the previous search round's answer: the transform moved
// the query only a little, so the search is down to one level and a few rows; the result is the exact nearest
// neighbour either way).
__device__ __forceinline__ void lo_nearest(const PairDesc &P, const float4 *__restrict__ sorted,
                                           const unsigned *__restrict__ cell_start, bool act, float4 p, int &idx,
                                           float &sqd, const float4 *__restrict__ cloud = nullptr, int seed = -1,
                                           uint2 *__restrict__ slab = nullptr)
{
    Best<1> B;
    B.init_radius(25.f);  // LO:607,758: a nearest point at 5 m or more is no correspondence
    if (act && seed >= 0) {
        const float4 c = cloud[seed];
        B.consider(sqdist(p.x, p.y, p.z, c.x, c.y, c.z), make_float4(c.x, c.y, c.z, __int_as_float(seed)), 0u);
    }
    if (!(GPSCAL_LO_DIAG & 2)) knn_query(P, sorted, cell_start, act, p.x, p.y, p.z, B, 0, slab);
    idx = B.i[0] == 0x7fffffff ? -1 : B.i[0];
    sqd = B.d[0];
}

// ===================================================================================
// laserMapping's sweep-to-map optimisation (laserMapping.cpp:748-1018): <= 10 iterations,
// every one with a fresh k=5 search of each stacked feature in the local map.
//   pointAssociateToMap                     LM:244-262
//   corner: 5-NN covariance, cv::eigen, line LM:757-858 (cyclic Jacobi 3x3, float64)
//   surf: 5-NN plane fit cv::solve QR        LM:860-920 (Householder 5x3, float64)
//   Jacobian / normal equations / solve      LM:922-968
//   degeneracy (eigenvalues < 100), update   LM:970-1018
// ===================================================================================

// two largest eigenvalues and the principal eigenvector of a symmetric 3x3
__device__ __forceinline__ void lm_eigen_sym3_top(const double *A_in, double &l1, double &l2, double *v1)
{
    double A[9], Q[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
#pragma unroll
    for (int i = 0; i < 9; ++i) A[i] = A_in[i];
    for (int sweep = 0; sweep < 50; ++sweep) {
        const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        // same stopping rule as lo_eigen_sym6: the restatement's 50 sweeps end in rotations below one ulp
        const bool conv = A[1] * A[1] <= 1e-30 * fabs(A[0] * A[4]) && A[2] * A[2] <= 1e-30 * fabs(A[0] * A[8]) &&
                          A[5] * A[5] <= 1e-30 * fabs(A[4] * A[8]);
        if (off < 1e-300 || conv) break;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                const double apq = A[3 * p + q];
                if (fabs(apq) < 1e-300) continue;
                const double tau = (A[4 * q] - A[4 * p]) / (2 * apq);
                const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1 + tau * tau));
                const double c = 1 / sqrt(1 + t * t), s = t * c;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[3 * k + p], akq = A[3 * k + q];
                    A[3 * k + p] = c * akp - s * akq;
                    A[3 * k + q] = s * akp + c * akq;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double apk = A[3 * p + k], aqk = A[3 * q + k];
                    A[3 * p + k] = c * apk - s * aqk;
                    A[3 * q + k] = s * apk + c * aqk;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double qkp = Q[3 * k + p], qkq = Q[3 * k + q];
                    Q[3 * k + p] = c * qkp - s * qkq;
                    Q[3 * k + q] = s * qkp + c * qkq;
                }
            }
    }
    // first maximum wins, then the first maximum of the rest (no dynamic indexing)
    const double e0 = A[0], e1 = A[4], e2 = A[8];
    int i1 = 0;
    if (e1 > e0) i1 = 1;
    if (e2 > (i1 == 1 ? e1 : e0)) i1 = 2;
    const double ea = i1 == 0 ? e1 : e0, eb = i1 == 2 ? e1 : e2;  // the other two, in index order
    l1 = i1 == 0 ? e0 : (i1 == 1 ? e1 : e2);
    l2 = eb > ea ? eb : ea;
#pragma unroll
    for (int k = 0; k < 3; ++k) v1[k] = i1 == 0 ? Q[3 * k] : (i1 == 1 ? Q[3 * k + 1] : Q[3 * k + 2]);
}

// least squares of the 5x3 system A x = -1 (LM:870-875)
__device__ __forceinline__ void lm_plane_fit5(double *A, double *x)
{
    double b[5] = {-1, -1, -1, -1, -1};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double nrm = 0;
#pragma unroll
        for (int i = k; i < 5; ++i) nrm += A[3 * i + k] * A[3 * i + k];
        nrm = sqrt(nrm);
        if (nrm == 0.0) continue;
        const double alpha = A[3 * k + k] > 0 ? -nrm : nrm;
        double v[5] = {0, 0, 0, 0, 0};
#pragma unroll
        for (int i = k; i < 5; ++i) v[i] = A[3 * i + k];
        v[k] -= alpha;
        double vv = 0;
#pragma unroll
        for (int i = k; i < 5; ++i) vv += v[i] * v[i];
        if (vv == 0.0) continue;
#pragma unroll
        for (int j = k; j < 3; ++j) {
            double d = 0;
#pragma unroll
            for (int i = k; i < 5; ++i) d += v[i] * A[3 * i + j];
            d = 2 * d / vv;
#pragma unroll
            for (int i = k; i < 5; ++i) A[3 * i + j] -= d * v[i];
        }
        double d = 0;
#pragma unroll
        for (int i = k; i < 5; ++i) d += v[i] * b[i];
        d = 2 * d / vv;
#pragma unroll
        for (int i = k; i < 5; ++i) b[i] -= d * v[i];
    }
    x[0] = x[1] = x[2] = 0.0;
#pragma unroll
    for (int i = 2; i >= 0; --i) {
        double acc = b[i];
#pragma unroll
        for (int j = i + 1; j < 3; ++j) acc -= A[3 * i + j] * x[j];
        x[i] = A[3 * i + i] != 0.0 ? acc / A[3 * i + i] : 0.0;
    }
}

struct LmTrig {
    float srx, crx, sry, cry, srz, crz, tx, ty, tz;
};

__device__ __forceinline__ float4 lm_to_map(const LmTrig &g, float4 p)
{
    // LM:244-262
    const float x1 = g.crz * p.x - g.srz * p.y;
    const float y1 = g.srz * p.x + g.crz * p.y;
    const float z1 = p.z;
    const float x2 = x1;
    const float y2 = g.crx * y1 - g.srx * z1;
    const float z2 = g.srx * y1 + g.crx * z1;
    return make_float4(g.cry * x2 + g.sry * z2 + g.tx, y2 + g.ty, -g.sry * x2 + g.cry * z2 + g.tz, p.w);
}

// one row of the mapping system (LM:940-966)
__device__ __forceinline__ void lm_row(const LmTrig &g, float4 pt, float4 cf, double *sum)
{
    const float srx = g.srx, crx = g.crx, sry = g.sry, cry = g.cry, srz = g.srz, crz = g.crz;
    const float px = pt.x, py = pt.y, pz = pt.z;
    float a[6];
    a[0] = (crx * sry * srz * px + crx * crz * sry * py - srx * sry * pz) * cf.x +
           (-srx * srz * px - crz * srx * py - crx * pz) * cf.y +
           (crx * cry * srz * px + crx * cry * crz * py - cry * srx * pz) * cf.z;
    a[1] = ((cry * srx * srz - crz * sry) * px + (sry * srz + cry * crz * srx) * py + crx * cry * pz) * cf.x +
           ((-cry * crz - srx * sry * srz) * px + (cry * srz - crz * srx * sry) * py - crx * sry * pz) * cf.z;
    a[2] = ((crz * srx * sry - cry * srz) * px + (-cry * crz - srx * sry * srz) * py) * cf.x +
           (crx * crz * px - crx * srz * py) * cf.y +
           ((sry * srz + cry * crz * srx) * px + (crz * sry - cry * srx * srz) * py) * cf.z;
    a[3] = cf.x;
    a[4] = cf.y;
    a[5] = cf.z;
    const float b = -cf.w;
    int k = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = r; c < 6; ++c) sum[k++] += (double)a[r] * (double)a[c];
#pragma unroll
    for (int r = 0; r < 6; ++r) sum[21 + r] += (double)a[r] * (double)b;
    sum[27] += 1.0;
}


// ---- laserMapping's loop as per-iteration launches.  One sweep has only a few thousand stacked
// features, so a single workgroup per sweep left the k = 5 searches latency bound on one CU;
// here every 256-point tile of every sweep is its own workgroup (lm_point_kernel), the 28 sums go
// through per-tile partials, and a one-wave kernel per sweep reduces them in tile order, solves
// and updates the transform (lm_solve_kernel).  Launches of sweeps that have converged return at
// once; the host enqueues all 10 iterations without reading anything back.
constexpr int PT_BLOCK = 256;
constexpr int PT_WAVES = PT_BLOCK / 64;

struct IterState {
    float tr[6];
    int done, degenerate, iters, nsel;
    double P[36];
    int rs_c[18], rs_s[18], mono, ring_ok;  // laserOdometry only: ring tables of the last clouds
};

__global__ void lm_init_kernel(const MapDesc *__restrict__ sweeps, int nsweeps, const float *__restrict__ tr_in,
                               IterState *__restrict__ st)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nsweeps) return;
    IterState &S = st[b];
    for (int k = 0; k < 6; ++k) S.tr[k] = tr_in[6 * b + k];
    S.done = !(sweeps[b].mc > 10 && sweeps[b].ms > 100);  // LM:748
    S.degenerate = 0;
    S.iters = 0;
    S.nsel = 0;
    for (int k = 0; k < 36; ++k) S.P[k] = (k % 7 == 0) ? 1.0 : 0.0;
}

// block partial of the 28 sums -> partial[(b * tiles_max + tile) * LSUMS + k]
// The 28 sums of a wave, lanes folded in a fixed order, written to out[0..LSUMS) by the lanes that end up holding
// them.  Eight accumulators at a time go through the wave's private LDS slab (8 x 64 doubles): lane (k, seg) adds
// eight consecutive lanes' copies of value k (4 x ds_read_b128), three DPP steps fold the eight segments -- ~130
// instructions against ~1 100 for 28 full DPP wave reductions (12 dpp moves, 6 adds, 2 readlanes and their hazard
// nops each), which were 45 % of an iteration of lo_iter_kernel.
__device__ __forceinline__ void wave_sums28(const double (&sum)[LSUMS], double *__restrict__ slab, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int g0 = 0; g0 < LSUMS; g0 += 8) {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (g0 + k < LSUMS) slab[k * 64 + lane] = sum[g0 + k];
        __builtin_amdgcn_wave_barrier();
        const int k = lane >> 3, seg = lane & 7;
        const double2 *row = reinterpret_cast<const double2 *>(slab + k * 64 + seg * 8);
        const double2 a0 = row[0], a1 = row[1], a2 = row[2], a3 = row[3];
        double v = ((a0.x + a0.y) + (a1.x + a1.y)) + ((a2.x + a2.y) + (a3.x + a3.y));
        v = dpp_add_f64<0x111, 0xF>(v);  // row_shr:1
        v = dpp_add_f64<0x112, 0xF>(v);  // row_shr:2
        v = dpp_add_f64<0x114, 0xF>(v);  // row_shr:4 -> lane 8k+7 holds value k
        if (seg == 7 && g0 + k < LSUMS) out[g0 + k] = v;
        __builtin_amdgcn_wave_barrier();
    }
}

// `wslab`: the calling wave's 4 KiB of LDS for the transposes (the search's run-list slab, free by now)
__device__ __forceinline__ void tile_partial(double (&sum)[LSUMS], double *__restrict__ out, double *__restrict__ wslab)
{
    __shared__ double red[PT_WAVES][LSUMS];
    const int wave = threadIdx.x >> 6;
    wave_sums28(sum, wslab, &red[wave][0]);
    __syncthreads();
    if (threadIdx.x < LSUMS) {
        double v = 0;
#pragma unroll
        for (int w = 0; w < PT_WAVES; ++w) v += red[w][threadIdx.x];
        out[threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(PT_BLOCK) void lm_point_kernel(
    const MapDesc *__restrict__ sweeps, const float4 *__restrict__ cstack, const float4 *__restrict__ sstack,
    const float4 *__restrict__ cmap, const float4 *__restrict__ smap, const PairDesc *__restrict__ cpairs,
    const float4 *__restrict__ csorted, const unsigned *__restrict__ ccells, const PairDesc *__restrict__ spairs,
    const float4 *__restrict__ ssorted, const unsigned *__restrict__ scells, const IterState *__restrict__ st,
    double *__restrict__ partial, int tiles_max, int *__restrict__ prev5, long long surf_base)
{
    const int b = blockIdx.y;
    if (st[b].done) return;
    const MapDesc D = sweeps[b];
    const int ct = (D.nc + PT_BLOCK - 1) / PT_BLOCK, stl = (D.ns + PT_BLOCK - 1) / PT_BLOCK;
    const int tile = blockIdx.x;
    if (tile >= ct + stl) return;
#if GPSCAL_LOAM_FLAT
    __shared__ uint2 s_slab[PT_BLOCK / 64][8 * 64];  // per wave: the run lists of block3_level_flat
    uint2 *slab = &s_slab[threadIdx.x >> 6][0];
#else
    uint2 *slab = nullptr;
#endif
    double sum[LSUMS];
#pragma unroll
    for (int k = 0; k < LSUMS; ++k) sum[k] = 0.0;
    LmTrig g;
    {
        const float *tr = st[b].tr;
        g.srx = sinf(tr[0]); g.crx = cosf(tr[0]);
        g.sry = sinf(tr[1]); g.cry = cosf(tr[1]);
        g.srz = sinf(tr[2]); g.crz = cosf(tr[2]);
        g.tx = tr[3]; g.ty = tr[4]; g.tz = tr[5];
    }
    if (tile < ct) {
        // ---- corner features: line through the 5 nearest map corners (LM:756-858)
        const float4 *cs = cstack + D.cstack_off, *cm = cmap + D.cmap_off;
        const int i = tile * PT_BLOCK + threadIdx.x;
        const bool act = i < D.nc;
        float4 po = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act) po = cs[i];
        const float4 ps = lm_to_map(g, po);
        Best<5> B;
        B.init_radius(1.0f);  // LM:762,865: a feature whose fifth neighbour is not within 1 m is dropped
        // last iteration's five neighbours first: the transform moved the query only a little, so they
        // bound the search to one grid level (the result is the exact k-NN either way)
        int *pv = prev5 + 5 * (D.cstack_off + i);
        if (act && pv[0] >= 0) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int pj = pv[j];
                const float4 c = cm[pj];
                B.consider(sqdist(ps.x, ps.y, ps.z, c.x, c.y, c.z), make_float4(c.x, c.y, c.z, __int_as_float(pj)), 0u);
            }
        }
        knn_query(cpairs[b], csorted, ccells, act, ps.x, ps.y, ps.z, B, 0, slab);
        if (act) {
#pragma unroll
            for (int j = 0; j < 5; ++j) pv[j] = B.i[4] == 0x7fffffff ? -1 : B.i[j];
        }
        if (act && B.d[4] < 1.0f) {
            float4 q[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) q[j] = cm[B.i[j]];
            float cx = 0, cy = 0, cz = 0;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                cx += q[j].x;
                cy += q[j].y;
                cz += q[j].z;
            }
            cx /= 5; cy /= 5; cz /= 5;
            float a11 = 0, a12 = 0, a13 = 0, a22 = 0, a23 = 0, a33 = 0;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float ax = q[j].x - cx, ay = q[j].y - cy, az = q[j].z - cz;
                a11 += ax * ax; a12 += ax * ay; a13 += ax * az;
                a22 += ay * ay; a23 += ay * az; a33 += az * az;
            }
            a11 /= 5; a12 /= 5; a13 /= 5; a22 /= 5; a23 /= 5; a33 /= 5;
            const double A1[9] = {a11, a12, a13, a12, a22, a23, a13, a23, a33};
            double l1, l2, v1[3];
            lm_eigen_sym3_top(A1, l1, l2, v1);
            if ((float)l1 > 3 * (float)l2) {  // LM:812
                const float x0 = ps.x, y0 = ps.y, z0 = ps.z;
                const float x1 = (float)((double)cx + 0.1 * (double)(float)v1[0]);
                const float y1 = (float)((double)cy + 0.1 * (double)(float)v1[1]);
                const float z1 = (float)((double)cz + 0.1 * (double)(float)v1[2]);
                const float x2 = (float)((double)cx - 0.1 * (double)(float)v1[0]);
                const float y2 = (float)((double)cy - 0.1 * (double)(float)v1[1]);
                const float z2 = (float)((double)cz - 0.1 * (double)(float)v1[2]);
                const float m11 = (x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1);
                const float m22 = (x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1);
                const float m33 = (y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1);
                const float a012 = sqrtf(m11 * m11 + m22 * m22 + m33 * m33);
                const float l12 = sqrtf((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
                const float la = ((y1 - y2) * m11 + (z1 - z2) * m22) / a012 / l12;
                const float lb = -((x1 - x2) * m11 - (z1 - z2) * m33) / a012 / l12;
                const float lc = -((x1 - x2) * m22 + (y1 - y2) * m33) / a012 / l12;
                const float ld2 = a012 / l12;
                const float s = (float)(1 - 0.9 * fabs((double)ld2));
                if (s > 0.1) lm_row(g, po, make_float4(s * la, s * lb, s * lc, s * ld2), sum);
            }
        }
    } else {
        // ---- surface features: plane through the 5 nearest map surfels (LM:860-920)
        const float4 *ss = sstack + D.sstack_off, *sm = smap + D.smap_off;
        const int i = (tile - ct) * PT_BLOCK + threadIdx.x;
        const bool act = i < D.ns;
        float4 po = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act) po = ss[i];
        const float4 ps = lm_to_map(g, po);
        Best<5> B;
        B.init_radius(1.0f);  // LM:762,865: a feature whose fifth neighbour is not within 1 m is dropped
        int *pv = prev5 + 5 * (surf_base + D.sstack_off + i);  // the surf entries follow the corner entries
        if (act && pv[0] >= 0) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int pj = pv[j];
                const float4 c = sm[pj];
                B.consider(sqdist(ps.x, ps.y, ps.z, c.x, c.y, c.z), make_float4(c.x, c.y, c.z, __int_as_float(pj)), 0u);
            }
        }
        knn_query(spairs[b], ssorted, scells, act, ps.x, ps.y, ps.z, B, 0, slab);
        if (act) {
#pragma unroll
            for (int j = 0; j < 5; ++j) pv[j] = B.i[4] == 0x7fffffff ? -1 : B.i[j];
        }
        if (act && B.d[4] < 1.0f) {
            float4 q[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) q[j] = sm[B.i[j]];
            double A0[15], x[3];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                A0[3 * j] = q[j].x;
                A0[3 * j + 1] = q[j].y;
                A0[3 * j + 2] = q[j].z;
            }
            lm_plane_fit5(A0, x);
            float pa = (float)x[0], pb = (float)x[1], pc = (float)x[2], pd = 1;
            const float pn = sqrtf(pa * pa + pb * pb + pc * pc);
            pa /= pn; pb /= pn; pc /= pn; pd /= pn;
            bool valid = true;
#pragma unroll
            for (int j = 0; j < 5; ++j)
                valid = valid && !((double)fabsf(pa * q[j].x + pb * q[j].y + pc * q[j].z + pd) > 0.2);
            if (valid) {
                const float pd2 = pa * ps.x + pb * ps.y + pc * ps.z + pd;
                const float s = (float)(1 - 0.9 * fabs((double)pd2) /
                                                (double)sqrtf(sqrtf(ps.x * ps.x + ps.y * ps.y + ps.z * ps.z)));
                if (s > 0.1) lm_row(g, po, make_float4(s * pa, s * pb, s * pc, s * pd2), sum);
            }
        }
    }
#if GPSCAL_LOAM_FLAT
    tile_partial(sum, partial + ((long long)b * tiles_max + tile) * LSUMS, reinterpret_cast<double *>(slab));
#else
    __shared__ double s_red_slab[PT_WAVES][8 * 64];
    tile_partial(sum, partial + ((long long)b * tiles_max + tile) * LSUMS, &s_red_slab[threadIdx.x >> 6][0]);
#endif
}

// one wave per sweep: partials in tile order -> solve -> update (LM:922-1017)
__global__ __launch_bounds__(64) void lm_solve_kernel(const MapDesc *__restrict__ sweeps, IterState *__restrict__ st,
                                                      const double *__restrict__ partial, int tiles_max, int it)
{
    const int b = blockIdx.x;
    IterState &S = st[b];
    if (S.done) return;
    __shared__ double s_tot[LSUMS];
    const MapDesc D = sweeps[b];
    const int tiles = (D.nc + PT_BLOCK - 1) / PT_BLOCK + (D.ns + PT_BLOCK - 1) / PT_BLOCK;
    if (threadIdx.x < LSUMS) {
        double v = 0;
        for (int t = 0; t < tiles; ++t) v += partial[((long long)b * tiles_max + t) * LSUMS + threadIdx.x];
        s_tot[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    double tot[LSUMS];
#pragma unroll
    for (int k = 0; k < LSUMS; ++k) tot[k] = s_tot[k];
    S.iters = it + 1;
    const int nsel = (int)tot[27];
    S.nsel = nsel;
    if (nsel >= 50) {  // LM:929-931; solve LM:968, degeneracy LM:970-997 (threshold 100)
        double x[6];
        solve_update(tot, it == 0, 100.0, S.P, &S.degenerate, x);
        float xf[6];
#pragma unroll
        for (int k2 = 0; k2 < 6; ++k2) {
            xf[k2] = (float)x[k2];
            S.tr[k2] = S.tr[k2] + xf[k2];
        }
        const double r2d = 180.0 / 3.14159265358979323846;
        const float dR = (float)sqrt((xf[0] * r2d) * (xf[0] * r2d) + (xf[1] * r2d) * (xf[1] * r2d) +
                                     (xf[2] * r2d) * (xf[2] * r2d));
        const float dT = (float)sqrt(((double)xf[3] * 100) * ((double)xf[3] * 100) +
                                     ((double)xf[4] * 100) * ((double)xf[4] * 100) +
                                     ((double)xf[5] * 100) * ((double)xf[5] * 100));
        if (dR < 0.05 && dT < 0.05) S.done = 1;  // LM:1015
    }
    if (it == 9) S.done = 1;  // LM:752
}

__global__ void iter_finish_kernel(const IterState *__restrict__ st, int nsweeps, float *__restrict__ tr_out,
                                   int *__restrict__ iters_out, int *__restrict__ nsel_out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nsweeps) return;
    for (int k = 0; k < 6; ++k) tr_out[6 * b + k] = st[b].tr[k];
    if (iters_out) iters_out[b] = st[b].iters;
    if (nsel_out) nsel_out[b] = st[b].nsel;
}

// ---- laserOdometry's loop as a few launches per sweep.  The correspondence searches of
// iterations 0, 5, 10, 15, 20 (three grid searches per feature) are spread over one workgroup per
// 256 features (lo_search_kernel); the five iterations that share a set of correspondences are
// cheap and sequential, and run in one workgroup per sweep (lo_iter_kernel).  Converged sweeps
// return at once, so the host enqueues the whole schedule without reading anything back.
__global__ __launch_bounds__(PT_BLOCK) void lo_init_kernel(const SweepDesc *__restrict__ sweeps,
                                                           const float4 *__restrict__ clast,
                                                           const float4 *__restrict__ slast, int *__restrict__ corr,
                                                           const float *__restrict__ tr_in, IterState *__restrict__ st,
                                                           const int *__restrict__ ring_cnt_c,
                                                           const int *__restrict__ ring_cnt_s)
{
    const int b = blockIdx.x;
    const SweepDesc D = sweeps[b];
    IterState &S = st[b];
    __shared__ int s_mono;
    int *ci1 = corr + D.corr_off;
    for (int i = threadIdx.x; i < 2 * D.nc + 3 * D.ns; i += PT_BLOCK) ci1[i] = -1;
    if (threadIdx.x < 6) S.tr[threadIdx.x] = tr_in[6 * b + threadIdx.x];
    if (threadIdx.x < 36) S.P[threadIdx.x] = (threadIdx.x % 7 == 0) ? 1.0 : 0.0;
    if (threadIdx.x < 18) {
        S.rs_c[threadIdx.x] = D.mc;
        S.rs_s[threadIdx.x] = D.ms;
    }
    if (threadIdx.x == 0) {
        S.done = !(D.mc > 10 && D.ms > 100);  // LO:569
        S.degenerate = 0;
        S.iters = 0;
        S.nsel = 0;
        s_mono = 1;
    }
    __syncthreads();
    // ring tables: rs[r] = first point with ring id >= r.  scanRegistration emits the clouds ring by
    // ring; anything else (ids outside 0..15, ids going down) takes the sequential walks.
    for (int pass = 0; pass < 2; ++pass) {
        const float4 *c = pass == 0 ? clast + D.clast_off : slast + D.slast_off;
        const int m = pass == 0 ? D.mc : D.ms;
        int *rs = pass == 0 ? S.rs_c : S.rs_s;
        for (int i = threadIdx.x; i < m; i += PT_BLOCK) {
            const int ri = (int)c[i].w, rp = i > 0 ? (int)c[i - 1].w : -1;
            if (ri < rp || ri < 0 || ri > 15) s_mono = 0;
            else
                for (int r = rp + 1; r <= ri; ++r) rs[r] = i;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        S.mono = s_mono;
        // per-ring grids were built from the caller's ring counts: use them only if they describe
        // the clouds' own ring ids
        int ok = s_mono && ring_cnt_c && ring_cnt_s;
        if (ok) {
            int ac = 0, as = 0;
            for (int r = 0; r < 16; ++r) {
                ok = ok && S.rs_c[r] == min(ac, D.mc) && S.rs_s[r] == min(as, D.ms);
                ac += ring_cnt_c[16 * b + r];
                as += ring_cnt_s[16 * b + r];
            }
            ok = ok && ac == D.mc && as == D.ms;
        }
        S.ring_ok = ok;
    }
}

// Adjacent-ring search on per-ring grids: every lane with 0 <= want <= 15 queries the grid of ring
// `want` of its sweep; the wave visits the distinct rings one after the other (a tile of features
// spans one or two rings).  R carries its intervals in cloud indices; a ring's grid numbers its
// points from 0, hence R.base.
__device__ __forceinline__ void ring_search(const PairDesc *__restrict__ rp, const float4 *__restrict__ rsorted,
                                            const unsigned *__restrict__ rcells, const int *rs, bool has, int want,
                                            float4 ps, BestRing &R, uint2 *__restrict__ slab = nullptr)
{
    if (GPSCAL_LO_DIAG & 1) return;
    // every lane searches the grid of the ring it wants, all of them in one pass (knn_query_lanes)
    const bool sel = has && want >= 0 && want <= 15;
    const int r = sel ? want : 0;
    if (sel) R.base = rs[r];
    knn_query_lanes(rp + r, rsorted, rcells, sel, ps.x, ps.y, ps.z, R, slab);
}

// the previous round's answer as the first candidate of a ring search (it still has to pass the record's
// interval filter: the closest point, and with it the rings, may have changed)
__device__ __forceinline__ void ring_seed(BestRing &R, bool has, const float4 *__restrict__ cloud, int seed, float4 ps)
{
    if (has && seed >= 0) {
        const float4 c = cloud[seed];
        R.base = 0;
        R.consider(sqdist(ps.x, ps.y, ps.z, c.x, c.y, c.z), make_float4(c.x, c.y, c.z, __int_as_float(seed)), 0u);
    }
}

// The nearest search of a tile's 64 features by all four waves of its workgroup: every wave holds the same 64
// queries; wave 0 scans the own row, then the waves take two of the other eight rows each; after each level the waves join
// their records through LDS (smaller squared distance, then smaller index: the record's own order), so that all of
// them decide together whether the level settled the query.  Must be called by the whole workgroup; `act`, `p` and
// the seed are the same in every wave.  One wave walking all nine rows was the longest chain of the kernel.
__device__ __forceinline__ void lo_nearest_split(const PairDesc &P, const float4 *__restrict__ sorted,
                                                 const unsigned *__restrict__ cell_start, bool act, float4 p, int &idx,
                                                 float &sqd, const float4 *__restrict__ cloud, int seed,
                                                 uint2 *__restrict__ slab, float (*s_d)[64], int (*s_i)[64])
{
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
    const unsigned rowmask = role == 0 ? 0x006u : (role == 1 ? 0x018u : (role == 2 ? 0x060u : 0x180u));  // rows 1-8, two per wave
    Best<1> B;
    B.init_radius(25.f);  // LO:607,758: a nearest point at 5 m or more is no correspondence
    if (act && seed >= 0) {
        const float4 c = cloud[seed];
        B.consider(sqdist(p.x, p.y, p.z, c.x, c.y, c.z), make_float4(c.x, c.y, c.z, __int_as_float(seed)), 0u);
    }
    float px = p.x, py = p.y, pz = p.z;
    if (!act) px = py = pz = 0.f;
    int lvl = 0;
    {
        const float w0 = B.worst();
        if (B.seeded()) {  // the seed's distance names the level (knn_query)
            lvl = P.nlevels - 1;
            for (int l = P.nlevels - 2; l >= 0; --l) {
                const float g = P.lv[l].h * 0.999f - P.lv[l].margin;
                if (g > 0.f && w0 <= g * g) lvl = l;
            }
        }
    }
    bool todo = act;
    for (int l = 0; l < P.nlevels; ++l) {
        if (__ballot(todo) == 0ull) break;  // the same in every wave: they hold identical records here
        const bool a = todo && l >= lvl;
        if (__ballot(a) == 0ull) continue;
        const GridDesc &G = P.lv[l];
        CellGeo C;
        C.set(G, px, py, pz);
        // the own row first, by wave 0 alone: its result is the bound with which the four waves then prune the
        // other eight rows (two each) -- without it they would scan every cell of their rows
        if (role == 0) {
            block3_level_flat(G, C, sorted, cell_start, a, px, py, pz, B, slab, 0x001u);
            s_d[0][lane] = B.d[0];
            s_i[0][lane] = B.i[0];
        }
        __syncthreads();
        if (role != 0) {
            const float d = s_d[0][lane];
            const int i = s_i[0][lane];
            if (d < B.d[0] || (d == B.d[0] && i < B.i[0])) {
                B.d[0] = d;
                B.i[0] = i;
            }
        }
        __syncthreads();
        block3_level_flat(G, C, sorted, cell_start, a, px, py, pz, B, slab, rowmask);
        s_d[role][lane] = B.d[0];
        s_i[role][lane] = B.i[0];
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d = s_d[r][lane];
            const int i = s_i[r][lane];
            if (d < B.d[0] || (d == B.d[0] && i < B.i[0])) {
                B.d[0] = d;
                B.i[0] = i;
            }
        }
        __syncthreads();
        if (a && B.worst() <= C.settled_r2(G)) todo = false;
    }
    idx = B.i[0] == 0x7fffffff ? -1 : B.i[0];
    sqd = B.d[0];
}

// A tile is 64 features and a workgroup its four waves, which all hold the same 64 queries.  The nearest point of
// the last cloud is searched by all of them (lo_nearest_split: wave 0 the own row of the 3x3x3 block, then two of the
// other eight rows per wave, records joined through LDS after every level).  The adjacent-ring searches need only
// that result and are independent of each other, so after a barrier waves 1..3 run one of them each (corner: rings
// scan-1 | scan+1 -> min2; surf: own ring -> min2, rings scan-1 | scan+1 -> min3) and wave 0 joins the partial
// records in the record's own order (squared distance, then the order the sequential walk would meet the
// candidates) and leaves the correspondence's geometry for lo_iter_kernel.  One after the other in one lane these
// searches were a chain of ~160 us at under one wave per SIMD.
constexpr int LO_TILE = 64;
__global__ __launch_bounds__(PT_BLOCK) void lo_search_kernel(
    const SweepDesc *__restrict__ sweeps, const float4 *__restrict__ sharp, const float4 *__restrict__ flat,
    const float4 *__restrict__ clast, const float4 *__restrict__ slast, const PairDesc *__restrict__ cpairs,
    const float4 *__restrict__ csorted, const unsigned *__restrict__ ccells, const PairDesc *__restrict__ spairs,
    const float4 *__restrict__ ssorted, const unsigned *__restrict__ scells, int *__restrict__ corr,
    const IterState *__restrict__ st, const PairDesc *__restrict__ rcpairs, const float4 *__restrict__ rcsorted,
    const unsigned *__restrict__ rccells, const PairDesc *__restrict__ rspairs, const float4 *__restrict__ rssorted,
    const unsigned *__restrict__ rscells, float4 *__restrict__ geo)
{
    static_assert(PT_BLOCK == 4 * LO_TILE, "four waves per tile of 64 features");
    const int b = blockIdx.y;
    if (st[b].done) return;
    const SweepDesc D = sweeps[b];
    const int ct = (D.nc + LO_TILE - 1) / LO_TILE, stl = (D.ns + LO_TILE - 1) / LO_TILE;
    const int tile = blockIdx.x;
    if (tile >= ct + stl) return;
    __shared__ float tr[6];
    __shared__ int s_rs_c[18], s_rs_s[18];
    __shared__ int s_closest[LO_TILE];
    __shared__ float s_rd[4][LO_TILE];
    __shared__ unsigned s_ro[4][LO_TILE];
    __shared__ int s_ri[4][LO_TILE];
#if GPSCAL_LOAM_FLAT
    __shared__ uint2 s_slab[PT_BLOCK / 64][8 * 64];  // per wave: the run lists of block3_level_flat
    uint2 *slab = &s_slab[threadIdx.x >> 6][0];
#else
    uint2 *slab = nullptr;
#endif
    if (threadIdx.x < 6) tr[threadIdx.x] = st[b].tr[threadIdx.x];
    if (threadIdx.x < 18) {
        s_rs_c[threadIdx.x] = st[b].rs_c[threadIdx.x];
        s_rs_s[threadIdx.x] = st[b].rs_s[threadIdx.x];
    }
    __syncthreads();
    const bool mono = st[b].mono != 0;
    const bool ring_grids = rcpairs != nullptr && st[b].ring_ok != 0;
    const int lane = threadIdx.x & 63, role = threadIdx.x >> 6;
#ifdef GPSCAL_STATS
    const unsigned long long t_kernel0 = wall_clock64();
#endif
    STAT_WAVE(2, 1);  // waves of lo_search_kernel
    const PairDesc &CP = cpairs[b];
    const PairDesc &SP = spairs[b];
    const float4 *sh = sharp + D.sharp_off, *fl = flat + D.flat_off, *cl = clast + D.clast_off,
                 *sl = slast + D.slast_off;
    int *ci1 = corr + D.corr_off, *ci2 = ci1 + D.nc;
    int *si1 = ci2 + D.nc, *si2 = si1 + D.ns, *si3 = si2 + D.ns;
    // forward ring scans are bounded by the CURRENT sweep's feature counts (LO:620,776)
    const int fwd_c = min(D.nc, D.mc), fwd_s = min(D.ns, D.ms);
    // joins two finished ring records: no candidate loses; else the smaller (d, ord)
    auto better = [](float da, unsigned oa, int ia, float db, unsigned ob, int ib) -> int {
        if (ia < 0) return ib;
        if (ib < 0) return ia;
        return (db < da || (db == da && ob < oa)) ? ib : ia;
    };
    if (tile < ct) {
        const int i = tile * LO_TILE + lane;
        const bool act = i < D.nc;
        float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act) pi = sh[i];
        const float4 ps = lo_to_start(tr, pi);
        const int prev2 = act ? ci2[i] : -1;  // -1 in a sweep's first round (lo_init_kernel)
        int closest = -1, min2 = -1;
        {
            int idx;
            float sqd;
            const int prev1 = act ? ci1[i] : -1;
            lo_nearest_split(CP, csorted, ccells, act, ps, idx, sqd, cl, prev1, slab, s_rd, s_ri);
            const bool has = act && idx >= 0 && sqd < 25;
            if (has) closest = idx;
            if (role == 0) s_closest[lane] = closest;
            if (role == 0 && !mono && has) {  // the sequential walks: one wave
            const int scan = (int)cl[closest].w;
            float d2min = 25;
            // the walks are sequential by definition (first strict minimum wins, stop at the
            // ring border); RS candidates are fetched per step so the loads overlap
            for (int j0 = closest + 1, stop = 0; j0 < fwd_c && !stop; j0 += RS) {
                float4 qq[RS];
#pragma unroll
                for (int u = 0; u < RS; ++u) qq[u] = cl[min(j0 + u, fwd_c - 1)];
#pragma unroll
                for (int u = 0; u < RS; ++u) {
                    const int j = j0 + u;
                    if (stop || j >= fwd_c) continue;
                    const float4 q = qq[u];
                    if ((int)q.w > scan + 1.5) {
                        stop = 1;
                        continue;
                    }
                    const float d = sq3(q, ps);
                    if ((int)q.w > scan && d < d2min) {
                        d2min = d;
                        min2 = j;
                    }
                }
            }
            for (int j0 = closest - 1, stop = 0; j0 >= 0 && !stop; j0 -= RS) {
                float4 qq[RS];
#pragma unroll
                for (int u = 0; u < RS; ++u) qq[u] = cl[max(j0 - u, 0)];
#pragma unroll
                for (int u = 0; u < RS; ++u) {
                    const int j = j0 - u;
                    if (stop || j < 0) continue;
                    const float4 q = qq[u];
                    if ((int)q.w < scan - 1.5) {
                        stop = 1;
                        continue;
                    }
                    const float d = sq3(q, ps);
                    if ((int)q.w < scan && d < d2min) {
                        d2min = d;
                        min2 = j;
                    }
                }
            }
        }
        }
        __syncthreads();
        if (mono && (role == 1 || role == 2)) {  // the two walks as searches filtered to the adjacent rings
            closest = s_closest[lane];
            const bool has = closest >= 0;
            int a0 = 0, a1 = 0, scan = -9;
            if (has) {
                scan = (int)cl[closest].w;
                if (role == 1) {
                    if (scan >= 1) {
                        a0 = s_rs_c[scan - 1];
                        a1 = s_rs_c[scan];
                    }
                } else {
                    a0 = s_rs_c[scan + 1];
                    a1 = min(s_rs_c[scan + 2], fwd_c);
                }
            }
            BestRing R;
            if (role == 1) R.init(25.f, closest, a0, a1, 0, 0);
            else R.init(25.f, closest, 0, 0, a0, a1);
            ring_seed(R, has, cl, prev2, ps);
            if (ring_grids) ring_search(rcpairs + 16 * b, rcsorted, rccells, s_rs_c, has && a1 > a0, role == 1 ? scan - 1 : scan + 1, ps, R, slab);
            else knn_query(CP, csorted, ccells, has && a1 > a0, ps.x, ps.y, ps.z, R, 0, slab);
            s_rd[role][lane] = R.d;
            s_ro[role][lane] = R.ord;
            s_ri[role][lane] = has ? R.i : -1;
        }
        __syncthreads();
        if (role == 0 && act) {
            if (mono) min2 = better(s_rd[1][lane], s_ro[1][lane], s_ri[1][lane], s_rd[2][lane], s_ro[2][lane], s_ri[2][lane]);
            ci1[i] = closest;
            ci2[i] = min2;
            // what the five iterations up to the next search need of this correspondence: the two points of the line
            // (the iteration kernel then reads two coalesced streams instead of chasing two indices per feature)
            float4 g1 = make_float4(0.f, 0.f, 0.f, 0.f), g2 = g1;
            if (closest >= 0 && min2 >= 0) {
                const float4 t1 = cl[closest], t2 = cl[min2];
                g1 = make_float4(t1.x, t1.y, t1.z, 1.f);
                g2 = make_float4(t2.x, t2.y, t2.z, 0.f);
            }
            float4 *gc = geo + D.corr_off;
            gc[2 * i] = g1;
            gc[2 * i + 1] = g2;
        }
    } else {
        const int i = (tile - ct) * LO_TILE + lane;
        const bool act = i < D.ns;
        float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
        if (act) pi = fl[i];
        const float4 ps = lo_to_start(tr, pi);
        const int prev2 = act ? si2[i] : -1, prev3 = act ? si3[i] : -1;
        int closest = -1, min2 = -1, min3 = -1;
        {
            int idx;
            float sqd;
            const int prev1 = act ? si1[i] : -1;
#ifdef GPSCAL_STATS
            const unsigned long long tk0 = wall_clock64();
#endif
            lo_nearest_split(SP, ssorted, scells, act, ps, idx, sqd, sl, prev1, slab, s_rd, s_ri);
#ifdef GPSCAL_STATS
            if (role == 0) {
                STAT_WAVE(21, wall_clock64() - tk0);  // surf tiles, wave 0: ticks (10 ns) in the split nearest search
                STAT_WAVE(10, 1);
            }
#endif
            const bool has = act && idx >= 0 && sqd < 25;
            if (has) closest = idx;
            if (role == 0) s_closest[lane] = closest;
            if (role == 0 && !mono && has) {  // the sequential walks: one wave
            const int scan = (int)sl[closest].w;
            float d2 = 25, d3 = 25;
            for (int j0 = closest + 1, stop = 0; j0 < fwd_s && !stop; j0 += RS) {
                float4 qq[RS];
#pragma unroll
                for (int u = 0; u < RS; ++u) qq[u] = sl[min(j0 + u, fwd_s - 1)];
#pragma unroll
                for (int u = 0; u < RS; ++u) {
                    const int j = j0 + u;
                    if (stop || j >= fwd_s) continue;
                    const float4 q = qq[u];
                    if ((int)q.w > scan + 1.5) {
                        stop = 1;
                        continue;
                    }
                    const float d = sq3(q, ps);
                    if ((int)q.w <= scan) {
                        if (d < d2) { d2 = d; min2 = j; }
                    } else {
                        if (d < d3) { d3 = d; min3 = j; }
                    }
                }
            }
            for (int j0 = closest - 1, stop = 0; j0 >= 0 && !stop; j0 -= RS) {
                float4 qq[RS];
#pragma unroll
                for (int u = 0; u < RS; ++u) qq[u] = sl[max(j0 - u, 0)];
#pragma unroll
                for (int u = 0; u < RS; ++u) {
                    const int j = j0 - u;
                    if (stop || j < 0) continue;
                    const float4 q = qq[u];
                    if ((int)q.w < scan - 1.5) {
                        stop = 1;
                        continue;
                    }
                    const float d = sq3(q, ps);
                    if ((int)q.w >= scan) {
                        if (d < d2) { d2 = d; min2 = j; }
                    } else {
                        if (d < d3) { d3 = d; min3 = j; }
                    }
                }
            }
        }
        }
        __syncthreads();
        if (mono && role >= 1) {  // own ring -> min2 (wave 1), adjacent rings -> min3 (waves 2 and 3)
            closest = s_closest[lane];
            const bool has = closest >= 0;
            int a0 = 0, a1 = 0, b0 = 0, b1 = 0, scan = -9, want = -9;
            if (has) {
                scan = (int)sl[closest].w;
                if (role == 1) {
                    a0 = s_rs_s[scan];
                    a1 = closest;
                    b0 = closest + 1;
                    b1 = min(s_rs_s[scan + 1], fwd_s);
                    want = scan;
                } else if (role == 2) {
                    if (scan >= 1) {
                        a0 = s_rs_s[scan - 1];
                        a1 = s_rs_s[scan];
                    }
                    want = scan - 1;
                } else {
                    b0 = s_rs_s[scan + 1];
                    b1 = min(s_rs_s[scan + 2], fwd_s);
                    want = scan + 1;
                }
            }
            BestRing R;
            R.init(25.f, closest, a0, a1, b0, b1);
            ring_seed(R, has, sl, role == 1 ? prev2 : prev3, ps);
            const bool any = has && (a1 > a0 || b1 > b0);
#ifdef GPSCAL_STATS
            const unsigned long long tr0 = wall_clock64();
#endif
            if (ring_grids) ring_search(rspairs + 16 * b, rssorted, rscells, s_rs_s, any, want, ps, R, slab);
            else knn_query(SP, ssorted, scells, any, ps.x, ps.y, ps.z, R, 0, slab);
#ifdef GPSCAL_STATS
            STAT_WAVE(22 + (role == 1 ? 0 : 1), wall_clock64() - tr0);  // 22: own-ring search (wave 1), 23: adjacent rings (waves 2, 3)
#endif
            s_rd[role][lane] = R.d;
            s_ro[role][lane] = R.ord;
            s_ri[role][lane] = has ? R.i : -1;
        }
        __syncthreads();
        if (role == 0 && act) {
            if (mono) {
                min2 = s_ri[1][lane];
                min3 = better(s_rd[2][lane], s_ro[2][lane], s_ri[2][lane], s_rd[3][lane], s_ro[3][lane], s_ri[3][lane]);
            }
            si1[i] = closest;
            si2[i] = min2;
            si3[i] = min3;
#ifdef GPSCAL_STATS
            STAT_WAVE(19, wall_clock64() - t_kernel0);  // surf tiles, wave 0: the whole tile
#endif
            // ... and the unit plane through the three points (LO:847-870), which no iteration changes
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);  // without a full correspondence: pd2 = 0, no row
            if (closest >= 0 && min2 >= 0 && min3 >= 0) {
                const float4 t1 = sl[closest], t2 = sl[min2], t3 = sl[min3];
                float pa = (t2.y - t1.y) * (t3.z - t1.z) - (t3.y - t1.y) * (t2.z - t1.z);
                float pb = (t2.z - t1.z) * (t3.x - t1.x) - (t3.z - t1.z) * (t2.x - t1.x);
                float pc = (t2.x - t1.x) * (t3.y - t1.y) - (t3.x - t1.x) * (t2.y - t1.y);
                float pd = -(pa * t1.x + pb * t1.y + pc * t1.z);
                const float pn = sqrtf(pa * pa + pb * pb + pc * pc);
                pa /= pn; pb /= pn; pc /= pn; pd /= pn;
                g = make_float4(pa, pb, pc, pd);
            }
            geo[D.corr_off + 2 * (long long)D.nc + i] = g;
        }
    }
}

// iterations it0 .. it0+4 of one sweep with the correspondences of the preceding search
__global__ __launch_bounds__(LBLOCK) void lo_iter_kernel(
    const SweepDesc *__restrict__ sweeps, const float4 *__restrict__ sharp, const float4 *__restrict__ flat,
    const float4 *__restrict__ clast, const float4 *__restrict__ slast, const int *__restrict__ corr,
    IterState *__restrict__ st, int it0, const float4 *__restrict__ geo)
{
    const int b = blockIdx.x;
    IterState &S = st[b];
    if (S.done) return;
    __shared__ float tr[6];
    __shared__ double red[LWAVES][LSUMS];
    __shared__ double s_slab[LWAVES][8 * 64];  // wave_sums28
    __shared__ int s_done;
    const SweepDesc D = sweeps[b];
    const float4 *sh = sharp + D.sharp_off, *fl = flat + D.flat_off;
    const float4 *gc = geo + D.corr_off, *gs = gc + 2 * (long long)D.nc;  // lo_search_kernel: line points | planes
    const int wave = threadIdx.x >> 6;
    if (threadIdx.x < 6) tr[threadIdx.x] = S.tr[threadIdx.x];
    if (threadIdx.x == 0) s_done = 0;
    __syncthreads();
    for (int it = it0; it < it0 + 5; ++it) {  // LO:585
        double sum[LSUMS];
#pragma unroll
        for (int k = 0; k < LSUMS; ++k) sum[k] = 0.0;
        LoTrig g;
        g.set(tr);
        // ---- corner features: point-to-line (LO:680-746)
        for (int i0 = 0; i0 < D.nc; i0 += LBLOCK) {
            const int i = i0 + threadIdx.x;
            const bool act = i < D.nc;
            float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
            if (act) pi = sh[i];
            const float4 ps = lo_to_start(tr, pi);
            float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = t1;
            if (act) {
                t1 = gc[2 * i];
                t2 = gc[2 * i + 1];
            }
            if (act && t1.w != 0.f) {  // a full correspondence (lo_search_kernel)
                const float x0 = ps.x, y0 = ps.y, z0 = ps.z;
                const float x1 = t1.x, y1 = t1.y, z1 = t1.z, x2 = t2.x, y2 = t2.y, z2 = t2.z;
                const float m11 = (x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1);
                const float m22 = (x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1);
                const float m33 = (y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1);
                const float a012 = sqrtf(m11 * m11 + m22 * m22 + m33 * m33);
                const float l12 = sqrtf((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
                const float la = ((y1 - y2) * m11 + (z1 - z2) * m22) / a012 / l12;
                const float lb = -((x1 - x2) * m11 - (z1 - z2) * m33) / a012 / l12;
                const float lc = -((x1 - x2) * m22 + (y1 - y2) * m33) / a012 / l12;
                const float ld2 = a012 / l12;
                float s = 1;
                if (it >= 5) s = (float)(1 - 1.8 * fabs((double)ld2));
                if (s > 0.1 && ld2 != 0) lo_row(g, pi, make_float4(s * la, s * lb, s * lc, s * ld2), sum);
            }
        }
        // ---- surface features: point-to-plane (LO:847-901)
        for (int i0 = 0; i0 < D.ns; i0 += LBLOCK) {
            const int i = i0 + threadIdx.x;
            const bool act = i < D.ns;
            float4 pi = make_float4(0.f, 0.f, 0.f, 0.f);
            if (act) pi = fl[i];
            const float4 ps = lo_to_start(tr, pi);
            float4 pl = make_float4(0.f, 0.f, 0.f, 0.f);
            if (act) pl = gs[i];  // the unit plane of the correspondence; zeros without one
            if (act && (pl.x != 0.f || pl.y != 0.f || pl.z != 0.f || pl.w != 0.f)) {
                const float pa = pl.x, pb = pl.y, pc = pl.z, pd = pl.w;
                const float pd2 = pa * ps.x + pb * ps.y + pc * ps.z + pd;
                float s = 1;
                if (it >= 5)
                    s = (float)(1 - 1.8 * fabs((double)pd2) / (double)sqrtf(sqrtf(ps.x * ps.x + ps.y * ps.y + ps.z * ps.z)));
                if (s > 0.1 && pd2 != 0) lo_row(g, pi, make_float4(s * pa, s * pb, s * pc, s * pd2), sum);
            }
        }
        // ---- block reduction of the 28 sums (fixed order)
        wave_sums28(sum, &s_slab[wave][0], &red[wave][0]);
        __syncthreads();
        if (threadIdx.x == 0) {
            double tot[LSUMS];
#pragma unroll
            for (int k = 0; k < LSUMS; ++k) {
                double v = 0;
#pragma unroll
                for (int w = 0; w < LWAVES; ++w) v += red[w][k];
                tot[k] = v;
            }
            S.iters = it + 1;
            const int nsel = (int)tot[27];
            S.nsel = nsel;
            if (nsel >= 10) {  // LO:905-907; solve LO:975, degeneracy LO:977-1003 (threshold 10)
                double x[6];
                solve_update(tot, it == 0, 10.0, S.P, &S.degenerate, x);
                float xf[6];
#pragma unroll
                for (int k2 = 0; k2 < 6; ++k2) {
                    xf[k2] = (float)x[k2];
                    float v = tr[k2] + xf[k2];
                    if (isnan(v)) v = 0;  // LO:1012-1015
                    tr[k2] = v;
                }
                const double r2d = 180.0 / 3.14159265358979323846;
                const float dR = (float)sqrt((xf[0] * r2d) * (xf[0] * r2d) + (xf[1] * r2d) * (xf[1] * r2d) +
                                             (xf[2] * r2d) * (xf[2] * r2d));
                const float dT = (float)sqrt(((double)xf[3] * 100) * ((double)xf[3] * 100) +
                                             ((double)xf[4] * 100) * ((double)xf[4] * 100) +
                                             ((double)xf[5] * 100) * ((double)xf[5] * 100));
                if (dR < 0.1 && dT < 0.1) s_done = 1;  // LO:1026
            }
            if (it == 24) s_done = 1;
        }
        __syncthreads();
        if (s_done) break;
    }
    if (threadIdx.x < 6) S.tr[threadIdx.x] = tr[threadIdx.x];
    if (threadIdx.x == 0 && s_done) S.done = 1;
}

// results + pose accumulation (LO:1035-1064 with zero IMU terms)
__global__ void lo_finish_kernel(const IterState *__restrict__ st, int nsweeps, float *__restrict__ tr_out,
                                 int *__restrict__ iters_out, int *__restrict__ nsel_out,
                                 const float *__restrict__ sum_in, float *__restrict__ sum_out)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nsweeps) return;
    float tr[6];
    for (int k = 0; k < 6; ++k) {
        tr[k] = st[b].tr[k];
        tr_out[6 * b + k] = tr[k];
    }
    if (iters_out) iters_out[b] = st[b].iters;
    if (nsel_out) nsel_out[b] = st[b].nsel;
    if (sum_in && sum_out) {
        // pose accumulation, LO:1035-1064 with zero IMU terms.  Every sine / cosine is evaluated once (the expression
        // as written names cosf(lx) ten times; inlined, that was 12 000 instructions on the odometry half's path)
        const float *S = sum_in + 6 * b;
        const float cx = S[0], cy = S[1], cz = S[2];
        const float lx = -tr[0], ly = (float)(-tr[1] * 1.05), lz = -tr[2];
        const float slx = sinf(lx), clx = cosf(lx), sly = sinf(ly), cly = cosf(ly), slz = sinf(lz), clz = cosf(lz);
        const float scx = sinf(cx), ccx = cosf(cx), scy = sinf(cy), ccy = cosf(cy), scz = sinf(cz), ccz = cosf(cz);
        const float srx = clx * ccx * sly * scz - ccx * ccz * slx - clx * cly * scx;
        const float ox = -asinf(srx);
        const float cox = cosf(ox);
        const float srycrx = slx * (ccy * scz - ccz * scx * scy) + clx * sly * (ccy * ccz + scx * scy * scz) +
                             clx * cly * ccx * scy;
        const float crycrx = clx * cly * ccx * ccy - clx * sly * (ccz * scy - ccy * scx * scz) -
                             slx * (scy * scz + ccy * ccz * scx);
        const float oy = atan2f(srycrx / cox, crycrx / cox);
        const float srzcrx = scx * (clz * sly - cly * slx * slz) + ccx * scz * (cly * clz + slx * sly * slz) +
                             clx * ccx * ccz * slz;
        const float crzcrx = clx * clz * ccx * ccz - ccx * scz * (cly * slz - clz * slx * sly) -
                             scx * (sly * slz + cly * clz * slx);
        const float oz = atan2f(srzcrx / cox, crzcrx / cox);
        const float rx = ox, ry = oy, rz = oz;
        const float s_rx = sinf(rx), c_rx = cox, s_ry = sinf(ry), c_ry = cosf(ry), s_rz = sinf(rz), c_rz = cosf(rz);
        const float x1 = c_rz * tr[3] - s_rz * tr[4];
        const float y1 = s_rz * tr[3] + c_rz * tr[4];
        const float z1 = (float)(tr[5] * 1.05);
        const float x2 = x1;
        const float y2 = c_rx * y1 - s_rx * z1;
        const float z2 = s_rx * y1 + c_rx * z1;
        float *O = sum_out + 6 * b;
        const float acx = -asinf(-s_rx);
        const float cacx = cosf(acx);
        O[0] = acx;
        O[1] = atan2f(c_rx * s_ry / cacx, c_rx * c_ry / cacx);
        O[2] = atan2f(c_rx * s_rz / cacx, c_rx * c_rz / cacx);
        O[3] = S[3] - (c_ry * x2 + s_ry * z2);
        O[4] = S[4] - y2;
        O[5] = S[5] - (-s_ry * x2 + c_ry * z2);
    }
}

__global__ void loam_to_end_kernel(const float *__restrict__ tr6, const float4 *__restrict__ in, int n,
                                   float4 *__restrict__ out, int to_end)
{
    __shared__ float tr[6];
    if (threadIdx.x < 6) tr[threadIdx.x] = tr6[threadIdx.x];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = to_end ? lo_to_end(tr, in[i]) : lo_to_start(tr, in[i]);
}

}  // namespace gpscal

using namespace gpscal;

#ifdef GPSCAL_STATS
__global__ void loam_stat_slot_kernel(int slot) { g_stat_iter = slot; }
// instrumented builds only: the search counters of this translation unit (slot 1 = lo_search_kernel, 2 = lm_point_kernel)
extern "C" int gpscal_debug_stats_loam(unsigned long long *out, int n)
{
    std::vector<unsigned long long> h(NSTAT * STAT_ITERS, 0ull);
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stats), h.size() * 8) != hipSuccess) return GPSCAL_EHIP;
    for (int i = 0; i < n && i < (int)h.size(); ++i) out[i] = h[i];
    std::fill(h.begin(), h.end(), 0ull);
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stats), h.data(), h.size() * 8) != hipSuccess) return GPSCAL_EHIP;
    return GPSCAL_OK;
}
#endif

namespace gpscal {

int loam_odometry_device(gpscal_ctx *ctx, int nsweeps, const SweepDesc *descs, const float4 *d_sharp,
                         const float4 *d_flat, const float4 *d_clast, const float4 *d_slast, const long long *coff,
                         const long long *soff, const float *d_tr_in, float *d_tr_out, int *d_iters, int *d_nsel,
                         const float *d_sum_in, float *d_sum_out, const int *ring_cnt_c, const int *ring_cnt_s)
{
    // the kd-trees of the last sweep (setInputCloud, LO:538-539,1119-1120) = two index sets, and one more grid per
    // ring of each last cloud for the adjacent-ring searches (LO:613-677, 769-844) -- all in ONE GridSet: the four
    // sources are concatenated (build_grids_multi), which makes their ~40 small launches ~12
    GridSet all;
    all.pooled = pool_grids();
    DevBuf<int> d_ringc, d_rings;
    bool ring_grids = ring_cnt_c && ring_cnt_s;
    std::vector<long long> roc, ros;
    if (ring_grids) {
        roc.resize((size_t)nsweeps * 16 + 1);
        ros.resize((size_t)nsweeps * 16 + 1);
        for (int b = 0; b < nsweeps && ring_grids; ++b) {
            long long ac = coff[b], as = soff[b];
            for (int r = 0; r < 16; ++r) {
                roc[16 * (size_t)b + r] = ac;
                ros[16 * (size_t)b + r] = as;
                ac += ring_cnt_c[16 * b + r];
                as += ring_cnt_s[16 * b + r];
            }
            ring_grids = ac == coff[b + 1] && as == soff[b + 1];
        }
        roc[(size_t)nsweeps * 16] = coff[nsweeps];
        ros[(size_t)nsweeps * 16] = soff[nsweeps];
    }
    {
        // all four index sets in one call: one host wait (their bounding boxes) instead of four
        // (the per-ring grids keep the even-surface cell size although a ring is a curve: its level 0 then
        // covers laserOdometry's 5 m radius in one pass; 10x / 100x / 1000x finer cells were measured 2 % / 10 % / 64 % slower)
        // (the two cloud sets as well: 1.5 / 0.75 / 0.4 / 0.2 points per cell instead of 3: +2 / 0 / -2 / -8 %)
        GridSource srcs[4] = {{d_clast, coff, nsweeps, &all, 0.f},
                              {d_slast, soff, nsweeps, &all, 0.f},
                              {d_clast, roc.data(), nsweeps * 16, &all, 0.f},
                              {d_slast, ros.data(), nsweeps * 16, &all, 0.f}};
        int rc = build_grids_multi(ctx, ring_grids ? 4 : 2, srcs, 16, 0.f, MAX_LEVELS);
        if (rc) return rc;
    }
    // a source's pairs inside the set: corner clouds | surf clouds | corner rings | surf rings
    const PairDesc *cgp = all.pairs.p, *sgp = all.pairs.p + nsweeps, *rcgp = all.pairs.p + 2 * (size_t)nsweeps,
                   *rsgp = all.pairs.p + 18 * (size_t)nsweeps;
    if (ring_grids) {
        GPSCAL_HIP(ctx, d_ringc.alloc_async((size_t)nsweeps * 16, ctx->stream));
        GPSCAL_HIP(ctx, d_rings.alloc_async((size_t)nsweeps * 16, ctx->stream));
        GPSCAL_HIP(ctx, hipMemcpyAsync(d_ringc.p, ring_cnt_c, sizeof(int) * 16 * nsweeps, hipMemcpyHostToDevice, ctx->stream));
        GPSCAL_HIP(ctx, hipMemcpyAsync(d_rings.p, ring_cnt_s, sizeof(int) * 16 * nsweeps, hipMemcpyHostToDevice, ctx->stream));
    }
    // correspondence indices (ci1, ci2 | si1, si2, si3) live in one region per entry of `descs`: two
    // entries may name the same sweep (two replay passes of one bag at the same message)
    std::vector<SweepDesc> hd(descs, descs + nsweeps);
    long long corr_total = 0;
    int tiles_max = 1, search_tiles = 1;
    for (int b = 0; b < nsweeps; ++b) {
        hd[b].corr_off = corr_total;
        corr_total += 2ll * hd[b].nc + 3ll * hd[b].ns;
        tiles_max = std::max(tiles_max, div_up(hd[b].nc, PT_BLOCK) + div_up(hd[b].ns, PT_BLOCK));
        search_tiles = std::max(search_tiles, div_up(hd[b].nc, LO_TILE) + div_up(hd[b].ns, LO_TILE));
    }
    DevBuf<SweepDesc> d_sw;
    DevBuf<IterState> d_st;
    DevBuf<int> corr;
    GPSCAL_HIP(ctx, d_sw.alloc_async(nsweeps, ctx->stream));
    GPSCAL_HIP(ctx, d_st.alloc_async(nsweeps, ctx->stream));
    GPSCAL_HIP(ctx, corr.alloc_async((size_t)corr_total + 8, ctx->stream));
    DevBuf<float4> geo;  // per feature: the line points (2 float4 per sharp point) | the plane (1 per flat point), at corr_off
    GPSCAL_HIP(ctx, geo.alloc_async((size_t)corr_total + 8, ctx->stream));
    descs = hd.data();
    GPSCAL_HIP(ctx, hipMemcpyAsync(d_sw.p, descs, sizeof(SweepDesc) * nsweeps, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(lo_init_kernel, dim3(nsweeps), dim3(PT_BLOCK), 0, ctx->stream, d_sw.p, d_clast, d_slast, corr.p,
                       d_tr_in, d_st.p, ring_grids ? d_ringc.p : nullptr, ring_grids ? d_rings.p : nullptr);
#ifdef GPSCAL_STATS
    hipLaunchKernelGGL(loam_stat_slot_kernel, dim3(1), dim3(1), 0, ctx->stream, 1);
#endif
    for (int it0 = 0; it0 < 25; it0 += 5) {  // LO:585: a search every fifth iteration (LO:592)
        hipLaunchKernelGGL(lo_search_kernel, dim3(search_tiles, nsweeps), dim3(PT_BLOCK), 0, ctx->stream, d_sw.p, d_sharp,
                           d_flat, d_clast, d_slast, cgp, all.sorted.p, all.cell_start, sgp, all.sorted.p,
                           all.cell_start, corr.p, d_st.p, ring_grids ? rcgp : nullptr, all.sorted.p, all.cell_start,
                           ring_grids ? rsgp : nullptr, all.sorted.p, all.cell_start, geo.p);
        hipLaunchKernelGGL(lo_iter_kernel, dim3(nsweeps), dim3(LBLOCK), 0, ctx->stream, d_sw.p, d_sharp, d_flat,
                           d_clast, d_slast, corr.p, d_st.p, it0, geo.p);
    }
    hipLaunchKernelGGL(lo_finish_kernel, dim3(div_up(nsweeps, 64)), dim3(64), 0, ctx->stream, d_st.p, nsweeps, d_tr_out,
                       d_iters, d_nsel, d_sum_in, d_sum_out);
    GPSCAL_HIP(ctx, hipGetLastError());
    // the grid sets die with this scope: pooled ones go back to the stream's block cache in stream order
    if (!pool_grids()) GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}

}  // namespace gpscal

extern "C" int gpscal_loam_odometry_batched(gpscal_ctx *ctx, int nsweeps, const float *sharp_xyzi,
                                            const int *sharp_off, const float *flat_xyzi, const int *flat_off,
                                            const float *corner_last_xyzi, const int *corner_last_off,
                                            const float *surf_last_xyzi, const int *surf_last_off,
                                            const float *transform_in, float *transform_out, int *iters_out,
                                            int *nsel_out, const float *transform_sum_in, float *transform_sum_out)
{
    if (!ctx || nsweeps < 1 || !sharp_xyzi || !sharp_off || !flat_xyzi || !flat_off || !corner_last_xyzi ||
        !corner_last_off || !surf_last_xyzi || !surf_last_off || !transform_in || !transform_out)
        return fail(ctx, GPSCAL_EINVAL, "gpscal_loam_odometry_batched: bad argument");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    const int tc = sharp_off[nsweeps], tf = flat_off[nsweeps], tcl = corner_last_off[nsweeps],
              tsl = surf_last_off[nsweeps];
    std::vector<SweepDesc> hs(nsweeps);
    std::vector<long long> coff(nsweeps + 1), soff(nsweeps + 1);
    for (int b = 0; b <= nsweeps; ++b) {
        coff[b] = corner_last_off[b];
        soff[b] = surf_last_off[b];
    }
    for (int b = 0; b < nsweeps; ++b) {
        SweepDesc &D = hs[b];
        D.sharp_off = sharp_off[b];
        D.flat_off = flat_off[b];
        D.clast_off = corner_last_off[b];
        D.slast_off = surf_last_off[b];
        D.nc = sharp_off[b + 1] - sharp_off[b];
        D.ns = flat_off[b + 1] - flat_off[b];
        D.mc = corner_last_off[b + 1] - corner_last_off[b];
        D.ms = surf_last_off[b + 1] - surf_last_off[b];
        if (D.nc < 0 || D.ns < 0 || D.mc < 0 || D.ms < 0) return fail(ctx, GPSCAL_EINVAL, "bad offsets");
    }
    InArg<float> a_sh, a_fl, a_cl, a_sl, a_tr, a_sum;
    OutArg<float> o_tr, o_sum;
    OutArg<int> o_it, o_ns;
    GPSCAL_HIP(ctx, a_sh.bind(ctx, sharp_xyzi, (size_t)std::max(tc, 1) * 4));
    GPSCAL_HIP(ctx, a_fl.bind(ctx, flat_xyzi, (size_t)std::max(tf, 1) * 4));
    GPSCAL_HIP(ctx, a_cl.bind(ctx, corner_last_xyzi, (size_t)std::max(tcl, 1) * 4));
    GPSCAL_HIP(ctx, a_sl.bind(ctx, surf_last_xyzi, (size_t)std::max(tsl, 1) * 4));
    GPSCAL_HIP(ctx, a_tr.bind(ctx, transform_in, (size_t)nsweeps * 6));
    GPSCAL_HIP(ctx, a_sum.bind(ctx, transform_sum_in, transform_sum_in ? (size_t)nsweeps * 6 : 0));
    GPSCAL_HIP(ctx, o_tr.bind(ctx, transform_out, (size_t)nsweeps * 6));
    GPSCAL_HIP(ctx, o_sum.bind(ctx, transform_sum_out, transform_sum_out ? (size_t)nsweeps * 6 : 0));
    GPSCAL_HIP(ctx, o_it.bind(ctx, iters_out, iters_out ? nsweeps : 0));
    GPSCAL_HIP(ctx, o_ns.bind(ctx, nsel_out, nsel_out ? nsweeps : 0));
    int rc = loam_odometry_device(ctx, nsweeps, hs.data(), reinterpret_cast<const float4 *>(a_sh.dev),
                                  reinterpret_cast<const float4 *>(a_fl.dev), reinterpret_cast<const float4 *>(a_cl.dev),
                                  reinterpret_cast<const float4 *>(a_sl.dev), coff.data(), soff.data(), a_tr.dev,
                                  o_tr.dev, o_it.dev, o_ns.dev, a_sum.dev, o_sum.dev);
    if (rc) return rc;
    bool sync = true;
    GPSCAL_HIP(ctx, o_tr.commit(ctx, &sync));
    GPSCAL_HIP(ctx, o_sum.commit(ctx, &sync));
    GPSCAL_HIP(ctx, o_it.commit(ctx, &sync));
    GPSCAL_HIP(ctx, o_ns.commit(ctx, &sync));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}

namespace gpscal {

int loam_mapping_device(gpscal_ctx *ctx, int nsweeps, const MapDesc *descs, const float4 *d_cstack,
                        const float4 *d_sstack, const float4 *d_cmap, const float4 *d_smap, const long long *cmoff,
                        const long long *smoff, const float *d_tr_in, float *d_tr_out, int *d_iters, int *d_nsel)
{
    // kdtreeCornerFromMap / kdtreeSurfFromMap (setInputCloud, LM:749-750) = two grid sets
    GridSet all;  // both in one set (concatenated sources): corner maps | surf maps
    all.pooled = pool_grids();
    {
        GridSource srcs[2] = {{d_cmap, cmoff, nsweeps, &all, 0.f}, {d_smap, smoff, nsweeps, &all, 0.f}};
        int rc = build_grids_multi(ctx, 2, srcs, 16, 0.f, MAX_LEVELS);
        if (rc) return rc;
    }
    const PairDesc *cgp = all.pairs.p, *sgp = all.pairs.p + nsweeps;
    DevBuf<MapDesc> d_sw;
    DevBuf<IterState> d_st;
    DevBuf<double> d_part;
    DevBuf<int> d_prev5;  // last iteration's five neighbours of every stacked feature
    int tiles_max = 1;
    long long ext_c = 0, ext_s = 0;
    for (int b = 0; b < nsweeps; ++b) {
        tiles_max = std::max(tiles_max, div_up(descs[b].nc, PT_BLOCK) + div_up(descs[b].ns, PT_BLOCK));
        ext_c = std::max(ext_c, descs[b].cstack_off + descs[b].nc);
        ext_s = std::max(ext_s, descs[b].sstack_off + descs[b].ns);
    }
    GPSCAL_HIP(ctx, d_prev5.alloc_async((size_t)(ext_c + ext_s) * 5 + 8, ctx->stream));
    GPSCAL_HIP(ctx, hipMemsetAsync(d_prev5.p, 0xff, sizeof(int) * ((size_t)(ext_c + ext_s) * 5 + 8), ctx->stream));
    GPSCAL_HIP(ctx, d_sw.alloc_async(nsweeps, ctx->stream));
    GPSCAL_HIP(ctx, d_st.alloc_async(nsweeps, ctx->stream));
    GPSCAL_HIP(ctx, d_part.alloc_async((size_t)nsweeps * tiles_max * LSUMS, ctx->stream));
    GPSCAL_HIP(ctx, hipMemcpyAsync(d_sw.p, descs, sizeof(MapDesc) * nsweeps, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(lm_init_kernel, dim3(div_up(nsweeps, 64)), dim3(64), 0, ctx->stream, d_sw.p, nsweeps, d_tr_in,
                       d_st.p);
#ifdef GPSCAL_STATS
    hipLaunchKernelGGL(loam_stat_slot_kernel, dim3(1), dim3(1), 0, ctx->stream, 2);
#endif
    for (int it = 0; it < 10; ++it) {  // LM:752; converged sweeps return at once
        hipLaunchKernelGGL(lm_point_kernel, dim3(tiles_max, nsweeps), dim3(PT_BLOCK), 0, ctx->stream, d_sw.p, d_cstack,
                           d_sstack, d_cmap, d_smap, cgp, all.sorted.p, all.cell_start, sgp, all.sorted.p,
                           all.cell_start, d_st.p, d_part.p, tiles_max, d_prev5.p, ext_c);
        hipLaunchKernelGGL(lm_solve_kernel, dim3(nsweeps), dim3(64), 0, ctx->stream, d_sw.p, d_st.p, d_part.p,
                           tiles_max, it);
    }
    hipLaunchKernelGGL(iter_finish_kernel, dim3(div_up(nsweeps, 64)), dim3(64), 0, ctx->stream, d_st.p, nsweeps,
                       d_tr_out, d_iters, d_nsel);
    GPSCAL_HIP(ctx, hipGetLastError());
    if (!pool_grids()) GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));  // see loam_odometry_device
    return GPSCAL_OK;
}

}  // namespace gpscal

extern "C" int gpscal_loam_mapping_batched(gpscal_ctx *ctx, int nsweeps, const float *corner_stack_xyzi,
                                           const int *corner_stack_off, const float *surf_stack_xyzi,
                                           const int *surf_stack_off, const float *corner_map_xyzi,
                                           const int *corner_map_off, const float *surf_map_xyzi,
                                           const int *surf_map_off, const float *transform_in, float *transform_out,
                                           int *iters_out, int *nsel_out)
{
    if (!ctx || nsweeps < 1 || !corner_stack_xyzi || !corner_stack_off || !surf_stack_xyzi || !surf_stack_off ||
        !corner_map_xyzi || !corner_map_off || !surf_map_xyzi || !surf_map_off || !transform_in || !transform_out)
        return fail(ctx, GPSCAL_EINVAL, "gpscal_loam_mapping_batched: bad argument");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    const int tc = corner_stack_off[nsweeps], ts = surf_stack_off[nsweeps], tcm = corner_map_off[nsweeps],
              tsm = surf_map_off[nsweeps];
    std::vector<MapDesc> hs(nsweeps);
    std::vector<long long> coff(nsweeps + 1), soff(nsweeps + 1);
    for (int b = 0; b <= nsweeps; ++b) {
        coff[b] = corner_map_off[b];
        soff[b] = surf_map_off[b];
    }
    for (int b = 0; b < nsweeps; ++b) {
        MapDesc &D = hs[b];
        D.cstack_off = corner_stack_off[b];
        D.sstack_off = surf_stack_off[b];
        D.cmap_off = corner_map_off[b];
        D.smap_off = surf_map_off[b];
        D.nc = corner_stack_off[b + 1] - corner_stack_off[b];
        D.ns = surf_stack_off[b + 1] - surf_stack_off[b];
        D.mc = corner_map_off[b + 1] - corner_map_off[b];
        D.ms = surf_map_off[b + 1] - surf_map_off[b];
        if (D.nc < 0 || D.ns < 0 || D.mc < 0 || D.ms < 0) return fail(ctx, GPSCAL_EINVAL, "bad offsets");
    }
    InArg<float> a_cs, a_ss, a_cm, a_sm, a_tr;
    OutArg<float> o_tr;
    OutArg<int> o_it, o_ns;
    GPSCAL_HIP(ctx, a_cs.bind(ctx, corner_stack_xyzi, (size_t)std::max(tc, 1) * 4));
    GPSCAL_HIP(ctx, a_ss.bind(ctx, surf_stack_xyzi, (size_t)std::max(ts, 1) * 4));
    GPSCAL_HIP(ctx, a_cm.bind(ctx, corner_map_xyzi, (size_t)std::max(tcm, 1) * 4));
    GPSCAL_HIP(ctx, a_sm.bind(ctx, surf_map_xyzi, (size_t)std::max(tsm, 1) * 4));
    GPSCAL_HIP(ctx, a_tr.bind(ctx, transform_in, (size_t)nsweeps * 6));
    GPSCAL_HIP(ctx, o_tr.bind(ctx, transform_out, (size_t)nsweeps * 6));
    GPSCAL_HIP(ctx, o_it.bind(ctx, iters_out, iters_out ? nsweeps : 0));
    GPSCAL_HIP(ctx, o_ns.bind(ctx, nsel_out, nsel_out ? nsweeps : 0));
    int rc = loam_mapping_device(ctx, nsweeps, hs.data(), reinterpret_cast<const float4 *>(a_cs.dev),
                                 reinterpret_cast<const float4 *>(a_ss.dev), reinterpret_cast<const float4 *>(a_cm.dev),
                                 reinterpret_cast<const float4 *>(a_sm.dev), coff.data(), soff.data(), a_tr.dev,
                                 o_tr.dev, o_it.dev, o_ns.dev);
    if (rc) return rc;
    bool sync = true;
    GPSCAL_HIP(ctx, o_tr.commit(ctx, &sync));
    GPSCAL_HIP(ctx, o_it.commit(ctx, &sync));
    GPSCAL_HIP(ctx, o_ns.commit(ctx, &sync));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}

extern "C" int gpscal_loam_transform(gpscal_ctx *ctx, const float *tr6, const float *pts_xyzi, int n, float *out_xyzi,
                                     int to_end)
{
    if (!ctx || !tr6 || !pts_xyzi || !out_xyzi || n < 1) return fail(ctx, GPSCAL_EINVAL, "gpscal_loam_transform: bad argument");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    InArg<float> t, p;
    OutArg<float> o;
    GPSCAL_HIP(ctx, t.bind(ctx, tr6, 6));
    GPSCAL_HIP(ctx, p.bind(ctx, pts_xyzi, (size_t)n * 4));
    GPSCAL_HIP(ctx, o.bind(ctx, out_xyzi, (size_t)n * 4));
    hipLaunchKernelGGL(loam_to_end_kernel, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, t.dev,
                       reinterpret_cast<const float4 *>(p.dev), n, reinterpret_cast<float4 *>(o.dev), to_end);
    GPSCAL_HIP(ctx, hipGetLastError());
    bool sync = true;
    GPSCAL_HIP(ctx, o.commit(ctx, &sync));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}
