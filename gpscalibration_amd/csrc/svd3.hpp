// svd3.hpp -- float64 3x3 two-sided Jacobi SVD and the Kabsch rotation, device
// code.  Semantics of Eigen::JacobiSVD + R = V U^T + reflection fix as used by
// trackCalibration::BFTWithWeight (track_calibration.cc:506-523).
#pragma once
#include <hip/hip_runtime.h>

namespace gpscal {

// 1/sqrt(x), sqrt(x) and 1/x in float64 for the Jacobi steps: the hardware estimates (v_rsq_f64 / v_rcp_f64, ~2^-27)
// + two Newton steps each, ~8 instructions against the ~25-30 of the library routines (which also serve denormals,
// infinities and the last half ulp).  The solve kernel is ONE lane working through ~15 Jacobi steps of five such
// operations each, every instruction a 4-cycle issue slot on a dependent chain: they were half of its SVD time.  The
// arguments here are >= 1 (1 + t^2), sums of squares of a pre-scaled matrix's entries that the callers have compared
// against `tiny`, or 2b with |b| >= tiny; results are good to ~1 ulp.
__device__ __forceinline__ double fast_rsqrt(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    y = y * __fma_rn(-0.5 * x * y, y, 1.5);
    y = y * __fma_rn(-0.5 * x * y, y, 1.5);
    return y;
}
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = r * __fma_rn(-x, r, 2.0);
    r = r * __fma_rn(-x, r, 2.0);
    return r;
}

__device__ __forceinline__ void rot_rows_T(double *W, int i, int j, double c, double s)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double a = W[3 * i + k], b = W[3 * j + k];
        W[3 * i + k] = c * a + s * b;
        W[3 * j + k] = -s * a + c * b;
    }
}
__device__ __forceinline__ void rot_cols(double *M, int i, int j, double c, double s)
{
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double a = M[3 * k + i], b = M[3 * k + j];
        M[3 * k + i] = c * a + s * b;
        M[3 * k + j] = -s * a + c * b;
    }
}

// One (i,j) Kogbetliantz step; returns true when a rotation was applied.
__device__ __forceinline__ bool jacobi_pair(double *W, double *U, double *V, int i, int j)
{
    const double prec = 4.440892098500626e-16;  // 2 * DBL_EPSILON
    const double tiny = 2.2250738585072014e-308;
    double wii = W[3 * i + i], wjj = W[3 * j + j];
    double wij = W[3 * i + j], wji = W[3 * j + i];
    double thr = prec * fmax(fabs(wii), fabs(wjj));
    thr = fmax(thr, tiny);
    if (!(fabs(wij) > thr || fabs(wji) > thr)) return false;
    double t = wii + wjj, d = wji - wij;
    double c1 = 1.0, s1 = 0.0;
    if (fabs(d) >= tiny) {
        double rh = fast_rsqrt(t * t + d * d);  // matrix is pre-scaled: no overflow
        c1 = t * rh;
        s1 = d * rh;
    }
    double a = c1 * wii + s1 * wji;
    double b = c1 * wij + s1 * wjj;
    double e = -s1 * wij + c1 * wjj;
    double cj = 1.0, sj = 0.0;
    if (fabs(b) >= tiny) {
        double tau = (e - a) * fast_rcp(2.0 * b);
        tau = fmin(fmax(tau, -1e300), 1e300);  // (b may be as small as `tiny`: the product can overflow)
        double tj;
        if (fabs(tau) > 1e150) {  // 1 + tau^2 would overflow: the limit of the formula below
            tj = 0.5 * fast_rcp(tau);
        } else {
            const double q = 1.0 + tau * tau;
            tj = (tau >= 0.0 ? 1.0 : -1.0) * fast_rcp(fabs(tau) + q * fast_rsqrt(q));
        }
        cj = fast_rsqrt(1.0 + tj * tj);
        sj = tj * cj;
    }
    double cl = c1 * cj + s1 * sj;
    double sl = s1 * cj - c1 * sj;
    rot_rows_T(W, i, j, cl, sl);
    rot_cols(W, i, j, cj, -sj);
    rot_cols(U, i, j, cl, sl);
    rot_cols(V, i, j, cj, -sj);
    return true;
}

template <int K, int L>
__device__ __forceinline__ void sv_cswap(double *U, double *S, double *V)
{
    if (S[L] > S[K]) {
        double ts = S[K];
        S[K] = S[L];
        S[L] = ts;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            double tu = U[3 * r + K];
            U[3 * r + K] = U[3 * r + L];
            U[3 * r + L] = tu;
            double tv = V[3 * r + K];
            V[3 * r + K] = V[3 * r + L];
            V[3 * r + L] = tv;
        }
    }
}

// A = U diag(S) V^T, S descending and >= 0.  All matrices row-major.
__device__ inline void svd3(const double *A, double *U, double *S, double *V)
{
    double W[9];
    double scale = 0.0;
#pragma unroll
    for (int k = 0; k < 9; ++k) scale = fmax(scale, fabs(A[k]));
    if (scale == 0.0) scale = 1.0;
    double inv = 1.0 / scale;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        W[k] = A[k] * inv;
        U[k] = V[k] = (k % 4 == 0) ? 1.0 : 0.0;
    }
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool any = false;
        any |= jacobi_pair(W, U, V, 0, 1);
        any |= jacobi_pair(W, U, V, 0, 2);
        any |= jacobi_pair(W, U, V, 1, 2);
        if (!any) break;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double s = W[4 * k];
        if (s < 0.0) {
            s = -s;
            U[k] = -U[k];
            U[3 + k] = -U[3 + k];
            U[6 + k] = -U[6 + k];
        }
        S[k] = s * scale;
    }
    // descending order: 3-element sorting network with column swaps (static
    // indices only, so everything stays in registers)
    sv_cswap<0, 1>(U, S, V);
    sv_cswap<1, 2>(U, S, V);
    sv_cswap<0, 1>(U, S, V);
}

__device__ __forceinline__ double det3(const double *M)
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// R = V U^T; if det R < 0 negate column 2 of V and recompute (TC:513-523).
__device__ inline void kabsch_from_H(const double *H, double *R)
{
    double U[9], S[3], V[9];
    svd3(H, U, S, V);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            R[3 * r + c] = V[3 * r] * U[3 * c] + V[3 * r + 1] * U[3 * c + 1] + V[3 * r + 2] * U[3 * c + 2];
    if (det3(R) < 0.0) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                R[3 * r + c] = V[3 * r] * U[3 * c] + V[3 * r + 1] * U[3 * c + 1] - V[3 * r + 2] * U[3 * c + 2];
    }
}

}  // namespace gpscal
