// track.hip -- weighted GPS<->SLAM track alignment, one workgroup per segment,
// float64 throughout, segment state resident in LDS.  gfx950 only.
//
// Replaces (reference files under /root/reference/src/gpsCalibration/src/):
//   WeightCoeCal::ICPWeightCoeCal x2   gps_calibration/weight_calculation.cc:4-27, 30-78
//   trackCalibration ctor/doICP/doCalibration
//                                      gps_calibration/track_calibration.cc:4-37,40-94,
//                                      97-201, 366-545, 555-588, 591-625, 631-689
//   longDisTrackPro body               long_distance_track_process/long_distance_track_process.cpp:58-83
//
// What is restated rather than transliterated:
//   * every row of the reference's N x 4 matrices has z = 1, so H's third row and
//     column are exactly zero and the 3x3 JacobiSVD + reflection fix reduces to a
//     2-D orthogonal Procrustes fit: R2 = V2 U2^T is a rotation when det H2 >= 0
//     and a REFLECTION otherwise (the reference's det fix only flips R(2,2));
//   * calibrateGPSWithSLAMTrack's O(N^2) loop equals
//     out_i = ((mean_j(E_j - S_j) + S_i) + S_i)/2 + E0, computed in O(N).
// Per-segment traffic: 5*8*N bytes in, <= 9*8*N out: negligible; the
// kernel is latency/sync-bound, so the lever is one launch for ALL segments.
#include "common.hpp"
#include "wave_reduce.hpp"

#include <algorithm>

namespace gpscal {

constexpr int TBLOCK = 256;
constexpr int TWAVES = TBLOCK / 64;
constexpr int TARRAYS = 8;  // ex ey cx cy w sp px py

template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double (*red)[8])
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double x = wave_sum(v[k]);
        if (lane == 0) red[wave][k] = x;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double x = 0.0;
#pragma unroll
        for (int w = 0; w < TWAVES; ++w) x += red[w][k];
        v[k] = x;
    }
    __syncthreads();
}

struct Fit2D {
    double r00, r01, r10, r11, t0, t1;
    bool reflection;
};

// BFTWithWeight on (ax,ay,1) -> (bx,by,1), TC:366-545.  Optionally also returns
// sum_i ||a_i - b_i|| (the pre-update distances of nearestNeighbor, TC:578-583).
template <bool WITH_DIST, class FA>
__device__ __forceinline__ Fit2D bft2d(FA a_of, const double *bx, const double *by, const double *w, int n,
                                       double (*red)[8], double *dist_sum)
{
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += TBLOCK) {
        double ax, ay;
        a_of(i, ax, ay);
        double wi = w[i], qx = bx[i], qy = by[i];
        s[0] += wi;          // TC:439
        s[1] += ax * wi;     // TC:418,430
        s[2] += ay * wi;
        s[3] += qx * wi;
        s[4] += qy * wi;
        if (WITH_DIST) {
            double dx = ax - qx, dy = ay - qy;
            s[5] += sqrt(dx * dx + dy * dy);
        }
    }
    block_sum<6>(s, red);
    if (WITH_DIST) *dist_sum = s[5];
    const double cax = s[1] / s[0], cay = s[2] / s[0];  // TC:451-457
    const double cbx = s[3] / s[0], cby = s[4] / s[0];
    double h[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < n; i += TBLOCK) {
        double ax, ay;
        a_of(i, ax, ay);
        double wi = w[i];
        double aax = (ax - cax) * wi, aay = (ay - cay) * wi;      // TC:490-503
        double bbx = (bx[i] - cbx) * wi, bby = (by[i] - cby) * wi;
        h[0] += aax * bbx;  // H = AA^T BB, TC:506
        h[1] += aax * bby;
        h[2] += aay * bbx;
        h[3] += aay * bby;
    }
    block_sum<4>(h, red);
    Fit2D F;
    // 2-D orthogonal Procrustes = V2 U2^T of the reference's SVD (TC:508-523)
    const double det = h[0] * h[3] - h[1] * h[2];
    double c, sn;
    F.reflection = det < 0.0;
    if (!F.reflection) {
        c = h[0] + h[3];
        sn = h[1] - h[2];
    } else {
        c = h[0] - h[3];
        sn = h[1] + h[2];
    }
    double q = sqrt(c * c + sn * sn);
    if (q > 0.0) {
        c /= q;
        sn /= q;
    } else {
        c = 1.0;
        sn = 0.0;
        F.reflection = false;
    }
    if (!F.reflection) {
        F.r00 = c;  F.r01 = -sn;
        F.r10 = sn; F.r11 = c;
    } else {
        F.r00 = c;  F.r01 = sn;
        F.r10 = sn; F.r11 = -c;
    }
    F.t0 = cbx - (F.r00 * cax + F.r01 * cay);  // TC:526
    F.t1 = cby - (F.r10 * cax + F.r11 * cay);
    return F;
}

__device__ __forceinline__ double speed_weight(const double *slam, int n, int i)
{
    // WC:10-22; the slot one past the end reads as (0,0) (zero-filled spare
    // vector capacity, SURVEY 8c -- same rule as the oracle).
    if (i == 0) return 1.0;
    double nx = 0.0, ny = 0.0;
    if (i + 1 < n) {
        nx = slam[4 * (size_t)(i + 1) + 0];
        ny = slam[4 * (size_t)(i + 1) + 1];
    }
    double dx = nx - slam[4 * (size_t)i + 0], dy = ny - slam[4 * (size_t)i + 1];
    double v = sqrt(dx * dx + dy * dy) / 2.2;
    return v < 1.0 ? v : 1.0;
}

// LONG = false: one fit with the caller's weights (trackCalibration API).
// LONG = true : speed weights -> fit -> irls_iters x {IRLS weights -> re-fit}.
template <bool LONG>
__global__ __launch_bounds__(TBLOCK) void track_fit_kernel(const double *__restrict__ slam,
                                                            const double *__restrict__ enu,
                                                            const double *__restrict__ w_in,
                                                            const int *__restrict__ seg_off, int irls_iters,
                                                            double *__restrict__ T_out, double *__restrict__ rot_out,
                                                            double *__restrict__ cal_out, double *__restrict__ w_out,
                                                            double *__restrict__ scratch, int lds_cap)
{
    extern __shared__ double lds[];
    __shared__ double red[TWAVES][8];
    const int seg = blockIdx.x;
    const int o = seg_off[seg];
    const int n = seg_off[seg + 1] - o;
    if (n <= 0) return;
    // segment state: LDS when it fits, else this block's slice of the HBM scratch
    double *base = (n <= lds_cap) ? lds : scratch + (size_t)o * TARRAYS;
    const int cap = (n <= lds_cap) ? lds_cap : n;
    double *ex = base, *ey = base + cap, *cx = base + 2 * (size_t)cap, *cy = base + 3 * (size_t)cap;
    double *w = base + 4 * (size_t)cap, *sp = base + 5 * (size_t)cap, *px = base + 6 * (size_t)cap,
           *py = base + 7 * (size_t)cap;
    const double *S = slam + 4 * (size_t)o, *E = enu + 4 * (size_t)o;
    const double E0x = E[0], E0y = E[1];  // ENUX0/ENUY0, TC:62-63

    for (int i = threadIdx.x; i < n; i += TBLOCK) {
        ex[i] = E[4 * (size_t)i + 0] - E0x;  // TC:64-68
        ey[i] = E[4 * (size_t)i + 1] - E0y;
        px[i] = S[4 * (size_t)i + 0];
        py[i] = S[4 * (size_t)i + 1];
        if (LONG) {
            sp[i] = speed_weight(S, n, i);  // LD:60
            w[i] = sp[i];
        } else {
            w[i] = w_in[o + i];
        }
    }
    __syncthreads();

    const int nfit = LONG ? 1 + irls_iters : 1;
    for (int f = 0; f < nfit; ++f) {
        if (LONG && f > 0) {
            // IRLS weights (WC:68-75) against the previous calibrated track (LD:76)
            for (int i = threadIdx.x; i < n; i += TBLOCK) {
                double dx = E[4 * (size_t)i + 0] - px[i], dy = E[4 * (size_t)i + 1] - py[i];
                double d = sqrt(dx * dx + dy * dy);
                w[i] = sp[i] * 1.0 / (d > 0.01 ? d : 0.01);
            }
        }
        const double S0x = px[0], S0y = py[0];  // TC:56-57
        __syncthreads();
        auto s_of = [&](int i, double &ax, double &ay) {
            ax = px[i] - S0x;
            ay = py[i] - S0y;
        };
        for (int i = threadIdx.x; i < n; i += TBLOCK) s_of(i, cx[i], cy[i]);  // src, TC:123-134
        __syncthreads();
        auto c_of = [&](int i, double &ax, double &ay) {
            ax = cx[i];
            ay = cy[i];
        };
        double prev = 0.0;
        for (int it = 0; it < 2; ++it) {  // TC:145-181
            double dsum = 0.0;
            Fit2D F = bft2d<true>(c_of, ex, ey, w, n, red, &dsum);
            for (int i = threadIdx.x; i < n; i += TBLOCK) {  // src = src * T^T, TC:165
                double x = cx[i], y = cy[i];
                cx[i] = (x * F.r00 + y * F.r01) + F.t0;
                cy[i] = (x * F.r10 + y * F.r11) + F.t1;
            }
            __syncthreads();
            double mean = dsum / (double)n;
            if (fabs(prev - mean) < 0.003) break;  // TC:176 (block-uniform)
            prev = mean;
        }
        Fit2D F = bft2d<false>(s_of, cx, cy, w, n, red, nullptr);  // TC:189
        // rotated SLAM track (TC:622) overwrites the working copy
        for (int i = threadIdx.x; i < n; i += TBLOCK) {
            double x, y;
            s_of(i, x, y);
            cx[i] = (x * F.r00 + y * F.r01) + F.t0;
            cy[i] = (x * F.r10 + y * F.r11) + F.t1;
        }
        __syncthreads();
        // calibration (TC:631-689) in closed form
        double m[2] = {0, 0};
        for (int i = threadIdx.x; i < n; i += TBLOCK) {
            m[0] += ex[i] - cx[i];
            m[1] += ey[i] - cy[i];
        }
        block_sum<2>(m, red);
        const double mx = m[0] / (double)n, my = m[1] / (double)n;
        const bool last = f == nfit - 1;
        for (int i = threadIdx.x; i < n; i += TBLOCK) {
            double rx = cx[i], ry = cy[i];
            double calx = ((mx + rx) + rx) / 2.0 + E0x;  // TC:670,680
            double caly = ((my + ry) + ry) / 2.0 + E0y;
            px[i] = calx;  // next fit's source (LD:78)
            py[i] = caly;
            if (last) {
                if (rot_out) {
                    rot_out[3 * (size_t)(o + i) + 0] = rx;
                    rot_out[3 * (size_t)(o + i) + 1] = ry;
                    rot_out[3 * (size_t)(o + i) + 2] = 1.0;  // z row: R22 + t_z == 1 always
                }
                if (cal_out) {
                    cal_out[4 * (size_t)(o + i) + 0] = calx;
                    cal_out[4 * (size_t)(o + i) + 1] = caly;
                    cal_out[4 * (size_t)(o + i) + 2] = E[4 * (size_t)i + 2];  // TC:84,682
                    cal_out[4 * (size_t)(o + i) + 3] = E[4 * (size_t)i + 3];  // TC:85,683
                }
                if (w_out) w_out[o + i] = w[i];
            }
        }
        if (last && T_out && threadIdx.x == 0) {
            // 4x4 as the reference assembles it (TC:529-542): the z block is
            // R22 = -1, t_z = 2 in the reflection case, else 1, 0.
            double *T = T_out + 16 * (size_t)seg;
            const double r22 = F.reflection ? -1.0 : 1.0;
            T[0] = F.r00; T[1] = F.r01; T[2] = 0.0; T[3] = F.t0;
            T[4] = F.r10; T[5] = F.r11; T[6] = 0.0; T[7] = F.t1;
            T[8] = 0.0;   T[9] = 0.0;   T[10] = r22; T[11] = 1.0 - r22;
            T[12] = 0.0;  T[13] = 0.0;  T[14] = 0.0; T[15] = 1.0;
        }
        __syncthreads();
    }
}

__global__ void weights_speed_kernel(const double *__restrict__ slam, int n, double *__restrict__ w)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[i] = speed_weight(slam, n, i);
}

__global__ void weights_irls_kernel(const double *__restrict__ slam, const double *__restrict__ enu,
                                    const double *__restrict__ fit, int n, double *__restrict__ w)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double dx = enu[4 * (size_t)i] - fit[4 * (size_t)i], dy = enu[4 * (size_t)i + 1] - fit[4 * (size_t)i + 1];
    double d = sqrt(dx * dx + dy * dy);
    w[i] = speed_weight(slam, n, i) * 1.0 / (d > 0.01 ? d : 0.01);
}

// TM:116-157: sequential by construction (running sum); one lane per chain.
__global__ void height_compensate_kernel(const double *__restrict__ p, int n, double *__restrict__ out)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double ppx = 0, ppy = 0, ppz = 0, tx = 0, ty = 0;
    for (int i = 0; i < n; ++i) {
        double cx = p[4 * (size_t)i + 2], cy = p[4 * (size_t)i + 0], cz = p[4 * (size_t)i + 1];
        if (i == 0) {
            tx = cx;
            ty = cy;
        } else {
            double dx = cx - ppx, dy = cy - ppy, dz = cz - ppz;
            double n3 = sqrt(dx * dx + dy * dy + dz * dz), n2 = sqrt(dx * dx + dy * dy);
            tx += dx * n3 / n2;
            ty += dy * n3 / n2;
        }
        ppx = cx; ppy = cy; ppz = cz;
        out[4 * (size_t)i + 0] = tx;
        out[4 * (size_t)i + 1] = ty;
        out[4 * (size_t)i + 2] = 10.0;
        out[4 * (size_t)i + 3] = p[4 * (size_t)i + 3];
    }
}

static int run_track(gpscal_ctx *ctx, bool is_long, const double *slam, const double *enu, const double *w,
                     const int *seg_off_host, int nseg, int irls_iters, double *T, double *rot, double *cal,
                     double *w_out)
{
    if (!ctx || !slam || !enu || !seg_off_host || nseg < 1 || (!is_long && !w) || irls_iters < 0)
        return fail(ctx, GPSCAL_EINVAL, "track fit: bad argument");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    const int total = seg_off_host[nseg];
    int nmax = 0;
    for (int s = 0; s < nseg; ++s) {
        int n = seg_off_host[s + 1] - seg_off_host[s];
        if (n < 1) return fail(ctx, GPSCAL_ESIZE, "track fit: empty segment");
        nmax = std::max(nmax, n);
    }
    InArg<double> a_slam, a_enu, a_w;
    OutArg<double> o_T, o_rot, o_cal, o_w;
    DevBuf<int> d_off;
    GPSCAL_HIP(ctx, a_slam.bind(ctx, slam, (size_t)total * 4));
    GPSCAL_HIP(ctx, a_enu.bind(ctx, enu, (size_t)total * 4));
    GPSCAL_HIP(ctx, a_w.bind(ctx, w, is_long ? 0 : (size_t)total));
    GPSCAL_HIP(ctx, o_T.bind(ctx, T, (size_t)nseg * 16));
    GPSCAL_HIP(ctx, o_rot.bind(ctx, rot, (size_t)total * 3));
    GPSCAL_HIP(ctx, o_cal.bind(ctx, cal, (size_t)total * 4));
    GPSCAL_HIP(ctx, o_w.bind(ctx, w_out, (size_t)total));
    GPSCAL_HIP(ctx, d_off.alloc(nseg + 1));
    GPSCAL_HIP(ctx, hipMemcpyAsync(d_off.p, seg_off_host, sizeof(int) * (nseg + 1), hipMemcpyHostToDevice, ctx->stream));
    // LDS budget: 8 arrays of nmax doubles, capped below the 160 KiB CU limit
    const int lds_limit = 144 * 1024;
    int cap = std::min(nmax, lds_limit / (TARRAYS * 8));
    size_t lds_bytes = (size_t)cap * TARRAYS * 8;
    DevBuf<double> scratch;
    if (nmax > cap) GPSCAL_HIP(ctx, scratch.alloc((size_t)total * TARRAYS));
    if (is_long) {
        GPSCAL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&track_fit_kernel<true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL(track_fit_kernel<true>, dim3(nseg), dim3(TBLOCK), lds_bytes, ctx->stream, a_slam.dev,
                           a_enu.dev, a_w.dev, d_off.p, irls_iters, o_T.dev, o_rot.dev, o_cal.dev, o_w.dev, scratch.p,
                           cap);
    } else {
        GPSCAL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(&track_fit_kernel<false>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        hipLaunchKernelGGL(track_fit_kernel<false>, dim3(nseg), dim3(TBLOCK), lds_bytes, ctx->stream, a_slam.dev,
                           a_enu.dev, a_w.dev, d_off.p, 0, o_T.dev, o_rot.dev, o_cal.dev, o_w.dev, scratch.p, cap);
    }
    GPSCAL_HIP(ctx, hipGetLastError());
    bool sync = true;  // staged inputs / d_off / scratch die with this frame
    GPSCAL_HIP(ctx, o_T.commit(ctx, &sync));
    GPSCAL_HIP(ctx, o_rot.commit(ctx, &sync));
    GPSCAL_HIP(ctx, o_cal.commit(ctx, &sync));
    GPSCAL_HIP(ctx, o_w.commit(ctx, &sync));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}

}  // namespace gpscal

using namespace gpscal;

extern "C" int gpscal_weights_speed(gpscal_ctx *ctx, const double *slam, int n, double *w)
{
    if (!ctx || !slam || !w || n < 1) return fail(ctx, GPSCAL_EINVAL, "gpscal_weights_speed: bad argument");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    InArg<double> a;
    OutArg<double> o;
    GPSCAL_HIP(ctx, a.bind(ctx, slam, (size_t)n * 4));
    GPSCAL_HIP(ctx, o.bind(ctx, w, n));
    hipLaunchKernelGGL(weights_speed_kernel, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, a.dev, n, o.dev);
    GPSCAL_HIP(ctx, hipGetLastError());
    bool sync = true;
    GPSCAL_HIP(ctx, o.commit(ctx, &sync));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}

extern "C" int gpscal_weights_irls(gpscal_ctx *ctx, const double *slam, const double *enu, const double *fit, int n,
                                   double *w)
{
    if (!ctx || !slam || !enu || !fit || !w || n < 1) return fail(ctx, GPSCAL_EINVAL, "gpscal_weights_irls: bad argument");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    InArg<double> a, b, c;
    OutArg<double> o;
    GPSCAL_HIP(ctx, a.bind(ctx, slam, (size_t)n * 4));
    GPSCAL_HIP(ctx, b.bind(ctx, enu, (size_t)n * 4));
    GPSCAL_HIP(ctx, c.bind(ctx, fit, (size_t)n * 4));
    GPSCAL_HIP(ctx, o.bind(ctx, w, n));
    hipLaunchKernelGGL(weights_irls_kernel, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, a.dev, b.dev, c.dev, n,
                       o.dev);
    GPSCAL_HIP(ctx, hipGetLastError());
    bool sync = true;
    GPSCAL_HIP(ctx, o.commit(ctx, &sync));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}

extern "C" int gpscal_track_fit(gpscal_ctx *ctx, const double *slam, const double *enu, const double *w, int n,
                                double *T, double *rot, double *cal)
{
    int off[2] = {0, n};
    if (n < 1) return fail(ctx, GPSCAL_ESIZE, "gpscal_track_fit: n < 1");
    return run_track(ctx, false, slam, enu, w, off, 1, 0, T, rot, cal, nullptr);
}

extern "C" int gpscal_track_fit_batched(gpscal_ctx *ctx, const double *slam, const double *enu, const double *w,
                                        const int *seg_offsets, int nseg, double *T, double *rot, double *cal)
{
    return run_track(ctx, false, slam, enu, w, seg_offsets, nseg, 0, T, rot, cal, nullptr);
}

extern "C" int gpscal_long_segment(gpscal_ctx *ctx, const double *slam, const double *enu, int n, int irls_iters,
                                   double *w_out, double *fit_out)
{
    int off[2] = {0, n};
    if (n < 1) return fail(ctx, GPSCAL_ESIZE, "gpscal_long_segment: n < 1");
    return run_track(ctx, true, slam, enu, nullptr, off, 1, irls_iters, nullptr, nullptr, fit_out, w_out);
}

extern "C" int gpscal_long_segment_batched(gpscal_ctx *ctx, const double *slam, const double *enu,
                                           const int *seg_offsets, int nseg, int irls_iters, double *w_out,
                                           double *fit_out)
{
    return run_track(ctx, true, slam, enu, nullptr, seg_offsets, nseg, irls_iters, nullptr, nullptr, fit_out, w_out);
}

extern "C" int gpscal_height_compensate(gpscal_ctx *ctx, const double *loam, int n, double *out)
{
    if (!ctx || !loam || !out || n < 1) return fail(ctx, GPSCAL_EINVAL, "gpscal_height_compensate: bad argument");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    InArg<double> a;
    OutArg<double> o;
    GPSCAL_HIP(ctx, a.bind(ctx, loam, (size_t)n * 4));
    GPSCAL_HIP(ctx, o.bind(ctx, out, (size_t)n * 4));
    hipLaunchKernelGGL(height_compensate_kernel, dim3(1), dim3(64), 0, ctx->stream, a.dev, n, o.dev);
    GPSCAL_HIP(ctx, hipGetLastError());
    bool sync = true;
    GPSCAL_HIP(ctx, o.commit(ctx, &sync));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}
