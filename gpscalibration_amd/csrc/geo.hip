// geo.hip -- WGS84 <-> local projection (UTM / Gauss-Krueger as coded by the
// reference) and GPS-at-SLAM-stamp interpolation, one lane per sample, float64.
// gfx950 only.
//
// Replaces (under /root/reference/src/gpsCalibration/src/gps_calibration/):
//   GPSPro::UTMTransform            gps_process.cc:851-908
//   GPSPro::GaussionTransform       gps_process.cc:953-1007 (+ arcLength :38-56)
//   GPSPro::UTMReverseTransform     gps_process.cc:1010-1058
//   GPSPro::GaussionReverseTransform gps_process.cc:911-950
//   GPSPro::interPolate             gps_process.cc:59-110
//   tail of GPSPro::GPSToENU        gps_process.cc:498-518
// Reference quirks kept: PI = 3.141592653589 (common.h:17); WGS84 b = 6356752.314
// (gps_process.cc:1114); UTM northing's A^6 term outside N*tan (gps_process.cc:899);
// band number from the first fix only (gps_process.cc:869-877).
#include "common.hpp"

#include <algorithm>

namespace gpscal {

constexpr double REF_PI = 3.141592653589;
constexpr double WGS_A = 6378137.0;
constexpr double WGS_B = 6356752.314;

struct Ellipsoid {
    double e1, e2, c;  // WGSParameter::E1, E2, C (gps_process.cc:1115-1117)
};
__host__ __device__ inline Ellipsoid ellipsoid()
{
    Ellipsoid e;
    double d = sqrt(WGS_A * WGS_A - WGS_B * WGS_B);
    e.e1 = d / WGS_A;
    e.e2 = d / WGS_B;
    e.c = WGS_A * WGS_A / WGS_B;
    return e;
}

__device__ __forceinline__ int band_of(double lon, int band_type)
{
    if (band_type == 3) {
        int band = (int)(lon / 3);
        double tmp = lon / 3;
        if (tmp - band > 0.5) band += 1;
        return band;
    }
    return (int)lon / 6 + 1;
}

// "if (0 == bandNum)" re-evaluates the band until it becomes non-zero
// (gps_process.cc:869-877): band of sample i = first non-zero band among 0..i.
__device__ __forceinline__ int sticky_band(const double *lon, int i, int band_type)
{
    int b = band_of(lon[0], band_type);
    for (int j = 1; b == 0 && j <= i; ++j) b = band_of(lon[j], band_type);
    return b;
}

__device__ __forceinline__ double pw2(double x) { return x * x; }
__device__ __forceinline__ double pw3(double x) { return x * x * x; }
__device__ __forceinline__ double pw4(double x) { double y = x * x; return y * y; }
__device__ __forceinline__ double pw5(double x) { double y = x * x; return y * y * x; }
__device__ __forceinline__ double pw6(double x) { double y = x * x; return y * y * y; }

__device__ inline double arc_length(double latitude, const Ellipsoid &el)
{
    double e1s = pw2(el.e1);
    double m0 = WGS_A * (1 - e1s);
    double m2 = 3.0 / 2.0 * e1s * m0;
    double m4 = 5.0 / 4.0 * e1s * m2;
    double m6 = 7.0 / 6.0 * e1s * m4;
    double m8 = 9.0 / 8.0 * e1s * m6;
    double a0 = m0 + 1.0 / 2.0 * m2 + 3.0 / 8.0 * m4 + 5.0 / 16.0 * m6 + 35.0 / 128.0 * m8;
    double a2 = 1.0 / 2.0 * m2 + 1.0 / 2.0 * m4 + 15.0 / 32.0 * m6 + 7.0 / 16.0 * m8;
    double a4 = 1.0 / 8.0 * m4 + 3.0 / 16.0 * m6 + 7.0 / 32.0 * m8;
    double a6 = 1.0 / 32.0 * m6 + 1.0 / 16.0 * m8;
    double a8 = 1.0 / 128.0 * m8;
    double rB = latitude * REF_PI / 180.0;
    return a0 * rB - a2 / 2.0 * sin(2 * rB) + a4 / 4.0 * sin(4 * rB) - a6 / 6.0 * sin(6 * rB) +
           a8 / 8.0 * sin(8 * rB);
}

__device__ inline void project_fwd(int method, int band_type, int band, double lat, double lon, double &x, double &y)
{
    const Ellipsoid el = ellipsoid();
    double meridian = band_type == 3 ? 3.0 * band : (double)(6 * band - 3);
    double rB = lat * REF_PI / 180.0;
    if (method == GPSCAL_METHOD_UTM) {
        const double k0 = 0.9996;
        double tn = tan(rB), cs = cos(rB), sn = sin(rB);
        double t = tn * tn;
        double c = pw2(el.e2) * pw2(cs);
        double A = (lon - meridian) * REF_PI / 180.0 * cs;
        double N = WGS_A / sqrt(1 - el.e1 * el.e1 * sn * sn);
        double e12 = pw2(el.e1), e14 = pw4(el.e1), e16 = pw6(el.e1);
        double M = WGS_A * ((1 - e12 / 4.0 - 3.0 * e14 / 64.0 - 5.0 * e16 / 256.0) * rB -
                            (3.0 * e12 / 8.0 + 3.0 * e14 / 32.0 + 45.0 * e16 / 1024.0) * sin(2 * rB) +
                            (15.0 * e14 / 256.0 + 45.0 * e16 / 1024.0) * sin(4 * rB) -
                            35.0 * e16 / 3072.0 * sin(6 * rB));
        x = k0 * (M + N * tn * (A * A / 2.0 + (5 - t + 9 * c + 4 * c * c) * pw4(A) / 24.0) +
                  (61 - 58 * t + t * t + 600 * c - 330 * el.e2 * el.e2) * pw6(A) / 720.0);
        y = k0 * N * (A + (1 - t + c) * pw3(A) / 6.0 +
                      (5 - 18 * t + t * t + 72 * c - 58 * el.e2 * el.e2) * pw5(A) / 120.0) +
            500000;
    } else {
        double t = tan(rB), cs = cos(rB);
        double ng2 = pw2(el.e2) * pw2(cs);
        double N = el.c / sqrt(1 + ng2);
        double m = cs * REF_PI / 180.0 * (lon - meridian);
        double ml = arc_length(lat, el);
        x = ml + N * t * (1.0 / 2.0 * m * m + 1.0 / 24.0 * (5 - t * t + 9 * ng2 + 4 * ng2 * ng2) * pw4(m) +
                          1.0 / 720.0 * (61 - 58 * t * t + pw4(t) + 270 * ng2 - 330 * ng2 * t * t) * pw6(m));
        y = N * (m + 1.0 / 6.0 * (1 - t * t + ng2) * pw3(m) +
                 1.0 / 120.0 * (5 - 18 * t * t + pw4(t) + 14 * ng2 - 58 * ng2 * t * t) * pw5(m)) +
            500000;
    }
    y += (double)band * 10000000;
}

__global__ void wgs_to_enu_kernel(int method, int band_type, const double *__restrict__ lat,
                                  const double *__restrict__ lon, int n, double *__restrict__ xy)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int band = sticky_band(lon, i, band_type);
    double x, y;
    project_fwd(method, band_type, band, lat[i], lon[i], x, y);
    xy[2 * (size_t)i] = x;
    xy[2 * (size_t)i + 1] = y;
}

__global__ void enu_to_wgs_kernel(int method, int band_type, const double *__restrict__ enu, int n,
                                  double *__restrict__ lonlat, double *__restrict__ alt)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Ellipsoid el = ellipsoid();
    double ex = enu[5 * (size_t)i], ey = enu[5 * (size_t)i + 1];
    int band = (int)(ey / 10000000);
    double meridian = band_type == 3 ? 3.0 * band : (double)(6 * band - 3);
    double ly = ey - (double)band * 10000000 - 500000;
    const double k0 = method == GPSCAL_METHOD_UTM ? 0.9996 : 1.0;
    double X = method == GPSCAL_METHOD_UTM ? ex / k0 : ex;
    double e12 = pw2(el.e1), e14 = pw4(el.e1), e16 = pw6(el.e1);
    double fi = X / (WGS_A * (1 - e12 / 4 - 3 * e14 / 64 - 5 * e16 / 256));
    double e = (1 - WGS_B / WGS_A) / (1 + WGS_B / WGS_A);
    double Bf = fi + (3 * e / 2 - 27 * pw3(e) / 32) * sin(2 * fi) + (21 * e * e / 16 - 55 * pw4(e) / 32) * sin(4 * fi) +
                151 * pw3(e) / 96 * sin(6 * fi);
    double sB = sin(Bf), cB = cos(Bf), tB = tan(Bf);
    double q = 1 - el.e1 * el.e1 * sB * sB;
    double Nf = WGS_A / sqrt(q);
    double Rf = WGS_A * (1 - el.e1 * el.e1) / (q * sqrt(q));
    double Cf = el.e2 * el.e2 * cB * cB;
    double Tf = tB * tB;
    double latitude, longitude;
    if (method == GPSCAL_METHOD_UTM) {
        double D = ly / (k0 * Nf);
        latitude = Bf - Nf * tB / Rf *
                            (D * D / 2 - (5 + 3 * Tf + 10 * Cf - 4 * Cf * Cf - 9 * el.e2 * el.e2) * pw4(D) / 24.0 +
                             (61 + 90 * Tf + 298 * Cf + 45 * Tf * Tf - 252 * el.e2 * el.e2 - 3 * Cf * Cf) * pw6(D) / 720);
        longitude = meridian + (1.0 / cB *
                                (D - (1 + 2 * Tf + Cf) * pw3(D) / 6.0 +
                                 (5 - 2 * Cf + 28 * Tf - 3 * Cf * Cf + 8 * el.e2 * el.e2 + 24 * Tf * Tf) * pw5(D) / 120.0)) *
                                   180 / REF_PI;
    } else {
        double D = ly / Nf;
        latitude = Bf - Nf * tB / Rf *
                            (D * D / 2 - (5 + 3 * Tf + Cf - 9 * Tf * Cf) * pw4(D) / 24 +
                             (61 + 90 * Tf + 45 * Tf * Tf) * pw6(D) / 720);
        longitude = meridian + (1.0 / cB *
                                (D - (1 + 2 * Tf + Cf) * pw3(D) / 6 +
                                 (5 + 28 * Tf + 6 * Cf + 8 * Tf * Cf + 24 * Tf * Tf) * pw5(D) / 120)) *
                                   180 / REF_PI;
    }
    latitude = latitude * 180 / REF_PI;
    lonlat[2 * (size_t)i] = longitude;  // gps_process.cc:1053: (lon, lat)
    lonlat[2 * (size_t)i + 1] = latitude;
    alt[i] = enu[5 * (size_t)i + 2];
}

// interPolate (gps_process.cc:85-107) for time-ordered logs: stamp r belongs to
// the first interval s with slam_t[r] <= gps_t[s+1]; stamps past the last fix
// are dropped (counted through *n_kept).  Output rows gps_process.cc:510-518.
__global__ void interp_kernel(const double *__restrict__ xy, const double *__restrict__ gt, int ngps,
                              const double *__restrict__ slam, int nslam, double *__restrict__ enu,
                              int *__restrict__ n_kept)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nslam) return;
    double tr = slam[4 * (size_t)r + 3];
    if (ngps < 2 || tr > gt[ngps - 1]) return;  // dropped (gps_process.cc:99)
    int lo = 0, hi = ngps - 2;                  // smallest s with gt[s+1] >= tr
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (gt[mid + 1] >= tr) hi = mid; else lo = mid + 1;
    }
    int s = lo;
    double s1 = gt[s], s2 = gt[s + 1], s3 = s2 - s1;
    double c1 = (tr - s1) / s3, c2 = 1.0 - c1;
    enu[4 * (size_t)r + 0] = c1 * xy[2 * (size_t)(s + 1)] + c2 * xy[2 * (size_t)s];
    enu[4 * (size_t)r + 1] = c1 * xy[2 * (size_t)(s + 1) + 1] + c2 * xy[2 * (size_t)s + 1];
    enu[4 * (size_t)r + 2] = slam[4 * (size_t)r + 2];
    enu[4 * (size_t)r + 3] = tr;
    atomicAdd(n_kept, 1);
}

// Batched form: segment s owns fixes [goff[s], goff[s+1]) and stamps [soff[s], soff[s+1]).
// The band number is taken from each segment's first fix, as one GPSToENU call per segment
// would (gps_process.cc:869-877).  Dropped stamps (after the segment's last fix) are only
// ever a suffix of the segment: kept[s] counts the survivors.
__global__ void wgs_to_enu_batched_kernel(int method, int band_type, const double *__restrict__ lat,
                                          const double *__restrict__ lon, const int *__restrict__ goff, int nseg,
                                          double *__restrict__ xy)
{
    const int s = blockIdx.y;
    const int g0 = goff[s], ng = goff[s + 1] - g0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < ng; i += gridDim.x * blockDim.x) {
        int band = sticky_band(lon + g0, i, band_type);
        double x, y;
        project_fwd(method, band_type, band, lat[g0 + i], lon[g0 + i], x, y);
        xy[2 * (size_t)(g0 + i)] = x;
        xy[2 * (size_t)(g0 + i) + 1] = y;
    }
}

__global__ void interp_batched_kernel(const double *__restrict__ xy, const double *__restrict__ gt,
                                      const int *__restrict__ goff, const double *__restrict__ slam,
                                      const int *__restrict__ soff, double *__restrict__ enu,
                                      int *__restrict__ kept)
{
    const int s = blockIdx.y;
    const int g0 = goff[s], ng = goff[s + 1] - g0, r0 = soff[s], nr = soff[s + 1] - r0;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nr; k += gridDim.x * blockDim.x) {
        const size_t r = (size_t)(r0 + k);
        const double tr = slam[4 * r + 3];
        if (ng < 2 || tr > gt[g0 + ng - 1]) continue;
        int lo = 0, hi = ng - 2;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (gt[g0 + mid + 1] >= tr) hi = mid; else lo = mid + 1;
        }
        const size_t a = (size_t)(g0 + lo);
        const double s1 = gt[a], s2 = gt[a + 1], s3 = s2 - s1;
        const double c1 = (tr - s1) / s3, c2 = 1.0 - c1;
        enu[4 * r + 0] = c1 * xy[2 * (a + 1)] + c2 * xy[2 * a];
        enu[4 * r + 1] = c1 * xy[2 * (a + 1) + 1] + c2 * xy[2 * a + 1];
        enu[4 * r + 2] = slam[4 * r + 2];
        enu[4 * r + 3] = tr;
        atomicAdd(&kept[s], 1);
    }
}

}  // namespace gpscal

using namespace gpscal;

static int check_proj(gpscal_ctx *ctx, int method, int band_type)
{
    if (!ctx) return GPSCAL_EINVAL;
    if (method != GPSCAL_METHOD_UTM && method != GPSCAL_METHOD_GAUSS)
        return fail(ctx, GPSCAL_EINVAL, "method must be GPSCAL_METHOD_UTM or GPSCAL_METHOD_GAUSS");
    if (band_type != 3 && band_type != 6) return fail(ctx, GPSCAL_EINVAL, "band_type must be 3 or 6");
    return GPSCAL_OK;
}

extern "C" int gpscal_wgs_to_enu(gpscal_ctx *ctx, int method, int band_type, const double *lat, const double *lon,
                                 int n, double *xy)
{
    int rc = check_proj(ctx, method, band_type);
    if (rc) return rc;
    if (!lat || !lon || !xy || n < 1) return fail(ctx, GPSCAL_EINVAL, "gpscal_wgs_to_enu: bad argument");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    InArg<double> a, b;
    OutArg<double> o;
    GPSCAL_HIP(ctx, a.bind(ctx, lat, n));
    GPSCAL_HIP(ctx, b.bind(ctx, lon, n));
    GPSCAL_HIP(ctx, o.bind(ctx, xy, (size_t)n * 2));
    hipLaunchKernelGGL(wgs_to_enu_kernel, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, method, band_type, a.dev,
                       b.dev, n, o.dev);
    GPSCAL_HIP(ctx, hipGetLastError());
    bool sync = true;
    GPSCAL_HIP(ctx, o.commit(ctx, &sync));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}

extern "C" int gpscal_enu_to_wgs(gpscal_ctx *ctx, int method, int band_type, const double *enu, int n, double *lonlat,
                                 double *alt)
{
    int rc = check_proj(ctx, method, band_type);
    if (rc) return rc;
    if (!enu || !lonlat || !alt || n < 1) return fail(ctx, GPSCAL_EINVAL, "gpscal_enu_to_wgs: bad argument");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    InArg<double> a;
    OutArg<double> o1, o2;
    GPSCAL_HIP(ctx, a.bind(ctx, enu, (size_t)n * 5));
    GPSCAL_HIP(ctx, o1.bind(ctx, lonlat, (size_t)n * 2));
    GPSCAL_HIP(ctx, o2.bind(ctx, alt, n));
    hipLaunchKernelGGL(enu_to_wgs_kernel, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, method, band_type, a.dev, n,
                       o1.dev, o2.dev);
    GPSCAL_HIP(ctx, hipGetLastError());
    bool sync = true;
    GPSCAL_HIP(ctx, o1.commit(ctx, &sync));
    GPSCAL_HIP(ctx, o2.commit(ctx, &sync));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}

// IMGPS{b, l, w} per calibrated point (short_distance_track_process.cpp:299-304): l = longitude and b = latitude of
// the inverse projection, w = the merged weight.
__global__ void imgps_kernel(int method, int band_type, const double *__restrict__ enu, int n, double *__restrict__ ll,
                             double *__restrict__ blw)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    blw[3 * i] = ll[2 * i + 1];
    blw[3 * i + 1] = ll[2 * i];
    blw[3 * i + 2] = enu[5 * (size_t)i + 4];
}

extern "C" int gpscal_imgps_message(gpscal_ctx *ctx, int method, int band_type, const double *enu, int n, double *blw)
{
    int rc = check_proj(ctx, method, band_type);
    if (rc) return rc;
    if (!enu || !blw || n < 1) return fail(ctx, GPSCAL_EINVAL, "gpscal_imgps_message: bad argument");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    InArg<double> a;
    OutArg<double> o;
    DevBuf<double> ll, alt;
    GPSCAL_HIP(ctx, a.bind(ctx, enu, (size_t)n * 5));
    GPSCAL_HIP(ctx, o.bind(ctx, blw, (size_t)n * 3));
    GPSCAL_HIP(ctx, ll.alloc_async((size_t)n * 2, ctx->stream));
    GPSCAL_HIP(ctx, alt.alloc_async((size_t)n, ctx->stream));
    hipLaunchKernelGGL(enu_to_wgs_kernel, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, method, band_type, a.dev, n,
                       ll.p, alt.p);
    hipLaunchKernelGGL(imgps_kernel, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, method, band_type, a.dev, n, ll.p,
                       o.dev);
    GPSCAL_HIP(ctx, hipGetLastError());
    bool sync = a.tmp.p != nullptr;  // a staged input must outlive the kernels
    GPSCAL_HIP(ctx, o.commit(ctx, &sync));
    if (sync) GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}

extern "C" int gpscal_gps_to_enu(gpscal_ctx *ctx, int method, int band_type, const double *lat, const double *lon,
                                 const double *gps_t, int ngps, const double *slam, int nslam, double *enu, int *n_out)
{
    int rc = check_proj(ctx, method, band_type);
    if (rc) return rc;
    if (!lat || !lon || !gps_t || !slam || !enu || !n_out || ngps < 1 || nslam < 1)
        return fail(ctx, GPSCAL_EINVAL, "gpscal_gps_to_enu: bad argument");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    InArg<double> a, b, t, s;
    OutArg<double> o;
    DevBuf<double> xy;
    DevBuf<int> kept;
    GPSCAL_HIP(ctx, a.bind(ctx, lat, ngps));
    GPSCAL_HIP(ctx, b.bind(ctx, lon, ngps));
    GPSCAL_HIP(ctx, t.bind(ctx, gps_t, ngps));
    GPSCAL_HIP(ctx, s.bind(ctx, slam, (size_t)nslam * 4));
    GPSCAL_HIP(ctx, o.bind(ctx, enu, (size_t)nslam * 4));
    GPSCAL_HIP(ctx, xy.alloc((size_t)ngps * 2));
    GPSCAL_HIP(ctx, kept.alloc(1));
    GPSCAL_HIP(ctx, hipMemsetAsync(kept.p, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(wgs_to_enu_kernel, dim3(div_up(ngps, 256)), dim3(256), 0, ctx->stream, method, band_type,
                       a.dev, b.dev, ngps, xy.p);
    hipLaunchKernelGGL(interp_kernel, dim3(div_up(nslam, 256)), dim3(256), 0, ctx->stream, xy.p, t.dev, ngps, s.dev,
                       nslam, o.dev, kept.p);
    GPSCAL_HIP(ctx, hipGetLastError());
    int k = 0;
    GPSCAL_HIP(ctx, hipMemcpyAsync(&k, kept.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    bool sync = false;
    GPSCAL_HIP(ctx, o.commit(ctx, &sync, (size_t)k * 4));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = k;
    return GPSCAL_OK;
}

extern "C" int gpscal_gps_to_enu_batched(gpscal_ctx *ctx, int method, int band_type, const double *lat,
                                         const double *lon, const double *gps_t, const int *gps_off,
                                         const double *slam, const int *slam_off, int nseg, double *enu, int *n_out)
{
    int rc = check_proj(ctx, method, band_type);
    if (rc) return rc;
    if (!lat || !lon || !gps_t || !gps_off || !slam || !slam_off || !enu || !n_out || nseg < 1)
        return fail(ctx, GPSCAL_EINVAL, "gpscal_gps_to_enu_batched: bad argument");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    const int ngps = gps_off[nseg], nslam = slam_off[nseg];
    if (ngps < 1 || nslam < 1) return fail(ctx, GPSCAL_EINVAL, "gpscal_gps_to_enu_batched: empty input");
    int gmax = 1, smax = 1;
    for (int s = 0; s < nseg; ++s) {
        gmax = std::max(gmax, gps_off[s + 1] - gps_off[s]);
        smax = std::max(smax, slam_off[s + 1] - slam_off[s]);
    }
    InArg<double> a, b, t, sl;
    InArg<int> go, so;
    OutArg<double> o;
    OutArg<int> ko;
    DevBuf<double> xy;
    GPSCAL_HIP(ctx, a.bind(ctx, lat, ngps));
    GPSCAL_HIP(ctx, b.bind(ctx, lon, ngps));
    GPSCAL_HIP(ctx, t.bind(ctx, gps_t, ngps));
    GPSCAL_HIP(ctx, sl.bind(ctx, slam, (size_t)nslam * 4));
    GPSCAL_HIP(ctx, go.bind(ctx, gps_off, nseg + 1));
    GPSCAL_HIP(ctx, so.bind(ctx, slam_off, nseg + 1));
    GPSCAL_HIP(ctx, o.bind(ctx, enu, (size_t)nslam * 4));
    GPSCAL_HIP(ctx, ko.bind(ctx, n_out, nseg));
    GPSCAL_HIP(ctx, xy.alloc_async((size_t)ngps * 2, ctx->stream));
    GPSCAL_HIP(ctx, hipMemsetAsync(ko.dev, 0, sizeof(int) * nseg, ctx->stream));
    hipLaunchKernelGGL(wgs_to_enu_batched_kernel, dim3(div_up(gmax, 256), nseg), dim3(256), 0, ctx->stream, method,
                       band_type, a.dev, b.dev, go.dev, nseg, xy.p);
    hipLaunchKernelGGL(interp_batched_kernel, dim3(div_up(smax, 256), nseg), dim3(256), 0, ctx->stream, xy.p, t.dev,
                       go.dev, sl.dev, so.dev, o.dev, ko.dev);
    GPSCAL_HIP(ctx, hipGetLastError());
    bool sync = true;
    GPSCAL_HIP(ctx, o.commit(ctx, &sync));
    GPSCAL_HIP(ctx, ko.commit(ctx, &sync));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}

// ------------------------------------------------------------ GCJ-02 / BD-09
// GPSPro::GPSToGCJ / GCJToBD / BDToGCJ (gps_process.cc:526-595) with transform2Mars, bd_encrypt,
// bd_decrypt (:1127-1207): the "Mars" offset polynomial of the Chinese map datums on {longitude,
// latitude} pairs, float64, one lane per point.  PI is the reference's truncated 3.141592653589
// (common.h:17), X_PI its 3.1415926535897932384626 * 3000 / 180 (common.h:27).
namespace gpscal {

constexpr double MARS_PI = 3.141592653589;
constexpr double MARS_A = 6378245.0, MARS_B = 6356863.0188;
constexpr double MARS_XPI = 3.1415926535897932384626 * 3000.0 / 180.0;

__device__ __forceinline__ double mars_lat(double x, double y)
{
    double ret = -100.0 + 2.0 * x + 3.0 * y + 0.2 * y * y + 0.1 * x * y + 0.2 * sqrt(fabs(x));
    ret += (20.0 * sin(6.0 * x * MARS_PI) + 20.0 * sin(2.0 * x * MARS_PI)) * 2.0 / 3.0;
    ret += (20.0 * sin(y * MARS_PI) + 40.0 * sin(y / 3.0 * MARS_PI)) * 2.0 / 3.0;
    ret += (160.0 * sin(y / 12.0 * MARS_PI) + 320 * sin(y * MARS_PI / 30.0)) * 2.0 / 3.0;
    return ret;
}

__device__ __forceinline__ double mars_lon(double x, double y)
{
    double ret = 300.0 + x + 2.0 * y + 0.1 * x * x + 0.1 * x * y + 0.1 * sqrt(fabs(x));
    ret += (20.0 * sin(6.0 * x * MARS_PI) + 20.0 * sin(2.0 * x * MARS_PI)) * 2.0 / 3.0;
    ret += (20.0 * sin(x * MARS_PI) + 40.0 * sin(x / 3.0 * MARS_PI)) * 2.0 / 3.0;
    ret += (150.0 * sin(x / 12.0 * MARS_PI) + 300.0 * sin(x / 30.0 * MARS_PI)) * 2.0 / 3.0;
    return ret;
}

// mode 0: WGS-84 -> GCJ-02, 1: GCJ-02 -> BD-09, 2: BD-09 -> GCJ-02
__global__ void mars_kernel(int mode, const double *__restrict__ in, int n, double *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double lon = in[2 * i], lat = in[2 * i + 1];
    double olon, olat;
    if (mode == 0) {
        if (lon < 72.004 || lon > 137.8347 || lat < 0.8293 || lat > 55.8271) {  // outOfChina, :1127-1138
            olat = lat;
            olon = lon;
        } else {
            const double ee = (MARS_A * MARS_A - MARS_B * MARS_B) / (MARS_A * MARS_A);  // common.h:29
            double dLat = mars_lat(lon - 105.0, lat - 35.0), dLon = mars_lon(lon - 105.0, lat - 35.0);
            const double radLat = lat / 180.0 * MARS_PI;
            double magic = sin(radLat);
            magic = 1 - ee * magic * magic;
            const double sqrtMagic = sqrt(magic);
            dLat = (dLat * 180.0) / ((MARS_A * (1 - ee)) / (magic * sqrtMagic) * MARS_PI);
            dLon = (dLon * 180.0) / (MARS_A / sqrtMagic * cos(radLat) * MARS_PI);
            olat = lat + dLat;
            olon = lon + dLon;
        }
    } else if (mode == 1) {
        const double x = lon, y = lat;
        const double z = sqrt(x * x + y * y) + 0.00002 * sin(y * MARS_XPI);
        const double theta = atan2(y, x) + 0.000003 * cos(x * MARS_XPI);
        olon = z * cos(theta) + 0.0065;
        olat = z * sin(theta) + 0.006;
    } else {
        const double x = lon - 0.0065, y = lat - 0.006;
        const double z = sqrt(x * x + y * y) - 0.00002 * sin(y * MARS_XPI);
        const double theta = atan2(y, x) - 0.000003 * cos(x * MARS_XPI);
        olon = z * cos(theta);
        olat = z * sin(theta);
    }
    out[2 * i] = olon;
    out[2 * i + 1] = olat;
}

static int run_mars(gpscal_ctx *ctx, int mode, const double *lonlat, int n, double *out, const char *who)
{
    if (!ctx || !lonlat || !out || n < 1) return fail(ctx, GPSCAL_EINVAL, who);
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    InArg<double> a;
    OutArg<double> o;
    GPSCAL_HIP(ctx, a.bind(ctx, lonlat, (size_t)n * 2));
    GPSCAL_HIP(ctx, o.bind(ctx, out, (size_t)n * 2));
    hipLaunchKernelGGL(mars_kernel, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, mode, a.dev, n, o.dev);
    GPSCAL_HIP(ctx, hipGetLastError());
    bool sync = true;
    GPSCAL_HIP(ctx, o.commit(ctx, &sync));
    GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return GPSCAL_OK;
}

}  // namespace gpscal

extern "C" int gpscal_gps_to_gcj(gpscal_ctx *ctx, const double *lonlat, int n, double *gcj_lonlat)
{
    return gpscal::run_mars(ctx, 0, lonlat, n, gcj_lonlat, "gpscal_gps_to_gcj: bad argument");
}
extern "C" int gpscal_gcj_to_bd(gpscal_ctx *ctx, const double *gcj_lonlat, int n, double *bd_lonlat)
{
    return gpscal::run_mars(ctx, 1, gcj_lonlat, n, bd_lonlat, "gpscal_gcj_to_bd: bad argument");
}
extern "C" int gpscal_bd_to_gcj(gpscal_ctx *ctx, const double *bd_lonlat, int n, double *gcj_lonlat)
{
    return gpscal::run_mars(ctx, 2, bd_lonlat, n, gcj_lonlat, "gpscal_bd_to_gcj: bad argument");
}
