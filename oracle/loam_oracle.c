/*
 * loam_oracle.c -- CPU restatement of LOAM's sweep-to-sweep scan matching
 * (laserOdometry): correspondences, point-to-line / point-to-plane residuals,
 * 6-DoF Gauss-Newton with the degeneracy projection, pose accumulation.
 * TEST INFRASTRUCTURE ONLY (see gpscal_oracle.h).
 *   LO = src/gpsCalibration/src/lidar_slam/loam/laserOdometry.cpp
 * Third-party pieces restated (absent from /root/reference, parity unpinned):
 *   pcl::KdTreeFLANN::nearestKSearch k=1 (LO:603,758)  -> exact search, ties by index;
 *   cv::solve(DECOMP_QR) on the 6x6 normal equations (LO:975) and cv::eigen
 *   (LO:982; eigenvalues descending, eigenvectors as rows) -> Householder QR and
 *   cyclic Jacobi in float64 (OpenCV computes both in float32; its summation order
 *   inside matAt*matA is not pinned either, so the normal equations are summed in
 *   float64 here).  Per-point arithmetic is float32 exactly as coded.
 * The IMU terms are identically zero under run.sh (nothing publishes /imu/data,
 * input_data.cpp:259-262) and are omitted.
 * Points are float[4] {x, y, z, intensity}; intensity = ring id + relative time
 * (scanRegistration.cpp:340-362).
 */
#include "gpscal_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

void orc_lo_transform_to_start(const float tr[6], const float *pi, float *po)
{
    /* LO:123-150 */
    float s = 10 * (pi[3] - (int)pi[3]);
    float rx = s * tr[0], ry = s * tr[1], rz = s * tr[2];
    float tx = s * tr[3], ty = s * tr[4], tz = s * tr[5];
    float x1 = cosf(rz) * (pi[0] - tx) + sinf(rz) * (pi[1] - ty);
    float y1 = -sinf(rz) * (pi[0] - tx) + cosf(rz) * (pi[1] - ty);
    float z1 = (pi[2] - tz);
    float x2 = x1;
    float y2 = cosf(rx) * y1 + sinf(rx) * z1;
    float z2 = -sinf(rx) * y1 + cosf(rx) * z1;
    po[0] = cosf(ry) * x2 - sinf(ry) * z2;
    po[1] = y2;
    po[2] = sinf(ry) * x2 + cosf(ry) * z2;
    po[3] = pi[3];
}

void orc_lo_transform_to_end(const float tr[6], const float *pi, float *po)
{
    /* LO:156-227 with the IMU terms at zero (x7..x11 are identities then) */
    float p3[4];
    orc_lo_transform_to_start(tr, pi, p3);
    float rx = tr[0], ry = tr[1], rz = tr[2], tx = tr[3], ty = tr[4], tz = tr[5];
    float x4 = cosf(ry) * p3[0] + sinf(ry) * p3[2];
    float y4 = p3[1];
    float z4 = -sinf(ry) * p3[0] + cosf(ry) * p3[2];
    float x5 = x4;
    float y5 = cosf(rx) * y4 - sinf(rx) * z4;
    float z5 = sinf(rx) * y4 + cosf(rx) * z4;
    po[0] = cosf(rz) * x5 - sinf(rz) * y5 + tx;
    po[1] = sinf(rz) * x5 + cosf(rz) * y5 + ty;
    po[2] = z5 + tz;
    po[3] = (float)(int)pi[3];
}

/* ---- small dense helpers (float64) ---- */

/* least squares A x = b, A 6x6, Householder QR (cv::solve DECOMP_QR semantics) */
static void solve_qr6(const double A_in[36], const double b_in[6], double x[6])
{
    double A[36], b[6];
    memcpy(A, A_in, sizeof A);
    memcpy(b, b_in, sizeof b);
    for (int k = 0; k < 6; ++k) {
        double nrm = 0;
        for (int i = k; i < 6; ++i) nrm += A[6 * i + k] * A[6 * i + k];
        nrm = sqrt(nrm);
        if (nrm == 0.0) continue;
        double alpha = A[6 * k + k] > 0 ? -nrm : nrm;
        double v[6] = {0, 0, 0, 0, 0, 0};
        for (int i = k; i < 6; ++i) v[i] = A[6 * i + k];
        v[k] -= alpha;
        double vv = 0;
        for (int i = k; i < 6; ++i) vv += v[i] * v[i];
        if (vv == 0.0) continue;
        for (int j = k; j < 6; ++j) {
            double d = 0;
            for (int i = k; i < 6; ++i) d += v[i] * A[6 * i + j];
            d = 2 * d / vv;
            for (int i = k; i < 6; ++i) A[6 * i + j] -= d * v[i];
        }
        double d = 0;
        for (int i = k; i < 6; ++i) d += v[i] * b[i];
        d = 2 * d / vv;
        for (int i = k; i < 6; ++i) b[i] -= d * v[i];
    }
    for (int i = 5; i >= 0; --i) {
        double acc = b[i];
        for (int j = i + 1; j < 6; ++j) acc -= A[6 * i + j] * x[j];
        x[i] = A[6 * i + i] != 0.0 ? acc / A[6 * i + i] : 0.0;
    }
}

/* symmetric 6x6 eigen-decomposition, cyclic Jacobi; eigenvalues descending,
 * eigenvectors as ROWS of V (cv::eigen convention) */
static void eigen_sym6(const double A_in[36], double E[6], double V[36])
{
    double A[36], Q[36];
    memcpy(A, A_in, sizeof A);
    for (int i = 0; i < 36; ++i) Q[i] = (i % 7 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0;
        for (int p = 0; p < 6; ++p)
            for (int q = p + 1; q < 6; ++q) off += A[6 * p + q] * A[6 * p + q];
        if (off < 1e-300) break;
        for (int p = 0; p < 6; ++p)
            for (int q = p + 1; q < 6; ++q) {
                double apq = A[6 * p + q];
                if (fabs(apq) < 1e-300) continue;
                double tau = (A[6 * q + q] - A[6 * p + p]) / (2 * apq);
                double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1 + tau * tau));
                double c = 1 / sqrt(1 + t * t), s = t * c;
                for (int k = 0; k < 6; ++k) {
                    double akp = A[6 * k + p], akq = A[6 * k + q];
                    A[6 * k + p] = c * akp - s * akq;
                    A[6 * k + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 6; ++k) {
                    double apk = A[6 * p + k], aqk = A[6 * q + k];
                    A[6 * p + k] = c * apk - s * aqk;
                    A[6 * q + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 6; ++k) {
                    double qkp = Q[6 * k + p], qkq = Q[6 * k + q];
                    Q[6 * k + p] = c * qkp - s * qkq;
                    Q[6 * k + q] = s * qkp + c * qkq;
                }
            }
    }
    int order[6] = {0, 1, 2, 3, 4, 5};
    for (int i = 0; i < 6; ++i)
        for (int j = i + 1; j < 6; ++j)
            if (A[7 * order[j]] > A[7 * order[i]]) {
                int t = order[i];
                order[i] = order[j];
                order[j] = t;
            }
    for (int i = 0; i < 6; ++i) {
        E[i] = A[7 * order[i]];
        for (int k = 0; k < 6; ++k) V[6 * i + k] = Q[6 * k + order[i]];
    }
}

/* P = V^-1 * V2 (LO:996); V has orthonormal rows, so V^-1 = V^T */
static void degeneracy_projector(const double E[6], const double V[36], double P[36], int *degenerate)
{
    double V2[36];
    memcpy(V2, V, sizeof V2);
    *degenerate = 0;
    for (int i = 5; i >= 0; --i) { /* LO:987-995, threshold 10 */
        if (E[i] < 10.0) {
            for (int j = 0; j < 6; ++j) V2[6 * i + j] = 0;
            *degenerate = 1;
        } else {
            break;
        }
    }
    for (int r = 0; r < 6; ++r)
        for (int c = 0; c < 6; ++c) {
            double acc = 0;
            for (int k = 0; k < 6; ++k) acc += V[6 * k + r] * V2[6 * k + c];
            P[6 * r + c] = acc;
        }
}

static float sq3(const float *a, const float *b)
{
    /* plain expression of LO:627-632 (no fused multiply-add) */
    float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return dx * dx + dy * dy + dz * dz;
}

int orc_lo_match(const float *sharp, int nc, const float *flat, int ns, const float *cornerLast, int mc,
                 const float *surfLast, int ms, const float tr_in[6], float tr_out[6], int *iters_out,
                 int *nsel_out)
{
    float tr[6];
    memcpy(tr, tr_in, sizeof tr);
    int iters = 0, nsel_last = 0;
    if (!(mc > 10 && ms > 100)) { /* LO:569 */
        memcpy(tr_out, tr, sizeof tr);
        if (iters_out) *iters_out = 0;
        if (nsel_out) *nsel_out = 0;
        return 0;
    }
    /* packed xyz copies for the exact 1-NN */
    float *c3 = (float *)malloc(sizeof(float) * 3 * (size_t)mc), *s3 = (float *)malloc(sizeof(float) * 3 * (size_t)ms);
    for (int i = 0; i < mc; ++i) memcpy(c3 + 3 * i, cornerLast + 4 * i, 12);
    for (int i = 0; i < ms; ++i) memcpy(s3 + 3 * i, surfLast + 4 * i, 12);
    orc_kdtree *kc = orc_kdtree_build(c3, mc), *ks = orc_kdtree_build(s3, ms);
    int *ci1 = (int *)malloc(sizeof(int) * (size_t)(nc + 1)), *ci2 = (int *)malloc(sizeof(int) * (size_t)(nc + 1));
    int *si1 = (int *)malloc(sizeof(int) * (size_t)(ns + 1)), *si2 = (int *)malloc(sizeof(int) * (size_t)(ns + 1)),
        *si3 = (int *)malloc(sizeof(int) * (size_t)(ns + 1));
    for (int i = 0; i < nc; ++i) ci1[i] = ci2[i] = -1;
    for (int i = 0; i < ns; ++i) si1[i] = si2[i] = si3[i] = -1;
    float *ori = (float *)malloc(sizeof(float) * 4 * (size_t)(nc + ns + 1));
    float *coef = (float *)malloc(sizeof(float) * 4 * (size_t)(nc + ns + 1));
    int degenerate = 0;
    double P[36];
    for (int i = 0; i < 36; ++i) P[i] = (i % 7 == 0) ? 1.0 : 0.0;
    /* the forward scans are bounded by the CURRENT sweep's feature count (LO:620,776);
     * clamped to the last cloud's size so the read stays in bounds */
    const int fwd_c = nc < mc ? nc : mc, fwd_s = ns < ms ? ns : ms;

    for (int it = 0; it < 25; ++it) { /* LO:585 */
        ++iters;
        int nsel = 0;
        for (int i = 0; i < nc; ++i) { /* LO:592-746 */
            float ps[4];
            orc_lo_transform_to_start(tr, sharp + 4 * i, ps);
            if (it % 5 == 0) {
                int32_t idx;
                float sqd;
                orc_kdtree_search(kc, ps, 1, 1, &idx, &sqd);
                int closest = -1, min2 = -1;
                if (sqd < 25) {
                    closest = idx;
                    int scan = (int)cornerLast[4 * closest + 3];
                    float d2min = 25;
                    for (int j = closest + 1; j < fwd_c; ++j) {
                        if ((int)cornerLast[4 * j + 3] > scan + 1.5) break;
                        float d = sq3(cornerLast + 4 * j, ps);
                        if ((int)cornerLast[4 * j + 3] > scan && d < d2min) {
                            d2min = d;
                            min2 = j;
                        }
                    }
                    for (int j = closest - 1; j >= 0; --j) {
                        if ((int)cornerLast[4 * j + 3] < scan - 1.5) break;
                        float d = sq3(cornerLast + 4 * j, ps);
                        if ((int)cornerLast[4 * j + 3] < scan && d < d2min) {
                            d2min = d;
                            min2 = j;
                        }
                    }
                }
                ci1[i] = closest;
                ci2[i] = min2;
            }
            if (ci2[i] >= 0) { /* LO:680-746 */
                const float *t1 = cornerLast + 4 * ci1[i], *t2 = cornerLast + 4 * ci2[i];
                float x0 = ps[0], y0 = ps[1], z0 = ps[2];
                float x1 = t1[0], y1 = t1[1], z1 = t1[2], x2 = t2[0], y2 = t2[1], z2 = t2[2];
                float m11 = (x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1);
                float m22 = (x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1);
                float m33 = (y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1);
                float a012 = sqrtf(m11 * m11 + m22 * m22 + m33 * m33);
                float l12 = sqrtf((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
                float la = ((y1 - y2) * m11 + (z1 - z2) * m22) / a012 / l12;
                float lb = -((x1 - x2) * m11 - (z1 - z2) * m33) / a012 / l12;
                float lc = -((x1 - x2) * m22 + (y1 - y2) * m33) / a012 / l12;
                float ld2 = a012 / l12;
                float s = 1;
                if (it >= 5) s = 1 - 1.8 * fabs(ld2);
                if (s > 0.1 && ld2 != 0) {
                    memcpy(ori + 4 * nsel, sharp + 4 * i, 16);
                    coef[4 * nsel + 0] = s * la;
                    coef[4 * nsel + 1] = s * lb;
                    coef[4 * nsel + 2] = s * lc;
                    coef[4 * nsel + 3] = s * ld2;
                    ++nsel;
                }
            }
        }
        for (int i = 0; i < ns; ++i) { /* LO:752-901 */
            float ps[4];
            orc_lo_transform_to_start(tr, flat + 4 * i, ps);
            if (it % 5 == 0) {
                int32_t idx;
                float sqd;
                orc_kdtree_search(ks, ps, 1, 1, &idx, &sqd);
                int closest = -1, min2 = -1, min3 = -1;
                if (sqd < 25) {
                    closest = idx;
                    int scan = (int)surfLast[4 * closest + 3];
                    float d2 = 25, d3 = 25;
                    for (int j = closest + 1; j < fwd_s; ++j) {
                        if ((int)surfLast[4 * j + 3] > scan + 1.5) break;
                        float d = sq3(surfLast + 4 * j, ps);
                        if ((int)surfLast[4 * j + 3] <= scan) {
                            if (d < d2) { d2 = d; min2 = j; }
                        } else {
                            if (d < d3) { d3 = d; min3 = j; }
                        }
                    }
                    for (int j = closest - 1; j >= 0; --j) {
                        if ((int)surfLast[4 * j + 3] < scan - 1.5) break;
                        float d = sq3(surfLast + 4 * j, ps);
                        if ((int)surfLast[4 * j + 3] >= scan) {
                            if (d < d2) { d2 = d; min2 = j; }
                        } else {
                            if (d < d3) { d3 = d; min3 = j; }
                        }
                    }
                }
                si1[i] = closest;
                si2[i] = min2;
                si3[i] = min3;
            }
            if (si2[i] >= 0 && si3[i] >= 0) { /* LO:847-901 */
                const float *t1 = surfLast + 4 * si1[i], *t2 = surfLast + 4 * si2[i], *t3 = surfLast + 4 * si3[i];
                float pa = (t2[1] - t1[1]) * (t3[2] - t1[2]) - (t3[1] - t1[1]) * (t2[2] - t1[2]);
                float pb = (t2[2] - t1[2]) * (t3[0] - t1[0]) - (t3[2] - t1[2]) * (t2[0] - t1[0]);
                float pc = (t2[0] - t1[0]) * (t3[1] - t1[1]) - (t3[0] - t1[0]) * (t2[1] - t1[1]);
                float pd = -(pa * t1[0] + pb * t1[1] + pc * t1[2]);
                float pn = sqrtf(pa * pa + pb * pb + pc * pc);
                pa /= pn; pb /= pn; pc /= pn; pd /= pn;
                float pd2 = pa * ps[0] + pb * ps[1] + pc * ps[2] + pd;
                float s = 1;
                if (it >= 5)
                    s = 1 - 1.8 * fabs(pd2) / sqrtf(sqrtf(ps[0] * ps[0] + ps[1] * ps[1] + ps[2] * ps[2]));
                if (s > 0.1 && pd2 != 0) {
                    memcpy(ori + 4 * nsel, flat + 4 * i, 16);
                    coef[4 * nsel + 0] = s * pa;
                    coef[4 * nsel + 1] = s * pb;
                    coef[4 * nsel + 2] = s * pc;
                    coef[4 * nsel + 3] = s * pd2;
                    ++nsel;
                }
            }
        }
        nsel_last = nsel;
        if (nsel < 10) continue; /* LO:905-907 */

        double AtA[36], AtB[6];
        memset(AtA, 0, sizeof AtA);
        memset(AtB, 0, sizeof AtB);
        const float srx = sinf(tr[0]), crx = cosf(tr[0]), sry = sinf(tr[1]), cry = cosf(tr[1]);
        const float srz = sinf(tr[2]), crz = cosf(tr[2]), tx = tr[3], ty = tr[4], tz = tr[5];
        for (int i = 0; i < nsel; ++i) { /* LO:916-971, s = 1 */
            const float px = ori[4 * i], py = ori[4 * i + 1], pz = ori[4 * i + 2];
            const float cx = coef[4 * i], cy = coef[4 * i + 1], cz = coef[4 * i + 2], d2 = coef[4 * i + 3];
            float a[6];
            a[0] = (-crx * sry * srz * px + crx * crz * sry * py + srx * sry * pz + tx * crx * sry * srz -
                    ty * crx * crz * sry - tz * srx * sry) * cx +
                   (srx * srz * px - crz * srx * py + crx * pz + ty * crz * srx - tz * crx - tx * srx * srz) * cy +
                   (crx * cry * srz * px - crx * cry * crz * py - cry * srx * pz + tz * cry * srx +
                    ty * crx * cry * crz - tx * crx * cry * srz) * cz;
            a[1] = ((-crz * sry - cry * srx * srz) * px + (cry * crz * srx - sry * srz) * py - crx * cry * pz +
                    tx * (crz * sry + cry * srx * srz) + ty * (sry * srz - cry * crz * srx) + tz * crx * cry) * cx +
                   ((cry * crz - srx * sry * srz) * px + (cry * srz + crz * srx * sry) * py - crx * sry * pz +
                    tz * crx * sry - ty * (cry * srz + crz * srx * sry) - tx * (cry * crz - srx * sry * srz)) * cz;
            a[2] = ((-cry * srz - crz * srx * sry) * px + (cry * crz - srx * sry * srz) * py +
                    tx * (cry * srz + crz * srx * sry) - ty * (cry * crz - srx * sry * srz)) * cx +
                   (-crx * crz * px - crx * srz * py + ty * crx * srz + tx * crx * crz) * cy +
                   ((cry * crz * srx - sry * srz) * px + (crz * sry + cry * srx * srz) * py +
                    tx * (sry * srz - cry * crz * srx) - ty * (crz * sry + cry * srx * srz)) * cz;
            a[3] = -(cry * crz - srx * sry * srz) * cx + crx * srz * cy - (crz * sry + cry * srx * srz) * cz;
            a[4] = -(cry * srz + crz * srx * sry) * cx - crx * crz * cy - (sry * srz - cry * crz * srx) * cz;
            a[5] = crx * sry * cx - srx * cy - crx * cry * cz;
            const float b = -0.05 * d2; /* LO:970 */
            for (int r = 0; r < 6; ++r) {
                for (int c = 0; c < 6; ++c) AtA[6 * r + c] += (double)a[r] * (double)a[c];
                AtB[r] += (double)a[r] * (double)b;
            }
        }
        double X[6];
        solve_qr6(AtA, AtB, X); /* LO:975 */
        if (it == 0) {         /* LO:977-997 */
            double E[6], V[36];
            eigen_sym6(AtA, E, V);
            degeneracy_projector(E, V, P, &degenerate);
        }
        if (degenerate) { /* LO:999-1003 */
            double X2[6];
            memcpy(X2, X, sizeof X2);
            for (int r = 0; r < 6; ++r) {
                double acc = 0;
                for (int c = 0; c < 6; ++c) acc += P[6 * r + c] * X2[c];
                X[r] = acc;
            }
        }
        float xf[6];
        for (int k = 0; k < 6; ++k) {
            xf[k] = (float)X[k];
            tr[k] += xf[k];
            if (isnan(tr[k])) tr[k] = 0; /* LO:1012-1015 */
        }
        const double r2d = 180.0 / M_PI;
        float deltaR = sqrt(pow(xf[0] * r2d, 2) + pow(xf[1] * r2d, 2) + pow(xf[2] * r2d, 2));
        float deltaT = sqrt(pow(xf[3] * 100, 2) + pow(xf[4] * 100, 2) + pow(xf[5] * 100, 2));
        if (deltaR < 0.1 && deltaT < 0.1) break; /* LO:1026 */
    }
    memcpy(tr_out, tr, sizeof tr);
    if (iters_out) *iters_out = iters;
    if (nsel_out) *nsel_out = nsel_last;
    orc_kdtree_free(kc);
    orc_kdtree_free(ks);
    free(c3); free(s3); free(ci1); free(ci2); free(si1); free(si2); free(si3); free(ori); free(coef);
    return 0;
}

void orc_lo_accumulate(const float sum_in[6], const float tr[6], float sum_out[6])
{
    /* LO:1035-1064 with the IMU terms at zero.  AccumulateRotation (LO:287-304) */
    float cx = sum_in[0], cy = sum_in[1], cz = sum_in[2];
    float lx = -tr[0], ly = -tr[1] * 1.05, lz = -tr[2];
    float srx = cosf(lx) * cosf(cx) * sinf(ly) * sinf(cz) - cosf(cx) * cosf(cz) * sinf(lx) - cosf(lx) * cosf(ly) * sinf(cx);
    float ox = -asinf(srx);
    float srycrx = sinf(lx) * (cosf(cy) * sinf(cz) - cosf(cz) * sinf(cx) * sinf(cy)) +
                   cosf(lx) * sinf(ly) * (cosf(cy) * cosf(cz) + sinf(cx) * sinf(cy) * sinf(cz)) +
                   cosf(lx) * cosf(ly) * cosf(cx) * sinf(cy);
    float crycrx = cosf(lx) * cosf(ly) * cosf(cx) * cosf(cy) -
                   cosf(lx) * sinf(ly) * (cosf(cz) * sinf(cy) - cosf(cy) * sinf(cx) * sinf(cz)) -
                   sinf(lx) * (sinf(cy) * sinf(cz) + cosf(cy) * cosf(cz) * sinf(cx));
    float oy = atan2f(srycrx / cosf(ox), crycrx / cosf(ox));
    float srzcrx = sinf(cx) * (cosf(lz) * sinf(ly) - cosf(ly) * sinf(lx) * sinf(lz)) +
                   cosf(cx) * sinf(cz) * (cosf(ly) * cosf(lz) + sinf(lx) * sinf(ly) * sinf(lz)) +
                   cosf(lx) * cosf(cx) * cosf(cz) * sinf(lz);
    float crzcrx = cosf(lx) * cosf(lz) * cosf(cx) * cosf(cz) -
                   cosf(cx) * sinf(cz) * (cosf(ly) * sinf(lz) - cosf(lz) * sinf(lx) * sinf(ly)) -
                   sinf(cx) * (sinf(ly) * sinf(lz) + cosf(ly) * cosf(lz) * sinf(lx));
    float oz = atan2f(srzcrx / cosf(ox), crzcrx / cosf(ox));
    float rx = ox, ry = oy, rz = oz;
    /* LO:1039-1052 */
    float x1 = cosf(rz) * tr[3] - sinf(rz) * tr[4];
    float y1 = sinf(rz) * tr[3] + cosf(rz) * tr[4];
    float z1 = tr[5] * 1.05;
    float x2 = x1;
    float y2 = cosf(rx) * y1 - sinf(rx) * z1;
    float z2 = sinf(rx) * y1 + cosf(rx) * z1;
    float tx = sum_in[3] - (cosf(ry) * x2 + sinf(ry) * z2);
    float ty = sum_in[4] - y2;
    float tz = sum_in[5] - (-sinf(ry) * x2 + cosf(ry) * z2);
    /* PluginIMURotation (LO:228-282) with zero IMU angles reduces to: */
    float sx = -sinf(rx);
    float acx = -asinf(sx);
    float acy = atan2f(cosf(rx) * sinf(ry) / cosf(acx), cosf(rx) * cosf(ry) / cosf(acx));
    float acz = atan2f(cosf(rx) * sinf(rz) / cosf(acx), cosf(rx) * cosf(rz) / cosf(acx));
    sum_out[0] = acx;
    sum_out[1] = acy;
    sum_out[2] = acz;
    sum_out[3] = tx;
    sum_out[4] = ty;
    sum_out[5] = tz;
}

/* =====================================================================
 * laserMapping's sweep-to-map optimisation loop.
 *   LM = src/gpsCalibration/src/lidar_slam/loam/laserMapping.cpp
 * pointAssociateToMap LM:244-262; k=5 search + line test LM:757-858; plane fit
 * LM:860-920; 6x6 system, solve, degeneracy (threshold 100), stop 0.05 LM:922-1018.
 * cv::eigen on the 3x3 covariance and cv::solve(DECOMP_QR) on the 5x3 plane system are
 * restated in float64 (cyclic Jacobi / Householder QR); parity unpinned (OpenCV absent).
 * ===================================================================== */

static void lm_to_map(const float tr[6], const float *pi, float *po)
{
    /* LM:244-262 */
    float x1 = cosf(tr[2]) * pi[0] - sinf(tr[2]) * pi[1];
    float y1 = sinf(tr[2]) * pi[0] + cosf(tr[2]) * pi[1];
    float z1 = pi[2];
    float x2 = x1;
    float y2 = cosf(tr[0]) * y1 - sinf(tr[0]) * z1;
    float z2 = sinf(tr[0]) * y1 + cosf(tr[0]) * z1;
    po[0] = cosf(tr[1]) * x2 + sinf(tr[1]) * z2 + tr[3];
    po[1] = y2 + tr[4];
    po[2] = -sinf(tr[1]) * x2 + cosf(tr[1]) * z2 + tr[5];
    po[3] = pi[3];
}

/* largest eigenpair test of a symmetric 3x3: returns l1, l2 (two largest) and v1 */
static void eigen_sym3_top(const double A_in[9], double *l1, double *l2, double v1[3])
{
    double A[9], Q[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(A, A_in, sizeof A);
    for (int sweep = 0; sweep < 50; ++sweep) {
        double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        if (off < 1e-300) break;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double apq = A[3 * p + q];
                if (fabs(apq) < 1e-300) continue;
                double tau = (A[4 * q] - A[4 * p]) / (2 * apq);
                double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1 + tau * tau));
                double c = 1 / sqrt(1 + t * t), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    double akp = A[3 * k + p], akq = A[3 * k + q];
                    A[3 * k + p] = c * akp - s * akq;
                    A[3 * k + q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    double apk = A[3 * p + k], aqk = A[3 * q + k];
                    A[3 * p + k] = c * apk - s * aqk;
                    A[3 * q + k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double qkp = Q[3 * k + p], qkq = Q[3 * k + q];
                    Q[3 * k + p] = c * qkp - s * qkq;
                    Q[3 * k + q] = s * qkp + c * qkq;
                }
            }
    }
    int i1 = 0;
    for (int i = 1; i < 3; ++i)
        if (A[4 * i] > A[4 * i1]) i1 = i;
    int i2 = -1;
    for (int i = 0; i < 3; ++i)
        if (i != i1 && (i2 < 0 || A[4 * i] > A[4 * i2])) i2 = i;
    *l1 = A[4 * i1];
    *l2 = A[4 * i2];
    for (int k = 0; k < 3; ++k) v1[k] = Q[3 * k + i1];
}

/* least squares of the 5x3 system A x = -1 (LM:870-875), Householder QR */
static void plane_fit5(const double P[15], double x[3])
{
    double A[15], b[5] = {-1, -1, -1, -1, -1};
    memcpy(A, P, sizeof A);
    for (int k = 0; k < 3; ++k) {
        double nrm = 0;
        for (int i = k; i < 5; ++i) nrm += A[3 * i + k] * A[3 * i + k];
        nrm = sqrt(nrm);
        if (nrm == 0.0) continue;
        double alpha = A[3 * k + k] > 0 ? -nrm : nrm;
        double v[5] = {0, 0, 0, 0, 0};
        for (int i = k; i < 5; ++i) v[i] = A[3 * i + k];
        v[k] -= alpha;
        double vv = 0;
        for (int i = k; i < 5; ++i) vv += v[i] * v[i];
        if (vv == 0.0) continue;
        for (int j = k; j < 3; ++j) {
            double d = 0;
            for (int i = k; i < 5; ++i) d += v[i] * A[3 * i + j];
            d = 2 * d / vv;
            for (int i = k; i < 5; ++i) A[3 * i + j] -= d * v[i];
        }
        double d = 0;
        for (int i = k; i < 5; ++i) d += v[i] * b[i];
        d = 2 * d / vv;
        for (int i = k; i < 5; ++i) b[i] -= d * v[i];
    }
    for (int i = 2; i >= 0; --i) {
        double acc = b[i];
        for (int j = i + 1; j < 3; ++j) acc -= A[3 * i + j] * x[j];
        x[i] = A[3 * i + i] != 0.0 ? acc / A[3 * i + i] : 0.0;
    }
}

int orc_lm_match(const float *cornerStack, int nc, const float *surfStack, int ns, const float *cornerMap, int mc,
                 const float *surfMap, int ms, const float tr_in[6], float tr_out[6], int *iters_out, int *nsel_out)
{
    float tr[6];
    memcpy(tr, tr_in, sizeof tr);
    int iters = 0, nsel_last = 0;
    if (!(mc > 10 && ms > 100)) { /* LM:748 */
        memcpy(tr_out, tr, sizeof tr);
        if (iters_out) *iters_out = 0;
        if (nsel_out) *nsel_out = 0;
        return 0;
    }
    float *c3 = (float *)malloc(sizeof(float) * 3 * (size_t)mc), *s3 = (float *)malloc(sizeof(float) * 3 * (size_t)ms);
    for (int i = 0; i < mc; ++i) memcpy(c3 + 3 * i, cornerMap + 4 * i, 12);
    for (int i = 0; i < ms; ++i) memcpy(s3 + 3 * i, surfMap + 4 * i, 12);
    orc_kdtree *kc = orc_kdtree_build(c3, mc), *ks = orc_kdtree_build(s3, ms);
    int degenerate = 0;
    double P[36];
    for (int i = 0; i < 36; ++i) P[i] = (i % 7 == 0) ? 1.0 : 0.0;
    for (int it = 0; it < 10; ++it) { /* LM:753 */
        ++iters;
        double AtA[36], AtB[6];
        memset(AtA, 0, sizeof AtA);
        memset(AtB, 0, sizeof AtB);
        int nsel = 0;
        const float srx = sinf(tr[0]), crx = cosf(tr[0]), sry = sinf(tr[1]), cry = cosf(tr[1]);
        const float srz = sinf(tr[2]), crz = cosf(tr[2]);
        for (int pass = 0; pass < 2; ++pass) {
            const float *stack = pass == 0 ? cornerStack : surfStack;
            const float *map = pass == 0 ? cornerMap : surfMap;
            const int n = pass == 0 ? nc : ns;
            const orc_kdtree *kd = pass == 0 ? kc : ks;
            for (int i = 0; i < n; ++i) {
                const float *po = stack + 4 * i;
                float ps[4];
                lm_to_map(tr, po, ps);
                int32_t idx[5];
                float sqd[5];
                orc_kdtree_search(kd, ps, 1, 5, idx, sqd);
                if (!(sqd[4] < 1.0)) continue; /* LM:762,869 */
                float cf[4];
                int ok = 0;
                if (pass == 0) { /* LM:763-857 */
                    float cx = 0, cy = 0, cz = 0;
                    for (int j = 0; j < 5; ++j) {
                        cx += map[4 * idx[j]];
                        cy += map[4 * idx[j] + 1];
                        cz += map[4 * idx[j] + 2];
                    }
                    cx /= 5; cy /= 5; cz /= 5;
                    float a11 = 0, a12 = 0, a13 = 0, a22 = 0, a23 = 0, a33 = 0;
                    for (int j = 0; j < 5; ++j) {
                        float ax = map[4 * idx[j]] - cx, ay = map[4 * idx[j] + 1] - cy, az = map[4 * idx[j] + 2] - cz;
                        a11 += ax * ax; a12 += ax * ay; a13 += ax * az;
                        a22 += ay * ay; a23 += ay * az; a33 += az * az;
                    }
                    a11 /= 5; a12 /= 5; a13 /= 5; a22 /= 5; a23 /= 5; a33 /= 5;
                    double A1[9] = {a11, a12, a13, a12, a22, a23, a13, a23, a33}, l1, l2, v1[3];
                    eigen_sym3_top(A1, &l1, &l2, v1);
                    if ((float)l1 > 3 * (float)l2) { /* LM:812 */
                        float x0 = ps[0], y0 = ps[1], z0 = ps[2];
                        float x1 = cx + 0.1 * (float)v1[0], y1 = cy + 0.1 * (float)v1[1], z1 = cz + 0.1 * (float)v1[2];
                        float x2 = cx - 0.1 * (float)v1[0], y2 = cy - 0.1 * (float)v1[1], z2 = cz - 0.1 * (float)v1[2];
                        float m11 = (x0 - x1) * (y0 - y2) - (x0 - x2) * (y0 - y1);
                        float m22 = (x0 - x1) * (z0 - z2) - (x0 - x2) * (z0 - z1);
                        float m33 = (y0 - y1) * (z0 - z2) - (y0 - y2) * (z0 - z1);
                        float a012 = sqrtf(m11 * m11 + m22 * m22 + m33 * m33);
                        float l12 = sqrtf((x1 - x2) * (x1 - x2) + (y1 - y2) * (y1 - y2) + (z1 - z2) * (z1 - z2));
                        float la = ((y1 - y2) * m11 + (z1 - z2) * m22) / a012 / l12;
                        float lb = -((x1 - x2) * m11 - (z1 - z2) * m33) / a012 / l12;
                        float lc = -((x1 - x2) * m22 + (y1 - y2) * m33) / a012 / l12;
                        float ld2 = a012 / l12;
                        float s = 1 - 0.9 * fabs(ld2);
                        cf[0] = s * la; cf[1] = s * lb; cf[2] = s * lc; cf[3] = s * ld2;
                        ok = s > 0.1;
                    }
                } else { /* LM:866-919 */
                    double A0[15], x[3];
                    for (int j = 0; j < 5; ++j)
                        for (int k = 0; k < 3; ++k) A0[3 * j + k] = map[4 * idx[j] + k];
                    plane_fit5(A0, x);
                    float pa = (float)x[0], pb = (float)x[1], pc = (float)x[2], pd = 1;
                    float pn = sqrtf(pa * pa + pb * pb + pc * pc);
                    pa /= pn; pb /= pn; pc /= pn; pd /= pn;
                    int valid = 1;
                    for (int j = 0; j < 5; ++j)
                        if (fabs(pa * map[4 * idx[j]] + pb * map[4 * idx[j] + 1] + pc * map[4 * idx[j] + 2] + pd) > 0.2) {
                            valid = 0;
                            break;
                        }
                    if (valid) {
                        float pd2 = pa * ps[0] + pb * ps[1] + pc * ps[2] + pd;
                        float s = 1 - 0.9 * fabs(pd2) / sqrtf(sqrtf(ps[0] * ps[0] + ps[1] * ps[1] + ps[2] * ps[2]));
                        cf[0] = s * pa; cf[1] = s * pb; cf[2] = s * pc; cf[3] = s * pd2;
                        ok = s > 0.1;
                    }
                }
                if (!ok) continue;
                ++nsel;
                /* LM:940-966 */
                const float px = po[0], py = po[1], pz = po[2];
                float a[6];
                a[0] = (crx * sry * srz * px + crx * crz * sry * py - srx * sry * pz) * cf[0] +
                       (-srx * srz * px - crz * srx * py - crx * pz) * cf[1] +
                       (crx * cry * srz * px + crx * cry * crz * py - cry * srx * pz) * cf[2];
                a[1] = ((cry * srx * srz - crz * sry) * px + (sry * srz + cry * crz * srx) * py + crx * cry * pz) * cf[0] +
                       ((-cry * crz - srx * sry * srz) * px + (cry * srz - crz * srx * sry) * py - crx * sry * pz) * cf[2];
                a[2] = ((crz * srx * sry - cry * srz) * px + (-cry * crz - srx * sry * srz) * py) * cf[0] +
                       (crx * crz * px - crx * srz * py) * cf[1] +
                       ((sry * srz + cry * crz * srx) * px + (crz * sry - cry * srx * srz) * py) * cf[2];
                a[3] = cf[0]; a[4] = cf[1]; a[5] = cf[2];
                const float b = -cf[3];
                for (int r = 0; r < 6; ++r) {
                    for (int c = 0; c < 6; ++c) AtA[6 * r + c] += (double)a[r] * (double)a[c];
                    AtB[r] += (double)a[r] * (double)b;
                }
            }
        }
        nsel_last = nsel;
        if (nsel < 50) continue; /* LM:929-931 */
        double X[6];
        solve_qr6(AtA, AtB, X);
        if (it == 0) { /* LM:970-991, threshold 100 */
            double E[6], V[36], V2[36];
            eigen_sym6(AtA, E, V);
            memcpy(V2, V, sizeof V2);
            degenerate = 0;
            for (int i = 5; i >= 0; --i) {
                if (E[i] < 100.0) {
                    for (int j = 0; j < 6; ++j) V2[6 * i + j] = 0;
                    degenerate = 1;
                } else break;
            }
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 6; ++c) {
                    double acc = 0;
                    for (int k = 0; k < 6; ++k) acc += V[6 * k + r] * V2[6 * k + c];
                    P[6 * r + c] = acc;
                }
        }
        if (degenerate) {
            double X2[6];
            memcpy(X2, X, sizeof X2);
            for (int r = 0; r < 6; ++r) {
                double acc = 0;
                for (int c = 0; c < 6; ++c) acc += P[6 * r + c] * X2[c];
                X[r] = acc;
            }
        }
        float xf[6];
        for (int k = 0; k < 6; ++k) {
            xf[k] = (float)X[k];
            tr[k] += xf[k];
        }
        const double r2d = 180.0 / M_PI;
        float deltaR = sqrt(pow(xf[0] * r2d, 2) + pow(xf[1] * r2d, 2) + pow(xf[2] * r2d, 2));
        float deltaT = sqrt(pow(xf[3] * 100, 2) + pow(xf[4] * 100, 2) + pow(xf[5] * 100, 2));
        if (deltaR < 0.05 && deltaT < 0.05) break; /* LM:1015 */
    }
    memcpy(tr_out, tr, sizeof tr);
    if (iters_out) *iters_out = iters;
    if (nsel_out) *nsel_out = nsel_last;
    orc_kdtree_free(kc);
    orc_kdtree_free(ks);
    free(c3);
    free(s3);
    return 0;
}
