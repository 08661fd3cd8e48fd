/*
 * track_oracle.c -- CPU restatement of the GPS<->SLAM track alignment.
 * TEST INFRASTRUCTURE ONLY (see gpscal_oracle.h).  Follows, with citations:
 *   WC = src/gpsCalibration/src/gps_calibration/weight_calculation.cc
 *   TC = src/gpsCalibration/src/gps_calibration/track_calibration.cc
 *   LD = src/gpsCalibration/src/long_distance_track_process/long_distance_track_process.cpp
 *   SD = src/gpsCalibration/src/short_distance_track_process/short_distance_track_process.cpp
 *   TM = src/gpsCalibration/src/lidar_slam/loam/transformMaintenance.cpp
 * Third-party arithmetic restated from its published algorithm:
 *   Eigen3 JacobiSVD<MatrixXd> (two-sided Jacobi, TC:508) -- version pinned
 *   only as "libeigen3-dev" by install/install_u1604_basic.sh:42.
 */
#include "gpscal_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SPEED_NORM 2.2 /* WC.h:6 */
#define DELTA_MIN 0.01 /* WC.h:7 */

/* ---------------------------------------------------------------- weights */

static double speed_weight(const double *p, int n, int i)
{
    /* WC:10-22.  i==0 -> 1.0; else distance to the NEXT sample.  At i==n-1 the
     * reference indexes one past the end; that slot is defined here as (0,0)
     * (zero-filled spare vector capacity). */
    if (i == 0) return 1.0;
    double nx = 0.0, ny = 0.0;
    if (i + 1 < n) {
        nx = p[4 * (i + 1) + 0];
        ny = p[4 * (i + 1) + 1];
    }
    double dx = nx - p[4 * i + 0];
    double dy = ny - p[4 * i + 1];
    double d = sqrt(dx * dx + dy * dy);
    double v = d / SPEED_NORM;
    return v < 1.0 ? v : 1.0;
}

int orc_weights_speed(const double *slam, int n, double *w)
{
    for (int i = 0; i < n; ++i) w[i] = speed_weight(slam, n, i);
    return 1; /* WC:26 */
}

int orc_weights_irls(const double *slam, const double *enu, const double *fit,
                     int n, double *w)
{
    for (int i = 0; i < n; ++i) w[i] = speed_weight(slam, n, i); /* WC:35-47 */
    for (int i = 0; i < n; ++i) {                                 /* WC:68-75 */
        double dx = enu[4 * i + 0] - fit[4 * i + 0];
        double dy = enu[4 * i + 1] - fit[4 * i + 1];
        double d = sqrt(dx * dx + dy * dy);
        double den = d > DELTA_MIN ? d : DELTA_MIN;
        w[i] = w[i] * 1.0 / den;
    }
    return 1;
}

/* ------------------------------------------------------------- 3x3 SVD   */

/* W <- G^T W on rows i,j with G = [[c,-s],[s,c]] embedded at (i,j). */
static void rot_rows_T(double W[9], int i, int j, double c, double s)
{
    for (int k = 0; k < 3; ++k) {
        double a = W[3 * i + k], b = W[3 * j + k];
        W[3 * i + k] = c * a + s * b;
        W[3 * j + k] = -s * a + c * b;
    }
}
/* M <- M G on columns i,j with G = [[c,-s],[s,c]]. */
static void rot_cols(double M[9], int i, int j, double c, double s)
{
    for (int k = 0; k < 3; ++k) {
        double a = M[3 * k + i], b = M[3 * k + j];
        M[3 * k + i] = c * a + s * b;
        M[3 * k + j] = -s * a + c * b;
    }
}

int orc_svd3(const double A[9], double U[9], double S[3], double V[9])
{
    double W[9];
    double scale = 0.0;
    for (int k = 0; k < 9; ++k) {
        double a = fabs(A[k]);
        if (a > scale) scale = a;
    }
    if (scale == 0.0) scale = 1.0;
    for (int k = 0; k < 9; ++k) W[k] = A[k] / scale;
    for (int k = 0; k < 9; ++k) U[k] = V[k] = (k % 4 == 0) ? 1.0 : 0.0;

    const double prec = 2.0 * DBL_EPSILON;
    for (int sweep = 0; sweep < 100; ++sweep) {
        int finished = 1;
        for (int j = 1; j < 3; ++j) {
            for (int i = 0; i < j; ++i) {
                double wii = W[3 * i + i], wjj = W[3 * j + j];
                double wij = W[3 * i + j], wji = W[3 * j + i];
                double big = fabs(wii) > fabs(wjj) ? fabs(wii) : fabs(wjj);
                double thr = prec * big;
                if (thr < DBL_MIN) thr = DBL_MIN;
                if (!(fabs(wij) > thr || fabs(wji) > thr)) continue;
                finished = 0;
                /* step 1: rotation G making G^T M symmetric */
                double t = wii + wjj, d = wji - wij;
                double c1 = 1.0, s1 = 0.0;
                if (fabs(d) >= DBL_MIN) {
                    double h = hypot(t, d);
                    c1 = t / h;
                    s1 = d / h;
                }
                /* symmetric S = G^T M */
                double a = c1 * wii + s1 * wji;
                double b = c1 * wij + s1 * wjj;
                double e = -s1 * wij + c1 * wjj;
                /* step 2: Jacobi rotation J=[[cj,sj],[-sj,cj]] diagonalising S */
                double cj = 1.0, sj = 0.0;
                if (fabs(b) >= DBL_MIN) {
                    double tau = (e - a) / (2.0 * b);
                    double tj = (tau >= 0.0 ? 1.0 : -1.0) /
                                (fabs(tau) + sqrt(1.0 + tau * tau));
                    cj = 1.0 / sqrt(1.0 + tj * tj);
                    sj = tj * cj;
                }
                /* L = G J with J = rot(c=cj, s=-sj) in the [[c,-s],[s,c]] form */
                double cl = c1 * cj + s1 * sj; /* cos(th1 - phi) */
                double sl = s1 * cj - c1 * sj; /* sin(th1 - phi) */
                rot_rows_T(W, i, j, cl, sl);
                rot_cols(W, i, j, cj, -sj);
                rot_cols(U, i, j, cl, sl);
                rot_cols(V, i, j, cj, -sj);
            }
        }
        if (finished) break;
    }
    for (int k = 0; k < 3; ++k) {
        double s = W[4 * k];
        if (s < 0.0) {
            s = -s;
            for (int r = 0; r < 3; ++r) U[3 * r + k] = -U[3 * r + k];
        }
        S[k] = s * scale;
    }
    /* descending order by column swaps (selection sort, stable for ties) */
    for (int k = 0; k < 2; ++k) {
        int best = k;
        for (int l = k + 1; l < 3; ++l)
            if (S[l] > S[best]) best = l;
        if (best != k) {
            double ts = S[k];
            S[k] = S[best];
            S[best] = ts;
            for (int r = 0; r < 3; ++r) {
                double tu = U[3 * r + k];
                U[3 * r + k] = U[3 * r + best];
                U[3 * r + best] = tu;
                double tv = V[3 * r + k];
                V[3 * r + k] = V[3 * r + best];
                V[3 * r + best] = tv;
            }
        }
    }
    return 0;
}

static double det3(const double M[9])
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) -
           M[1] * (M[3] * M[8] - M[5] * M[6]) +
           M[2] * (M[3] * M[7] - M[4] * M[6]);
}

static void mul_V_Ut(const double V[9], const double U[9], double R[9])
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc += V[3 * r + k] * U[3 * c + k];
            R[3 * r + c] = acc;
        }
}

void orc_kabsch_from_H(const double H[9], double R[9])
{
    double U[9], S[3], V[9];
    orc_svd3(H, U, S, V);
    mul_V_Ut(V, U, R); /* TC:513 */
    if (det3(R) < 0.0) { /* TC:516-523 */
        for (int r = 0; r < 3; ++r) V[3 * r + 2] = -V[3 * r + 2];
        mul_V_Ut(V, U, R);
    }
}

/* --------------------------------------------------------- track alignment */

int orc_bft_weighted(const double *A, const double *B, const double *w, int n,
                     double T[16])
{
    double sA[3] = {0, 0, 0}, sB[3] = {0, 0, 0}, sW = 0.0;
    for (int i = 0; i < n; ++i) { /* TC:416-440 */
        for (int j = 0; j < 3; ++j) {
            sA[j] += A[4 * i + j] * w[i];
            sB[j] += B[4 * i + j] * w[i];
        }
        sW += w[i];
    }
    double cA[3], cB[3];
    for (int j = 0; j < 3; ++j) { /* TC:451-457 */
        cA[j] = sA[j] / sW;
        cB[j] = sB[j] / sW;
    }
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) { /* TC:490-506 */
        double aa[3], bb[3];
        for (int j = 0; j < 3; ++j) {
            aa[j] = (A[4 * i + j] - cA[j]) * w[i];
            bb[j] = (B[4 * i + j] - cB[j]) * w[i];
        }
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) H[3 * r + c] += aa[r] * bb[c];
    }
    double R[9];
    orc_kabsch_from_H(H, R); /* TC:508-523 */
    for (int k = 0; k < 16; ++k) T[k] = (k % 5 == 0) ? 1.0 : 0.0;
    for (int r = 0; r < 3; ++r) { /* TC:526-541 */
        double rc = 0.0;
        for (int k = 0; k < 3; ++k) rc += R[3 * r + k] * cA[k];
        for (int c = 0; c < 3; ++c) T[4 * r + c] = R[3 * r + c];
        T[4 * r + 3] = cB[r] - rc;
    }
    return 1;
}

int orc_track_fit(const double *slam, const double *enu, const double *w,
                  int n, double T[16], double *rotated, double *calibrated,
                  int *iters, int quadratic)
{
    if (n <= 0) return -1;
    double *S4 = (double *)malloc(sizeof(double) * 4 * (size_t)n);
    double *E4 = (double *)malloc(sizeof(double) * 4 * (size_t)n);
    double *src = (double *)malloc(sizeof(double) * 4 * (size_t)n);
    double *dist = (double *)malloc(sizeof(double) * (size_t)n);
    double *rot = rotated ? rotated
                          : (double *)malloc(sizeof(double) * 3 * (size_t)n);
    const double ex0 = enu[0], ey0 = enu[1]; /* TC:62-63 */
    for (int i = 0; i < n; ++i) {            /* TC:53-68 */
        S4[4 * i + 0] = slam[4 * i + 0] - slam[0];
        S4[4 * i + 1] = slam[4 * i + 1] - slam[1];
        S4[4 * i + 2] = 1.0;
        S4[4 * i + 3] = 1.0;
        E4[4 * i + 0] = enu[4 * i + 0] - ex0;
        E4[4 * i + 1] = enu[4 * i + 1] - ey0;
        E4[4 * i + 2] = 1.0;
        E4[4 * i + 3] = 1.0;
    }
    memcpy(src, S4, sizeof(double) * 4 * (size_t)n); /* TC:123-134 */
    double prev = 0.0;
    double Tk[16];
    int passes = 0;
    for (int it = 0; it < 2; ++it) { /* TC:145-181 */
        ++passes;
        for (int i = 0; i < n; ++i) { /* TC:578-583 */
            double dx = src[4 * i + 0] - E4[4 * i + 0];
            double dy = src[4 * i + 1] - E4[4 * i + 1];
            dist[i] = sqrt(dx * dx + dy * dy);
        }
        orc_bft_weighted(src, E4, w, n, Tk); /* TC:157 */
        for (int i = 0; i < n; ++i) {        /* TC:165 src = src * T^T */
            double v[4];
            for (int j = 0; j < 4; ++j) {
                double acc = 0.0;
                for (int k = 0; k < 4; ++k) acc += src[4 * i + k] * Tk[4 * j + k];
                v[j] = acc;
            }
            memcpy(src + 4 * i, v, sizeof v);
        }
        double mean = 0.0;
        for (int i = 0; i < n; ++i) mean += dist[i]; /* TC:170-174 */
        mean /= (double)n;
        if (fabs(prev - mean) < 0.003) break; /* TC:176 */
        prev = mean;
    }
    orc_bft_weighted(S4, src, w, n, T); /* TC:189 */
    for (int i = 0; i < n; ++i)         /* TC:603-622 */
        for (int j = 0; j < 3; ++j) {
            double acc = 0.0;
            for (int k = 0; k < 3; ++k) acc += S4[4 * i + k] * T[4 * j + k];
            rot[3 * i + j] = acc + T[4 * j + 3];
        }
    if (calibrated) { /* TC:631-689 */
        if (quadratic) {
            for (int a = 0; a < n; ++a) {
                double ax = 0.0, ay = 0.0;
                for (int b = 0; b < n; ++b) {
                    double dx = rot[3 * b + 0] - rot[3 * a + 0];
                    double dy = rot[3 * b + 1] - rot[3 * a + 1];
                    ax += E4[4 * b + 0] - dx;
                    ay += E4[4 * b + 1] - dy;
                }
                ax /= (double)n;
                ay /= (double)n;
                calibrated[4 * a + 0] = (ax + rot[3 * a + 0]) / 2.0 + ex0;
                calibrated[4 * a + 1] = (ay + rot[3 * a + 1]) / 2.0 + ey0;
                calibrated[4 * a + 2] = enu[4 * a + 2]; /* TC:84,682 */
                calibrated[4 * a + 3] = enu[4 * a + 3]; /* TC:85,683 */
            }
        } else {
            double mx = 0.0, my = 0.0;
            for (int b = 0; b < n; ++b) {
                mx += E4[4 * b + 0] - rot[3 * b + 0];
                my += E4[4 * b + 1] - rot[3 * b + 1];
            }
            mx /= (double)n;
            my /= (double)n;
            for (int a = 0; a < n; ++a) {
                calibrated[4 * a + 0] =
                    ((mx + rot[3 * a + 0]) + rot[3 * a + 0]) / 2.0 + ex0;
                calibrated[4 * a + 1] =
                    ((my + rot[3 * a + 1]) + rot[3 * a + 1]) / 2.0 + ey0;
                calibrated[4 * a + 2] = enu[4 * a + 2];
                calibrated[4 * a + 3] = enu[4 * a + 3];
            }
        }
    }
    if (iters) *iters = passes;
    free(S4);
    free(E4);
    free(src);
    free(dist);
    if (!rotated) free(rot);
    return 1;
}

int orc_long_segment(const double *slam, const double *enu, int n,
                     int irls_iters, double *w_out, double *fit_out,
                     int quadratic)
{
    if (n <= 0) return -1;
    double *w = (double *)malloc(sizeof(double) * (size_t)n);
    double *pro = (double *)malloc(sizeof(double) * 4 * (size_t)n);
    double *nxt = (double *)malloc(sizeof(double) * 4 * (size_t)n);
    double T[16];
    orc_weights_speed(slam, n, w);                                 /* LD:60 */
    orc_track_fit(slam, enu, w, n, T, NULL, pro, NULL, quadratic); /* LD:65-69 */
    for (int it = 1; it <= irls_iters; ++it) {                     /* LD:72-82 */
        orc_weights_irls(slam, enu, pro, n, w);                    /* LD:76 */
        orc_track_fit(pro, enu, w, n, T, NULL, nxt, NULL, quadratic);
        memcpy(pro, nxt, sizeof(double) * 4 * (size_t)n);
    }
    memcpy(w_out, w, sizeof(double) * (size_t)n); /* LD:83 */
    if (fit_out) memcpy(fit_out, pro, sizeof(double) * 4 * (size_t)n);
    free(w);
    free(pro);
    free(nxt);
    return 1;
}

/* ------------------------------------------------------- short-pass glue  */

int orc_match_gps(const double *gps, int ngps, const double *slam, int nslam,
                  double *slam_out, double *gps_out, double *w_out)
{
    /* SD:39-70 */
    int i = 0, m = 0;
    for (int g = 0; g < ngps; ++g) {
        if (i >= nslam) break;
        double dt = gps[5 * g + 3] - slam[4 * i + 3];
        if (fabs(dt) < 0.000001) {
            memcpy(gps_out + 4 * m, gps + 5 * g, 4 * sizeof(double));
            w_out[m] = gps[5 * g + 4];
            memcpy(slam_out + 4 * m, slam + 4 * i, 4 * sizeof(double));
            ++m;
            ++i;
        } else if (gps[5 * g + 3] > slam[4 * i + 3]) {
            ++i;
            --g;
        }
    }
    return m;
}

int orc_merge_short(double *acc, int *nacc, int cap, const double *seg,
                    const double *segw, int nseg)
{
    /* SD:73-158 */
    int na = *nacc;
    if (na == 0) {
        if (nseg > cap) return -1;
        for (int i = 0; i < nseg; ++i) {
            memcpy(acc + 5 * i, seg + 4 * i, 4 * sizeof(double));
            acc[5 * i + 4] = segw[i];
        }
        *nacc = nseg;
        return 0;
    }
    int it = 0, num = 1, sm = -1, op = -1, overlap = 0;
    int *lost = (int *)malloc(sizeof(int) * (size_t)(na > 0 ? na : 1));
    int nlost = 0;
    double c1 = 0.0, c2 = 0.0;
    for (int a = 0; a < na; ++a) {
        /* NB: the reference indexes slamTrack[indexTmp] without a bound check
         * (SD:101); beyond the end it reads spare capacity.  Defined here as
         * "no match". */
        int match = it < nseg &&
                    fabs(acc[5 * a + 3] - seg[4 * it + 3]) < 0.000001;
        if (match) {
            overlap = 1;
            if (op == -1) { /* SD:104-109 */
                nlost = 0;
                op = na - a;
                sm = op / 2;
            }
            if (num <= sm) { /* SD:110-124 */
                c1 = 1.0 - num / (2.0 * sm);
                c2 = num / (2.0 * sm);
            } else if (num > sm && num <= op - sm) {
                c1 = 0.5;
                c2 = 0.5;
            } else if (num > op - sm) {
                c1 = (op - num + 1) / (2.0 * sm);
                c2 = 1.0 - (op - num + 1) / (2.0 * sm);
            }
            for (int k = 0; k < 3; ++k)
                acc[5 * a + k] = acc[5 * a + k] * c1 + seg[4 * it + k] * c2;
            acc[5 * a + 4] = acc[5 * a + 4] * c1 + segw[it] * c2;
            ++it;
            ++num;
        } else {
            lost[nlost++] = a; /* SD:134 */
        }
    }
    if (!overlap) nlost = 0; /* SD:137-140 */
    int total = na;
    for (; it < nseg; ++it) { /* SD:141-150 */
        if (total >= cap) {
            free(lost);
            return -1;
        }
        memcpy(acc + 5 * total, seg + 4 * it, 4 * sizeof(double));
        acc[5 * total + 4] = segw[it];
        ++total;
    }
    while (nlost > 0) { /* SD:151-156 */
        int idx = lost[--nlost];
        memmove(acc + 5 * idx, acc + 5 * (idx + 1),
                sizeof(double) * 5 * (size_t)(total - idx - 1));
        --total;
    }
    free(lost);
    *nacc = total;
    return 0;
}

/* ------------------------------------------------------ height (TM:116-157) */

int orc_height_compensate(const double *p, int n, double *out)
{
    double px = 0, py = 0, pz = 0, tx = 0, ty = 0;
    for (int i = 0; i < n; ++i) {
        /* LOAM axes: reference x := pos.z, y := pos.x, z := pos.y (TM:120-122) */
        double cx = p[4 * i + 2], cy = p[4 * i + 0], cz = p[4 * i + 1];
        if (i == 0) {
            tx = cx;
            ty = cy;
        } else {
            double dx = cx - px, dy = cy - py, dz = cz - pz;
            double n3 = sqrt(pow(dx, 2) + pow(dy, 2) + pow(dz, 2));
            double n2 = sqrt(pow(dx, 2) + pow(dy, 2));
            tx += dx * n3 / n2; /* TM:131-132 (0/0 -> NaN as coded) */
            ty += dy * n3 / n2;
        }
        px = cx;
        py = cy;
        pz = cz;
        out[4 * i + 0] = tx;
        out[4 * i + 1] = ty;
        out[4 * i + 2] = 10.0; /* HEIGHT, CM.h:16, TM:149 */
        out[4 * i + 3] = p[4 * i + 3];
    }
    return 0;
}
