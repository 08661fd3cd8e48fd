/*
 * sr_oracle.c -- CPU restatement of scanRegistration's feature extraction and of the
 * pcl::VoxelGrid filter LOAM applies to its clouds.  TEST INFRASTRUCTURE ONLY (see
 * gpscal_oracle.h).
 *
 *   SR = src/gpsCalibration/src/lidar_slam/loam/scanRegistration.cpp
 *   ring / time tagging   SR:262-363 (IMU block SR:364-433 inactive: no /imu/data under run.sh)
 *   ring concatenation    SR:444-447
 *   curvature, ring spans SR:455-490
 *   occlusion / parallel  SR:492-548
 *   per-sector picking    SR:558-665
 *   VoxelGrid 0.2         SR:667-673 (PCL 1.8.0 voxel_grid.hpp applyFilter, restated)
 *
 * PARITY UNPINNED: PCL is absent and the reference ships no fixtures for this path.
 * Choices made where the C++ is toolchain dependent:
 *   - common.h:31 says `using namespace std;`, so unqualified sqrt / atan / atan2 / fabs on
 *     float arguments are the float overloads (sqrtf, atanf, atan2f);
 *   - atanf / atan2f are restated as the correctly rounded float of the double function
 *     (what glibc returns except in rare 1-ulp cases), so that this file and the HIP kernel,
 *     which have different libms, agree bit for bit;
 *   - the global work arrays (cloudCurvature, cloudSortInd, cloudNeighborPicked, cloudLabel)
 *     are zero outside [5, cloudSize-5), i.e. the state of a first sweep;
 *   - std::sort in VoxelGrid leaves the order of equal cell ids unspecified; the
 *     restatement keeps input order (a stable sort), which fixes the float summation order.
 */
#include "gpscal_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define N_SCANS 16

static float atan2_f(float y, float x) { return (float)atan2((double)y, (double)x); }
static float atan_f(float x) { return (float)atan((double)x); }

static int ring_of(int roundedAngle)
{
    switch (roundedAngle) { /* SR:307-325 */
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

typedef struct {
    int idx; /* cell id */
    int pos; /* input position */
} vg_item;

static int vg_cmp(const void *a, const void *b)
{
    const vg_item *x = (const vg_item *)a, *y = (const vg_item *)b;
    if (x->idx != y->idx) return x->idx < y->idx ? -1 : 1;
    return x->pos < y->pos ? -1 : (x->pos > y->pos);
}

int orc_voxel_grid(const float *pts, int n, float leaf, float *out, int *n_out)
{
    *n_out = 0;
    if (n <= 0) return 0;
    const float inv = 1.0f / leaf;
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int nfin = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = pts + 4 * i;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        ++nfin;
        for (int a = 0; a < 3; ++a) {
            if (p[a] < mn[a]) mn[a] = p[a];
            if (p[a] > mx[a]) mx[a] = p[a];
        }
    }
    if (!nfin) return 0;
    int64_t d[3];
    for (int a = 0; a < 3; ++a) d[a] = (int64_t)((mx[a] - mn[a]) * inv) + 1;
    if (d[0] * d[1] * d[2] > (int64_t)2147483647) { /* PCL: warn and copy the input */
        memcpy(out, pts, sizeof(float) * 4 * (size_t)n);
        *n_out = n;
        return 1;
    }
    int minb[3], divb[3], mul[3];
    for (int a = 0; a < 3; ++a) {
        minb[a] = (int)floorf(mn[a] * inv);
        divb[a] = (int)floorf(mx[a] * inv) - minb[a] + 1;
    }
    mul[0] = 1;
    mul[1] = divb[0];
    mul[2] = divb[0] * divb[1];
    vg_item *it = (vg_item *)malloc(sizeof(vg_item) * (size_t)n);
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const float *p = pts + 4 * i;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        int ijk[3];
        for (int a = 0; a < 3; ++a) ijk[a] = (int)(floorf(p[a] * inv) - (float)minb[a]);
        it[m].idx = ijk[0] * mul[0] + ijk[1] * mul[1] + ijk[2] * mul[2];
        it[m].pos = i;
        ++m;
    }
    qsort(it, (size_t)m, sizeof(vg_item), vg_cmp);
    int k = 0, no = 0;
    while (k < m) {
        int e = k + 1;
        while (e < m && it[e].idx == it[k].idx) ++e;
        float s[4] = {0, 0, 0, 0};
        for (int j = k; j < e; ++j)
            for (int a = 0; a < 4; ++a) s[a] += pts[4 * it[j].pos + a];
        const float cnt = (float)(e - k);
        for (int a = 0; a < 4; ++a) out[4 * no + a] = s[a] / cnt;
        ++no;
        k = e;
    }
    free(it);
    *n_out = no;
    return 0;
}

int orc_sr_extract(const float *xyz, int n_in, float *full, int *n_full, float *sharp, int *n_sharp,
                   float *less_sharp, int *n_less_sharp, float *flat, int *n_flat, float *less_flat,
                   int *n_less_flat)
{
    *n_full = *n_sharp = *n_less_sharp = *n_flat = *n_less_flat = 0;
    /* removeNaNFromPointCloud, SR:265-266 */
    float *in = (float *)malloc(sizeof(float) * 3 * (size_t)(n_in > 0 ? n_in : 1));
    int n = 0;
    for (int i = 0; i < n_in; ++i) {
        const float *p = xyz + 3 * i;
        if (isfinite(p[0]) && isfinite(p[1]) && isfinite(p[2])) {
            memcpy(in + 3 * n, p, 12);
            ++n;
        }
    }
    if (n < 1) {
        free(in);
        return 0;
    }
    int cloudSize = n;
    float startOri = -atan2_f(in[1], in[0]);                                                 /* SR:270 */
    float endOri = (float)((double)(-atan2_f(in[3 * (n - 1) + 1], in[3 * (n - 1)])) + 2 * M_PI); /* SR:272 */
    if ((double)(endOri - startOri) > 3 * M_PI) endOri = (float)((double)endOri - 2 * M_PI);
    else if ((double)(endOri - startOri) < M_PI) endOri = (float)((double)endOri + 2 * M_PI);
    int halfPassed = 0, count = cloudSize;
    float *tagged = (float *)malloc(sizeof(float) * 4 * (size_t)n);
    int *ring = (int *)malloc(sizeof(int) * (size_t)n);
    int ring_cnt[N_SCANS] = {0};
    for (int i = 0; i < cloudSize; ++i) {
        float px = in[3 * i + 1], py = in[3 * i + 2], pz = in[3 * i]; /* SR:295-297 */
        float angle = (float)((double)(atan_f(py / sqrtf(px * px + pz * pz)) * 180) / M_PI);
        int roundedAngle = (int)(angle + (angle < 0.0 ? -0.5 : +0.5));
        int scanID = ring_of(roundedAngle);
        ring[i] = scanID;
        if (scanID == -1) {
            --count;
            continue;
        }
        float ori = -atan2_f(px, pz);
        if (!halfPassed) { /* SR:341-351 */
            if ((double)ori < (double)startOri - M_PI / 2) ori = (float)((double)ori + 2 * M_PI);
            else if ((double)ori > (double)startOri + M_PI * 3 / 2) ori = (float)((double)ori - 2 * M_PI);
            if ((double)(ori - startOri) > M_PI) halfPassed = 1;
        } else { /* SR:352-359 */
            ori = (float)((double)ori + 2 * M_PI);
            if ((double)ori < (double)endOri - M_PI * 3 / 2) ori = (float)((double)ori + 2 * M_PI);
            else if ((double)ori > (double)endOri + M_PI / 2) ori = (float)((double)ori - 2 * M_PI);
        }
        float relTime = (ori - startOri) / (endOri - startOri);
        tagged[4 * i] = px;
        tagged[4 * i + 1] = py;
        tagged[4 * i + 2] = pz;
        tagged[4 * i + 3] = (float)(scanID + 0.1 * (double)relTime); /* SR:362 */
        ++ring_cnt[scanID];
    }
    cloudSize = count;
    /* SR:444-447: rings concatenated, each in arrival order */
    float *cloud = full;
    {
        int off[N_SCANS + 1];
        off[0] = 0;
        for (int r = 0; r < N_SCANS; ++r) off[r + 1] = off[r] + ring_cnt[r];
        int fill[N_SCANS];
        memcpy(fill, off, sizeof fill);
        for (int i = 0; i < n; ++i)
            if (ring[i] >= 0) memcpy(cloud + 4 * (size_t)fill[ring[i]]++, tagged + 4 * i, 16);
    }
    *n_full = cloudSize;
    free(tagged);
    free(ring);
    free(in);
    const size_t cap = (size_t)cloudSize + 16;
    float *curv = (float *)calloc(cap, sizeof(float));
    int *sortInd = (int *)calloc(cap, sizeof(int));
    int *picked = (int *)calloc(cap, sizeof(int));
    int *label = (int *)calloc(cap, sizeof(int));
    int scanStart[N_SCANS] = {0}, scanEnd[N_SCANS] = {0};
#define P(i, a) cloud[4 * (size_t)(i) + (a)]
    int scanCount = -1;
    for (int i = 5; i < cloudSize - 5; ++i) { /* SR:455-489 */
        float dif[3];
        for (int a = 0; a < 3; ++a)
            dif[a] = P(i - 5, a) + P(i - 4, a) + P(i - 3, a) + P(i - 2, a) + P(i - 1, a) - 10 * P(i, a) + P(i + 1, a) +
                     P(i + 2, a) + P(i + 3, a) + P(i + 4, a) + P(i + 5, a);
        curv[i] = dif[0] * dif[0] + dif[1] * dif[1] + dif[2] * dif[2];
        sortInd[i] = i;
        picked[i] = 0;
        label[i] = 0;
        if ((int)P(i, 3) != scanCount) {
            scanCount = (int)P(i, 3);
            if (scanCount > 0 && scanCount < N_SCANS) {
                scanStart[scanCount] = i + 5;
                scanEnd[scanCount - 1] = i - 5;
            }
        }
    }
    scanStart[0] = 5;
    scanEnd[N_SCANS - 1] = cloudSize - 5;
    for (int i = 5; i < cloudSize - 6; ++i) { /* SR:492-548 */
        float dX = P(i + 1, 0) - P(i, 0), dY = P(i + 1, 1) - P(i, 1), dZ = P(i + 1, 2) - P(i, 2);
        float diff = dX * dX + dY * dY + dZ * dZ;
        if ((double)diff > 0.1) {
            float depth1 = sqrtf(P(i, 0) * P(i, 0) + P(i, 1) * P(i, 1) + P(i, 2) * P(i, 2));
            float depth2 = sqrtf(P(i + 1, 0) * P(i + 1, 0) + P(i + 1, 1) * P(i + 1, 1) + P(i + 1, 2) * P(i + 1, 2));
            if (depth1 > depth2) {
                dX = P(i + 1, 0) - P(i, 0) * depth2 / depth1;
                dY = P(i + 1, 1) - P(i, 1) * depth2 / depth1;
                dZ = P(i + 1, 2) - P(i, 2) * depth2 / depth1;
                if ((double)(sqrtf(dX * dX + dY * dY + dZ * dZ) / depth2) < 0.1)
                    for (int l = -5; l <= 0; ++l) picked[i + l] = 1;
            } else {
                dX = P(i + 1, 0) * depth1 / depth2 - P(i, 0);
                dY = P(i + 1, 1) * depth1 / depth2 - P(i, 1);
                dZ = P(i + 1, 2) * depth1 / depth2 - P(i, 2);
                if ((double)(sqrtf(dX * dX + dY * dY + dZ * dZ) / depth1) < 0.1)
                    for (int l = 1; l <= 6; ++l) picked[i + l] = 1;
            }
        }
        float d2X = P(i, 0) - P(i - 1, 0), d2Y = P(i, 1) - P(i - 1, 1), d2Z = P(i, 2) - P(i - 1, 2);
        float diff2 = d2X * d2X + d2Y * d2Y + d2Z * d2Z;
        float dis = P(i, 0) * P(i, 0) + P(i, 1) * P(i, 1) + P(i, 2) * P(i, 2);
        if ((double)diff > 0.0002 * (double)dis && (double)diff2 > 0.0002 * (double)dis) picked[i] = 1;
    }
    float *lfs = (float *)malloc(sizeof(float) * 4 * cap);
    for (int i = 0; i < N_SCANS; ++i) { /* SR:558-674 */
        int nl = 0;
        for (int j = 0; j < 6; ++j) {
            int sp = (scanStart[i] * (6 - j) + scanEnd[i] * j) / 6;
            int ep = (scanStart[i] * (5 - j) + scanEnd[i] * (j + 1)) / 6 - 1;
            for (int k = sp + 1; k <= ep; ++k) /* SR:567-575 */
                for (int l = k; l >= sp + 1; --l)
                    if (curv[sortInd[l]] < curv[sortInd[l - 1]]) {
                        int t = sortInd[l - 1];
                        sortInd[l - 1] = sortInd[l];
                        sortInd[l] = t;
                    }
            int largest = 0;
            for (int k = ep; k >= sp; --k) { /* SR:578-619 */
                int ind = sortInd[k];
                if (picked[ind] == 0 && (double)curv[ind] > 0.1) {
                    ++largest;
                    if (largest <= 16) {
                        label[ind] = 2;
                        memcpy(sharp + 4 * (size_t)(*n_sharp)++, &P(ind, 0), 16);
                        memcpy(less_sharp + 4 * (size_t)(*n_less_sharp)++, &P(ind, 0), 16);
                    } else if (largest <= 20) {
                        label[ind] = 1;
                        memcpy(less_sharp + 4 * (size_t)(*n_less_sharp)++, &P(ind, 0), 16);
                    } else
                        break;
                    picked[ind] = 1;
                    for (int l = 1; l <= 5; ++l) {
                        if (ind + l >= cloudSize) break; /* guard: the reference would read past the cloud */
                        float aX = P(ind + l, 0) - P(ind + l - 1, 0), aY = P(ind + l, 1) - P(ind + l - 1, 1),
                              aZ = P(ind + l, 2) - P(ind + l - 1, 2);
                        if ((double)(aX * aX + aY * aY + aZ * aZ) > 0.05) break;
                        picked[ind + l] = 1;
                    }
                    for (int l = -1; l >= -5; --l) {
                        if (ind + l < 0) break; /* guard: the reference would read before the cloud (UB) */
                        float aX = P(ind + l, 0) - P(ind + l + 1, 0), aY = P(ind + l, 1) - P(ind + l + 1, 1),
                              aZ = P(ind + l, 2) - P(ind + l + 1, 2);
                        if ((double)(aX * aX + aY * aY + aZ * aZ) > 0.05) break;
                        picked[ind + l] = 1;
                    }
                }
            }
            int smallest = 0;
            for (int k = sp; k <= ep; ++k) { /* SR:621-657 */
                int ind = sortInd[k];
                if (picked[ind] == 0 && (double)curv[ind] < 0.1) {
                    label[ind] = -1;
                    memcpy(flat + 4 * (size_t)(*n_flat)++, &P(ind, 0), 16);
                    ++smallest;
                    if (smallest >= 32) break;
                    picked[ind] = 1;
                    for (int l = 1; l <= 5; ++l) {
                        if (ind + l >= cloudSize) break; /* guard: the reference would read past the cloud */
                        float aX = P(ind + l, 0) - P(ind + l - 1, 0), aY = P(ind + l, 1) - P(ind + l - 1, 1),
                              aZ = P(ind + l, 2) - P(ind + l - 1, 2);
                        if ((double)(aX * aX + aY * aY + aZ * aZ) > 0.05) break;
                        picked[ind + l] = 1;
                    }
                    for (int l = -1; l >= -5; --l) {
                        if (ind + l < 0) break; /* guard: the reference would read before the cloud (UB) */
                        float aX = P(ind + l, 0) - P(ind + l + 1, 0), aY = P(ind + l, 1) - P(ind + l + 1, 1),
                              aZ = P(ind + l, 2) - P(ind + l + 1, 2);
                        if ((double)(aX * aX + aY * aY + aZ * aZ) > 0.05) break;
                        picked[ind + l] = 1;
                    }
                }
            }
            for (int k = sp; k <= ep; ++k) /* SR:659-663 */
                if (label[k] <= 0) memcpy(lfs + 4 * (size_t)nl++, &P(k, 0), 16);
        }
        int nds = 0;
        orc_voxel_grid(lfs, nl, 0.2f, less_flat + 4 * (size_t)(*n_less_flat), &nds); /* SR:667-673 */
        *n_less_flat += nds;
    }
#undef P
    free(lfs);
    free(curv);
    free(sortInd);
    free(picked);
    free(label);
    return 0;
}
