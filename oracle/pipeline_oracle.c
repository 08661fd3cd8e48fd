/*
 * pipeline_oracle.c -- CPU restatement of the four LOAM nodes run in lock step on the raw
 * sweeps of ONE SLAM segment: scanRegistration -> laserOdometry -> laserMapping ->
 * transformMaintenance, i.e. the chain that turns /velodyne_points into the
 * /true_odometry_to_init track the calibration consumes.  TEST INFRASTRUCTURE ONLY.
 *
 *   SR = loam/scanRegistration.cpp   (orc_sr_extract)
 *   LO = loam/laserOdometry.cpp      state handling :495-563, 1030-1124 (+ orc_lo_match)
 *   LM = loam/laserMapping.cpp       :116-203, 244-283, 420-745, 1019-1079 (+ orc_lm_match)
 *   TM = loam/transformMaintenance.cpp :113-157, 178-265, 267-337
 *
 * Schedule: the nodes are separate processes joined by topics; this restatement is the
 * schedule in which every node finishes a sweep before the next sweep arrives (what the
 * reference's 1 Hz bag playback gives): for sweep t, LO runs, TM consumes the odometry with
 * the mapping correction of the PREVIOUS mapped sweep, then LM runs (every second sweep,
 * skipFrameNum = 1) and its correction reaches TM before sweep t+1.
 * The odometry message carries transformSum through a tf quaternion (LO:1068, LM:321-330);
 * that round trip is the identity for |rx| < pi/2 and is restated as such (tf absent).
 * IMU terms are zero (nothing publishes /imu/data under run.sh).
 * laserMapping keeps isDegenerate / matP across sweeps; orc_lm_match starts each sweep with
 * "not degenerate", which differs only if iteration 0 of a sweep selects < 50 points.
 * PARITY UNPINNED (PCL / OpenCV / tf absent, no fixtures).
 */
#include "gpscal_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    float *p; /* float[4] points */
    int n, cap;
} cl;

static void cl_reserve(cl *c, int n)
{
    if (n > c->cap) {
        int cap = c->cap ? c->cap : 64;
        while (cap < n) cap *= 2;
        c->p = (float *)realloc(c->p, sizeof(float) * 4 * (size_t)cap);
        c->cap = cap;
    }
}
static void cl_push(cl *c, const float *pt)
{
    cl_reserve(c, c->n + 1);
    memcpy(c->p + 4 * (size_t)c->n, pt, 16);
    ++c->n;
}
static void cl_append(cl *d, const cl *s)
{
    cl_reserve(d, d->n + s->n);
    if (s->n) memcpy(d->p + 4 * (size_t)d->n, s->p, sizeof(float) * 4 * (size_t)s->n);
    d->n += s->n;
}
static void cl_free(cl *c)
{
    free(c->p);
    c->p = NULL;
    c->n = c->cap = 0;
}
static void cl_voxel(const cl *src, float leaf, cl *dst)
{
    dst->n = 0;
    cl_reserve(dst, src->n > 0 ? src->n : 1);
    int no = 0;
    orc_voxel_grid(src->p, src->n, leaf, dst->p, &no);
    dst->n = no;
}

/* transformAssociateToMap, LM:116-203 == TM:178-265 */
static void assoc_to_map(const float sum[6], const float bef[6], const float aft[6], float incre[6], float out[6])
{
    float x1 = cosf(sum[1]) * (bef[3] - sum[3]) - sinf(sum[1]) * (bef[5] - sum[5]);
    float y1 = bef[4] - sum[4];
    float z1 = sinf(sum[1]) * (bef[3] - sum[3]) + cosf(sum[1]) * (bef[5] - sum[5]);
    float x2 = x1;
    float y2 = cosf(sum[0]) * y1 + sinf(sum[0]) * z1;
    float z2 = -sinf(sum[0]) * y1 + cosf(sum[0]) * z1;
    incre[3] = cosf(sum[2]) * x2 + sinf(sum[2]) * y2;
    incre[4] = -sinf(sum[2]) * x2 + cosf(sum[2]) * y2;
    incre[5] = z2;
    float sbcx = sinf(sum[0]), cbcx = cosf(sum[0]), sbcy = sinf(sum[1]), cbcy = cosf(sum[1]);
    float sbcz = sinf(sum[2]), cbcz = cosf(sum[2]);
    float sblx = sinf(bef[0]), cblx = cosf(bef[0]), sbly = sinf(bef[1]), cbly = cosf(bef[1]);
    float sblz = sinf(bef[2]), cblz = cosf(bef[2]);
    float salx = sinf(aft[0]), calx = cosf(aft[0]), saly = sinf(aft[1]), caly = cosf(aft[1]);
    float salz = sinf(aft[2]), calz = cosf(aft[2]);
    float srx = -sbcx * (salx * sblx + calx * cblx * salz * sblz + calx * calz * cblx * cblz) -
                cbcx * sbcy * (calx * calz * (cbly * sblz - cblz * sblx * sbly) -
                               calx * salz * (cbly * cblz + sblx * sbly * sblz) + cblx * salx * sbly) -
                cbcx * cbcy * (calx * salz * (cblz * sbly - cbly * sblx * sblz) -
                               calx * calz * (sbly * sblz + cbly * cblz * sblx) + cblx * cbly * salx);
    out[0] = -asinf(srx);
    float srycrx = sbcx * (cblx * cblz * (caly * salz - calz * salx * saly) -
                           cblx * sblz * (caly * calz + salx * saly * salz) + calx * saly * sblx) -
                   cbcx * cbcy * ((caly * calz + salx * saly * salz) * (cblz * sbly - cbly * sblx * sblz) +
                                  (caly * salz - calz * salx * saly) * (sbly * sblz + cbly * cblz * sblx) -
                                  calx * cblx * cbly * saly) +
                   cbcx * sbcy * ((caly * calz + salx * saly * salz) * (cbly * cblz + sblx * sbly * sblz) +
                                  (caly * salz - calz * salx * saly) * (cbly * sblz - cblz * sblx * sbly) +
                                  calx * cblx * saly * sbly);
    float crycrx = sbcx * (cblx * sblz * (calz * saly - caly * salx * salz) -
                           cblx * cblz * (saly * salz + caly * calz * salx) + calx * caly * sblx) +
                   cbcx * cbcy * ((saly * salz + caly * calz * salx) * (sbly * sblz + cbly * cblz * sblx) +
                                  (calz * saly - caly * salx * salz) * (cblz * sbly - cbly * sblx * sblz) +
                                  calx * caly * cblx * cbly) -
                   cbcx * sbcy * ((saly * salz + caly * calz * salx) * (cbly * sblz - cblz * sblx * sbly) +
                                  (calz * saly - caly * salx * salz) * (cbly * cblz + sblx * sbly * sblz) -
                                  calx * caly * cblx * sbly);
    out[1] = atan2f(srycrx / cosf(out[0]), crycrx / cosf(out[0]));
    float srzcrx = (cbcz * sbcy - cbcy * sbcx * sbcz) * (calx * salz * (cblz * sbly - cbly * sblx * sblz) -
                                                         calx * calz * (sbly * sblz + cbly * cblz * sblx) +
                                                         cblx * cbly * salx) -
                   (cbcy * cbcz + sbcx * sbcy * sbcz) * (calx * calz * (cbly * sblz - cblz * sblx * sbly) -
                                                         calx * salz * (cbly * cblz + sblx * sbly * sblz) +
                                                         cblx * salx * sbly) +
                   cbcx * sbcz * (salx * sblx + calx * cblx * salz * sblz + calx * calz * cblx * cblz);
    float crzcrx = (cbcy * sbcz - cbcz * sbcx * sbcy) * (calx * calz * (cbly * sblz - cblz * sblx * sbly) -
                                                         calx * salz * (cbly * cblz + sblx * sbly * sblz) +
                                                         cblx * salx * sbly) -
                   (sbcy * sbcz + cbcy * cbcz * sbcx) * (calx * salz * (cblz * sbly - cbly * sblx * sblz) -
                                                         calx * calz * (sbly * sblz + cbly * cblz * sblx) +
                                                         cblx * cbly * salx) +
                   cbcx * cbcz * (salx * sblx + calx * cblx * salz * sblz + calx * calz * cblx * cblz);
    out[2] = atan2f(srzcrx / cosf(out[0]), crzcrx / cosf(out[0]));
    x1 = cosf(out[2]) * incre[3] - sinf(out[2]) * incre[4];
    y1 = sinf(out[2]) * incre[3] + cosf(out[2]) * incre[4];
    z1 = incre[5];
    x2 = x1;
    y2 = cosf(out[0]) * y1 - sinf(out[0]) * z1;
    z2 = sinf(out[0]) * y1 + cosf(out[0]) * z1;
    out[3] = aft[3] - (cosf(out[1]) * x2 + sinf(out[1]) * z2);
    out[4] = aft[4] - y2;
    out[5] = aft[5] - (-sinf(out[1]) * x2 + cosf(out[1]) * z2);
}

void orc_assoc_to_map(const float sum[6], const float bef[6], const float aft[6], float out[6])
{
    float incre[6] = {0, 0, 0, 0, 0, 0};
    assoc_to_map(sum, bef, aft, incre, out);
}

static void to_map(const float tr[6], const float *pi, float *po)
{
    /* pointAssociateToMap, LM:244-262 */
    float x1 = cosf(tr[2]) * pi[0] - sinf(tr[2]) * pi[1];
    float y1 = sinf(tr[2]) * pi[0] + cosf(tr[2]) * pi[1];
    float z1 = pi[2];
    float x2 = x1;
    float y2 = cosf(tr[0]) * y1 - sinf(tr[0]) * z1;
    float z2 = sinf(tr[0]) * y1 + cosf(tr[0]) * z1;
    float ox = cosf(tr[1]) * x2 + sinf(tr[1]) * z2 + tr[3];
    float oy = y2 + tr[4];
    float oz = -sinf(tr[1]) * x2 + cosf(tr[1]) * z2 + tr[5];
    po[0] = ox;
    po[1] = oy;
    po[2] = oz;
    po[3] = pi[3];
}

static void to_be_mapped(const float tr[6], const float *pi, float *po)
{
    /* pointAssociateTobeMapped, LM:264-283 */
    float x1 = cosf(tr[1]) * (pi[0] - tr[3]) - sinf(tr[1]) * (pi[2] - tr[5]);
    float y1 = pi[1] - tr[4];
    float z1 = sinf(tr[1]) * (pi[0] - tr[3]) + cosf(tr[1]) * (pi[2] - tr[5]);
    float x2 = x1;
    float y2 = cosf(tr[0]) * y1 + sinf(tr[0]) * z1;
    float z2 = -sinf(tr[0]) * y1 + cosf(tr[0]) * z1;
    float ox = cosf(tr[2]) * x2 + sinf(tr[2]) * y2;
    float oy = -sinf(tr[2]) * x2 + cosf(tr[2]) * y2;
    po[0] = ox;
    po[1] = oy;
    po[2] = z2;
    po[3] = pi[3];
}

#define LW 21
#define LH 11
#define LD 21
#define LNUM (LW * LH * LD)

typedef struct {
    cl *corner[LNUM], *surf[LNUM];
    int cenW, cenH, cenD;
    float tSum[6], tIncre[6], tTobe[6], tBef[6], tAft[6];
    int inited;
} lm_state;

static void lm_reset(lm_state *S)
{
    for (int i = 0; i < LNUM; ++i) {
        S->corner[i]->n = 0;
        S->surf[i]->n = 0;
    }
    S->cenW = 10;
    S->cenH = 5;
    S->cenD = 10;
    for (int i = 0; i < 6; ++i) S->tIncre[i] = S->tTobe[i] = S->tBef[i] = S->tAft[i] = 0;
}

#define CUBE(i, j, k) ((i) + LW * (j) + LW * LH * (k))

/* one axis shift of the cube ring (LM:501-651): dir = +1 moves contents towards higher
 * indices (the centre was below 3), dir = -1 towards lower */
static void shift_cubes(cl **arr, int axis, int dir)
{
    const int n[3] = {LW, LH, LD};
    int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
    for (int u = 0; u < n[a1]; ++u)
        for (int v = 0; v < n[a2]; ++v) {
            int idx[3];
            idx[a1] = u;
            idx[a2] = v;
            if (dir > 0) {
                idx[axis] = n[axis] - 1;
                cl *last = arr[CUBE(idx[0], idx[1], idx[2])];
                for (int i = n[axis] - 1; i >= 1; --i) {
                    int d[3] = {idx[0], idx[1], idx[2]}, s[3] = {idx[0], idx[1], idx[2]};
                    d[axis] = i;
                    s[axis] = i - 1;
                    arr[CUBE(d[0], d[1], d[2])] = arr[CUBE(s[0], s[1], s[2])];
                }
                idx[axis] = 0;
                arr[CUBE(idx[0], idx[1], idx[2])] = last;
                last->n = 0;
            } else {
                idx[axis] = 0;
                cl *first = arr[CUBE(idx[0], idx[1], idx[2])];
                for (int i = 0; i < n[axis] - 1; ++i) {
                    int d[3] = {idx[0], idx[1], idx[2]}, s[3] = {idx[0], idx[1], idx[2]};
                    d[axis] = i;
                    s[axis] = i + 1;
                    arr[CUBE(d[0], d[1], d[2])] = arr[CUBE(s[0], s[1], s[2])];
                }
                idx[axis] = n[axis] - 1;
                arr[CUBE(idx[0], idx[1], idx[2])] = first;
                first->n = 0;
            }
        }
}

static int cube_of(float v, int cen)
{
    int c = (int)(((double)v + 25.0) / 50.0) + cen; /* LM:1025-1031 */
    if ((double)v + 25.0 < 0) --c;
    return c;
}

/* one laserMapping cycle, LM:420-1148 without the visualisation outputs */
static void lm_step(lm_state *S, const cl *cornerLast, const cl *surfLast, const float odomSum[6], int *iters_out)
{
    if (fabs((double)odomSum[3]) < 0.000001 && fabs((double)odomSum[4]) < 0.000001 && fabs((double)odomSum[5]) < 0.000001)
        S->inited = 0; /* LM:316-319 */
    memcpy(S->tSum, odomSum, sizeof S->tSum);
    if (!S->inited) { /* LM:435-461 */
        S->inited = 1;
        lm_reset(S);
    }
    assoc_to_map(S->tSum, S->tBef, S->tAft, S->tIncre, S->tTobe); /* LM:465 */
    cl cStack2 = {0, 0, 0}, sStack2 = {0, 0, 0}, cStack = {0, 0, 0}, sStack = {0, 0, 0}, cMap = {0, 0, 0},
       sMap = {0, 0, 0};
    float q[4];
    for (int i = 0; i < cornerLast->n; ++i) {
        to_map(S->tTobe, cornerLast->p + 4 * i, q);
        cl_push(&cStack2, q);
    }
    for (int i = 0; i < surfLast->n; ++i) {
        to_map(S->tTobe, surfLast->p + 4 * i, q);
        cl_push(&sStack2, q);
    }
    float onY[4] = {0.0f, 10.0f, 0.0f, 0.0f}, pY[4];
    to_map(S->tTobe, onY, pY); /* LM:483-487 */
    int cI = cube_of(S->tTobe[3], S->cenW), cJ = cube_of(S->tTobe[4], S->cenH), cK = cube_of(S->tTobe[5], S->cenD);
    while (cI < 3) {
        shift_cubes(S->corner, 0, +1);
        shift_cubes(S->surf, 0, +1);
        ++cI;
        ++S->cenW;
    }
    while (cI >= LW - 3) {
        shift_cubes(S->corner, 0, -1);
        shift_cubes(S->surf, 0, -1);
        --cI;
        --S->cenW;
    }
    while (cJ < 3) {
        shift_cubes(S->corner, 1, +1);
        shift_cubes(S->surf, 1, +1);
        ++cJ;
        ++S->cenH;
    }
    while (cJ >= LH - 3) {
        shift_cubes(S->corner, 1, -1);
        shift_cubes(S->surf, 1, -1);
        --cJ;
        --S->cenH;
    }
    while (cK < 3) {
        shift_cubes(S->corner, 2, +1);
        shift_cubes(S->surf, 2, +1);
        ++cK;
        ++S->cenD;
    }
    while (cK >= LD - 3) {
        shift_cubes(S->corner, 2, -1);
        shift_cubes(S->surf, 2, -1);
        --cK;
        --S->cenD;
    }
    int valid[125], nvalid = 0;
    for (int i = cI - 2; i <= cI + 2; ++i) /* LM:653-712 */
        for (int j = cJ - 2; j <= cJ + 2; ++j)
            for (int k = cK - 2; k <= cK + 2; ++k) {
                if (!(i >= 0 && i < LW && j >= 0 && j < LH && k >= 0 && k < LD)) continue;
                float centerX = 50.0 * (i - S->cenW), centerY = 50.0 * (j - S->cenH), centerZ = 50.0 * (k - S->cenD);
                int inFOV = 0;
                for (int ii = -1; ii <= 1; ii += 2)
                    for (int jj = -1; jj <= 1; jj += 2)
                        for (int kk = -1; kk <= 1; kk += 2) {
                            float cornerX = centerX + 25.0 * ii, cornerY = centerY + 25.0 * jj,
                                  cornerZ = centerZ + 25.0 * kk;
                            float s1 = (S->tTobe[3] - cornerX) * (S->tTobe[3] - cornerX) +
                                       (S->tTobe[4] - cornerY) * (S->tTobe[4] - cornerY) +
                                       (S->tTobe[5] - cornerZ) * (S->tTobe[5] - cornerZ);
                            float s2 = (pY[0] - cornerX) * (pY[0] - cornerX) + (pY[1] - cornerY) * (pY[1] - cornerY) +
                                       (pY[2] - cornerZ) * (pY[2] - cornerZ);
                            float check1 = 100.0 + s1 - s2 - 10.0 * sqrt(3.0) * sqrtf(s1);
                            float check2 = 100.0 + s1 - s2 + 10.0 * sqrt(3.0) * sqrtf(s1);
                            if (check1 < 0 && check2 > 0) inFOV = 1;
                        }
                if (inFOV) valid[nvalid++] = CUBE(i, j, k);
            }
    for (int i = 0; i < nvalid; ++i) { /* LM:714-719 */
        cl_append(&cMap, S->corner[valid[i]]);
        cl_append(&sMap, S->surf[valid[i]]);
    }
    for (int i = 0; i < cStack2.n; ++i) to_be_mapped(S->tTobe, cStack2.p + 4 * i, cStack2.p + 4 * i); /* LM:723-731 */
    for (int i = 0; i < sStack2.n; ++i) to_be_mapped(S->tTobe, sStack2.p + 4 * i, sStack2.p + 4 * i);
    cl_voxel(&cStack2, 0.2f, &cStack); /* LM:733-741 */
    cl_voxel(&sStack2, 0.4f, &sStack);
    int iters = 0;
    if (cMap.n > 10 && sMap.n > 100) { /* LM:748 */
        float tr[6];
        orc_lm_match(cStack.p, cStack.n, sStack.p, sStack.n, cMap.p, cMap.n, sMap.p, sMap.n, S->tTobe, tr, &iters, NULL);
        memcpy(S->tTobe, tr, sizeof tr);
        for (int i = 0; i < 6; ++i) { /* transformUpdate, LM:238-241 */
            S->tBef[i] = S->tSum[i];
            S->tAft[i] = S->tTobe[i];
        }
    }
    if (iters_out) *iters_out = iters;
    for (int pass = 0; pass < 2; ++pass) { /* LM:1022-1058 */
        const cl *st = pass == 0 ? &cStack : &sStack;
        cl **arr = pass == 0 ? S->corner : S->surf;
        for (int i = 0; i < st->n; ++i) {
            to_map(S->tTobe, st->p + 4 * i, q);
            int a = cube_of(q[0], S->cenW), b = cube_of(q[1], S->cenH), c = cube_of(q[2], S->cenD);
            if (a >= 0 && a < LW && b >= 0 && b < LH && c >= 0 && c < LD) cl_push(arr[CUBE(a, b, c)], q);
        }
    }
    cl tmp = {0, 0, 0};
    for (int i = 0; i < nvalid; ++i) { /* LM:1060-1078 */
        int ind = valid[i];
        cl_voxel(S->corner[ind], 0.2f, &tmp);
        S->corner[ind]->n = 0;
        cl_append(S->corner[ind], &tmp);
        cl_voxel(S->surf[ind], 0.4f, &tmp);
        S->surf[ind]->n = 0;
        cl_append(S->surf[ind], &tmp);
    }
    cl_free(&tmp);
    cl_free(&cStack2);
    cl_free(&sStack2);
    cl_free(&cStack);
    cl_free(&sStack);
    cl_free(&cMap);
    cl_free(&sMap);
}

/* ---- the node chain as a steppable object: one call per published sweep, and laserOdometry's
 * /control_command reset (LO:411-415) between segments */
typedef struct {
    float transform[6], transformSum[6];
    cl cornerLast, surfLast;
    int lastNumC, lastNumS, frameCount, lo_inited;
    lm_state *S;
    cl *store;
    float mSum[6], mIncre[6], mMapped[6], mBef[6], mAft[6];
    double pre[4], tmpd[4];
} chain;

static chain *chain_new(void)
{
    chain *C = (chain *)calloc(1, sizeof(chain));
    C->frameCount = 1; /* skipFrameNum, LO:495 */
    C->S = (lm_state *)calloc(1, sizeof(lm_state));
    C->store = (cl *)calloc(2 * LNUM, sizeof(cl));
    for (int i = 0; i < LNUM; ++i) {
        C->S->corner[i] = &C->store[i];
        C->S->surf[i] = &C->store[LNUM + i];
    }
    C->S->inited = 0;
    lm_reset(C->S);
    return C;
}

static void chain_free(chain *C)
{
    for (int i = 0; i < 2 * LNUM; ++i) cl_free(&C->store[i]);
    free(C->store);
    free(C->S);
    cl_free(&C->cornerLast);
    cl_free(&C->surfLast);
    free(C);
}

/* /control_command with systemInited = false (ID:283-286,327-330; LO:411-415) */
static void chain_control_reset(chain *C) { C->lo_inited = 0; }

/* one raw sweep through the four nodes.  Returns 1 when /true_odometry_to_init was published
 * (track = {x, y, HEIGHT, stamp}), 0 for the sweep that (re)initialises laserOdometry. */
static int chain_step(chain *C, const float *xyz, int n, double stamp, float *lo_sum, float *lm_aft,
                      float *tm_mapped, double *track, int *lm_iters)
{
    const size_t cap = (size_t)4 * (n > 0 ? n : 1) + 16;
    float *full = (float *)malloc(sizeof(float) * 4 * cap), *sharp = (float *)malloc(sizeof(float) * 4 * cap),
          *lsharp = (float *)malloc(sizeof(float) * 4 * cap), *flat = (float *)malloc(sizeof(float) * 4 * cap),
          *lflat = (float *)malloc(sizeof(float) * 4 * cap);
    int nf, nsh, nls, nfl, nlf, published = 0;
    orc_sr_extract(xyz, n, full, &nf, sharp, &nsh, lsharp, &nls, flat, &nfl, lflat, &nlf);
    if (lm_aft)
        for (int k = 0; k < 6; ++k) lm_aft[k] = NAN;
    if (tm_mapped)
        for (int k = 0; k < 6; ++k) tm_mapped[k] = NAN;
    if (track)
        for (int k = 0; k < 4; ++k) track[k] = NAN;
    if (lm_iters) *lm_iters = -1;
    if (!C->lo_inited) { /* LO:519-562: this sweep only seeds the "last" clouds */
        C->cornerLast.n = C->surfLast.n = 0;
        for (int i = 0; i < nls; ++i) cl_push(&C->cornerLast, lsharp + 4 * i);
        for (int i = 0; i < nlf; ++i) cl_push(&C->surfLast, lflat + 4 * i);
        for (int k = 0; k < 6; ++k) C->transform[k] = C->transformSum[k] = 0;
        C->lastNumC = C->lastNumS = 0; /* LO:520-521: the counters are NOT set from the clouds here */
        C->lo_inited = 1;
        if (lo_sum) memcpy(lo_sum, C->transformSum, sizeof C->transformSum);
    } else {
        if (C->lastNumC > 10 && C->lastNumS > 100) { /* LO:571 */
            float tr[6];
            orc_lo_match(sharp, nsh, flat, nfl, C->cornerLast.p, C->cornerLast.n, C->surfLast.p, C->surfLast.n,
                         C->transform, tr, NULL, NULL);
            memcpy(C->transform, tr, sizeof tr);
        }
        float ns[6];
        orc_lo_accumulate(C->transformSum, C->transform, ns); /* LO:1035-1066 */
        memcpy(C->transformSum, ns, sizeof ns);
        if (lo_sum) memcpy(lo_sum, C->transformSum, sizeof C->transformSum);
        C->cornerLast.n = C->surfLast.n = 0; /* LO:1087-1114 */
        float q[4];
        for (int i = 0; i < nls; ++i) {
            orc_lo_transform_to_end(C->transform, lsharp + 4 * i, q);
            cl_push(&C->cornerLast, q);
        }
        for (int i = 0; i < nlf; ++i) {
            orc_lo_transform_to_end(C->transform, lflat + 4 * i, q);
            cl_push(&C->surfLast, q);
        }
        C->lastNumC = C->cornerLast.n;
        C->lastNumS = C->surfLast.n;
        ++C->frameCount;
        int publish = C->frameCount >= 2; /* skipFrameNum + 1, LO:1126 */
        if (publish) C->frameCount = 0;
        /* ---- TM: laserOdometryHandler, TM:267-314 */
        if (fabs((double)C->transformSum[3]) < 0.000001 && fabs((double)C->transformSum[4]) < 0.000001 &&
            fabs((double)C->transformSum[5]) < 0.000001) {
            C->pre[3] = 0;
            for (int k = 0; k < 6; ++k) C->mSum[k] = C->mIncre[k] = C->mMapped[k] = C->mBef[k] = C->mAft[k] = 0;
        }
        memcpy(C->mSum, C->transformSum, sizeof C->mSum);
        assoc_to_map(C->mSum, C->mBef, C->mAft, C->mIncre, C->mMapped);
        if (tm_mapped) memcpy(tm_mapped, C->mMapped, sizeof C->mMapped);
        { /* SaveTrailWithTimeTotxt, TM:113-157 */
            double px = C->mMapped[5], py = C->mMapped[3], pz = C->mMapped[4];
            double *pre = C->pre, *tmpd = C->tmpd;
            if (pre[3] == 0) {
                pre[0] = px;
                pre[1] = py;
                pre[2] = pz;
                pre[3] = stamp;
                memcpy(tmpd, pre, sizeof C->pre);
            } else {
                double dX = px - pre[0], dY = py - pre[1], dZ = pz - pre[2];
                double dX1 = dX * sqrt(pow(dX, 2) + pow(dY, 2) + pow(dZ, 2)) / sqrt(pow(dX, 2) + pow(dY, 2));
                double dY1 = dY * sqrt(pow(dX, 2) + pow(dY, 2) + pow(dZ, 2)) / sqrt(pow(dX, 2) + pow(dY, 2));
                tmpd[0] += dX1;
                tmpd[1] += dY1;
                tmpd[2] = pz;
                tmpd[3] = stamp;
                pre[0] = px;
                pre[1] = py;
                pre[2] = pz;
                pre[3] = stamp;
            }
            if (track) {
                track[0] = tmpd[0];
                track[1] = tmpd[1];
                track[2] = 10.0; /* HEIGHT, common.h:16 */
                track[3] = tmpd[3];
            }
            published = 1;
        }
        /* ---- LM: every (skipFrameNum+1)-th sweep */
        if (publish) {
            int it = 0;
            lm_step(C->S, &C->cornerLast, &C->surfLast, C->transformSum, &it);
            if (lm_aft) memcpy(lm_aft, C->S->tAft, sizeof C->S->tAft);
            if (lm_iters) *lm_iters = it;
            /* odomAftMappedHandler, TM:316-337 */
            memcpy(C->mAft, C->S->tAft, sizeof C->mAft);
            memcpy(C->mBef, C->S->tBef, sizeof C->mBef);
        }
    }
    free(full);
    free(sharp);
    free(lsharp);
    free(flat);
    free(lflat);
    return published;
}

int orc_loam_run(const float *xyz, const int *sweep_off, int nsweeps, const double *stamps, float *lo_sum,
                 float *lm_aft, float *tm_mapped, double *track, int *lm_iters)
{
    chain *C = chain_new();
    for (int t = 0; t < nsweeps; ++t)
        chain_step(C, xyz + 3 * (size_t)sweep_off[t], sweep_off[t + 1] - sweep_off[t], stamps[t], lo_sum + 6 * t,
                   lm_aft + 6 * t, tm_mapped + 6 * t, track + 4 * t, lm_iters ? lm_iters + t : NULL);
    chain_free(C);
    return 0;
}

/* =====================================================================
 * input_data's replay + segmentation (ID = lidar_slam/input_data/input_data.cpp:78-124, 266-444)
 * around the node chain, for ONE bag (a list of sweeps) and ONE pass (`times`): long pass =
 * (slam_distance, overlap 0), short pass = (slam_distance, overlap > 0).  Emits the tracks
 * input_data publishes on /slam_track, in order: seg_first[k] .. seg_last[k] are the 1-based
 * message indices replayed for track k, track rows {x, y, z, t} as collected by
 * subOdometryHandler (ID:80-88).  Returns the number of tracks (<= cap_tracks), or -1 when the
 * row buffer is too small.
 * ===================================================================== */
int orc_input_data_pass(const float *xyz, const int *sweep_off, int nsweeps, const double *stamps, double slam_distance,
                        double overlap, int cap_tracks, int *seg_first, int *seg_last, int *track_off,
                        double *track_xyzt, int cap_rows)
{
    typedef struct {
        int idx;
        double distance, timestamp;
    } loc;
    chain *C = chain_new();
    loc *all = (loc *)malloc(sizeof(loc) * (size_t)(nsweeps + 4));
    int nall = 0, ntracks = 0, nrows = 0;
    loc pub = {0, 0, 0};
    all[nall++] = pub; /* ID:269-273 */
    double total = 0;
    /* the two most recent tracks are still queued when the replay ends (ID:348-352) */
    int cursor = 0; /* messages published so far in file order = the next message index - 1 */
    int have_pre = 0;
    double prex = 0, prey = 0, prez = 0;
    chain_control_reset(C); /* ID:283-286 */
    track_off[0] = 0;
#define EMIT_BEGIN(first)            \
    do {                             \
        if (ntracks >= cap_tracks) { \
            ntracks = -1;            \
            goto done;               \
        }                            \
        seg_first[ntracks] = (first); \
    } while (0)
    int exhausted = 0;
    while (!exhausted) { /* ID:288 */
        int end = 0;
        EMIT_BEGIN(pub.idx + 1);
        int last_msg = pub.idx;
        for (int i = pub.idx + 1; i <= nsweeps; ++i) { /* ID:311-340: messages after pubLocation */
            double tr[4];
            last_msg = i;
            int got = chain_step(C, xyz + 3 * (size_t)sweep_off[i - 1], sweep_off[i] - sweep_off[i - 1], stamps[i - 1],
                                 NULL, NULL, NULL, tr, NULL);
            if (got) { /* subOdometryHandler, ID:78-118 */
                if (nrows >= cap_rows) {
                    ntracks = -1;
                    goto done;
                }
                memcpy(track_xyzt + 4 * (size_t)nrows, tr, sizeof tr);
                ++nrows;
                loc t;
                t.idx = i;
                t.distance = have_pre ? sqrt(pow(tr[0] - prex, 2) + pow(tr[1] - prey, 2) + pow(tr[2] - prez, 2)) + total : 0;
                t.timestamp = tr[3];
                have_pre = 1;
                prex = tr[0];
                prey = tr[1];
                prez = tr[2];
                if (t.distance <= slam_distance - overlap) pub = t;
                else if (all[nall - 1].timestamp != pub.timestamp) all[nall++] = pub;
                total = t.distance;
            }
            if (total > slam_distance) { /* ID:332-339 */
                total = 0;
                end = 1;
                break;
            }
        }
        if (end) chain_control_reset(C); /* ID:342-346 */
        else exhausted = 1;
        seg_last[ntracks] = last_msg;
        ++ntracks;
        track_off[ntracks] = nrows;
        have_pre = 0; /* preOdometry = NULL, ID:351 */
        cursor = last_msg;
        if (pub.idx >= nsweeps) exhausted = 1;
    }
    (void)cursor;
    /* the rest is too short: replay from the start of the previous track to the end (ID:366-414) */
    if (nall > 1 && total < slam_distance / 3.0 /* IMREST */) {
        loc tmp = all[nall - 2];
        /* the queue holds the last two tracks (or one): they are dropped (ID:377-381) */
        int drop = ntracks >= 2 ? 2 : ntracks;
        ntracks -= drop;
        nrows = track_off[ntracks];
        chain_control_reset(C);
        EMIT_BEGIN(tmp.idx + 1);
        int last_msg = tmp.idx;
        for (int i = tmp.idx + 1; i <= nsweeps; ++i) {
            double tr[4];
            last_msg = i;
            int got = chain_step(C, xyz + 3 * (size_t)sweep_off[i - 1], sweep_off[i] - sweep_off[i - 1], stamps[i - 1],
                                 NULL, NULL, NULL, tr, NULL);
            if (got) {
                if (nrows >= cap_rows) {
                    ntracks = -1;
                    goto done;
                }
                memcpy(track_xyzt + 4 * (size_t)nrows, tr, sizeof tr);
                ++nrows;
            }
        }
        seg_last[ntracks] = last_msg;
        if (nrows > track_off[ntracks]) { /* ID:419-424: only a non-empty track is queued */
            ++ntracks;
            track_off[ntracks] = nrows;
        }
    }
done:
    free(all);
    chain_free(C);
    return ntracks;
}
