// loam_pipeline.hip -- the four LOAM nodes (scanRegistration -> laserOdometry -> laserMapping
// -> transformMaintenance) for many SLAM segments in lock step, device resident.  Segments are
// independent (LOAM is reset per segment), sweeps inside a segment are sequential, so step t
// of the pipeline processes sweep t of every segment with one launch per stage:
//
//   once      scan_registration_kernel over every sweep of every segment          (sr.hip)
//   per t     loam_odometry_kernel (one workgroup per segment)                     (loam.hip)
//             lo_post_kernel: TransformToEnd of the less-sharp / less-flat clouds  LO:1087-1114
//             tm_kernel: transformMaintenance's handler + height compensation      TM:113-157, 267-314
//   odd t     lm_prepare_kernel: transformAssociateToMap, cube ring shift, FOV cube list,
//             map assembly, stack voxel filters                                    LM:420-745
//             loam_mapping_kernel (one workgroup per segment)                      (loam.hip)
//             lm_insert_kernel / lm_filter_kernel / lm_rebuild_kernel: cube insertion and
//             the per-cube voxel filters                                           LM:1019-1079
//
// The 21 x 11 x 21 cube ring of laserMapping (LM:69-75) is one point pool per segment and cloud
// type plus a (start, count) table per cube; a ring shift permutes the table, the per-sweep
// rebuild writes the pool compactly into its double buffer.  Schedule and simplifications are
// those DESIGN.md states for the CPU restatement (every node finishes a sweep before the next
// one arrives; the odometry message's quaternion round trip is the identity; no IMU).
#include "block_utils.hpp"
#include "common.hpp"
#include "knn_device.hpp"
#include "loam_internal.hpp"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cmath>

namespace gpscal {

// The context's worker stream (the second node thread of the LOAM chain enqueues on it): created once, so that its
// block cache serves every later run; gpscal_destroy retires it.
static hipError_t worker_stream_of(gpscal_ctx *ctx, hipStream_t *out)
{
    if (!ctx->worker_stream) {
        hipError_t e = hipStreamCreateWithFlags(&ctx->worker_stream, hipStreamNonBlocking);
        if (e != hipSuccess) return e;
        cache_revive(ctx->worker_stream);
    }
    *out = ctx->worker_stream;
    return hipSuccess;
}

constexpr int LW = 21, LH = 11, LDp = 21, LNUM = LW * LH * LDp;  // LM:72-75
constexpr int MAXVALID = 125;

struct SegState {
    float lo_tr[6], lo_sum[6];                   // laserOdometry: transform, transformSum
    float tSum[6], tTobe[6], tBef[6], tAft[6];   // laserMapping
    int inited, cenW, cenH, cenD;
    int cur;          // which pool / table buffer is current (both cloud types flip together)
    int nvalid;
    int valid[MAXVALID];
    int active, ran;
    float mBef[6], mAft[6];                      // transformMaintenance
    double pre[4], tmpd[4];
};

struct PipeDims {
    int nseg;
    int cap[2];        // pool capacity per segment: corner, surf
    int stack_cap[2];  // stack2 scratch capacity per segment
    int key_cap[2];    // global sort-key capacity per segment
};

// per segment and cloud type (0 corner, 1 surf) views into the big buffers
struct PipeBufs {
    float4 *pool[2][2];       // [type][buffer], nseg * cap[type]
    int *tab_start[2][2];     // [type][buffer], nseg * LNUM
    int *tab_cnt[2][2];
    float4 *frommap[2];       // nseg * cap[type]
    float4 *stack2[2];        // nseg * stack_cap[type]
    float4 *stack[2];         // nseg * stack_cap[type]
    float4 *newq[2];          // nseg * stack_cap[type]: stack points in the map frame, sorted by cube
    int *new_start[2], *new_cnt[2];  // nseg * LNUM
    float4 *vin[2], *vout[2];        // nseg * (cap + stack_cap)
    int *vin_off[2], *vin_cnt[2], *vout_cnt[2];  // nseg * MAXVALID
    unsigned long long *keys[2];     // nseg * key_cap[type]
    unsigned long long *vkeys[2];    // nseg * 2 * (cap + stack_cap)
};

// ---------------------------------------------------------------- small device math
__device__ __forceinline__ void dev_assoc_to_map(const float *sum, const float *bef, const float *aft, float *out)
{
    // transformAssociateToMap, LM:116-203 == TM:178-265
    float x1 = cosf(sum[1]) * (bef[3] - sum[3]) - sinf(sum[1]) * (bef[5] - sum[5]);
    float y1 = bef[4] - sum[4];
    float z1 = sinf(sum[1]) * (bef[3] - sum[3]) + cosf(sum[1]) * (bef[5] - sum[5]);
    float x2 = x1;
    float y2 = cosf(sum[0]) * y1 + sinf(sum[0]) * z1;
    float z2 = -sinf(sum[0]) * y1 + cosf(sum[0]) * z1;
    const float in3 = cosf(sum[2]) * x2 + sinf(sum[2]) * y2;
    const float in4 = -sinf(sum[2]) * x2 + cosf(sum[2]) * y2;
    const float in5 = z2;
    const float sbcx = sinf(sum[0]), cbcx = cosf(sum[0]), sbcy = sinf(sum[1]), cbcy = cosf(sum[1]);
    const float sbcz = sinf(sum[2]), cbcz = cosf(sum[2]);
    const float sblx = sinf(bef[0]), cblx = cosf(bef[0]), sbly = sinf(bef[1]), cbly = cosf(bef[1]);
    const float sblz = sinf(bef[2]), cblz = cosf(bef[2]);
    const float salx = sinf(aft[0]), calx = cosf(aft[0]), saly = sinf(aft[1]), caly = cosf(aft[1]);
    const float salz = sinf(aft[2]), calz = cosf(aft[2]);
    const float srx = -sbcx * (salx * sblx + calx * cblx * salz * sblz + calx * calz * cblx * cblz) -
                      cbcx * sbcy * (calx * calz * (cbly * sblz - cblz * sblx * sbly) -
                                     calx * salz * (cbly * cblz + sblx * sbly * sblz) + cblx * salx * sbly) -
                      cbcx * cbcy * (calx * salz * (cblz * sbly - cbly * sblx * sblz) -
                                     calx * calz * (sbly * sblz + cbly * cblz * sblx) + cblx * cbly * salx);
    out[0] = -asinf(srx);
    const float srycrx = sbcx * (cblx * cblz * (caly * salz - calz * salx * saly) -
                                 cblx * sblz * (caly * calz + salx * saly * salz) + calx * saly * sblx) -
                         cbcx * cbcy * ((caly * calz + salx * saly * salz) * (cblz * sbly - cbly * sblx * sblz) +
                                        (caly * salz - calz * salx * saly) * (sbly * sblz + cbly * cblz * sblx) -
                                        calx * cblx * cbly * saly) +
                         cbcx * sbcy * ((caly * calz + salx * saly * salz) * (cbly * cblz + sblx * sbly * sblz) +
                                        (caly * salz - calz * salx * saly) * (cbly * sblz - cblz * sblx * sbly) +
                                        calx * cblx * saly * sbly);
    const float crycrx = sbcx * (cblx * sblz * (calz * saly - caly * salx * salz) -
                                 cblx * cblz * (saly * salz + caly * calz * salx) + calx * caly * sblx) +
                         cbcx * cbcy * ((saly * salz + caly * calz * salx) * (sbly * sblz + cbly * cblz * sblx) +
                                        (calz * saly - caly * salx * salz) * (cblz * sbly - cbly * sblx * sblz) +
                                        calx * caly * cblx * cbly) -
                         cbcx * sbcy * ((saly * salz + caly * calz * salx) * (cbly * sblz - cblz * sblx * sbly) +
                                        (calz * saly - caly * salx * salz) * (cbly * cblz + sblx * sbly * sblz) -
                                        calx * caly * cblx * sbly);
    out[1] = atan2f(srycrx / cosf(out[0]), crycrx / cosf(out[0]));
    const float srzcrx = (cbcz * sbcy - cbcy * sbcx * sbcz) * (calx * salz * (cblz * sbly - cbly * sblx * sblz) -
                                                               calx * calz * (sbly * sblz + cbly * cblz * sblx) +
                                                               cblx * cbly * salx) -
                         (cbcy * cbcz + sbcx * sbcy * sbcz) * (calx * calz * (cbly * sblz - cblz * sblx * sbly) -
                                                               calx * salz * (cbly * cblz + sblx * sbly * sblz) +
                                                               cblx * salx * sbly) +
                         cbcx * sbcz * (salx * sblx + calx * cblx * salz * sblz + calx * calz * cblx * cblz);
    const float crzcrx = (cbcy * sbcz - cbcz * sbcx * sbcy) * (calx * calz * (cbly * sblz - cblz * sblx * sbly) -
                                                               calx * salz * (cbly * cblz + sblx * sbly * sblz) +
                                                               cblx * salx * sbly) -
                         (sbcy * sbcz + cbcy * cbcz * sbcx) * (calx * salz * (cblz * sbly - cbly * sblx * sblz) -
                                                               calx * calz * (sbly * sblz + cbly * cblz * sblx) +
                                                               cblx * cbly * salx) +
                         cbcx * cbcz * (salx * sblx + calx * cblx * salz * sblz + calx * calz * cblx * cblz);
    out[2] = atan2f(srzcrx / cosf(out[0]), crzcrx / cosf(out[0]));
    x1 = cosf(out[2]) * in3 - sinf(out[2]) * in4;
    y1 = sinf(out[2]) * in3 + cosf(out[2]) * in4;
    z1 = in5;
    x2 = x1;
    y2 = cosf(out[0]) * y1 - sinf(out[0]) * z1;
    z2 = sinf(out[0]) * y1 + cosf(out[0]) * z1;
    out[3] = aft[3] - (cosf(out[1]) * x2 + sinf(out[1]) * z2);
    out[4] = aft[4] - y2;
    out[5] = aft[5] - (-sinf(out[1]) * x2 + cosf(out[1]) * z2);
}

struct Trig6 {
    float s0, c0, s1, c1, s2, c2, t3, t4, t5;
};
__device__ __forceinline__ Trig6 trig_of(const float *tr)
{
    Trig6 g;
    g.s0 = sinf(tr[0]); g.c0 = cosf(tr[0]);
    g.s1 = sinf(tr[1]); g.c1 = cosf(tr[1]);
    g.s2 = sinf(tr[2]); g.c2 = cosf(tr[2]);
    g.t3 = tr[3]; g.t4 = tr[4]; g.t5 = tr[5];
    return g;
}
__device__ __forceinline__ float4 dev_to_map(const Trig6 &g, float4 p)
{
    // pointAssociateToMap, LM:244-262
    const float x1 = g.c2 * p.x - g.s2 * p.y;
    const float y1 = g.s2 * p.x + g.c2 * p.y;
    const float z1 = p.z;
    const float x2 = x1;
    const float y2 = g.c0 * y1 - g.s0 * z1;
    const float z2 = g.s0 * y1 + g.c0 * z1;
    return make_float4(g.c1 * x2 + g.s1 * z2 + g.t3, y2 + g.t4, -g.s1 * x2 + g.c1 * z2 + g.t5, p.w);
}
__device__ __forceinline__ float4 dev_to_be_mapped(const Trig6 &g, float4 p)
{
    // pointAssociateTobeMapped, LM:264-283
    const float x1 = g.c1 * (p.x - g.t3) - g.s1 * (p.z - g.t5);
    const float y1 = p.y - g.t4;
    const float z1 = g.s1 * (p.x - g.t3) + g.c1 * (p.z - g.t5);
    const float x2 = x1;
    const float y2 = g.c0 * y1 + g.s0 * z1;
    const float z2 = -g.s0 * y1 + g.c0 * z1;
    return make_float4(g.c2 * x2 + g.s2 * y2, -g.s2 * x2 + g.c2 * y2, z2, p.w);
}
__device__ __forceinline__ int dev_cube_of(float v, int cen)
{
    int c = (int)(((double)v + 25.0) / 50.0) + cen;  // LM:489-495, 1025-1031
    if ((double)v + 25.0 < 0) --c;
    return c;
}

// TransformToEnd with the IMU stages dropped (LO:156-227), as in loam.hip
__device__ __forceinline__ float4 dev_to_end(const float *tr, float4 p)
{
    const float s = 10 * (p.w - (int)p.w);
    float rx = s * tr[0], ry = s * tr[1], rz = s * tr[2];
    float tx = s * tr[3], ty = s * tr[4], tz = s * tr[5];
    const float x1 = cosf(rz) * (p.x - tx) + sinf(rz) * (p.y - ty);
    const float y1 = -sinf(rz) * (p.x - tx) + cosf(rz) * (p.y - ty);
    const float z1 = (p.z - tz);
    const float x2 = x1;
    const float y2 = cosf(rx) * y1 + sinf(rx) * z1;
    const float z2 = -sinf(rx) * y1 + cosf(rx) * z1;
    const float x3 = cosf(ry) * x2 - sinf(ry) * z2, y3 = y2, z3 = sinf(ry) * x2 + cosf(ry) * z2;
    rx = tr[0]; ry = tr[1]; rz = tr[2]; tx = tr[3]; ty = tr[4]; tz = tr[5];
    const float x4 = cosf(ry) * x3 + sinf(ry) * z3;
    const float y4 = y3;
    const float z4 = -sinf(ry) * x3 + cosf(ry) * z3;
    const float x5 = x4;
    const float y5 = cosf(rx) * y4 - sinf(rx) * z4;
    const float z5 = sinf(rx) * y4 + cosf(rx) * z4;
    return make_float4(cosf(rz) * x5 - sinf(rz) * y5 + tx, sinf(rz) * x5 + cosf(rz) * y5 + ty, z5 + tz,
                       (float)(int)p.w);
}

// ---------------------------------------------------------------- per-step kernels
struct PostDesc {
    long long src_c, src_s;  // less-sharp / less-flat of this sweep (float4 index)
    long long dst_c, dst_s;  // into the new "last" buffers
    int nc, ns, row, identity;
};

// LO:1087-1114 (identity != 0: the first sweep seeds the last clouds untransformed, LO:519-538);
// also stores laserOdometry's transformSum of the sweep.
__global__ void lo_post_kernel(const PostDesc *__restrict__ descs, SegState *__restrict__ st,
                               const float4 *__restrict__ lsharp, const float4 *__restrict__ lflat,
                               const float4 *__restrict__ clast_old, const float4 *__restrict__ slast_old,
                               float4 *__restrict__ clast, float4 *__restrict__ slast, float *__restrict__ lo_sum_out)
{
    const int s = blockIdx.y;
    const PostDesc D = descs[s];
    if (D.row == -1) return;
    const bool carry = D.row == -2;  // an idle stream keeps its last clouds
    __shared__ float tr[6];
    if (threadIdx.x < 6) {
        if (!carry && D.identity && blockIdx.x == 0) {  // laserOdometry (re)initialises: LO:556-562
            st[s].lo_tr[threadIdx.x] = 0.f;
            st[s].lo_sum[threadIdx.x] = 0.f;
        }
        tr[threadIdx.x] = (carry || D.identity) ? 0.f : st[s].lo_tr[threadIdx.x];
    }
    __syncthreads();
    const int n = D.nc + D.ns;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const bool corner = i < D.nc;
        const int j = corner ? i : i - D.nc;
        float4 p;
        if (carry) p = corner ? clast_old[D.src_c + j] : slast_old[D.src_s + j];
        else {
            p = corner ? lsharp[D.src_c + j] : lflat[D.src_s + j];
            if (!D.identity) p = dev_to_end(tr, p);
        }
        if (corner) clast[D.dst_c + j] = p;
        else slast[D.dst_s + j] = p;
    }
    if (!carry && blockIdx.x == 0 && threadIdx.x < 6 && lo_sum_out)
        lo_sum_out[6 * (long long)D.row + threadIdx.x] = D.identity ? 0.f : st[s].lo_sum[threadIdx.x];
}

// transformMaintenance: laserOdometryHandler (TM:267-314) + SaveTrailWithTimeTotxt (TM:113-157)
// `lo_sum` is the step's /laser_odom_to_init message (lo_post_kernel wrote it): laserOdometry may already be a
// sweep ahead when this runs (pipelined chain), SegState::lo_sum is its working copy.
__global__ void tm_kernel(SegState *__restrict__ st, const int *__restrict__ rows, const double *__restrict__ stamps,
                          int nseg, const float *__restrict__ lo_sum, float *__restrict__ tm_out,
                          double *__restrict__ track_out)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nseg) return;
    const int row = rows[s];
    if (row < 0) return;
    SegState &S = st[s];
    float sum[6];
    for (int k = 0; k < 6; ++k) sum[k] = lo_sum[6 * (long long)row + k];
    if (fabs((double)sum[3]) < 0.000001 && fabs((double)sum[4]) < 0.000001 && fabs((double)sum[5]) < 0.000001) {
        S.pre[3] = 0;
        for (int k = 0; k < 6; ++k) S.mBef[k] = S.mAft[k] = 0.f;
    }
    float mapped[6];
    dev_assoc_to_map(sum, S.mBef, S.mAft, mapped);
    if (tm_out)
        for (int k = 0; k < 6; ++k) tm_out[6 * (long long)row + k] = mapped[k];
    const double px = mapped[5], py = mapped[3], pz = mapped[4], stamp = stamps[row];
    if (S.pre[3] == 0) {
        S.pre[0] = px; S.pre[1] = py; S.pre[2] = pz; S.pre[3] = stamp;
        for (int k = 0; k < 4; ++k) S.tmpd[k] = S.pre[k];
    } else {
        const double dX = px - S.pre[0], dY = py - S.pre[1], dZ = pz - S.pre[2];
        const double n3 = sqrt(dX * dX + dY * dY + dZ * dZ), n2 = sqrt(dX * dX + dY * dY);
        S.tmpd[0] += dX * n3 / n2;
        S.tmpd[1] += dY * n3 / n2;
        S.tmpd[2] = pz;
        S.tmpd[3] = stamp;
        S.pre[0] = px; S.pre[1] = py; S.pre[2] = pz; S.pre[3] = stamp;
    }
    track_out[4 * (long long)row] = S.tmpd[0];
    track_out[4 * (long long)row + 1] = S.tmpd[1];
    track_out[4 * (long long)row + 2] = 10.0;  // HEIGHT, common.h:16
    track_out[4 * (long long)row + 3] = S.tmpd[3];
}

struct PrepDesc {
    long long clast_off, slast_off;
    int nc, ns, active, pad;
};

// LM:420-745 for one segment: state reset, transformAssociateToMap, ring shift, FOV cube list,
// map assembly, stack transform + voxel filters.  sizes[s] = {map corner, map surf, stack corner,
// stack surf}.
// The VoxelGrid filters of laserMapping (the stack of a sweep: 3-8 k points; a cube of the map: old + new points)
// sort their keys by one workgroup.  Up to LM_LDS_KEYS keys that happens in LDS (128 KiB of the CU's 160: these
// kernels run one workgroup per stream or per cube anyway); with the 4 096 keys of the other kernels' 32 KiB the
// larger clouds took the HBM path of block_bitonic_sort -- ~100 passes of one workgroup over global memory,
// 0.6 ms (`lm_prepare_kernel`) and 0.85 ms (`lm_filter_kernel`) per mapping step of the bench's bag -> KML run.
constexpr int LM_LDS_KEYS = 16384;
#ifndef GPSCAL_LM_DIAG
#define GPSCAL_LM_DIAG 0  // timing experiments only (wrong results): 1 = lm_prepare without its VoxelGrid filters
#endif
__global__ __launch_bounds__(SBLOCK) void lm_prepare_kernel(const PrepDesc *__restrict__ descs, SegState *__restrict__ st,
                                                            PipeDims dims, PipeBufs B, const float4 *__restrict__ clast,
                                                            const float4 *__restrict__ slast, int *__restrict__ sizes,
                                                            int *__restrict__ status, const float *__restrict__ lo_sum)
{
    extern __shared__ unsigned long long dyn_lds[];
    __shared__ BlockShared S;
    __shared__ float tTobe[6];
    __shared__ int s_shift[3], s_reset, s_valid[MAXVALID], s_vstart[MAXVALID], s_nvalid, s_voff[2][MAXVALID + 1], s_cnt, s_center[6];
    __shared__ float s_pY[3];
    const int s = blockIdx.x;
    const PrepDesc D = descs[s];
    SegState &G = st[s];
    if (threadIdx.x == 0) {
        G.active = D.active;
        S.overflow = 0;
    }
    if (!D.active) {
        if (threadIdx.x < 4) sizes[4 * s + threadIdx.x] = 0;
        if (threadIdx.x == 0) G.nvalid = 0;
        return;
    }
    const int cur = G.cur;
    if (threadIdx.x == 0) {
        float sum[6];
        for (int k = 0; k < 6; ++k) sum[k] = lo_sum[6 * (long long)s + k];  // the step's odometry message
        if (fabs((double)sum[3]) < 0.000001 && fabs((double)sum[4]) < 0.000001 && fabs((double)sum[5]) < 0.000001)
            G.inited = 0;  // LM:316-319
        for (int k = 0; k < 6; ++k) G.tSum[k] = sum[k];
        s_reset = 0;
        if (!G.inited) {  // LM:435-461
            G.inited = 1;
            s_reset = 1;
            G.cenW = 10; G.cenH = 5; G.cenD = 10;
            for (int k = 0; k < 6; ++k) G.tTobe[k] = G.tBef[k] = G.tAft[k] = 0.f;
        }
        float tobe[6];
        dev_assoc_to_map(sum, G.tBef, G.tAft, tobe);  // LM:465
        for (int k = 0; k < 6; ++k) {
            G.tTobe[k] = tobe[k];
            tTobe[k] = tobe[k];
        }
        const Trig6 g = trig_of(tobe);
        const float4 pY = dev_to_map(g, make_float4(0.f, 10.f, 0.f, 0.f));  // LM:483-487
        int cI = dev_cube_of(tobe[3], G.cenW), cJ = dev_cube_of(tobe[4], G.cenH), cK = dev_cube_of(tobe[5], G.cenD);
        int dI = 0, dJ = 0, dK = 0;  // LM:497-651
        while (cI < 3) { ++dI; ++cI; ++G.cenW; }
        while (cI >= LW - 3) { --dI; --cI; --G.cenW; }
        while (cJ < 3) { ++dJ; ++cJ; ++G.cenH; }
        while (cJ >= LH - 3) { --dJ; --cJ; --G.cenH; }
        while (cK < 3) { ++dK; ++cK; ++G.cenD; }
        while (cK >= LDp - 3) { --dK; --cK; --G.cenD; }
        s_shift[0] = dI; s_shift[1] = dJ; s_shift[2] = dK;
        s_center[0] = cI; s_center[1] = cJ; s_center[2] = cK;
        s_center[3] = G.cenW; s_center[4] = G.cenH; s_center[5] = G.cenD;
        s_pY[0] = pY.x; s_pY[1] = pY.y; s_pY[2] = pY.z;
    }
    __syncthreads();
    // ---- the 5 x 5 x 5 cubes around the sensor that the field of view touches (LM:653-712): one lane
    // per cube, kept in the reference's loop order (i, then j, then k) by an ordered compaction
    {
        const int c = threadIdx.x;
        bool inFOV = false;
        int cube = 0;
        if (c < 125) {
            const int i = s_center[0] - 2 + c / 25, j = s_center[1] - 2 + (c / 5) % 5, k = s_center[2] - 2 + c % 5;
            if (i >= 0 && i < LW && j >= 0 && j < LH && k >= 0 && k < LDp) {
                cube = i + LW * j + LW * LH * k;
                const float centerX = (float)(50.0 * (i - s_center[3])), centerY = (float)(50.0 * (j - s_center[4])),
                            centerZ = (float)(50.0 * (k - s_center[5]));
                for (int ii = -1; ii <= 1; ii += 2)
                    for (int jj = -1; jj <= 1; jj += 2)
                        for (int kk = -1; kk <= 1; kk += 2) {
                            const float cX = (float)((double)centerX + 25.0 * ii), cY = (float)((double)centerY + 25.0 * jj),
                                        cZ = (float)((double)centerZ + 25.0 * kk);
                            const float s1 = (tTobe[3] - cX) * (tTobe[3] - cX) + (tTobe[4] - cY) * (tTobe[4] - cY) +
                                             (tTobe[5] - cZ) * (tTobe[5] - cZ);
                            const float s2 = (s_pY[0] - cX) * (s_pY[0] - cX) + (s_pY[1] - cY) * (s_pY[1] - cY) +
                                             (s_pY[2] - cZ) * (s_pY[2] - cZ);
                            const float check1 = (float)(100.0 + (double)s1 - (double)s2 - 10.0 * sqrt(3.0) * (double)sqrtf(s1));
                            const float check2 = (float)(100.0 + (double)s1 - (double)s2 + 10.0 * sqrt(3.0) * (double)sqrtf(s1));
                            if (check1 < 0 && check2 > 0) inFOV = true;
                        }
            }
        }
        int tot;
        const int r = block_rank(S, inFOV, tot);
        if (inFOV) {
            s_valid[r] = cube;
            G.valid[r] = cube;
        }
        if (threadIdx.x == 0) {
            s_nvalid = tot;
            G.nvalid = tot;
        }
    }
    __syncthreads();
    // ---- ring shift of both tables (register staged, in place)
    {
        const int dI = s_shift[0], dJ = s_shift[1], dK = s_shift[2];
        for (int type = 0; type < 2; ++type) {
            int *ts = B.tab_start[type][cur] + (long long)s * LNUM, *tc = B.tab_cnt[type][cur] + (long long)s * LNUM;
            int rs[(LNUM + SBLOCK - 1) / SBLOCK], rc[(LNUM + SBLOCK - 1) / SBLOCK];
#pragma unroll
            for (int u = 0; u < (LNUM + SBLOCK - 1) / SBLOCK; ++u) {
                const int c = threadIdx.x + u * SBLOCK;
                rs[u] = 0;
                rc[u] = 0;
                if (c < LNUM && !s_reset) {
                    const int i = c % LW, j = (c / LW) % LH, k = c / (LW * LH);
                    const int si = i - dI, sj = j - dJ, sk = k - dK;
                    if (si >= 0 && si < LW && sj >= 0 && sj < LH && sk >= 0 && sk < LDp) {
                        const int src = si + LW * sj + LW * LH * sk;
                        rs[u] = ts[src];
                        rc[u] = tc[src];
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < (LNUM + SBLOCK - 1) / SBLOCK; ++u) {
                const int c = threadIdx.x + u * SBLOCK;
                if (c < LNUM) {
                    ts[c] = rs[u];
                    tc[c] = rc[u];
                }
            }
            __syncthreads();
        }
    }
    // ---- map assembly: the valid cubes, in list order (LM:714-719)
    const int nvalid = s_nvalid;
    if (threadIdx.x < 2) {
        const int type = threadIdx.x;
        const int *tc = B.tab_cnt[type][cur] + (long long)s * LNUM;
        int acc = 0;
        for (int k = 0; k < nvalid; ++k) {
            s_voff[type][k] = acc;
            acc += tc[s_valid[k]];
        }
        s_voff[type][nvalid] = acc;
    }
    __syncthreads();
    // the copy is spread over the workgroup by POINT, not by cube (a wave per cube left the largest cube -- 20 k
    // points of a street's ground -- to one wave: most of this kernel's 0.5 ms): the cube of an output slot by
    // bisection of the offsets in LDS
    for (int type = 0; type < 2; ++type) {
        const int *ts = B.tab_start[type][cur] + (long long)s * LNUM;
        const float4 *pool = B.pool[type][cur] + (long long)s * dims.cap[type];
        float4 *dst = B.frommap[type] + (long long)s * dims.cap[type];
        __syncthreads();
        if (threadIdx.x < nvalid) s_vstart[threadIdx.x] = ts[s_valid[threadIdx.x]];
        __syncthreads();
        const int total = s_voff[type][nvalid];
        for (int i = threadIdx.x; i < total; i += SBLOCK) {
            int lo = 0, hi = nvalid;  // last k with s_voff[k] <= i
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_voff[type][mid] <= i) lo = mid; else hi = mid;
            }
            dst[i] = pool[s_vstart[lo] + (i - s_voff[type][lo])];
        }
    }
    // ---- stack: last clouds -> map frame -> back (LM:466-478, 723-731), then VoxelGrid 0.2 / 0.4
    const Trig6 g = trig_of(tTobe);
    for (int type = 0; type < 2; ++type) {
        const float4 *src = type == 0 ? clast + D.clast_off : slast + D.slast_off;
        const int n = type == 0 ? D.nc : D.ns;
        float4 *s2 = B.stack2[type] + (long long)s * dims.stack_cap[type];
        for (int i = threadIdx.x; i < n; i += SBLOCK) s2[i] = dev_to_be_mapped(g, dev_to_map(g, src[i]));
        if (threadIdx.x == 0) s_cnt = 0;
        __syncthreads();
#if GPSCAL_LM_DIAG & 1
        if (threadIdx.x == 0) s_cnt = min(n, 64);
#else
        block_voxel_grid(S, s2, n, type == 0 ? 0.2f : 0.4f, B.stack[type] + (long long)s * dims.stack_cap[type],
                         dims.stack_cap[type], &s_cnt, dyn_lds, B.keys[type] + (long long)s * dims.key_cap[type],
                         dims.key_cap[type], LM_LDS_KEYS);
#endif
        __syncthreads();
        if (threadIdx.x == 0) sizes[4 * s + 2 + type] = s_cnt;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        sizes[4 * s] = s_voff[0][nvalid];
        sizes[4 * s + 1] = s_voff[1][nvalid];
        if (S.overflow) atomicOr(status, 4);
    }
}

// packs the per-segment map / stack slots into the contiguous arrays the grid build wants
struct PackDesc {
    long long dst[4];  // map corner, map surf, stack corner, stack surf
    int n[4];
};
__global__ void lm_pack_kernel(const PackDesc *__restrict__ descs, PipeDims dims, PipeBufs B, float4 *__restrict__ cmap,
                               float4 *__restrict__ smap, float4 *__restrict__ cstack, float4 *__restrict__ sstack)
{
    const int s = blockIdx.y;
    const PackDesc D = descs[s];
    for (int which = 0; which < 4; ++which) {
        const int type = which & 1;
        const float4 *src = which < 2 ? B.frommap[type] + (long long)s * dims.cap[type]
                                      : B.stack[type] + (long long)s * dims.stack_cap[type];
        float4 *dst = (which == 0 ? cmap : which == 1 ? smap : which == 2 ? cstack : sstack) + D.dst[which];
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D.n[which]; i += gridDim.x * blockDim.x) dst[i] = src[i];
    }
}

// LM:1019-1058 for one segment: transformUpdate, then the stack points go to the map frame and
// are grouped by cube (stable), and the inputs of the per-cube voxel filters are laid out.
__global__ __launch_bounds__(SBLOCK) void lm_insert_kernel(SegState *__restrict__ st, PipeDims dims, PipeBufs B,
                                                           const int *__restrict__ sizes, const float *__restrict__ tr_out,
                                                           const int *__restrict__ iters, const int *__restrict__ rows,
                                                           float *__restrict__ lm_aft_out, int *__restrict__ lm_iters_out,
                                                           int *__restrict__ status, int counting)
{
    extern __shared__ unsigned long long dyn_lds[];
    __shared__ float tTobe[6];
    __shared__ int s_voff[MAXVALID + 1], s_oc[MAXVALID], s_st0[MAXVALID], s_nn[MAXVALID], s_n0[MAXVALID];
    // counting sort of the new points by cube (see below)
    constexpr int DMAX = 64;
    __shared__ unsigned s_seen[(LNUM + 31) / 32];
    __shared__ unsigned char s_slot[LNUM];
    __shared__ int s_cube[DMAX], s_nd, s_hist[SWAVES][DMAX], s_off[SWAVES][DMAX];
    const int s = blockIdx.x;
    SegState &G = st[s];
    if (!G.active) return;
    const int cur = G.cur;
    if (threadIdx.x == 0) {
        const bool ran = sizes[4 * s] > 10 && sizes[4 * s + 1] > 100;  // LM:748
        G.ran = ran;
        if (ran) {
            for (int k = 0; k < 6; ++k) {  // transformUpdate, LM:238-241
                G.tTobe[k] = tr_out[6 * s + k];
                G.tBef[k] = G.tSum[k];
                G.tAft[k] = G.tTobe[k];
            }
        }
        for (int k = 0; k < 6; ++k) {
            tTobe[k] = G.tTobe[k];
            // odomAftMappedHandler, TM:316-337: the correction reaches transformMaintenance
            G.mAft[k] = G.tAft[k];
            G.mBef[k] = G.tBef[k];
        }
        const int row = rows[s];
        if (row >= 0) {
            if (lm_aft_out)
                for (int k = 0; k < 6; ++k) lm_aft_out[6 * (long long)row + k] = G.tAft[k];
            if (lm_iters_out) lm_iters_out[row] = ran ? iters[s] : 0;
        }
    }
    __syncthreads();
    const Trig6 g = trig_of(tTobe);
    const int nvalid = G.nvalid;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int type = 0; type < 2; ++type) {
        const int n = sizes[4 * s + 2 + type];
        const float4 *stack = B.stack[type] + (long long)s * dims.stack_cap[type];
        float4 *s2 = B.stack2[type] + (long long)s * dims.stack_cap[type];  // reused: map-frame points, stack order
        float4 *newq = B.newq[type] + (long long)s * dims.stack_cap[type];
        int *ns = B.new_start[type] + (long long)s * LNUM, *nc = B.new_cnt[type] + (long long)s * LNUM;
        const int np2 = next_pow2(max(n, 1));
        unsigned long long *K = np2 <= LDS_KEYS ? dyn_lds : B.keys[type] + (long long)s * dims.key_cap[type];
        if (np2 > LDS_KEYS && np2 > dims.key_cap[type]) {
            if (threadIdx.x == 0) atomicOr(status, 4);
            return;
        }
        for (int c = threadIdx.x; c < LNUM; c += SBLOCK) {
            ns[c] = 0;
            nc[c] = 0;
        }
        // The new points go into their cubes in stack order: a stable sort by cube id.  A sweep reaches only a
        // handful of the 50 m cubes, so instead of a bitonic sort of (cube, position) keys by this one workgroup
        // (27k keys through HBM: ~330 us) the points are COUNTED: distinct cubes -> slots (at most DMAX, else the
        // sort below), every wave takes a contiguous eighth of the stack, per-(wave, slot) counts give each wave its
        // offsets, and the wave places its points tile by tile with ballot ranks -- the same order, bit for bit.
        bool counted = false;
        if (counting && n > LDS_KEYS && n <= dims.key_cap[type]) {
            int *cubes = reinterpret_cast<int *>(B.keys[type] + (long long)s * dims.key_cap[type]);
            for (int w = threadIdx.x; w < (LNUM + 31) / 32; w += SBLOCK) s_seen[w] = 0u;
            for (int k = threadIdx.x; k < SWAVES * DMAX; k += SBLOCK) (&s_hist[0][0])[k] = 0;
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += SBLOCK) {
                const float4 q = dev_to_map(g, stack[i]);
                s2[i] = q;
                const int a = dev_cube_of(q.x, G.cenW), b = dev_cube_of(q.y, G.cenH), c = dev_cube_of(q.z, G.cenD);
                int cube = -1;
                if (a >= 0 && a < LW && b >= 0 && b < LH && c >= 0 && c < LDp) {
                    cube = a + LW * b + LW * LH * c;
                    atomicOr(&s_seen[cube >> 5], 1u << (cube & 31));
                }
                cubes[i] = cube;
            }
            __syncthreads();
            if (threadIdx.x == 0) {  // the cubes that occur, in increasing order
                int nd = 0;
                for (int w = 0; w < (LNUM + 31) / 32; ++w) {
                    unsigned m = s_seen[w];
                    while (m) {
                        const int bit = __ffs(m) - 1;
                        m &= m - 1;
                        if (nd < DMAX) {
                            s_cube[nd] = 32 * w + bit;
                            s_slot[32 * w + bit] = (unsigned char)nd;
                        }
                        ++nd;
                    }
                }
                s_nd = nd;
            }
            __syncthreads();
            if (s_nd <= DMAX) {
                counted = true;
                const int nd = s_nd;
                const int chunk = (n + SWAVES - 1) / SWAVES, c0 = wave * chunk, c1 = min(n, c0 + chunk);
                for (int i = c0 + lane; i < c1; i += 64) {
                    const int cube = cubes[i];
                    if (cube >= 0) atomicAdd(&s_hist[wave][s_slot[cube]], 1);
                }
                __syncthreads();
                if (threadIdx.x == 0) {
                    int acc = 0;
                    for (int d = 0; d < nd; ++d) {
                        ns[s_cube[d]] = acc;
                        for (int w = 0; w < SWAVES; ++w) {
                            s_off[w][d] = acc;
                            acc += s_hist[w][d];
                        }
                        nc[s_cube[d]] = acc - ns[s_cube[d]];
                    }
                }
                __syncthreads();
                for (int i0 = c0; i0 < c1; i0 += 64) {  // wave-uniform
                    const int i = i0 + lane;
                    int slot = -1;
                    if (i < c1) {
                        const int cube = cubes[i];
                        if (cube >= 0) slot = s_slot[cube];
                    }
                    unsigned long long todo = __ballot(slot >= 0);
                    while (todo) {
                        const int d = __shfl(slot, (int)__builtin_ctzll(todo));
                        const unsigned long long m = __ballot(slot == d);
                        todo &= ~m;
                        if (slot == d) newq[s_off[wave][d] + __popcll(m & ((1ull << lane) - 1ull))] = s2[i];
                        if (lane == 0) s_off[wave][d] += __popcll(m);
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                __syncthreads();
            }
        }
        if (!counted) {
        for (int i = threadIdx.x; i < np2; i += SBLOCK) {
            unsigned long long key = ~0ull;
            if (i < n) {
                const float4 q = dev_to_map(g, stack[i]);
                s2[i] = q;
                const int a = dev_cube_of(q.x, G.cenW), b = dev_cube_of(q.y, G.cenH), c = dev_cube_of(q.z, G.cenD);
                if (a >= 0 && a < LW && b >= 0 && b < LH && c >= 0 && c < LDp)
                    key = ((unsigned long long)(unsigned)(a + LW * b + LW * LH * c) << 32) | (unsigned)i;
            }
            K[i] = key;
        }
        __syncthreads();
        block_bitonic_sort(K, np2);
        for (int j = threadIdx.x; j < n; j += SBLOCK) {
            const unsigned long long k = K[j];
            if (k == ~0ull) continue;
            newq[j] = s2[(unsigned)k];
            const unsigned cube = (unsigned)(k >> 32);
            if (j == 0 || (unsigned)(K[j - 1] >> 32) != cube) ns[cube] = j;
        }
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += SBLOCK) {
            const unsigned long long k = K[j];
            if (k == ~0ull) continue;
            const unsigned cube = (unsigned)(k >> 32);
            const bool last = j + 1 >= n || K[j + 1] == ~0ull || (unsigned)(K[j + 1] >> 32) != cube;
            if (last) nc[cube] = j + 1 - ns[cube];
        }
        }  // !counted
        __syncthreads();
        // inputs of the voxel filters of the valid cubes: old points, then the new ones
        const int *ts = B.tab_start[type][cur] + (long long)s * LNUM, *tc = B.tab_cnt[type][cur] + (long long)s * LNUM;
        const long long vcap = (long long)dims.cap[type] + dims.stack_cap[type];
        int *voff = B.vin_off[type] + (long long)s * MAXVALID, *vcnt = B.vin_cnt[type] + (long long)s * MAXVALID;
        if (threadIdx.x < nvalid) {  // the cubes' old and new runs, for the copy below
            const int c = G.valid[threadIdx.x];
            s_oc[threadIdx.x] = tc[c];
            s_st0[threadIdx.x] = ts[c];
            s_nn[threadIdx.x] = nc[c];
            s_n0[threadIdx.x] = ns[c];
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int acc = 0;
            for (int k = 0; k < nvalid; ++k) {
                s_voff[k] = acc;
                voff[k] = acc;
                vcnt[k] = s_oc[k] + s_nn[k];
                acc += s_oc[k] + s_nn[k];
            }
            s_voff[nvalid] = acc;
        }
        __syncthreads();
        const float4 *pool = B.pool[type][cur] + (long long)s * dims.cap[type];
        float4 *vin = B.vin[type] + (long long)s * vcap;
        // by point over the whole workgroup (the cube of a slot by bisection), not a wave per cube: see lm_prepare_kernel
        const int vtotal = s_voff[nvalid];
        for (int i = threadIdx.x; i < vtotal; i += SBLOCK) {
            int lo = 0, hi = nvalid;  // last k with s_voff[k] <= i
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (s_voff[mid] <= i) lo = mid; else hi = mid;
            }
            const int j = i - s_voff[lo];
            vin[i] = j < s_oc[lo] ? pool[s_st0[lo] + j] : newq[s_n0[lo] + (j - s_oc[lo])];
        }
        __syncthreads();
    }
}

// downSizeFilterCorner / downSizeFilterSurf over one valid cube (LM:1060-1078)
__global__ __launch_bounds__(SBLOCK) void lm_filter_kernel(const SegState *__restrict__ st, PipeDims dims, PipeBufs B,
                                                           int *__restrict__ status)
{
    extern __shared__ unsigned long long dyn_lds[];
    __shared__ BlockShared S;
    __shared__ int s_cnt;
    const int k = blockIdx.x, s = blockIdx.y >> 1, type = blockIdx.y & 1;
    const SegState &G = st[s];
    if (!G.active || k >= G.nvalid) return;
    const long long vcap = (long long)dims.cap[type] + dims.stack_cap[type];
    const int off = B.vin_off[type][(long long)s * MAXVALID + k], n = B.vin_cnt[type][(long long)s * MAXVALID + k];
    if (threadIdx.x == 0) {
        s_cnt = 0;
        S.overflow = 0;
    }
    __syncthreads();
    block_voxel_grid(S, B.vin[type] + (long long)s * vcap + off, n, type == 0 ? 0.2f : 0.4f,
                     B.vout[type] + (long long)s * vcap + off, n, &s_cnt, dyn_lds,
                     B.vkeys[type] + 2 * ((long long)s * vcap + off), 2 * max(n, 1), LM_LDS_KEYS);
    __syncthreads();
    if (threadIdx.x == 0) {
        B.vout_cnt[type][(long long)s * MAXVALID + k] = s_cnt;
        if (S.overflow) atomicOr(status, 4);
    }
}

// writes the next pool compactly: filtered valid cubes, the others as old ++ new
__global__ __launch_bounds__(SBLOCK) void lm_rebuild_kernel(SegState *__restrict__ st, PipeDims dims, PipeBufs B,
                                                            int *__restrict__ status)
{
    __shared__ short kmap[LNUM];
    __shared__ int s_part[SBLOCK];
    __shared__ int s_total;
    __shared__ int s_ts2[LNUM + 1];  // the new table's starts (the bisection of the copy below)
    const int s = blockIdx.x >> 1, type = blockIdx.x & 1;
    const SegState &G = st[s];
    if (!G.active) return;
    const int cur = G.cur, nxt = cur ^ 1;
    const int *ts = B.tab_start[type][cur] + (long long)s * LNUM, *tc = B.tab_cnt[type][cur] + (long long)s * LNUM;
    int *ts2 = B.tab_start[type][nxt] + (long long)s * LNUM, *tc2 = B.tab_cnt[type][nxt] + (long long)s * LNUM;
    const int *ns = B.new_start[type] + (long long)s * LNUM, *nc = B.new_cnt[type] + (long long)s * LNUM;
    const long long vcap = (long long)dims.cap[type] + dims.stack_cap[type];
    const int *voff = B.vin_off[type] + (long long)s * MAXVALID, *vocnt = B.vout_cnt[type] + (long long)s * MAXVALID;
    for (int c = threadIdx.x; c < LNUM; c += SBLOCK) kmap[c] = -1;
    __syncthreads();
    if ((int)threadIdx.x < G.nvalid) kmap[G.valid[threadIdx.x]] = (short)threadIdx.x;
    __syncthreads();
    // exclusive scan of the final counts over the cubes: contiguous chunk per thread
    constexpr int PER = (LNUM + SBLOCK - 1) / SBLOCK;
    int loc[PER], sum = 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int c = threadIdx.x * PER + u;
        int f = 0;
        if (c < LNUM) f = kmap[c] >= 0 ? vocnt[kmap[c]] : tc[c] + nc[c];
        loc[u] = f;
        sum += f;
    }
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int i = 0; i < SBLOCK; ++i) {
            const int v = s_part[i];
            s_part[i] = acc;
            acc += v;
        }
        s_total = acc;
    }
    __syncthreads();
    if (s_total > dims.cap[type]) {
        if (threadIdx.x == 0) atomicOr(status, 8);  // pool capacity
        return;
    }
    int run = s_part[threadIdx.x];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int c = threadIdx.x * PER + u;
        if (c < LNUM) {
            ts2[c] = run;
            tc2[c] = loc[u];
            s_ts2[c] = run;
            run += loc[u];
        }
    }
    if (threadIdx.x == 0) s_ts2[LNUM] = s_total;
    __syncthreads();
    const float4 *pool = B.pool[type][cur] + (long long)s * dims.cap[type];
    float4 *pool2 = B.pool[type][nxt] + (long long)s * dims.cap[type];
    const float4 *newq = B.newq[type] + (long long)s * dims.stack_cap[type];
    const float4 *vout = B.vout[type] + (long long)s * vcap;
    // The new pool, by point over the whole workgroup: the cube of a slot by bisection of the new table's starts in
    // LDS.  (A wave per cube walked 600 cubes per wave behind a dependent load of each count, and left the largest
    // cube to one wave.)
    const int total = s_total;
    for (int i = threadIdx.x; i < total; i += SBLOCK) {
        int lo = 0, hi = LNUM;  // last c with s_ts2[c] <= i: empty cubes before it share its start
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (s_ts2[mid] <= i) lo = mid; else hi = mid;
        }
        const int c = lo, j = i - s_ts2[c], k = kmap[c];
        float4 v;
        if (k >= 0) {
            v = vout[voff[k] + j];
        } else {
            const int oc = tc[c];
            v = j < oc ? pool[ts[c] + j] : newq[ns[c] + (j - oc)];
        }
        pool2[i] = v;
    }
}

// moves laserOdometry's transform / transformSum between SegState and the flat arrays the loop
// kernels take; store: only streams that published take the results
__global__ void lo_state_kernel(SegState *__restrict__ st, int nseg, float *__restrict__ tr, float *__restrict__ sum,
                                const int *__restrict__ rows, int store)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseg * 6) return;
    const int s = i / 6, k = i - 6 * s;
    if (!store) {
        tr[i] = st[s].lo_tr[k];
        sum[i] = st[s].lo_sum[k];
    } else if (rows[s] >= 0) {
        st[s].lo_tr[k] = tr[i];
        st[s].lo_sum[k] = sum[i];
    }
}

__global__ void lm_flip_kernel(SegState *__restrict__ st, int nseg)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < nseg && st[s].active) st[s].cur ^= 1;
}

}  // namespace gpscal

using namespace gpscal;

namespace {

#ifndef GPSCAL_LOAM_RING
#define GPSCAL_LOAM_RING 4  // measured at 6 segments: 3 / 4 / 6 slots = 2 170 / 2 230 / 2 235 sweeps/s
#endif

// GPSCAL_LM_COUNTING=0: lm_insert_kernel sorts the new points by cube with the bitonic sort again
inline bool lm_counting()
{
    const char *e = getenv("GPSCAL_LM_COUNTING");
    return e ? atoi(e) != 0 : true;
}

// GPSCAL_LOAM_PIPELINE=0: gpscal_loam_run_batched runs the two halves of a step one after the other again
inline bool loam_pipelined()
{
    const char *e = getenv("GPSCAL_LOAM_PIPELINE");  // read per call: the tests compare both orders in one process
    return e ? atoi(e) != 0 : true;
}

// The node chain for `nstream` independent streams of sweeps, advanced one sweep per step.  All
// sweeps are registered (scanRegistration) up front; a step names, per stream, which sweep is
// published next (or none), and returns what the stream's nodes emitted for it.
struct LoamPipe {
    // ring of hand-over slots between the two halves of a step: laserOdometry may run NRING - 1 sweeps ahead of
    // transformMaintenance + laserMapping (which does its work in bursts, on every second sweep)
    static constexpr int NRING = GPSCAL_LOAM_RING;
    gpscal_ctx *ctx = nullptr;
    int nstream = 0, nsw = 0;
    const int *sweep_off = nullptr;
    std::vector<int> cnt;  // nsw x 5 feature counts
    int max_ls = 1, max_lf = 1;
    PipeDims dims{};
    PipeBufs B{};
    InArg<float> a_xyz;
    DevBuf<float4> d_sharp, d_lsharp, d_flat, d_lflat;
    DevBuf<int> d_counts, d_ring_counts;
    std::vector<int> ring_cnt;      // nsw x 32: less-sharp / less-flat points per ring (from scanRegistration)
    std::vector<int> last_sweep;    // per stream: the sweep whose clouds are the current "last" clouds
    std::vector<int> hring_c, hring_s;
    DevBuf<SegState> d_state;
    DevBuf<float4> b_pool[2][2], b_frommap[2], b_stack2[2], b_stack[2], b_newq[2], b_vin[2], b_vout[2];
    DevBuf<int> b_ts[2][2], b_tc[2][2], b_ns[2], b_nc[2], b_voff[2], b_vcnt[2], b_vocnt[2];
    DevBuf<unsigned long long> b_keys[2], b_vkeys[2];
    DevBuf<float4> d_clast[NRING], d_slast[NRING], d_cmap, d_smap, d_cstack, d_sstack;
    DevBuf<PostDesc> d_post;
    DevBuf<PrepDesc> d_prep;
    DevBuf<PackDesc> d_pack;
    DevBuf<int> d_rows, d_rows_o, d_sizes, d_status, d_iters, d_nsel;
    DevBuf<float> d_tr, d_tr2, d_mtr, d_mtr2, d_step_lo[NRING];  // odometry / mapping scratch
    DevBuf<double> d_step_stamp;
    // the mapping half's per-step outputs in one block (one fill per step): track | tm | lm | iterations
    DevBuf<char> d_step_out;
    size_t step_out_bytes = 0, tm_out_bytes = 0;
    double *d_step_track = nullptr;
    float *d_step_lm = nullptr, *d_step_tm = nullptr;
    int *d_step_it = nullptr;
    std::vector<char> h_step_out;
    std::vector<PostDesc> hpost;
    std::vector<PrepDesc> hprep;
    std::vector<PackDesc> hpack;
    std::vector<SweepDesc> hsw;
    std::vector<MapDesc> hmap;
    std::vector<int> hrows, hrows_o, hsizes;
    std::vector<long long> coff, soff, coff_new, soff_new, cmoff, smoff;
    std::vector<double> hstamp;
    // per-stream host state of laserOdometry's bookkeeping
    std::vector<int> local_t;      // sweeps since the last (re)initialisation; 0 = the next sweep seeds
    std::vector<int> frame_count;  // LO:495,1099-1127
    const double *h_stamps = nullptr;

    // Every buffer of the chain is a block of the context stream's cache: a second run of the same shape takes
    // them back without a single hipMalloc (the map pools alone are ~200 MB per stream of sweeps).
    int init(gpscal_ctx *c, int nstream_, const float *xyz, const int *sweep_off_, int nsw_, const double *stamps,
             int corner_cap, int surf_cap)
    {
        ctx = c;
        nstream = nstream_;
        nsw = nsw_;
        sweep_off = sweep_off_;
        h_stamps = stamps;
        hipStream_t q = ctx->stream;
        {
            // more than the default 64 KiB of dynamic LDS needs the attribute; set per run (it belongs to the current
            // device's copy of the kernel: a process with contexts on several GPUs needs it on each)
            const int bytes = (int)(sizeof(unsigned long long) * LM_LDS_KEYS);
            GPSCAL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(lm_prepare_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
            GPSCAL_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(lm_filter_kernel),
                                                hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        }
        const size_t npts = (size_t)std::max(sweep_off[nsw], 1);
        GPSCAL_HIP(ctx, a_xyz.bind(ctx, xyz, npts * 3));
        DevBuf<float4> d_full;
        GPSCAL_HIP(ctx, d_full.alloc_async(npts, q));
        GPSCAL_HIP(ctx, d_sharp.alloc_async((size_t)nsw * 1536, q));
        GPSCAL_HIP(ctx, d_lsharp.alloc_async((size_t)nsw * 1920, q));
        GPSCAL_HIP(ctx, d_flat.alloc_async((size_t)nsw * 3072, q));
        GPSCAL_HIP(ctx, d_lflat.alloc_async(npts, q));
        GPSCAL_HIP(ctx, d_counts.alloc_async((size_t)nsw * 5, q));
        GPSCAL_HIP(ctx, d_ring_counts.alloc_async((size_t)nsw * 32, q));
        int sr_status = 0;
        int rc = scan_registration_device(ctx, nsw, sweep_off, sweep_off, a_xyz.dev, d_full.p, d_sharp.p, d_lsharp.p,
                                          d_flat.p, d_lflat.p, d_counts.p, &sr_status, d_ring_counts.p);
        if (rc) return rc;
        if (sr_status & 2) return fail(ctx, GPSCAL_ESIZE, "LOAM chain: a sweep has more than 60000 ring points");
        if (sr_status & 1) return fail(ctx, GPSCAL_ERANGE, "LOAM chain: less-flat capacity exceeded (sweep with missing rings)");
        cnt.resize((size_t)nsw * 5);
        ring_cnt.resize((size_t)nsw * 32);
        GPSCAL_HIP(ctx, hipMemcpyAsync(cnt.data(), d_counts.p, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost, q));
        GPSCAL_HIP(ctx, hipMemcpyAsync(ring_cnt.data(), d_ring_counts.p, sizeof(int) * ring_cnt.size(), hipMemcpyDeviceToHost, q));
        GPSCAL_HIP(ctx, hipStreamSynchronize(q));
        for (int g = 0; g < nsw; ++g) {
            max_ls = std::max(max_ls, cnt[5 * g + 2]);
            max_lf = std::max(max_lf, cnt[5 * g + 4]);
        }
        dims.nseg = nstream;
        dims.cap[0] = corner_cap > 0 ? corner_cap : 1 << 18;
        dims.cap[1] = surf_cap > 0 ? surf_cap : 1 << 20;
        dims.stack_cap[0] = max_ls;
        dims.stack_cap[1] = max_lf;
        for (int t = 0; t < 2; ++t) {
            int np2 = 1;
            while (np2 < dims.stack_cap[t]) np2 <<= 1;
            dims.key_cap[t] = np2;
        }
        const int nseg = nstream;
        GPSCAL_HIP(ctx, d_state.alloc_async(nseg, q));
        GPSCAL_HIP(ctx, hipMemsetAsync(d_state.p, 0, sizeof(SegState) * nseg, q));
        for (int t = 0; t < 2; ++t) {
            const size_t vcap = (size_t)dims.cap[t] + dims.stack_cap[t];
            for (int k = 0; k < 2; ++k) {
                GPSCAL_HIP(ctx, b_pool[t][k].alloc_async((size_t)nseg * dims.cap[t], q));
                GPSCAL_HIP(ctx, b_ts[t][k].alloc_async((size_t)nseg * LNUM, q));
                GPSCAL_HIP(ctx, b_tc[t][k].alloc_async((size_t)nseg * LNUM, q));
                GPSCAL_HIP(ctx, hipMemsetAsync(b_ts[t][k].p, 0, sizeof(int) * (size_t)nseg * LNUM, q));
                GPSCAL_HIP(ctx, hipMemsetAsync(b_tc[t][k].p, 0, sizeof(int) * (size_t)nseg * LNUM, q));
                B.pool[t][k] = b_pool[t][k].p;
                B.tab_start[t][k] = b_ts[t][k].p;
                B.tab_cnt[t][k] = b_tc[t][k].p;
            }
            GPSCAL_HIP(ctx, b_frommap[t].alloc_async((size_t)nseg * dims.cap[t], q));
            GPSCAL_HIP(ctx, b_stack2[t].alloc_async((size_t)nseg * dims.stack_cap[t], q));
            GPSCAL_HIP(ctx, b_stack[t].alloc_async((size_t)nseg * dims.stack_cap[t], q));
            GPSCAL_HIP(ctx, b_newq[t].alloc_async((size_t)nseg * dims.stack_cap[t], q));
            GPSCAL_HIP(ctx, b_ns[t].alloc_async((size_t)nseg * LNUM, q));
            GPSCAL_HIP(ctx, b_nc[t].alloc_async((size_t)nseg * LNUM, q));
            GPSCAL_HIP(ctx, b_vin[t].alloc_async((size_t)nseg * vcap, q));
            GPSCAL_HIP(ctx, b_vout[t].alloc_async((size_t)nseg * vcap, q));
            GPSCAL_HIP(ctx, b_voff[t].alloc_async((size_t)nseg * MAXVALID, q));
            GPSCAL_HIP(ctx, b_vcnt[t].alloc_async((size_t)nseg * MAXVALID, q));
            GPSCAL_HIP(ctx, b_vocnt[t].alloc_async((size_t)nseg * MAXVALID, q));
            GPSCAL_HIP(ctx, b_keys[t].alloc_async((size_t)nseg * dims.key_cap[t], q));
            GPSCAL_HIP(ctx, b_vkeys[t].alloc_async((size_t)nseg * 2 * vcap, q));
            B.frommap[t] = b_frommap[t].p;
            B.stack2[t] = b_stack2[t].p;
            B.stack[t] = b_stack[t].p;
            B.newq[t] = b_newq[t].p;
            B.new_start[t] = b_ns[t].p;
            B.new_cnt[t] = b_nc[t].p;
            B.vin[t] = b_vin[t].p;
            B.vout[t] = b_vout[t].p;
            B.vin_off[t] = b_voff[t].p;
            B.vin_cnt[t] = b_vcnt[t].p;
            B.vout_cnt[t] = b_vocnt[t].p;
            B.keys[t] = b_keys[t].p;
            B.vkeys[t] = b_vkeys[t].p;
        }
        for (int k = 0; k < NRING; ++k) {
            GPSCAL_HIP(ctx, d_clast[k].alloc_async((size_t)nseg * max_ls, q));
            GPSCAL_HIP(ctx, d_slast[k].alloc_async((size_t)nseg * max_lf, q));
            GPSCAL_HIP(ctx, d_step_lo[k].alloc_async((size_t)nseg * 6, q));
            GPSCAL_HIP(ctx, hipMemsetAsync(d_step_lo[k].p, 0, sizeof(float) * 6 * nseg, q));
        }
        GPSCAL_HIP(ctx, d_cmap.alloc_async((size_t)nseg * dims.cap[0], q));
        GPSCAL_HIP(ctx, d_smap.alloc_async((size_t)nseg * dims.cap[1], q));
        GPSCAL_HIP(ctx, d_cstack.alloc_async((size_t)nseg * dims.stack_cap[0], q));
        GPSCAL_HIP(ctx, d_sstack.alloc_async((size_t)nseg * dims.stack_cap[1], q));
        GPSCAL_HIP(ctx, d_post.alloc_async(nseg, q));
        GPSCAL_HIP(ctx, d_prep.alloc_async(nseg, q));
        GPSCAL_HIP(ctx, d_pack.alloc_async(nseg, q));
        GPSCAL_HIP(ctx, d_rows.alloc_async(nseg, q));
        GPSCAL_HIP(ctx, d_rows_o.alloc_async(nseg, q));
        GPSCAL_HIP(ctx, d_sizes.alloc_async((size_t)nseg * 4, q));
        GPSCAL_HIP(ctx, d_status.alloc_async(1, q));
        GPSCAL_HIP(ctx, d_iters.alloc_async(nseg, q));
        GPSCAL_HIP(ctx, d_nsel.alloc_async(nseg, q));
        GPSCAL_HIP(ctx, d_tr.alloc_async((size_t)nseg * 6, q));
        GPSCAL_HIP(ctx, d_tr2.alloc_async((size_t)nseg * 6, q));
        GPSCAL_HIP(ctx, d_mtr.alloc_async((size_t)nseg * 6, q));
        GPSCAL_HIP(ctx, d_mtr2.alloc_async((size_t)nseg * 6, q));
        GPSCAL_HIP(ctx, d_step_stamp.alloc_async((size_t)nseg, q));
        step_out_bytes = (size_t)nseg * (4 * sizeof(double) + 12 * sizeof(float) + sizeof(int));
        GPSCAL_HIP(ctx, d_step_out.alloc_async(step_out_bytes, q));
        d_step_track = reinterpret_cast<double *>(d_step_out.p);
        d_step_tm = reinterpret_cast<float *>(d_step_track + (size_t)nseg * 4);
        d_step_lm = d_step_tm + (size_t)nseg * 6;
        d_step_it = reinterpret_cast<int *>(d_step_lm + (size_t)nseg * 6);
        tm_out_bytes = (size_t)nseg * (4 * sizeof(double) + 6 * sizeof(float));  // track | tm: transformMaintenance's part
        h_step_out.resize(step_out_bytes);
        GPSCAL_HIP(ctx, hipMemsetAsync(d_status.p, 0, sizeof(int), q));
        hpost.resize(nseg);
        hprep.resize(nseg);
        hpack.resize(nseg);
        hsw.resize(nseg);
        hmap.resize(nseg);
        hrows.resize(nseg);
        hrows_o.resize(nseg);
        hsizes.resize((size_t)nseg * 4);
        hstamp.resize(nseg);
        coff.assign(nseg + 1, 0);
        soff.assign(nseg + 1, 0);
        coff_new.assign(nseg + 1, 0);
        soff_new.assign(nseg + 1, 0);
        cmoff.assign(nseg + 1, 0);
        smoff.assign(nseg + 1, 0);
        local_t.assign(nseg, 0);
        last_sweep.assign(nseg, -1);
        hring_c.assign((size_t)nseg * 16, 0);
        hring_s.assign((size_t)nseg * 16, 0);
        frame_count.assign(nseg, 1);  // skipFrameNum, LO:495
        // the odometry half may run on another stream: the fills above are done before it starts
        GPSCAL_HIP(ctx, hipStreamSynchronize(q));
        return GPSCAL_OK;
    }

    // /control_command with systemInited = false (ID:283-286, 342-346; LO:411-415)
    void control_reset(int s) { local_t[s] = 0; }

    // What the odometry half of a step hands to the mapping half.  The new "last" clouds and the step's
    // transformSum live in ring entry `buf` of d_clast / d_slast / d_step_lo.
    struct StepSlot {
        std::vector<int> published, do_map, rows_tm;
        std::vector<double> stamp;
        std::vector<long long> coff, soff;
        bool any_tm = false, any_map = false;
        int buf = 0;
    };
    StepSlot slots[NRING];
    long long step_no = 0;  // steps started by step_odo

    double t_sec[6] = {0, 0, 0, 0, 0, 0};  // GPSCAL_LOAM_TIMING: host wall seconds per section of a step
    typedef std::chrono::steady_clock clk;
    void t_add(int k, clk::time_point a) { t_sec[k] += std::chrono::duration<double>(clk::now() - a).count(); }

    // Odometry half of a step, on c->stream (the chain's own context, or a clone of it with another stream
    // when the halves run concurrently): laserOdometry for the named sweeps, then the new "last" clouds
    // (LO:1087-1114) and the step's transformSum into ring entry L.buf.  Synchronises c->stream.
    int step_odo(gpscal_ctx *c, const int *sweep_idx, StepSlot &L)
    {
        auto t0 = clk::now();
        hipStream_t q = c->stream;
        const int nseg = nstream;
        SegState *S = d_state.p;
        const int newbuf = (int)(step_no % NRING), lastbuf = (int)((step_no + NRING - 1) % NRING);
        ++step_no;
        L.buf = newbuf;
        L.published.assign(nseg, 0);
        L.do_map.assign(nseg, 0);
        L.rows_tm.assign(nseg, -1);
        L.stamp.assign(nseg, 0.0);
        L.any_tm = L.any_map = false;
        coff_new[0] = soff_new[0] = 0;
        bool any_match = false;
        for (int s = 0; s < nseg; ++s) {
            const int g = sweep_idx[s];
            const bool act = g >= 0;
            const bool seed = act && local_t[s] == 0;
            hrows_o[s] = act && !seed ? s : -1;  // odometry (and everything after it) is published
            L.rows_tm[s] = hrows_o[s];
            L.stamp[s] = act ? h_stamps[g] : 0.0;
            PostDesc &P = hpost[s];
            P.row = act ? s : -1;
            P.nc = act ? cnt[5 * g + 2] : 0;
            P.ns = act ? cnt[5 * g + 4] : 0;
            P.src_c = act ? (long long)g * 1920 : 0;
            P.src_s = act ? sweep_off[g] : 0;
            P.dst_c = coff_new[s];
            P.dst_s = soff_new[s];
            P.identity = seed;
            // an idle stream keeps its last clouds: carry them over untouched
            if (!act) {
                P.nc = (int)(coff[s + 1] - coff[s]);
                P.ns = (int)(soff[s + 1] - soff[s]);
                P.row = -2;  // copy only
                P.src_c = coff[s];
                P.src_s = soff[s];
            }
            coff_new[s + 1] = coff_new[s] + P.nc;
            soff_new[s + 1] = soff_new[s] + P.ns;
            SweepDesc &D = hsw[s];
            D.sharp_off = act ? (long long)g * 1536 : 0;
            D.flat_off = act ? (long long)g * 3072 : 0;
            D.clast_off = coff[s];
            D.slast_off = soff[s];
            D.nc = act && !seed ? cnt[5 * g + 1] : 0;
            D.ns = act && !seed ? cnt[5 * g + 3] : 0;
            // LO:520-521,571: after a (re)initialisation the counters are still 0 for one sweep
            const bool matchable = act && local_t[s] >= 2;
            D.mc = matchable ? (int)(coff[s + 1] - coff[s]) : 0;
            D.ms = matchable ? (int)(soff[s + 1] - soff[s]) : 0;
            for (int r = 0; r < 16; ++r) {
                hring_c[16 * (size_t)s + r] = last_sweep[s] >= 0 ? ring_cnt[32 * (size_t)last_sweep[s] + r] : 0;
                hring_s[16 * (size_t)s + r] = last_sweep[s] >= 0 ? ring_cnt[32 * (size_t)last_sweep[s] + 16 + r] : 0;
            }
            any_match = any_match || (act && !seed);
            L.published[s] = act && !seed;
            L.any_tm = L.any_tm || L.published[s];
            if (L.published[s]) {
                if (++frame_count[s] >= 2) {  // skipFrameNum + 1, LO:1099-1127
                    frame_count[s] = 0;
                    L.do_map[s] = 1;
                    L.any_map = true;
                }
            }
        }
        GPSCAL_HIP(c, hipMemcpyAsync(d_rows_o.p, hrows_o.data(), sizeof(int) * nseg, hipMemcpyHostToDevice, q));
        GPSCAL_HIP(c, hipMemcpyAsync(d_post.p, hpost.data(), sizeof(PostDesc) * nseg, hipMemcpyHostToDevice, q));
        t_add(0, t0);
        t0 = clk::now();
        if (any_match) {
            // transform / transformSum live in SegState; the kernels take flat [nstream][6] arrays.  A
            // stream that idles or seeds has empty clouds here: its state passes through unchanged
            // (zero transform accumulates to itself only for a zero sum, so those rows are restored).
            hipLaunchKernelGGL(lo_state_kernel, dim3(div_up(nseg * 6, 64)), dim3(64), 0, q, S, nseg, d_tr.p, d_tr2.p, d_rows_o.p, 0);
            int rc = loam_odometry_device(c, nseg, hsw.data(), d_sharp.p, d_flat.p, d_clast[lastbuf].p, d_slast[lastbuf].p,
                                          coff.data(), soff.data(), d_tr.p, d_tr.p, nullptr, nullptr, d_tr2.p, d_tr2.p,
                                          hring_c.data(), hring_s.data());
            if (rc) return rc;
            hipLaunchKernelGGL(lo_state_kernel, dim3(div_up(nseg * 6, 64)), dim3(64), 0, q, S, nseg, d_tr.p, d_tr2.p, d_rows_o.p, 1);
        }
        {
            const int gx = std::max(1, std::min(div_up(std::max(max_ls + max_lf, 1), 256), 64));
            hipLaunchKernelGGL(lo_post_kernel, dim3(gx, nseg), dim3(256), 0, q, d_post.p, S, d_lsharp.p, d_lflat.p,
                               d_clast[lastbuf].p, d_slast[lastbuf].p, d_clast[newbuf].p, d_slast[newbuf].p,
                               d_step_lo[newbuf].p);
            GPSCAL_HIP(c, hipGetLastError());
        }
        coff = coff_new;
        soff = soff_new;
        L.coff = coff;
        L.soff = soff;
        for (int s = 0; s < nseg; ++s)
            if (sweep_idx[s] >= 0) {
                ++local_t[s];
                last_sweep[s] = sweep_idx[s];
            }
        GPSCAL_HIP(c, hipStreamSynchronize(q));  // the hand-over to the mapping half is a host-side one
        t_add(1, t0);
        return GPSCAL_OK;
    }

    // transformMaintenance's part of a step, on ctx->stream (TM:267-314, 113-157): needs the step's odometry and the
    // LAST mapping cycle's correction only, so the step's track sample is known before its own mapping cycle runs --
    // which is what lets input_data's cut decision (it reads the track) be taken while laserMapping still works.
    // Host outputs, nstream rows each (may be null): tm poses, track.  Synchronises the stream.
    int step_tm(const StepSlot &L, float *tm, double *track)
    {
        auto t0 = clk::now();
        hipStream_t q = ctx->stream;
        const int nseg = nstream;
        SegState *S = d_state.p;
        GPSCAL_HIP(ctx, hipMemsetAsync(d_step_out.p, 0xff, step_out_bytes, q));  // track | tm | lm | iterations
        if (L.any_tm) {
            GPSCAL_HIP(ctx, hipMemcpyAsync(d_rows.p, L.rows_tm.data(), sizeof(int) * nseg, hipMemcpyHostToDevice, q));
            GPSCAL_HIP(ctx, hipMemcpyAsync(d_step_stamp.p, L.stamp.data(), sizeof(double) * nseg, hipMemcpyHostToDevice, q));
            hipLaunchKernelGGL(tm_kernel, dim3(div_up(nseg, 64)), dim3(64), 0, q, S, d_rows.p, d_step_stamp.p, nseg,
                               d_step_lo[L.buf].p, d_step_tm, d_step_track);
            GPSCAL_HIP(ctx, hipGetLastError());
        }
        if (tm || track) {
            GPSCAL_HIP(ctx, hipMemcpyAsync(h_step_out.data(), d_step_out.p, tm_out_bytes, hipMemcpyDeviceToHost, q));
            GPSCAL_HIP(ctx, hipStreamSynchronize(q));
            const char *h = h_step_out.data();
            if (track) memcpy(track, h, sizeof(double) * 4 * nseg);
            if (tm) memcpy(tm, h + sizeof(double) * 4 * nseg, sizeof(float) * 6 * nseg);
        }
        t_add(2, t0);
        return GPSCAL_OK;
    }

    // laserMapping's part of a step, on ctx->stream (LM:420-1079, every second published sweep), after step_tm of
    // the same step.  Host outputs (may be null): lo / lm poses, iterations.  Synchronises the stream.
    int step_mapping(const StepSlot &L, float *lo, float *lm, int *iters, float *tm = nullptr, double *track = nullptr)
    {
        auto t0 = clk::now();
        hipStream_t q = ctx->stream;
        const int nseg = nstream;
        SegState *S = d_state.p;
        const int buf = L.buf;
        const size_t lds_keys = sizeof(unsigned long long) * LDS_KEYS, lm_lds = sizeof(unsigned long long) * LM_LDS_KEYS;
        const std::vector<long long> &coff = L.coff, &soff = L.soff;
        if (L.any_map) {
            for (int s = 0; s < nseg; ++s) {
                PrepDesc &P = hprep[s];
                P.clast_off = coff[s];
                P.slast_off = soff[s];
                P.nc = (int)(coff[s + 1] - coff[s]);
                P.ns = (int)(soff[s + 1] - soff[s]);
                P.active = L.do_map[s];
                P.pad = 0;
                hrows[s] = L.do_map[s] ? s : -1;
            }
            GPSCAL_HIP(ctx, hipMemcpyAsync(d_rows.p, hrows.data(), sizeof(int) * nseg, hipMemcpyHostToDevice, q));
            GPSCAL_HIP(ctx, hipMemcpyAsync(d_prep.p, hprep.data(), sizeof(PrepDesc) * nseg, hipMemcpyHostToDevice, q));
            hipLaunchKernelGGL(lm_prepare_kernel, dim3(nseg), dim3(SBLOCK), lm_lds, q, d_prep.p, S, dims, B,
                               d_clast[buf].p, d_slast[buf].p, d_sizes.p, d_status.p, d_step_lo[buf].p);
            GPSCAL_HIP(ctx, hipGetLastError());
            GPSCAL_HIP(ctx, hipMemcpyAsync(hsizes.data(), d_sizes.p, sizeof(int) * hsizes.size(), hipMemcpyDeviceToHost, q));
            GPSCAL_HIP(ctx, hipStreamSynchronize(q));
            cmoff[0] = smoff[0] = 0;
            long long cso = 0, sso = 0;
            int nmax = 1;
            for (int s = 0; s < nseg; ++s) {
                const int mc = hsizes[4 * s], ms = hsizes[4 * s + 1], nc = hsizes[4 * s + 2], ns = hsizes[4 * s + 3];
                PackDesc &P = hpack[s];
                P.dst[0] = cmoff[s]; P.dst[1] = smoff[s]; P.dst[2] = cso; P.dst[3] = sso;
                P.n[0] = mc; P.n[1] = ms; P.n[2] = nc; P.n[3] = ns;
                MapDesc &M = hmap[s];
                M.cmap_off = cmoff[s]; M.smap_off = smoff[s]; M.cstack_off = cso; M.sstack_off = sso;
                M.mc = mc; M.ms = ms; M.nc = nc; M.ns = ns;
                cmoff[s + 1] = cmoff[s] + mc;
                smoff[s + 1] = smoff[s] + ms;
                cso += nc;
                sso += ns;
                nmax = std::max(nmax, std::max(std::max(mc, ms), std::max(nc, ns)));
            }
            t_add(2, t0);
            t0 = clk::now();
            GPSCAL_HIP(ctx, hipMemcpyAsync(d_pack.p, hpack.data(), sizeof(PackDesc) * nseg, hipMemcpyHostToDevice, q));
            hipLaunchKernelGGL(lm_pack_kernel, dim3(std::max(1, std::min(div_up(nmax, 256), 128)), nseg), dim3(256), 0, q,
                               d_pack.p, dims, B, d_cmap.p, d_smap.p, d_cstack.p, d_sstack.p);
            GPSCAL_HIP(ctx, hipGetLastError());
            GPSCAL_HIP(ctx, hipMemcpy2DAsync(d_mtr.p, 24, &S[0].tTobe[0], sizeof(SegState), 24, nseg, hipMemcpyDeviceToDevice, q));
            int rc = loam_mapping_device(ctx, nseg, hmap.data(), d_cstack.p, d_sstack.p, d_cmap.p, d_smap.p, cmoff.data(),
                                         smoff.data(), d_mtr.p, d_mtr2.p, d_iters.p, d_nsel.p);
            if (rc) return rc;
            t_add(3, t0);
            t0 = clk::now();
            hipLaunchKernelGGL(lm_insert_kernel, dim3(nseg), dim3(SBLOCK), lds_keys, q, S, dims, B, d_sizes.p, d_mtr2.p,
                               d_iters.p, d_rows.p, d_step_lm, d_step_it, d_status.p, lm_counting() ? 1 : 0);
            hipLaunchKernelGGL(lm_filter_kernel, dim3(MAXVALID, nseg * 2), dim3(SBLOCK), lm_lds, q, S, dims, B, d_status.p);
            hipLaunchKernelGGL(lm_rebuild_kernel, dim3(nseg * 2), dim3(SBLOCK), 0, q, S, dims, B, d_status.p);
            hipLaunchKernelGGL(lm_flip_kernel, dim3(div_up(nseg, 64)), dim3(64), 0, q, S, nseg);
            GPSCAL_HIP(ctx, hipGetLastError());
        }
        t_add(4, t0);
        t0 = clk::now();
        // (tm / track given: transformMaintenance's outputs of this step come back with the same read-back --
        // one host wait per step when nothing on the host needs them earlier)
        const bool with_tm = tm || track;
        if (with_tm)
            GPSCAL_HIP(ctx, hipMemcpyAsync(h_step_out.data(), d_step_out.p, step_out_bytes, hipMemcpyDeviceToHost, q));
        else if (lm || iters)
            GPSCAL_HIP(ctx, hipMemcpyAsync(h_step_out.data() + tm_out_bytes, d_step_out.p + tm_out_bytes,
                                           step_out_bytes - tm_out_bytes, hipMemcpyDeviceToHost, q));
        if (lo) GPSCAL_HIP(ctx, hipMemcpyAsync(lo, d_step_lo[buf].p, sizeof(float) * 6 * nseg, hipMemcpyDeviceToHost, q));
        GPSCAL_HIP(ctx, hipStreamSynchronize(q));
        {
            const char *h = h_step_out.data() + tm_out_bytes;
            if (lm) memcpy(lm, h, sizeof(float) * 6 * nseg);
            if (iters) memcpy(iters, h + sizeof(float) * 6 * nseg, sizeof(int) * nseg);
            if (track) memcpy(track, h_step_out.data(), sizeof(double) * 4 * nseg);
            if (tm) memcpy(tm, h_step_out.data() + sizeof(double) * 4 * nseg, sizeof(float) * 6 * nseg);
        }
        t_add(5, t0);
        return GPSCAL_OK;
    }

    // Mapping half of a step: both parts.
    int step_map(const StepSlot &L, float *lo, float *lm, float *tm, double *track, int *iters)
    {
        int rc = step_tm(L, nullptr, nullptr);  // enqueued only: its outputs come back with the mapping part's
        if (rc) return rc;
        return step_mapping(L, lo, lm, iters, tm, track);
    }

    // One sweep per stream (sweep_idx[s] < 0: the stream idles), both halves one after the other on the
    // chain's stream.  Host outputs, nstream rows each: published[s] (odometry emitted), mapped[s]
    // (laserMapping ran), lo / lm / tm poses, track, iters.
    int step(const int *sweep_idx, int *published, int *mapped, float *lo, float *lm, float *tm, double *track,
             int *iters)
    {
        StepSlot &L = slots[step_no % NRING];
        int rc = step_odo(ctx, sweep_idx, L);
        if (rc) return rc;
        for (int s = 0; s < nstream; ++s) {
            published[s] = L.published[s];
            mapped[s] = L.do_map[s];
        }
        return step_map(L, lo, lm, tm, track, iters);
    }

    int finish()
    {
        if (getenv("GPSCAL_LOAM_TIMING"))
            fprintf(stderr, "LoamPipe host seconds: setup %.3f | odometry %.3f | post+tm+prepare %.3f | pack+mapping %.3f | map update enqueue %.3f | final sync %.3f\n",
                    t_sec[0], t_sec[1], t_sec[2], t_sec[3], t_sec[4], t_sec[5]);
        int st = 0;
        // on the library's stream: a null-stream copy would not wait for it (the stream is non-blocking)
        GPSCAL_HIP(ctx, hipMemcpyAsync(&st, d_status.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        GPSCAL_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (st & 8) return fail(ctx, GPSCAL_ENOMEM, "LOAM chain: map pool capacity exceeded (raise corner_pool_cap / surf_pool_cap)");
        if (st & 4) return fail(ctx, GPSCAL_ERANGE, "LOAM chain: internal scratch capacity exceeded");
        return GPSCAL_OK;
    }
};

}  // namespace

extern "C" int gpscal_loam_run_batched(gpscal_ctx *ctx, int nseg, const float *xyz, const int *sweep_off,
                                       const int *seg_sweep_off, const double *stamps, float *lo_sum_out,
                                       float *lm_aft_out, float *tm_mapped_out, double *track_xyzt, int *lm_iters_out,
                                       int corner_pool_cap, int surf_pool_cap)
{
    if (!ctx || nseg < 1 || !xyz || !sweep_off || !seg_sweep_off || !stamps || !track_xyzt)
        return fail(ctx, GPSCAL_EINVAL, "gpscal_loam_run_batched: bad argument");
    if (is_device_ptr(stamps) || is_device_ptr(track_xyzt) || is_device_ptr(lo_sum_out) || is_device_ptr(lm_aft_out) ||
        is_device_ptr(tm_mapped_out) || is_device_ptr(lm_iters_out))
        return fail(ctx, GPSCAL_EINVAL, "gpscal_loam_run_batched: stamps and the per-sweep outputs are host arrays");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    const int nsw = seg_sweep_off[nseg] - seg_sweep_off[0];
    if (nsw < 1 || seg_sweep_off[0] != 0) return fail(ctx, GPSCAL_EINVAL, "gpscal_loam_run_batched: bad segment offsets");
    for (int s = 0; s < nseg; ++s)
        if (seg_sweep_off[s + 1] < seg_sweep_off[s]) return fail(ctx, GPSCAL_EINVAL, "gpscal_loam_run_batched: bad segment offsets");
    // Segments are independent, so they run in groups sized to the free HBM: a stream holds its map
    // pools, filter scratch and packed clouds (~200 MB at the default pool capacities).
    const long long c0 = corner_pool_cap > 0 ? corner_pool_cap : 1 << 18, c1 = surf_pool_cap > 0 ? surf_pool_cap : 1 << 20;
    const double per_stream = 16.0 * 10.5 * (double)(c0 + c1);  // bytes, see LoamPipe::init
    size_t free_b = 0, total_b = 0;
    GPSCAL_HIP(ctx, hipMemGetInfo(&free_b, &total_b));
    int group = (int)std::max(1.0, std::min((double)nseg, 0.6 * (double)free_b / per_stream));
    if (const char *e = getenv("GPSCAL_LOAM_GROUP")) group = std::max(1, std::min(nseg, atoi(e)));
    const float fnan = std::nanf("");
    const double dnan = std::nan("");
    for (int s0 = 0; s0 < nseg; s0 += group) {
        const int ng = std::min(group, nseg - s0);
        const int w0 = seg_sweep_off[s0], w1 = seg_sweep_off[s0 + ng], nw = w1 - w0;
        if (nw < 1) continue;
        // the group's sweeps, offsets rebased to its first point
        std::vector<int> off(nw + 1);
        for (int k = 0; k <= nw; ++k) off[k] = sweep_off[w0 + k] - sweep_off[w0];
        int gmax = 0;
        for (int s = 0; s < ng; ++s) gmax = std::max(gmax, seg_sweep_off[s0 + s + 1] - seg_sweep_off[s0 + s]);
        LoamPipe P;
        int rc = P.init(ctx, ng, xyz + 3 * (size_t)sweep_off[w0], off.data(), nw, stamps + w0, corner_pool_cap, surf_pool_cap);
        if (rc) return rc;
        std::vector<int> idx(ng), pub(ng), mapd(ng), its(ng);
        std::vector<float> lo((size_t)ng * 6), lm((size_t)ng * 6), tm((size_t)ng * 6);
        std::vector<double> tr((size_t)ng * 4);
        auto sweeps_of_step = [&](int t, int *out) {
            for (int s = 0; s < ng; ++s) {
                const int a0 = seg_sweep_off[s0 + s], a1 = seg_sweep_off[s0 + s + 1];
                out[s] = t < a1 - a0 ? a0 + t - w0 : -1;
            }
        };
        auto scatter = [&]() {
            for (int s = 0; s < ng; ++s) {
                if (idx[s] < 0) continue;
                const size_t g = (size_t)idx[s] + w0;
                for (int k = 0; k < 6; ++k) {
                    if (lo_sum_out) lo_sum_out[6 * g + k] = lo[6 * s + k];
                    if (lm_aft_out) lm_aft_out[6 * g + k] = mapd[s] ? lm[6 * s + k] : fnan;
                    if (tm_mapped_out) tm_mapped_out[6 * g + k] = pub[s] ? tm[6 * s + k] : fnan;
                }
                for (int k = 0; k < 4; ++k) track_xyzt[4 * g + k] = pub[s] ? tr[4 * s + k] : dnan;
                if (lm_iters_out) lm_iters_out[g] = mapd[s] ? its[s] : -1;
            }
        };
        if (!loam_pipelined() || gmax < 2) {
            for (int t = 0; t < gmax; ++t) {
                sweeps_of_step(t, idx.data());
                rc = P.step(idx.data(), pub.data(), mapd.data(), lo.data(), lm.data(), tm.data(), tr.data(), its.data());
                if (rc) return rc;
                scatter();
            }
        } else {
            // The nodes run concurrently, as the reference's ROS nodes do: laserOdometry (this thread's twin, on
            // its own stream) works up to two sweeps ahead of transformMaintenance + laserMapping (this thread,
            // on the context's stream).  Nothing flows back from mapping to odometry (LO subscribes to
            // scanRegistration's topics only), so the results are those of the lock-step order.
            gpscal_ctx octx = *ctx;  // same device, another stream; errors are copied back
            hipStream_t os = nullptr;
            GPSCAL_HIP(ctx, worker_stream_of(ctx, &os));
            octx.stream = os;
            std::mutex mu;
            std::condition_variable cv;
            int produced = 0, consumed = 0, rc_o = 0;
            bool stop = false;
            std::thread odo([&] {
                (void)hipSetDevice(ctx->device);
                std::vector<int> oidx(ng);
                for (int t = 0; t < gmax; ++t) {
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return stop || consumed >= t - (LoamPipe::NRING - 1); });
                        if (stop) return;
                    }
                    sweeps_of_step(t, oidx.data());
                    const int r = P.step_odo(&octx, oidx.data(), P.slots[t % LoamPipe::NRING]);
                    {
                        std::lock_guard<std::mutex> lk(mu);
                        if (r) {
                            rc_o = r;
                            stop = true;
                        } else
                            produced = t + 1;
                    }
                    cv.notify_all();
                    if (r) return;
                }
            });
            int rc_m = 0;
            for (int t = 0; t < gmax; ++t) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return stop || produced > t; });
                    if (produced <= t) break;  // the odometry half failed
                }
                const LoamPipe::StepSlot &L = P.slots[t % LoamPipe::NRING];
                rc_m = P.step_map(L, lo.data(), lm.data(), tm.data(), tr.data(), its.data());
                if (!rc_m) {
                    sweeps_of_step(t, idx.data());
                    for (int s = 0; s < ng; ++s) {
                        pub[s] = L.published[s];
                        mapd[s] = L.do_map[s];
                    }
                    scatter();
                }
                {
                    std::lock_guard<std::mutex> lk(mu);
                    consumed = t + 1;
                    if (rc_m) stop = true;
                }
                cv.notify_all();
                if (rc_m) break;
            }
            odo.join();
            (void)hipStreamSynchronize(os);  // (the stream and its cached blocks stay with the context)
            if (rc_o) {
                ctx->last_error = octx.last_error;
                return rc_o;
            }
            if (rc_m) return rc_m;
        }
        rc = P.finish();
        if (rc) return rc;
    }
    return GPSCAL_OK;
}

// input_data's replay + segmentation (input_data.cpp:78-124, 266-444) for nbag independent bags and
// both passes: stream 2*b is bag b's long pass, stream 2*b+1 its short pass; all streams advance in
// lock step, each with its own cursor, rewinds and laserOdometry resets.
extern "C" int gpscal_input_data_run(gpscal_ctx *ctx, int nbag, const float *xyz, const int *sweep_off,
                                     const int *bag_sweep_off, const double *stamps, double long_distance,
                                     double short_distance, double overlap_distance, int cap_tracks, int *track_flag,
                                     int *track_bag, int *seg_first, int *seg_last, int *track_off,
                                     double *track_xyzt, int cap_rows, int *ntracks_out, int corner_pool_cap,
                                     int surf_pool_cap)
{
    if (!ctx || nbag < 1 || !xyz || !sweep_off || !bag_sweep_off || !stamps || !track_flag || !track_bag || !seg_first ||
        !seg_last || !track_off || !track_xyzt || !ntracks_out || cap_tracks < 1 || cap_rows < 1)
        return fail(ctx, GPSCAL_EINVAL, "gpscal_input_data_run: bad argument");
    if (!(long_distance > short_distance && short_distance > overlap_distance && overlap_distance > 0))  // ID:235-247
        return fail(ctx, GPSCAL_EINVAL, "gpscal_input_data_run: need long > short > overlap > 0");
    GPSCAL_HIP(ctx, hipSetDevice(ctx->device));
    const int nsw = bag_sweep_off[nbag];
    if (bag_sweep_off[0] != 0 || nsw < 1) return fail(ctx, GPSCAL_EINVAL, "gpscal_input_data_run: bad bag offsets");
    const int nstream = 2 * nbag;
    LoamPipe P;
    int rc = P.init(ctx, nstream, xyz, sweep_off, nsw, stamps, corner_pool_cap, surf_pool_cap);
    if (rc) return rc;
    struct Loc {
        int idx;
        double distance, timestamp;
    };
    struct Track {
        int first, last;
        std::vector<double> rows;
    };
    struct Stream {
        int bag = 0, n = 0, base = 0;  // sweeps of the bag: base .. base + n (message index i <-> sweep base + i - 1)
        double L = 0, ov = 0;
        std::vector<Loc> all;
        Loc pub{0, 0, 0};
        double total = 0;
        bool have_pre = false;
        double pre[3] = {0, 0, 0};
        int next = 1;        // next message index to publish
        int phase = 0;       // 0 = segments, 1 = rest replay, 2 = done
        bool crossed = false;
        Track cur;
        std::vector<Track> tracks;
        int seg_first = 1;
    };
    std::vector<Stream> st(nstream);
    for (int b = 0; b < nbag; ++b)
        for (int pass = 0; pass < 2; ++pass) {
            Stream &S = st[2 * b + pass];
            S.bag = b;
            S.base = bag_sweep_off[b];
            S.n = bag_sweep_off[b + 1] - bag_sweep_off[b];
            S.L = pass == 0 ? long_distance : short_distance;
            S.ov = pass == 0 ? 0.0 : overlap_distance;
            S.all.push_back(S.pub);  // ID:269-273
            S.next = 1;
            S.cur.first = 1;
            S.cur.last = 0;
            if (S.n < 1) S.phase = 2;
        }
    std::vector<int> idx(nstream), pub(nstream), mapd(nstream);
    std::vector<double> tr((size_t)nstream * 4);
    // ends the track being collected (ID:347-352) and decides what the stream does next
    auto close_track = [&](Stream &S, int s, bool end_by_distance) {
        S.tracks.push_back(S.cur);
        S.have_pre = false;
        if (S.phase == 1) {  // the rest replay is over
            S.phase = 2;
            return;
        }
        if (end_by_distance && S.pub.idx < S.n) {  // next segment: everything after pubLocation, fresh LOAM
            P.control_reset(s);
            S.next = S.pub.idx + 1;
            S.cur = Track{S.next, S.next - 1, {}};
            return;
        }
        // the bag is exhausted; is the rest too short?  (ID:366-414)
        if (S.all.size() > 1 && S.total < S.L / 3.0) {
            const Loc tmp = S.all[S.all.size() - 2];
            const int drop = S.tracks.size() >= 2 ? 2 : (int)S.tracks.size();
            S.tracks.resize(S.tracks.size() - drop);
            P.control_reset(s);
            S.phase = 1;
            S.next = tmp.idx + 1;
            S.cur = Track{S.next, S.next - 1, {}};
            if (S.next > S.n) S.phase = 2;
            return;
        }
        S.phase = 2;
    };
    // The nodes run concurrently here too, as far as input_data's feedback allows: a step's cut decisions read its
    // /true_odometry_to_init sample, which transformMaintenance computes from the step's odometry and the PREVIOUS
    // mapping cycle -- so the mapping cycle of step t (a worker thread, the context's stream) overlaps laserOdometry of
    // step t + 1 (this thread, its own stream).  The order of every node's inputs is that of the lock-step replay.
    struct Mapper {
        bool on = false;
        LoamPipe *P = nullptr;
        gpscal_ctx octx;
        hipStream_t os = nullptr;
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        const LoamPipe::StepSlot *job = nullptr;
        bool busy = false, quit = false;
        int rc = 0;
        void run()
        {
            (void)hipSetDevice(P->ctx->device);
            for (;;) {
                const LoamPipe::StepSlot *j;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return quit || job; });
                    if (!job) return;
                    j = job;
                }
                const int r = P->step_mapping(*j, nullptr, nullptr, nullptr);
                {
                    std::lock_guard<std::mutex> lk(mu);
                    if (r && !rc) rc = r;
                    job = nullptr;
                    busy = false;
                }
                cv.notify_all();
            }
        }
        void post(const LoamPipe::StepSlot *j)
        {
            {
                std::lock_guard<std::mutex> lk(mu);
                job = j;
                busy = true;
            }
            cv.notify_all();
        }
        int wait()  // until the worker is idle; its first error, if any
        {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return !busy; });
            return rc;
        }
        ~Mapper()
        {
            if (th.joinable()) {
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !busy; });
                    quit = true;
                }
                cv.notify_all();
                th.join();
            }
            if (os) (void)hipStreamSynchronize(os);  // (the stream and its cached blocks stay with the context)
        }
    } mapper;
    if (loam_pipelined()) {
        mapper.P = &P;
        mapper.octx = *ctx;
        GPSCAL_HIP(ctx, worker_stream_of(ctx, &mapper.os));
        mapper.octx.stream = mapper.os;
        mapper.th = std::thread([&mapper] { mapper.run(); });
        mapper.on = true;
    }
    for (;;) {
        bool any = false;
        for (int s = 0; s < nstream; ++s) {
            Stream &S = st[s];
            idx[s] = -1;
            if (S.phase == 2) continue;
            if (S.next > S.n) {  // ran off the end of the bag (ID:341: the inner loop ends without `end`)
                if (S.phase == 1 && S.cur.rows.empty()) S.phase = 2;  // ID:419: an empty track is not queued
                else close_track(S, s, false);
                if (S.phase == 2 || S.next > S.n) {
                    if (S.phase != 2 && S.next > S.n) S.phase = 2;
                    continue;
                }
            }
            idx[s] = S.base + S.next - 1;
            any = true;
        }
        if (!any) break;
        if (!mapper.on) {
            rc = P.step(idx.data(), pub.data(), mapd.data(), nullptr, nullptr, nullptr, tr.data(), nullptr);
            if (rc) return rc;
        } else {
            // laserOdometry of this step (own stream) while the worker still runs the previous step's mapping cycle;
            // transformMaintenance needs both, and the step's cut decisions need only its track sample
            LoamPipe::StepSlot &L = P.slots[P.step_no % LoamPipe::NRING];
            rc = P.step_odo(&mapper.octx, idx.data(), L);
            int rc_w = mapper.wait();
            if (rc) {
                ctx->last_error = mapper.octx.last_error;
                return rc;
            }
            if (rc_w) return rc_w;
            rc = P.step_tm(L, nullptr, tr.data());
            if (rc) return rc;
            for (int s = 0; s < nstream; ++s) {
                pub[s] = L.published[s];
                mapd[s] = L.do_map[s];
            }
            if (L.any_map) mapper.post(&L);
        }
        for (int s = 0; s < nstream; ++s) {
            Stream &S = st[s];
            if (idx[s] < 0) continue;
            const int i = S.next;
            S.cur.last = i;
            ++S.next;
            if (pub[s]) {  // subOdometryHandler, ID:78-118
                const double *t4 = tr.data() + 4 * s;
                S.cur.rows.insert(S.cur.rows.end(), t4, t4 + 4);
                Loc t;
                t.idx = i;
                t.distance = S.have_pre ? std::sqrt((t4[0] - S.pre[0]) * (t4[0] - S.pre[0]) + (t4[1] - S.pre[1]) * (t4[1] - S.pre[1]) +
                                                    (t4[2] - S.pre[2]) * (t4[2] - S.pre[2])) + S.total
                                        : 0.0;
                t.timestamp = t4[3];
                S.have_pre = true;
                S.pre[0] = t4[0]; S.pre[1] = t4[1]; S.pre[2] = t4[2];
                if (S.phase == 0) {
                    if (t.distance <= S.L - S.ov) S.pub = t;
                    else if (S.all.back().timestamp != S.pub.timestamp) S.all.push_back(S.pub);
                }
                S.total = t.distance;
            }
            if (S.phase == 0 && S.total > S.L) {  // ID:332-339
                S.total = 0;
                close_track(S, s, true);
            }
        }
    }
    if (mapper.on) {
        rc = mapper.wait();
        if (rc) return rc;
    }
    rc = P.finish();
    if (rc) return rc;
    int nt = 0, nrows = 0;
    track_off[0] = 0;
    for (int pass = 0; pass < 2; ++pass)  // input_data publishes every long track before the first short one
        for (int b = 0; b < nbag; ++b)
            for (const Track &T : st[2 * b + pass].tracks) {
                if (nt >= cap_tracks || nrows + (int)T.rows.size() / 4 > cap_rows)
                    return fail(ctx, GPSCAL_ESIZE, "gpscal_input_data_run: output capacity exceeded");
                track_flag[nt] = pass;
                track_bag[nt] = b;
                seg_first[nt] = T.first;
                seg_last[nt] = T.last;
                std::memcpy(track_xyzt + 4 * (size_t)nrows, T.rows.data(), sizeof(double) * T.rows.size());
                nrows += (int)T.rows.size() / 4;
                track_off[++nt] = nrows;
            }
    *ntracks_out = nt;
    return GPSCAL_OK;
}
