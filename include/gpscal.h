/*
 * gpscal.h -- C ABI of libgpscal_hip.so: the MI355X (gfx950) implementation of
 * gpsCalibration's scan-matching + GPS/SLAM track-alignment hot path.
 *
 * This is the drop-in boundary.  The reference has no
 * FFI seam of its own; the seams replaced are
 *   (i)  the C++ class API of libgpsCalibration.so (CMakeLists.txt:141-147),
 *        called from long_distance_track_process.cpp:58-83 and
 *        short_distance_track_process.cpp:240-244, and
 *   (ii) the pcl::KdTreeFLANN / Eigen calls inside the LOAM nodes.
 * Each entry point cites the reference interface it replaces.  Citations use
 * SURVEY.md's abbreviations, all under
 * /root/reference/src/gpsCalibration/:
 *   TC  = src/gps_calibration/track_calibration.cc   TC.h = include/gpsCalibration/track_calibration.h
 *   WC  = src/gps_calibration/weight_calculation.cc  WC.h = include/gpsCalibration/weight_calculation.h
 *   GP  = src/gps_calibration/gps_process.cc         GP.h = include/gpsCalibration/gps_process.h
 *   CM.h= include/gpsCalibration/common.h
 *   LD  = src/long_distance_track_process/long_distance_track_process.cpp
 *   SD  = src/short_distance_track_process/short_distance_track_process.cpp
 *   LO  = src/lidar_slam/loam/laserOdometry.cpp      LM = src/lidar_slam/loam/laserMapping.cpp
 *   TM  = src/lidar_slam/loam/transformMaintenance.cpp
 *
 * Conventions
 *   - every function returns 0 on success or a negative GPSCAL_E* code; the
 *     library never calls exit() (the reference does: TC:49, GP:489,494,608);
 *   - plain pointers and sizes only; the caller owns every buffer;
 *   - a pointer may address HOST memory or DEVICE (HBM) memory: the library asks
 *     the HIP runtime (hipPointerGetAttributes) and stages host buffers through
 *     the context.  Device pointers are used in place, with no copy;
 *   - all work is issued on the context's HIP stream (created non-blocking: it
 *     does not synchronise with the null stream); calls that hand results to
 *     host memory synchronise that stream before returning, calls whose outputs
 *     are all device pointers return as soon as the work is enqueued;
 *   - DEVICE-pointer arguments carry no ordering of their own.  A device INPUT
 *     must be complete, or ordered before the context's stream, when the call
 *     is made: either the producer has been synchronised, or
 *     gpscal_wait_for_stream(ctx, producer_stream) was called first.  A device
 *     OUTPUT is ready only after gpscal_sync(ctx), or for work enqueued on a
 *     stream that was passed to gpscal_make_stream_wait(ctx, consumer_stream)
 *     after the call, or on the stream gpscal_stream(ctx) returns;
 *   - a context is bound to one GPU and one CALLING thread at a time (the
 *     reference's nodes are single-threaded: LD:128-132, SD:223-231).  Two
 *     entry points run the LOAM nodes concurrently, as the reference's ROS
 *     nodes do: gpscal_loam_run_batched and gpscal_input_data_run start one
 *     worker thread and one extra stream for the duration of the call and
 *     join / destroy them before they return (GPSCAL_LOAM_PIPELINE=0 keeps
 *     everything on the calling thread; the results are identical);
 *   - device memory: per-call temporaries and the state of a scan batch are
 *     blocks of a cache the library keeps per stream (no hipMalloc / hipFree,
 *     and so no device-wide synchronisation, once the sizes have been seen);
 *     at most an eighth of the device memory (8 GiB at least) stays cached
 *     idle per stream, gpscal_destroy returns it.  A scan batch must be
 *     destroyed before its context;
 *   - there is NO CPU fallback: without a usable gfx950 device gpscal_create
 *     fails with GPSCAL_ENODEV and nothing else can be called.
 *
 * Layouts (identical to the reference's structs / messages)
 *   COORDXYZT  = double[4] {x,y,z,t}      CM.h:33-39, msg/IMLocalXYZT.msg
 *   COORDXYZTW = double[5] {x,y,z,t,w}    CM.h:41-48, msg/IMLocalXYZTW.msg
 *   clouds     = float xyz with a caller-given byte stride (12 = packed,
 *                16 = pcl::PointXYZ, 32 = pcl::PointXYZI)
 *   transforms = double[16], 4x4 row-major homogeneous (TC:529-542)
 */
#ifndef GPSCAL_H
#define GPSCAL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPSCAL_OK 0
#define GPSCAL_EINVAL (-1)  /* bad argument                                   */
#define GPSCAL_ENODEV (-2)  /* no gfx950 device / HIP runtime error at create */
#define GPSCAL_EHIP (-3)    /* HIP runtime error (see gpscal_last_error)      */
#define GPSCAL_ENOMEM (-4)  /* allocation failed                              */
#define GPSCAL_ESIZE (-5)   /* track sizes differ (the reference exit(1)s, TC:46-50) */
#define GPSCAL_ERANGE (-6)  /* output capacity too small / input beyond a size limit */
#define GPSCAL_ECOMM (-7)   /* RCCL error                                     */

#define GPSCAL_METHOD_UTM 0      /* "UTM"      (run.sh ctm, GP:498)  */
#define GPSCAL_METHOD_GAUSS 1    /* "Gaussion" (GP:502-505)          */

typedef struct gpscal_ctx gpscal_ctx;
typedef struct gpscal_knn_index gpscal_knn_index;
typedef struct gpscal_scan_batch gpscal_scan_batch;

/* ------------------------------------------------------------- lifecycle */
int gpscal_create(gpscal_ctx **ctx, int device_id, unsigned flags);
int gpscal_destroy(gpscal_ctx *ctx);
int gpscal_sync(gpscal_ctx *ctx);
/* The context's hipStream_t, for callers that bracket work with HIP events. */
void *gpscal_stream(gpscal_ctx *ctx);
/* Stream ordering for device-pointer arguments (hipStream_t passed as void*; NULL = the
 * null stream).  wait_for_stream: work enqueued on the context's stream after this call
 * starts only when everything queued on `producer_stream` so far has finished.
 * make_stream_wait: work enqueued on `consumer_stream` after this call starts only when
 * everything queued on the context's stream so far has finished.  Neither blocks the host. */
int gpscal_wait_for_stream(gpscal_ctx *ctx, void *producer_stream);
int gpscal_make_stream_wait(gpscal_ctx *ctx, void *consumer_stream);
const char *gpscal_strerror(int code);
const char *gpscal_last_error(gpscal_ctx *ctx);
/* "gfx950", CU count, library version; for logs. */
int gpscal_device_info(gpscal_ctx *ctx, char *buf, size_t cap);

/* --------------------------------------------------------------- weights */
/* Replaces WeightCoeCal::ICPWeightCoeCal(slam, w) (WC:4-27, WC.h:14).
 * slam_xyzt: n COORDXYZT; w: n doubles. */
int gpscal_weights_speed(gpscal_ctx *ctx, const double *slam_xyzt, int n,
                         double *w);
/* Replaces WeightCoeCal::ICPWeightCoeCal(slam, w, enuOri, slamRotated)
 * (WC:30-78, WC.h:16). */
int gpscal_weights_irls(gpscal_ctx *ctx, const double *slam_xyzt,
                        const double *enu_xyzt, const double *fit_xyzt, int n,
                        double *w);

/* ------------------------------------------------------- track alignment */
/* Replaces trackCalibration(slam, enu, w) + doICP() + doCalibration(out)
 * (TC:4-37; TC.h:12-23) for ONE segment.  Outputs (each may be NULL):
 *   T               final 4x4 (TC:189), rotated_xyz n x 3 (TC:622),
 *   calibrated_xyzt n COORDXYZT (TC:678-686). */
int gpscal_track_fit(gpscal_ctx *ctx, const double *slam_xyzt,
                     const double *enu_xyzt, const double *w, int n,
                     double *T, double *rotated_xyz, double *calibrated_xyzt);
/* Same for nseg independent segments in one launch.  seg_offsets has nseg+1
 * entries (segment s owns rows [seg_offsets[s], seg_offsets[s+1]) of every
 * array); T is nseg x 16.  This is what SD:234-245 does per queued track. */
int gpscal_track_fit_batched(gpscal_ctx *ctx, const double *slam_xyzt,
                             const double *enu_xyzt, const double *w,
                             const int *seg_offsets, int nseg, double *T,
                             double *rotated_xyz, double *calibrated_xyzt);
/* Replaces the body of longDisTrackPro for one flag-0 track once its ENU GPS is
 * known (LD:58-83): speed weights -> fit -> irls_iters x {IRLS weights ->
 * fit(previous fit -> ENU)}.  w_out: final weights (what LD:83 merges);
 * fit_out (may be NULL): last calibrated track, n COORDXYZT. */
int gpscal_long_segment(gpscal_ctx *ctx, const double *slam_xyzt,
                        const double *enu_xyzt, int n, int irls_iters,
                        double *w_out, double *fit_out);
int gpscal_long_segment_batched(gpscal_ctx *ctx, const double *slam_xyzt,
                                const double *enu_xyzt, const int *seg_offsets,
                                int nseg, int irls_iters, double *w_out,
                                double *fit_out);

/* -------------------------------------------------------------------- geo */
/* Replaces GPSPro::UTMTransform / GaussionTransform (GP:851-908, 953-1007).
 * lat/lon in degrees; band_type 3 or 6 (run.sh gdt); xy: n x {x = northing,
 * y = easting + 500000 + band * 1e7}.  The band number comes from the first
 * fix only, as in the reference (GP:869-877). */
int gpscal_wgs_to_enu(gpscal_ctx *ctx, int method, int band_type,
                      const double *lat, const double *lon, int n, double *xy);
/* Replaces GPSPro::UTMReverseTransform / GaussionReverseTransform
 * (GP:1010-1058, 911-950).  enu_xyztw: n COORDXYZTW; lonlat: n x {longitude,
 * latitude} (GP:1053 order); alt: n (= z). */
int gpscal_enu_to_wgs(gpscal_ctx *ctx, int method, int band_type,
                      const double *enu_xyztw, int n, double *lonlat,
                      double *alt);
/* Replaces the arithmetic half of GPSPro::GPSToENU (GP:476-521) after the text
 * has been parsed and dropouts filled on the host: projection of the ngps
 * fixes (GP:498-505), linear interpolation at the SLAM stamps (GP:59-110) and
 * assembly of {x, y, slam z, slam t} (GP:510-518).  *n_out receives the number
 * of samples produced (stamps after the last fix are dropped, GP:99). */
int gpscal_gps_to_enu(gpscal_ctx *ctx, int method, int band_type,
                      const double *lat, const double *lon,
                      const double *gps_t, int ngps, const double *slam_xyzt,
                      int nslam, double *enu_xyzt, int *n_out);
/* The same for nseg segments in one launch (what nseg GPSToENU calls do): segment s owns
 * fixes [gps_off[s], gps_off[s+1]) -- its own time window of the log, gap-filled on its
 * own -- and stamps [slam_off[s], slam_off[s+1]).  enu rows are written at the stamps'
 * positions; n_out[s] = stamps of segment s that survived (a prefix of the segment). */
int gpscal_gps_to_enu_batched(gpscal_ctx *ctx, int method, int band_type,
                              const double *lat, const double *lon,
                              const double *gps_t, const int *gps_off,
                              const double *slam_xyzt, const int *slam_off,
                              int nseg, double *enu_xyzt, int *n_out);
/* Replaces SaveTrailWithTimeTotxt's height compensation (TM:116-157) for a
 * whole pose chain: in n x {px,py,pz,t} in LOAM axes, out n COORDXYZT. */
int gpscal_height_compensate(gpscal_ctx *ctx, const double *loam_xyzt, int n,
                             double *out_xyzt);

/* ------------------------------------------------------------------ k-NN */
/* Replaces pcl::KdTreeFLANN<PointType>::setInputCloud (LO:538-539,1119-1120;
 * LM:750-751).  Builds the multi-level uniform-grid index over m points.
 * GPSCAL_ERANGE for a cloud of 2^27 points or more, or a set whose clouds hold 2^32 points over all levels (the
 * index addresses a cloud's points by signed 32-bit byte offsets, a set's by 32-bit positions); the same limits hold
 * for every cloud of gpscal_scan_batch_create. */
int gpscal_knn_build(gpscal_ctx *ctx, const float *xyz, int m,
                     int stride_bytes, float cell_size_or_0,
                     gpscal_knn_index **index);
/* Replaces nearestKSearch(point, k, idx, sqd) (LO:603,758 k=1; LM:760,867 k=5),
 * batched: all n queries of an iteration in one call.  Exact (eps = 0), results
 * ascending by (squared distance, index); idx/sqd are n x k; missing
 * neighbours (k > m) are idx -1 / sqd +inf.  1 <= k <= 8. */
int gpscal_knn_search(gpscal_knn_index *index, const float *query_xyz, int n,
                      int stride_bytes, int k, int32_t *idx, float *sqd);
int gpscal_knn_free(gpscal_knn_index *index);

/* ------------------------------------------------------------------- ICP */
/* The generic scan-matching iteration (superset of
 * TC:145-181): p_i = fl32(T) * src_i; exact 1-NN q_i of p_i in the target;
 * centroids weighted by w, 3x3 cross-covariance by w^2 (TC:416-506); 3x3 SVD,
 * R = V U^T with the reflection fix (TC:508-523), t = c_q - R c_p (TC:526);
 * T <- [R|t] * T; mean NN distance reported.  A scan batch holds npairs
 * independent (target, source) pairs resident in HBM: the unit that shards
 * one-per-GPU.
 * tgt_off / src_off: npairs+1 point offsets into the packed xyz arrays;
 * w (may be NULL): one weight per source point. */
int gpscal_scan_batch_create(gpscal_ctx *ctx, int npairs, const float *tgt_xyz,
                             const int64_t *tgt_off, const float *src_xyz,
                             const int64_t *src_off, const double *w,
                             float cell_size_or_0, gpscal_scan_batch **batch);
/* T0: npairs x 16 or NULL (identity). */
int gpscal_scan_batch_set_pose(gpscal_scan_batch *batch, const double *T0);
/* Runs `iters` iterations on every pair.  T_out npairs x 16; mean_err (may be
 * NULL) npairs x iters (pair-major); step_ms (may be NULL) iters floats: when
 * given, every launch of the correspondence kernel is bracketed by HIP events
 * on the context stream and its duration returned (profiling mode, no graph).
 * With step_ms == NULL the iteration chain is replayed from a captured
 * hipGraph. */
int gpscal_scan_batch_icp(gpscal_scan_batch *batch, int iters, double *T_out,
                          double *mean_err, float *step_ms);
/* Correspondences of the last iteration, in the caller's source order. */
int gpscal_scan_batch_correspondences(gpscal_scan_batch *batch, int32_t *idx,
                                      float *sqd);
/* Seconds spent building the target indices and sorting the sources. */
double gpscal_scan_batch_build_seconds(gpscal_scan_batch *batch);
int gpscal_scan_batch_destroy(gpscal_scan_batch *batch);

/* Single-pair conveniences over a prebuilt index (benchmark / LOAM shims). */
int gpscal_icp_iterate(gpscal_ctx *ctx, gpscal_knn_index *index,
                       const float *src_xyz, int n, int stride_bytes,
                       const double *w, const double *T_in, double *T_out,
                       double *mean_err);
int gpscal_icp_run(gpscal_ctx *ctx, gpscal_knn_index *index,
                   const float *src_xyz, int n, int stride_bytes,
                   const double *w, int iters, const double *T0, double *T_out,
                   double *mean_err_hist);

/* ------------------------------------------------------- LOAM odometry */
/* Replaces the per-sweep Gauss-Newton loop of laserOdometry (LO:585-1029) for nsweeps
 * INDEPENDENT sweeps (one per SLAM segment; LOAM is reset per segment, LO:519-563) in one
 * launch: TransformToStart (LO:123-150), kd-tree k=1 + adjacent-ring correspondence search
 * every 5th iteration (LO:592-677, 752-844), point-to-line / point-to-plane terms
 * (LO:680-746, 847-901), 6x6 normal equations + QR solve (LO:909-975), degeneracy
 * projection at iteration 0 (LO:977-1004), update and the 0.1 deg / 0.1 cm stop (LO:1005-1028).
 * Points are float[4] {x, y, z, intensity}, intensity = ring id + 0.1 * relative time
 * (scanRegistration.cpp:340-362), clouds ordered by ring as scanRegistration emits them;
 * *_off are nsweeps+1 point offsets.  sharp / flat = cornerPointsSharp / surfPointsFlat of the
 * current sweep, corner_last / surf_last = laserCloudCornerLast / laserCloudSurfLast.
 * transform_* are LOAM's float[6] {rx, ry, rz, tx, ty, tz} per sweep.  Optional:
 * iters_out, nsel_out (rows of the last linear system), transform_sum_in/out = pose
 * accumulation (LO:1035-1064).  The IMU terms are zero (nothing publishes /imu/data under
 * run.sh, input_data.cpp:259-262). */
int gpscal_loam_odometry_batched(gpscal_ctx *ctx, int nsweeps,
                                 const float *sharp_xyzi, const int *sharp_off,
                                 const float *flat_xyzi, const int *flat_off,
                                 const float *corner_last_xyzi, const int *corner_last_off,
                                 const float *surf_last_xyzi, const int *surf_last_off,
                                 const float *transform_in, float *transform_out,
                                 int *iters_out, int *nsel_out,
                                 const float *transform_sum_in, float *transform_sum_out);
/* Replaces the sweep-to-map optimisation loop of laserMapping (LM:748-1018) for nsweeps
 * INDEPENDENT sweeps (one per SLAM segment) in one launch: pointAssociateToMap (LM:244-262),
 * k=5 searches in the local corner / surface maps (kdtree*FromMap, LM:749-750,760,867), the
 * covariance-eigenvector line test (LM:763-857), the 5-point plane fit (LM:866-919), the 6x6
 * normal equations + QR solve (LM:922-968), the degeneracy projection with threshold 100 at
 * iteration 0 (LM:970-997), the update and the 0.05 deg / 0.05 cm stop (LM:999-1017); at most
 * 10 iterations, skipped when the maps hold <= 10 corner or <= 100 surface points (LM:748).
 * *_stack = laserCloudCornerStack / laserCloudSurfStack (the down-sampled features of the
 * sweep), *_map = laserCloudCornerFromMap / laserCloudSurfFromMap.  transform_in / _out are
 * transformTobeMapped before / after the loop (float[6] per sweep). */
int gpscal_loam_mapping_batched(gpscal_ctx *ctx, int nsweeps,
                                const float *corner_stack_xyzi, const int *corner_stack_off,
                                const float *surf_stack_xyzi, const int *surf_stack_off,
                                const float *corner_map_xyzi, const int *corner_map_off,
                                const float *surf_map_xyzi, const int *surf_map_off,
                                const float *transform_in, float *transform_out,
                                int *iters_out, int *nsel_out);
/* TransformToStart (to_end = 0, LO:123-150) / TransformToEnd (to_end = 1, LO:156-227)
 * of n points with one transform. */
int gpscal_loam_transform(gpscal_ctx *ctx, const float *transform6,
                          const float *pts_xyzi, int n, float *out_xyzi, int to_end);

/* GPSPro::GPSToGCJ / GCJToBD / BDToGCJ (GP:526-595; transform2Mars, bd_encrypt, bd_decrypt
 * GP:1127-1207): WGS-84 -> GCJ-02 (Gaode), GCJ-02 <-> BD-09 (Baidu) on n {longitude, latitude}
 * pairs as gpscal_enu_to_wgs emits them; the map outputs of result_control 2 / 3
 * (short_distance_track_process.cpp:271-291). */
int gpscal_gps_to_gcj(gpscal_ctx *ctx, const double *lonlat, int n, double *gcj_lonlat);
int gpscal_gcj_to_bd(gpscal_ctx *ctx, const double *gcj_lonlat, int n, double *bd_lonlat);
int gpscal_bd_to_gcj(gpscal_ctx *ctx, const double *bd_lonlat, int n, double *gcj_lonlat);
/* The payload of /imorpheus_gps (result_control 4; short_distance_track_process.cpp:295-309): for each of the
 * n calibrated points (COORDXYZTW, the merged short-pass result) one gpsCalibration/IMGPS record
 * {b = latitude, l = longitude, w = merged weight} (msg/IMGPS.msg, msg/IMMessage.msg: float64 b, l, w).
 * blw: n x 3 doubles.  The ROS side only has to copy them into IMMessage.track and publish. */
int gpscal_imgps_message(gpscal_ctx *ctx, int method, int band_type, const double *calibrated_xyztw, int n,
                         double *blw);

/* ------------------------------------------------ scanRegistration, VoxelGrid */
/* Replaces scanRegistration's laserCloudHandler (SR:238-674, IMU inactive) for nsweeps raw
 * sweeps in one launch: NaN removal, start / end orientation (SR:262-281), ring id from the
 * vertical angle table and relative time -> intensity (SR:284-363), ring concatenation
 * (SR:444-447), 11-tap curvature and ring spans (SR:455-490), occluded / parallel-beam
 * rejection (SR:492-548), the per-sector curvature sort and greedy picking of <= 16 sharp,
 * <= 20 less-sharp, <= 32 flat points with +-5 neighbour suppression (SR:558-657), and
 * VoxelGrid(0.2) of each ring's remaining points (SR:659-673).
 * xyz = packed float[3] points in the sensor frame, firing order; xyz_off = nsweeps+1 point
 * offsets.  Outputs are float[4] {x,y,z,intensity} in LOAM axes: full (laserCloud, sweep b at
 * xyz_off[b]), sharp (1536 slots per sweep), less_sharp (1920), flat (3072), less_flat (sweep
 * b at less_flat_off[b], capacity less_flat_off[b+1]-less_flat_off[b]; NULL = xyz_off).
 * counts = nsweeps x 5 {full, sharp, less_sharp, flat, less_flat}.  GPSCAL_ESIZE if a sweep
 * keeps more than 60000 points (POINTSNUM, common.h:15), GPSCAL_ERANGE on less_flat overflow. */
int gpscal_scan_registration_batched(gpscal_ctx *ctx, int nsweeps,
                                     const float *xyz, const int *xyz_off,
                                     float *full_xyzi, float *sharp_xyzi,
                                     float *less_sharp_xyzi, float *flat_xyzi,
                                     float *less_flat_xyzi, const int *less_flat_off,
                                     int *counts);
/* pcl::VoxelGrid<PointXYZI> with a cubic leaf (SR:667-673; LM:1044-1058 downSizeFilterCorner /
 * Surf / Map) over nclouds clouds: centroid of x, y, z, intensity per occupied cell, output
 * ordered by cell id, written at the cloud's own offset; counts[c] = points kept. */
int gpscal_voxel_grid_batched(gpscal_ctx *ctx, int nclouds, const float *pts_xyzi,
                              const int *off, float leaf, float *out_xyzi, int *counts);

/* ------------------------------------------------------ the LOAM node chain */
/* Replaces the four LOAM nodes of launch/loam_velodyne.launch -- scanRegistration,
 * laserOdometry, laserMapping, transformMaintenance -- for nseg pre-cut SLAM segments run in
 * lock step on the device: per sweep, scanRegistration's features (SR:238-674), laserOdometry's
 * state machine and loop (LO:495-1124), every second sweep laserMapping's cycle (LM:420-1148:
 * transformAssociateToMap, the 21x11x21 cube ring with its shifts, the field-of-view cube list,
 * the stack / cube voxel filters, the optimisation loop, transformUpdate), and
 * transformMaintenance's integration and height compensation (TM:113-157, 178-337) that
 * produces the /true_odometry_to_init samples input_data collects into the SLAM track
 * (ID:266-444).  Schedule: each node finishes a sweep before the next one arrives (the
 * reference's bag playback rate guarantees this); no IMU.  laserOdometry runs up to three sweeps
 * ahead of transformMaintenance + laserMapping (own thread and stream): nothing flows back from
 * mapping to odometry, so every node sees its inputs in that schedule's order.
 * xyz = packed float[3] raw points, sweep_off = nsweeps+1 point offsets over ALL sweeps,
 * seg_sweep_off = nseg+1 sweep-index offsets (seg_sweep_off[0] = 0), stamps = one per sweep.
 * Outputs, one row per sweep (optional unless noted): lo_sum = laserOdometry's transformSum,
 * lm_aft = transformAftMapped (NaN where laserMapping did not run), tm_mapped =
 * transformMaintenance's pose, track_xyzt (required) = {x, y, HEIGHT, stamp} (NaN for the first
 * sweep of a segment, which publishes no odometry, LO:519-562), lm_iters = mapping iterations
 * (-1 where it did not run).  *_pool_cap = map points kept per segment (0 = 262144 / 1048576);
 * GPSCAL_ENOMEM when a map outgrows them. */
int gpscal_loam_run_batched(gpscal_ctx *ctx, int nseg, const float *xyz,
                            const int *sweep_off, const int *seg_sweep_off,
                            const double *stamps, float *lo_sum, float *lm_aft,
                            float *tm_mapped, double *track_xyzt, int *lm_iters,
                            int corner_pool_cap, int surf_pool_cap);

/* Replaces input_data_node's replay + segmentation (input_data.cpp:78-124, 266-444) together with
 * the LOAM nodes it drives, for nbag independent bags: per bag a long pass (segments of
 * long_distance metres, no overlap) and a short pass (short_distance, restarting overlap_distance
 * before the cut), each cut online from the /true_odometry_to_init track exactly as
 * subOdometryHandler does (distance from the previous sample, pubLocation = last sample within
 * slam - overlap, laserOdometry reset through /control_command at every cut, ID:283-286,342-346),
 * including the final rule that replays from the start of the previous segment when the rest is
 * shorter than a third of the segment length (ID:366-414).  All 2 x nbag streams advance in lock
 * step on the device; within a step, the mapping cycle overlaps the NEXT step's laserOdometry (a cut
 * decision needs transformMaintenance's sample of its own step only, which depends on the previous
 * mapping cycle).  Inputs as gpscal_loam_run_batched, with bag_sweep_off = nbag+1 sweep-index
 * offsets.  Outputs (host arrays): per track its flag (0 long / 1 short, the IMTrack.track_flag of
 * ID:347), bag, first / last replayed message (1-based within the bag), and rows
 * track_off[k]..track_off[k+1] of track_xyzt = the IMLocalXYZT samples {x, y, z, t} (ID:80-88).
 * Long tracks of all bags come first, then the short ones. */
int gpscal_input_data_run(gpscal_ctx *ctx, int nbag, const float *xyz, const int *sweep_off,
                          const int *bag_sweep_off, const double *stamps,
                          double long_distance, double short_distance, double overlap_distance,
                          int cap_tracks, int *track_flag, int *track_bag, int *seg_first,
                          int *seg_last, int *track_off, double *track_xyzt, int cap_rows,
                          int *ntracks_out, int corner_pool_cap, int surf_pool_cap);

/* ------------------------------------------------------------- multi-GPU */
/* New (no reference counterpart): the one exchange of the sharded pipeline,
 * an RCCL all-gather of per-segment pose chains / fit results over xGMI
 * (SURVEY section 8e).  One process per GPU: rank 0 obtains an id with
 * gpscal_comm_unique_id and hands it to the other ranks out of band. */
#define GPSCAL_COMM_ID_BYTES 128
int gpscal_comm_unique_id(void *id_bytes);
int gpscal_comm_init(gpscal_ctx *ctx, const void *id_bytes, int rank,
                     int world);
/* local: count doubles on this rank; counts: world ints (doubles per rank);
 * all: sum(counts) doubles, rank-major.  Pointers host or device; with device
 * pointers for both the call returns after enqueue on the context's stream. */
int gpscal_allgather_chains(gpscal_ctx *ctx, const double *local,
                            const int *counts, double *all);
int gpscal_comm_destroy(gpscal_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* GPSCAL_H */
