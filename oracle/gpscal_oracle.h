/*
 * gpscal_oracle.h -- CPU restatement of the gpsCalibration hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load liboracle.so.  The product path (include/gpscal.h, libgpscal_hip.so)
 * never links, calls or falls back to anything declared here.
 *
 * PARITY STATUS: "parity unpinned" in the strict sense -- the reference ships
 * no tests, golden vectors or result files, and none of
 * its translation units compiles in this image without stand-in headers
 * (catkin-generated gpsCalibration/IMTrack.h, pcl/point_types.h, Eigen), so
 * there is no oracle/_ref build.  What pins this restatement instead:
 *   - line-by-line citations of the reference source below (path
 *     abbreviations as in SURVEY.md: TC, WC, GP, LD, SD, TM);
 *   - the one data fixture the reference ships (data/original_gps_data.txt,
 *     copied to tests/golden/ as an INPUT fixture);
 *   - the known-answer values recorded in SURVEY.md section 8(c) from a probe
 *     of the reference's own gps_process.cc (ENU / WGS84 / KML line);
 *   - independent mathematics: numpy.linalg.svd for the 3x3 SVD, a Krueger
 *     series for the transverse-Mercator forward/inverse, brute force for the
 *     kd-tree.
 *
 * Layouts: COORDXYZT  = double[4] {x,y,z,t}   (CM.h:33-39)
 *          COORDXYZTW = double[5] {x,y,z,t,w} (CM.h:41-48)
 *          point clouds = float[3] xyz, tightly packed.
 */
#ifndef GPSCAL_ORACLE_H
#define GPSCAL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- weights */
/* WC:4-27.  w[0]=1; w[i]=min(||p[i+1]-p[i]||/2.2, 1).  The reference reads
 * p[n] one past the end at i=n-1; the restatement defines that slot as the
 * zero-filled spare capacity of the vector, i.e. p[n]=(0,0). */
int orc_weights_speed(const double *slam_xyzt, int n, double *w);
/* WC:30-78.  speed weight times 1/max(0.01, ||enu-fit||). */
int orc_weights_irls(const double *slam_xyzt, const double *enu_xyzt,
                     const double *fit_xyzt, int n, double *w);

/* ------------------------------------------------------------- 3x3 SVD   */
/* Two-sided Jacobi SVD of a row-major 3x3, A = U diag(S) V^T, S descending,
 * S>=0 (semantics of Eigen::JacobiSVD as used at TC:508-511). */
int orc_svd3(const double A[9], double U[9], double S[3], double V[9]);
/* R = V U^T with the reference's reflection fix: if det R < 0 negate column 2
 * of V and recompute (TC:513-523). */
void orc_kabsch_from_H(const double H[9], double R[9]);

/* --------------------------------------------------------- track alignment */
/* TC:366-545 on N x 4 row-major homogeneous rows. T is 4x4 row-major. */
int orc_bft_weighted(const double *A4, const double *B4, const double *w,
                     int n, double T[16]);
/* trackCalibration ctor + doICP + doCalibration (TC:4-37,40-94,97-201,
 * 555-588,591-625,631-689).  Outputs: T (final 4x4), rotated (n x 3),
 * calibrated (n x 4 COORDXYZT), *iters = loop passes executed (1 or 2).
 * quadratic!=0 runs the calibration exactly as coded (O(n^2) double loop);
 * quadratic==0 uses the algebraically equal O(n) form (for big sweeps). */
int orc_track_fit(const double *slam_xyzt, const double *enu_xyzt,
                  const double *w, int n, double T[16], double *rotated_xyz,
                  double *calibrated_xyzt, int *iters, int quadratic);
/* LD:57-83: speed weights -> fit -> 5 x {IRLS weights -> fit(prev fit)}.
 * w_out = final weights (n); fit_out (n x 4, may be NULL) = last fit. */
int orc_long_segment(const double *slam_xyzt, const double *enu_xyzt, int n,
                     int irls_iters, double *w_out, double *fit_out,
                     int quadratic);

/* ------------------------------------------------------------------- geo  */
/* GP:161-229.  Parses a whole GPRMC log held in memory (text, len bytes),
 * keeping fixes with (long)t in [(long)(t0-1), (long)(t1+1)].  Returns the
 * number of fixes written (<= cap) or <0.  lat/lon use the (90,180) sentinel
 * for status 'V' (GP:169,176-179). */
int orc_parse_gprmc(const char *text, size_t len, double t0, double t1,
                    double *lat, double *lon, double *t, int cap);
/* GP:113-159: the sentence named by the first line's second field selects the parser -- $GPRMC
 * (above), $GPGGA (GP:231-299), $GPGLL (GP:300-372).  Returns fixes written, 0 for an unsupported
 * sentence, <0 on error. */
int orc_parse_gps_log(const char *text, size_t len, double t0, double t1, double *lat, double *lon, double *t,
                      int cap);
/* GP:526-595, 1127-1207 on n {longitude, latitude} pairs. */
void orc_gps_to_gcj(const double *lonlat, int n, double *out);
void orc_gcj_to_bd(const double *lonlat, int n, double *out);
void orc_bd_to_gcj(const double *lonlat, int n, double *out);
/* createJSON, GP:1210-1250 (returns bytes needed, like snprintf). */
long orc_json(char *buf, size_t cap, const double *lonlat, int n, int flag, const int *seg_end, const uint32_t *rgb,
              int nseg);
/* GP:389-473.  In-place dropout fill; returns the reference's return value. */
int orc_gap_fill(double *lat, double *lon, const double *t, int n);
/* GP:851-908 (method 0, "UTM") / GP:953-1007 (method 1, "Gaussion").
 * band_type 3 or 6.  out = n x {x(northing), y(easting + 5e5 + band*1e7)}. */
int orc_wgs_to_local(int method, int band_type, const double *lat,
                     const double *lon, int n, double *xy);
/* GP:1010-1058 / GP:911-950.  in = n x COORDXYZTW; out lonlat = n x 2
 * {longitude, latitude} (GP:1053), alt = z. */
int orc_local_to_wgs(int method, int band_type, const double *enu_xyztw,
                     int n, double *lonlat, double *alt);
/* GP:59-110.  Returns number of interpolated samples written. */
int orc_interpolate(const double *xy, const double *gps_t, int ngps,
                    const double *slam_t, int nslam, double *out_xy);
/* GP:476-521 chain on pre-parsed fixes: gap fill -> project -> interpolate ->
 * {x,y,slam z,slam t}.  Returns samples written (may be < nslam, GP:85-107). */
int orc_gps_to_enu(int method, int band_type, double *lat, double *lon,
                   const double *gps_t, int ngps, const double *slam_xyzt,
                   int nslam, double *enu_xyzt);
/* GP:600-626 + GP:692-756.  seg_end[k] = index, rgb[k] = 0xRRGGBB.
 * Returns the number of colour segments. */
int orc_colour_segments(const double *enu_xyztw, int n, int *seg_end,
                        uint32_t *rgb, int cap);
/* GP:759-847.  Writes the KML text into buf (returns bytes needed, like
 * snprintf).  flag 0 = original track, 1 = calibrated (per-colour placemarks,
 * including the reference's quirks: last point never written, GP:832). */
long orc_kml(char *buf, size_t cap, const double *lonlat, const double *alt,
             int n, int flag, const int *seg_end, const uint32_t *rgb,
             int nseg);

/* SD:39-70 two-pointer time match.  Returns matched count. */
int orc_match_gps(const double *gps_xyztw, int ngps, const double *slam_xyzt,
                  int nslam, double *slam_out_xyzt, double *gps_out_xyzt,
                  double *w_out);
/* SD:73-158 overlap cross-fade.  acc (capacity cap rows of COORDXYZTW) holds
 * *nacc rows on entry; returns 0 and updates *nacc. */
int orc_merge_short(double *acc_xyztw, int *nacc, int cap,
                    const double *seg_xyzt, const double *seg_w, int nseg);

/* TM:116-157 height compensation of a pose chain (LOAM axes z,x,y -> x,y,z).
 * in: n x {px,py,pz,t} LOAM frame; out n x COORDXYZT with z = HEIGHT(10). */
int orc_height_compensate(const double *loam_xyzt, int n, double *out_xyzt);

/* --------------------------------------------------------------- k-NN/ICP */
/* Squared distance exactly as every implementation must compute it:
 * fmaf(dz,dz, fmaf(dy,dy, dx*dx)) in float32.  Ordering: (d2, index). */
float orc_sqdist(const float *a, const float *b);
/* Exact brute-force k-NN (pcl::KdTreeFLANN::nearestKSearch semantics,
 * LO:603,758; LM:760,867; ties -> lower index).  idx/sqd are n x k. */
int orc_knn_brute(const float *tgt, int m, const float *q, int n, int k,
                  int32_t *idx, float *sqd);
/* Exact kd-tree (CPU baseline; same results as brute force). */
typedef struct orc_kdtree orc_kdtree;
orc_kdtree *orc_kdtree_build(const float *tgt, int m);
void orc_kdtree_free(orc_kdtree *t);
int orc_kdtree_search(const orc_kdtree *t, const float *q, int n, int k,
                      int32_t *idx, float *sqd);
/* One generic ICP iteration: p_i = fl32(T_in) * src_i, exact
 * 1-NN into the tree, centroids weighted by w, covariance by w^2 (the
 * BFTWithWeight convention, TC:416-506), Kabsch with reflection fix,
 * T_out = dT * T_in.  w may be NULL (all ones).  mean_err = mean NN distance
 * (sqrt of sqd) before the update.  idx/sqd (n) may be NULL. */
int orc_icp_iterate(const orc_kdtree *t, const float *src, int n,
                    const double *w, const double T_in[16], double T_out[16],
                    double *mean_err, int32_t *idx, float *sqd);
int orc_icp_run(const orc_kdtree *t, const float *src, int n, const double *w,
                int iters, const double T0[16], double T_out[16],
                double *mean_err_hist);
/* Apply the float32 rounding of T to a cloud: the exact arithmetic the GPU
 * kernel and the oracle share (fmaf chains). */
void orc_transform_f32(const double T[16], const float *src, int n,
                       float *dst);

/* ----------------------------------------------------- LOAM odometry (LO) */
/* Points are float[4] {x,y,z,intensity}; transforms are LOAM's float[6]
 * {rx,ry,rz,tx,ty,tz}.  laserOdometry.cpp:123-150 / 156-227 (IMU terms zero). */
void orc_lo_transform_to_start(const float tr[6], const float *pi, float *po);
void orc_lo_transform_to_end(const float tr[6], const float *pi, float *po);
/* The Gauss-Newton loop of one sweep (laserOdometry.cpp:585-1029). */
int orc_lo_match(const float *sharp, int nc, const float *flat, int ns, const float *cornerLast, int mc,
                 const float *surfLast, int ms, const float tr_in[6], float tr_out[6], int *iters_out,
                 int *nsel_out);
/* laserMapping's sweep-to-map optimisation loop (laserMapping.cpp:244-262, 748-1018):
 * k=5 search, covariance-eigen line test / 5-point plane fit, 6x6 Gauss-Newton. */
int orc_lm_match(const float *cornerStack, int nc, const float *surfStack, int ns, const float *cornerMap, int mc,
                 const float *surfMap, int ms, const float tr_in[6], float tr_out[6], int *iters_out, int *nsel_out);
/* Pose accumulation (laserOdometry.cpp:1035-1064, IMU terms zero). */
void orc_lo_accumulate(const float sum_in[6], const float tr[6], float sum_out[6]);

/* ------------------------------------------- scanRegistration (SR) + VoxelGrid */
/* scanRegistration.cpp:238-674 for one raw sweep (n_in x float[3], sensor axes, NaNs
 * allowed): ring / time tagging, ring concatenation, curvature, occlusion rejection,
 * per-sector picking, VoxelGrid(0.2) of the less-flat points per ring.  Outputs are
 * float[4] {x,y,z,intensity} in LOAM axes; every buffer needs room for 4*n_in points
 * (quirk: a missing ring makes the next ring index re-scan earlier rings). */
int orc_sr_extract(const float *xyz, int n_in, float *full, int *n_full, float *sharp, int *n_sharp,
                   float *less_sharp, int *n_less_sharp, float *flat, int *n_flat, float *less_flat,
                   int *n_less_flat);
/* pcl::VoxelGrid<PointXYZI>::filter with setLeafSize(leaf,leaf,leaf) (PCL 1.8.0
 * voxel_grid.hpp applyFilter; SR:667-673, LM:1044-1058).  Output ordered by cell id;
 * points of a cell are summed in input order.  Returns 1 when PCL would give up
 * (more than INT_MAX cells) and copy the input. */
int orc_voxel_grid(const float *pts_xyzi, int n, float leaf, float *out_xyzi, int *n_out);

/* ------------------------------------------------------- the four LOAM nodes */
/* transformAssociateToMap (laserMapping.cpp:116-203 == transformMaintenance.cpp:178-265). */
void orc_assoc_to_map(const float sum[6], const float bef[6], const float aft[6], float out[6]);
/* scanRegistration -> laserOdometry -> laserMapping -> transformMaintenance in lock step
 * over the raw sweeps of one segment (see pipeline_oracle.c for the schedule).  Per sweep t:
 * lo_sum = laserOdometry's transformSum, lm_aft = transformAftMapped (NaN when laserMapping
 * did not run), tm_mapped = transformMaintenance's integrated pose, track = the
 * /true_odometry_to_init sample {x, y, HEIGHT, stamp} (NaN for sweep 0), lm_iters = mapping
 * iterations (-1 when it did not run).  Arrays are nsweeps x 6 / x 4 / x 1. */
int orc_loam_run(const float *xyz, const int *sweep_off, int nsweeps, const double *stamps, float *lo_sum,
                 float *lm_aft, float *tm_mapped, double *track, int *lm_iters);

/* input_data's replay + segmentation state machine (input_data.cpp:78-124, 266-444) around the
 * node chain, one bag, one pass (long pass: overlap 0; short pass: overlap > 0).  Track k replays
 * messages seg_first[k]..seg_last[k] (1-based) and holds rows track_off[k]..track_off[k+1] of
 * track_xyzt ({x, y, z, t} as collected at ID:80-88).  Returns the number of tracks or -1. */
int orc_input_data_pass(const float *xyz, const int *sweep_off, int nsweeps, const double *stamps, double slam_distance,
                        double overlap, int cap_tracks, int *seg_first, int *seg_last, int *track_off,
                        double *track_xyzt, int cap_rows);

#ifdef __cplusplus
}
#endif
#endif
