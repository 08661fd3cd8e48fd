// loam_internal.hpp -- device-pointer entry points of the LOAM kernels, shared between the
// C-ABI wrappers (loam.hip, sr.hip) and the segment pipeline (loam_pipeline.hip).  Every
// pointer named d_* is HBM; offsets and descriptors are host arrays.  The functions enqueue
// on ctx->stream; the grid builds inside them synchronise the stream.
#pragma once
#include "common.hpp"

namespace gpscal {

struct SweepDesc {
    long long sharp_off, flat_off, clast_off, slast_off;  // into the float4 arrays
    long long corr_off;                                   // filled by loam_odometry_device
    int nc, ns, mc, ms;
};

struct MapDesc {
    long long cstack_off, sstack_off, cmap_off, smap_off;  // into the float4 arrays
    int nc, ns, mc, ms;
};

// laserOdometry's loop for nsweeps sweeps.  coff / soff: nsweeps+1 CONTIGUOUS point offsets of the
// last-sweep clouds inside d_clast / d_slast (descs[b].clast_off == coff[b]).  ring_cnt_c / ring_cnt_s
// (optional host arrays, nsweeps x 16): points per ring of the last clouds, which then get one grid per
// ring for the adjacent-ring searches.
int loam_odometry_device(gpscal_ctx *ctx, int nsweeps, const SweepDesc *descs, const float4 *d_sharp,
                         const float4 *d_flat, const float4 *d_clast, const float4 *d_slast, const long long *coff,
                         const long long *soff, const float *d_tr_in, float *d_tr_out, int *d_iters, int *d_nsel,
                         const float *d_sum_in, float *d_sum_out, const int *ring_cnt_c = nullptr,
                         const int *ring_cnt_s = nullptr);

// laserMapping's loop; cmoff / smoff as above for the map clouds.
int loam_mapping_device(gpscal_ctx *ctx, int nsweeps, const MapDesc *descs, const float4 *d_cstack,
                        const float4 *d_sstack, const float4 *d_cmap, const float4 *d_smap, const long long *cmoff,
                        const long long *smoff, const float *d_tr_in, float *d_tr_out, int *d_iters, int *d_nsel);

// scanRegistration; xyz_off / lf_off: nsweeps+1 host offsets (input points / less-flat capacity).
// *status receives the kernel's flags (1 = less-flat overflow, 2 = more than POINTSNUM points).
// d_ring_counts (optional, nsweeps x 32): less-sharp [0,16) and less-flat [16,32) points per ring.
int scan_registration_device(gpscal_ctx *ctx, int nsweeps, const int *xyz_off, const int *lf_off, const float *d_xyz,
                             float4 *d_full, float4 *d_sharp, float4 *d_lsharp, float4 *d_flat, float4 *d_lflat,
                             int *d_counts, int *status, int *d_ring_counts = nullptr);

}  // namespace gpscal
