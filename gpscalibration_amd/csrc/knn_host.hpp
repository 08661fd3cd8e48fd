// knn_host.hpp -- host side of the grid index shared between translation units.
#pragma once
#include <vector>

#include "common.hpp"
#include "knn_device.hpp"

namespace gpscal {

struct GridSet {
    gpscal_ctx *ctx = nullptr;
    int npairs = 0;
    std::vector<long long> off;  // npairs + 1 point offsets
    std::vector<PairDesc> hpairs;
    DevBuf<PairDesc> pairs;
    DevBuf<float4> pts4;    // caller order
    DevBuf<float4> sorted;  // all (pair, level) blocks
    DevBuf<float4> nbr;     // per original point: its 4 nearest other points (lazy, ICP only)
    DevBuf<float2> pt_r2;   // per original point: (r_a^2, r_b^2) certified radii
    DevBuf<unsigned> cell_start_buf;  // 4 pad + cells + 1 + 4 pad
    unsigned *cell_start = nullptr;
    long long total_cells = 0, total_sorted = 0;
    bool pooled = false;  // per-call grid sets: stream-ordered allocations, no device-wide sync on release
};

// Builds the level ladders and the counting-sorted copies for npairs clouds
// (xyz at `stride` bytes, `off` = npairs+1 point offsets).  max_levels == 1 gives the
// tiled single-level grouping used for source clouds.
int build_grids(gpscal_ctx *ctx, const void *xyz, int stride, const long long *off, int npairs, float cell,
                int max_levels, GridSet &gs);
// Several clouds-of-clouds in one call, with one host wait for all of them (the bounding boxes).
struct GridSource {
    const void *xyz;
    const long long *off;
    int npairs;
    GridSet *gs;
    float cell = 0.f;  // this set's level-0 cell size (0: the call's); < 0: -cell points per footprint cell
};
int build_grids_multi(gpscal_ctx *ctx, int nsrc, const GridSource *src, int stride, float cell, int max_levels);
// Neighbour lists + certified radii (ICP only).
// `side` + `done`: the kernel runs on that stream behind what ctx->stream holds now, `done` is recorded behind it and
// the CALLER makes its stream wait for it (gpscal_scan_batch_create: the source grouping runs beside the kernel).
int ensure_safe_radius(gpscal_ctx *ctx, GridSet &gs, hipStream_t side = nullptr, hipEvent_t done = nullptr);

}  // namespace gpscal

struct gpscal_knn_index {
    gpscal_ctx *ctx;
    gpscal::GridSet gs;
};
