// common.h -- host-side structs and constants of the gpsCalibration interface,
// mirroring include/gpsCalibration/common.h:15-48 of the reference so that callers
// written against it compile unchanged (ROS message converters excluded: no ROS here).
#ifndef GPSCAL_HOST_COMMON_H
#define GPSCAL_HOST_COMMON_H

#include <string>
#include <utility>
#include <vector>

#include "../../include/gpscal.h"

#define POINTSNUM 60000
#define HEIGHT 10
#define IMLDLEN 512
#define IMSDLEN 512
#define IMDP 15
#define IMTHREEBANDS 3
#define IMSIXBANDS 6

typedef struct {
    double x;
    double y;
    double z;
    double t;
} COORDXYZT;

typedef struct {
    double x;
    double y;
    double z;
    double t;
    double w;
} COORDXYZTW;

// gpsCalibration/IMGPS (msg/IMGPS.msg): one element of IMMessage.track, the /imorpheus_gps payload
typedef struct {
    double b;  // latitude
    double l;  // longitude
    double w;  // confidence (merged weight)
} IMGPS;

static_assert(sizeof(COORDXYZT) == 32 && sizeof(COORDXYZTW) == 40 && sizeof(IMGPS) == 24, "layouts are part of the C ABI");

namespace gpscal_host {
// The reference's classes take no context argument; they share one process-wide
// context (device GPSCAL_DEVICE, default 0), created on first use.  Throws
// std::runtime_error when no gfx950 device is usable: there is no CPU fallback.
gpscal_ctx *default_ctx();
void check(int rc, const char *what);
}  // namespace gpscal_host

#endif
