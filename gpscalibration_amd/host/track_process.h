// track_process.h -- ROS-free bodies of the two track-processing nodes, same control
// flow as long_distance_track_process.cpp:41-88 and short_distance_track_process.cpp:
// 39-158,223-247.  Segments arrive as vectors instead of IMTrack messages.
#ifndef GPSCAL_HOST_TRACK_PROCESS_H
#define GPSCAL_HOST_TRACK_PROCESS_H
#include "common.h"
#include "gps_process.h"

#define MAXITERATOR 5

class LongDistanceTrackProcess {
public:
    explicit LongDistanceTrackProcess(GPSPro &gps) : gps_(gps) {}
    // one flag-0 track (long_distance_track_process.cpp:57-83)
    void process(const std::vector<COORDXYZT> &slamTrack);
    // all queued flag-0 tracks in ONE batched launch (what the GPU is for)
    void processBatch(const std::vector<std::vector<COORDXYZT> > &slamTracks);
    // what the end marker publishes on gps_weight (long_distance_track_process.cpp:45-53)
    const std::vector<COORDXYZTW> &totalTrack() const { return total_; }

private:
    GPSPro &gps_;
    std::vector<COORDXYZTW> total_;
};

class ShortDistanceTrackProcess {
public:
    // gps = the whole-run ENU GPS + weights received on gps_weight
    void setGPS(const std::vector<COORDXYZTW> &gps);
    void process(const std::vector<COORDXYZT> &slamTrack);  // short_distance_track_process.cpp:236-244
    void processBatch(const std::vector<std::vector<COORDXYZT> > &slamTracks);
    const std::vector<COORDXYZTW> &result() const { return out_; }
    const std::vector<COORDXYZTW> &gps() const { return gps_; }
    // exposed for tests
    static void getGPS(const std::vector<COORDXYZTW> &gps, const std::vector<COORDXYZT> &slamTrack,
                       std::vector<COORDXYZT> &slamWithGPS, std::vector<COORDXYZT> &GPSWithSlam,
                       std::vector<double> &weight, bool gpsTimeOrdered = false);
    void merge(const std::vector<COORDXYZT> &slamTrack, const std::vector<double> &weight);

private:
    std::vector<COORDXYZTW> gps_, out_;
    bool gps_sorted_ = false;  // gps_ is in time order (lets getGPS start at the segment's first stamp)
    bool sorted_ = true;  // out_ is in time order (lets merge() skip the untouched head)
};
#endif
