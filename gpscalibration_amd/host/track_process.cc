// track_process.cc -- ROS-free long / short track processing (reference:
// long_distance_track_process.cpp:21-88, short_distance_track_process.cpp:39-158,234-245).
#include "track_process.h"

#include <algorithm>
#include <cmath>
#include <stdexcept>

#include "track_calibration.h"

using gpscal_host::check;
using gpscal_host::default_ctx;

void LongDistanceTrackProcess::process(const std::vector<COORDXYZT> &slamTrack)
{
    std::vector<std::vector<COORDXYZT> > one(1, slamTrack);
    processBatch(one);
}

void LongDistanceTrackProcess::processBatch(const std::vector<std::vector<COORDXYZT> > &tracks)
{
    // ENU GPS per segment (host parse + device projection/interpolation), then the whole
    // speed-weight -> fit -> 5 x IRLS chain of every segment in one launch.
    std::vector<COORDXYZT> slam, enu;
    std::vector<int> off(1, 0);
    const std::vector<std::vector<COORDXYZT> > enus = gps_.GPSToENUBatch(tracks);
    for (size_t k = 0; k < tracks.size(); ++k) {
        const auto &trk = tracks[k];
        if (trk.empty()) continue;
        const std::vector<COORDXYZT> &e = enus[k];
        if (e.empty()) throw std::runtime_error("WARN: cannot find GPS information corresponding to slam track time");
        // interPolate drops stamps after the last fix (gps_process.cc:99): keep the matched prefix
        slam.insert(slam.end(), trk.begin(), trk.begin() + e.size());
        enu.insert(enu.end(), e.begin(), e.end());
        off.push_back((int)slam.size());
    }
    const int nseg = (int)off.size() - 1;
    if (nseg == 0) return;
    std::vector<double> w(slam.size());
    check(gpscal_long_segment_batched(default_ctx(), &slam[0].x, &enu[0].x, off.data(), nseg, MAXITERATOR, w.data(),
                                      nullptr),
          "gpscal_long_segment_batched");
    // merge(localCoor, weightCoe): the ENU GPS itself with the final weights (LD:83)
    for (size_t i = 0; i < enu.size(); ++i) {
        COORDXYZTW p = {enu[i].x, enu[i].y, enu[i].z, enu[i].t, w[i]};
        total_.push_back(p);
    }
}

void ShortDistanceTrackProcess::setGPS(const std::vector<COORDXYZTW> &gps)
{
    gps_ = gps;
    gps_sorted_ = std::is_sorted(gps_.begin(), gps_.end(), [](const COORDXYZTW &a, const COORDXYZTW &b) { return a.t < b.t; });
}

void ShortDistanceTrackProcess::getGPS(const std::vector<COORDXYZTW> &gps, const std::vector<COORDXYZT> &slamTrack,
                                       std::vector<COORDXYZT> &slamWithGPS, std::vector<COORDXYZT> &GPSWithSlam,
                                       std::vector<double> &weight, bool gpsTimeOrdered)
{
    // two-pointer match on |dt| < 1e-6 (short_distance_track_process.cpp:39-70).  The
    // reference walks the whole-run track from its first sample for every segment; on a
    // time-ordered track (long segments are appended in time order, LD:28-37) nothing
    // before the segment's first stamp can match, so the walk starts there.
    size_t i = 0, g0 = 0;
    if (!slamTrack.empty() && gpsTimeOrdered) {
        const double t0 = slamTrack[0].t - 0.000001;
        g0 = std::lower_bound(gps.begin(), gps.end(), t0, [](const COORDXYZTW &a, double t) { return a.t < t; }) - gps.begin();
    }
    for (size_t g = g0; g < gps.size() && i < slamTrack.size();) {
        const double dt = gps[g].t - slamTrack[i].t;
        if (std::fabs(dt) < 0.000001) {
            COORDXYZT p = {gps[g].x, gps[g].y, gps[g].z, gps[g].t};
            GPSWithSlam.push_back(p);
            weight.push_back(gps[g].w);
            slamWithGPS.push_back(slamTrack[i]);
            ++i;
            ++g;
        } else if (dt > 0) {
            ++i;  // this SLAM stamp has no GPS sample
        } else {
            ++g;
        }
    }
}

void ShortDistanceTrackProcess::merge(const std::vector<COORDXYZT> &seg, const std::vector<double> &weight)
{
    // overlap cross-fade (short_distance_track_process.cpp:73-158)
    if (out_.empty()) {
        for (size_t i = 0; i < seg.size(); ++i) {
            COORDXYZTW p = {seg[i].x, seg[i].y, seg[i].z, seg[i].t, weight[i]};
            out_.push_back(p);
        }
        return;
    }
    size_t it = 0;
    int num = 1, sm = -1, op = -1;
    bool overlap = false;
    std::vector<size_t> lost;
    const size_t na = out_.size();
    double c1 = 0.0, c2 = 0.0;
    // Samples before the first match only ever enter the "lost" list that the first match
    // clears (SD:104-107) -- or, without any match, that SD:137-140 clears -- so on a
    // time-ordered track the scan can start at the segment's first stamp.
    size_t a0 = 0;
    if (!seg.empty() && sorted_) {
        const double t0 = seg[0].t - 0.000001;
        a0 = std::lower_bound(out_.begin(), out_.end(), t0, [](const COORDXYZTW &p, double t) { return p.t < t; }) - out_.begin();
    }
    for (size_t a = a0; a < na; ++a) {
        const bool match = it < seg.size() && std::fabs(out_[a].t - seg[it].t) < 0.000001;
        if (!match) {
            lost.push_back(a);
            continue;
        }
        overlap = true;
        if (op == -1) {
            lost.clear();
            op = (int)(na - a);
            sm = op / 2;
        }
        if (num <= sm) {
            c1 = 1.0 - num / (2.0 * sm);
            c2 = num / (2.0 * sm);
        } else if (num > sm && num <= op - sm) {
            c1 = c2 = 0.5;
        } else if (num > op - sm) {
            c1 = (op - num + 1) / (2.0 * sm);
            c2 = 1.0 - (op - num + 1) / (2.0 * sm);
        }
        out_[a].x = out_[a].x * c1 + seg[it].x * c2;
        out_[a].y = out_[a].y * c1 + seg[it].y * c2;
        out_[a].z = out_[a].z * c1 + seg[it].z * c2;
        out_[a].w = out_[a].w * c1 + weight[it] * c2;
        ++it;
        ++num;
    }
    if (!overlap) lost.clear();
    for (; it < seg.size(); ++it) {
        COORDXYZTW p = {seg[it].x, seg[it].y, seg[it].z, seg[it].t, weight[it]};
        out_.push_back(p);
    }
    for (size_t k = lost.size(); k-- > 0;) out_.erase(out_.begin() + lost[k]);
    // keep track of time order for the shortcut above
    if (sorted_)
        for (size_t k = (a0 > 0 ? a0 : 1); k < out_.size() && sorted_; ++k) sorted_ = out_[k - 1].t <= out_[k].t;
}

void ShortDistanceTrackProcess::process(const std::vector<COORDXYZT> &slamTrack)
{
    std::vector<std::vector<COORDXYZT> > one(1, slamTrack);
    processBatch(one);
}

void ShortDistanceTrackProcess::processBatch(const std::vector<std::vector<COORDXYZT> > &tracks)
{
    if (gps_.empty()) throw std::runtime_error("WARN: total GPS track from long_distance_track_process is NULL.");
    // every queued segment is matched on the host, all fits run in ONE launch, then the
    // sequential overlap merge (segment k+1 blends into the accumulated track)
    std::vector<COORDXYZT> slam, enu;
    std::vector<double> w;
    std::vector<int> off(1, 0);
    for (const auto &trk : tracks) {
        std::vector<COORDXYZT> s, g;
        std::vector<double> ww;
        getGPS(gps_, trk, s, g, ww, gps_sorted_);
        if (s.empty()) continue;
        slam.insert(slam.end(), s.begin(), s.end());
        enu.insert(enu.end(), g.begin(), g.end());
        w.insert(w.end(), ww.begin(), ww.end());
        off.push_back((int)slam.size());
    }
    const int nseg = (int)off.size() - 1;
    if (nseg == 0) return;
    std::vector<COORDXYZT> cal(slam.size());
    check(gpscal_track_fit_batched(default_ctx(), &slam[0].x, &enu[0].x, w.data(), off.data(), nseg, nullptr, nullptr,
                                   &cal[0].x),
          "gpscal_track_fit_batched");
    for (int s = 0; s < nseg; ++s) {
        std::vector<COORDXYZT> c(cal.begin() + off[s], cal.begin() + off[s + 1]);
        std::vector<double> ww(w.begin() + off[s], w.begin() + off[s + 1]);
        merge(c, ww);
    }
}
