// pipeline.cc -- the whole track pipeline (what the long and short nodes plus the KML
// writer do between the end of SLAM and the files on disk) as one in-process C call, so
// benchmarks and tests can time it without process start-up.  Same steps as gpscal_run.
#include <chrono>
#include <cstdio>
#include <stdexcept>

#include "gps_process.h"
#include "rosbag_reader.h"
#include "track_process.h"

extern "C" int gpscal_host_pipeline(const char *gps_log, const double *long_xyzt, const int *long_off, int nlong,
                                    const double *short_xyzt, const int *short_off, int nshort, const char *method,
                                    int band_type, const char *kml_original, const char *kml_calibrated,
                                    double *seconds /* [4]: long pass, short pass, wgs+kml, total */,
                                    int *n_points /* [2]: whole-run GPS points, calibrated points */)
{
    typedef std::chrono::steady_clock clk;
    auto sec = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    try {
        auto unpack = [](const double *xyzt, const int *off, int n) {
            std::vector<std::vector<COORDXYZT> > v(n);
            for (int s = 0; s < n; ++s) {
                const COORDXYZT *p = reinterpret_cast<const COORDXYZT *>(xyzt) + off[s];
                v[s].assign(p, p + (off[s + 1] - off[s]));
            }
            return v;
        };
        const auto longs = unpack(long_xyzt, long_off, nlong);
        const auto shorts = unpack(short_xyzt, short_off, nshort);
        GPSPro gps;
        gps.setGPSPath(gps_log);
        gps.setMethod(method);
        gps.setType(band_type);
        gps.setKMLConfigPath("/nonexistent");  // shipped defaults
        const auto t0 = clk::now();
        LongDistanceTrackProcess lp(gps);
        lp.processBatch(longs);
        if (lp.totalTrack().empty()) return -1;
        const auto t1 = clk::now();
        ShortDistanceTrackProcess sp;
        sp.setGPS(lp.totalTrack());
        sp.processBatch(shorts);
        const auto t2 = clk::now();
        std::vector<std::pair<double, double> > oriWGSBL, impWGSBL;
        std::vector<double> oriAlt, impAlt;
        std::vector<std::pair<int, std::string> > oriCol, impCol;
        gps.ENUToGPS(sp.gps(), oriWGSBL, oriAlt, oriCol);
        gps.ENUToGPS(sp.result(), impWGSBL, impAlt, impCol);
        if (kml_original && *kml_original) gps.createKML(kml_original, oriWGSBL, oriAlt, 0, oriCol);
        if (kml_calibrated && *kml_calibrated) gps.createKML(kml_calibrated, impWGSBL, impAlt, 1, impCol);
        const auto t3 = clk::now();
        if (seconds) {
            seconds[0] = sec(t0, t1);
            seconds[1] = sec(t1, t2);
            seconds[2] = sec(t2, t3);
            seconds[3] = sec(t0, t3);
        }
        if (n_points) {
            n_points[0] = (int)sp.gps().size();
            n_points[1] = (int)sp.result().size();
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "gpscal_host_pipeline: %s\n", e.what());
        return 1;
    }
    return 0;
}

// Raw sweeps -> KML: input_data's replay + segmentation with the LOAM nodes on the device
// (gpscal_input_data_run), then the long / short track nodes and the KML writer as above.  This is
// what `./run.sh` does between the bags and the two KML files (run.sh:63-75, launch/*.launch).
extern "C" int gpscal_host_pipeline_sweeps(const char *gps_log, int nbag, const float *xyz, const int *sweep_off,
                                           const int *bag_sweep_off, const double *stamps, double long_distance,
                                           double short_distance, double overlap_distance, const char *method,
                                           int band_type, const char *kml_original, const char *kml_calibrated,
                                           double *seconds /* [5]: slam, long pass, short pass, wgs+kml, total */,
                                           int *n_out /* [4]: long tracks, short tracks, GPS points, calibrated points */)
{
    typedef std::chrono::steady_clock clk;
    auto sec = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    try {
        const int nsw = bag_sweep_off[nbag];
        const int cap_t = 2 * nsw + 8, cap_r = 8 * nsw + 16;
        std::vector<int> flag(cap_t), bag(cap_t), first(cap_t), last(cap_t), toff(cap_t + 1);
        std::vector<double> rows((size_t)cap_r * 4);
        int nt = 0;
        const auto t0 = clk::now();
        gpscal_host::check(gpscal_input_data_run(gpscal_host::default_ctx(), nbag, xyz, sweep_off, bag_sweep_off, stamps,
                                            long_distance, short_distance, overlap_distance, cap_t, flag.data(),
                                            bag.data(), first.data(), last.data(), toff.data(), rows.data(), cap_r, &nt,
                                            0, 0),
                      "gpscal_input_data_run");
        std::vector<std::vector<COORDXYZT> > longs, shorts;
        for (int k = 0; k < nt; ++k) {
            const COORDXYZT *p = reinterpret_cast<const COORDXYZT *>(rows.data()) + toff[k];
            std::vector<COORDXYZT> v(p, p + (toff[k + 1] - toff[k]));
            if (v.empty()) continue;
            (flag[k] == 0 ? longs : shorts).push_back(v);
        }
        const auto t1 = clk::now();
        GPSPro gps;
        gps.setGPSPath(gps_log);
        gps.setMethod(method);
        gps.setType(band_type);
        gps.setKMLConfigPath("/nonexistent");
        LongDistanceTrackProcess lp(gps);
        lp.processBatch(longs);
        if (lp.totalTrack().empty()) return -1;
        const auto t2 = clk::now();
        ShortDistanceTrackProcess sp;
        sp.setGPS(lp.totalTrack());
        sp.processBatch(shorts);
        const auto t3 = clk::now();
        std::vector<std::pair<double, double> > oriWGSBL, impWGSBL;
        std::vector<double> oriAlt, impAlt;
        std::vector<std::pair<int, std::string> > oriCol, impCol;
        gps.ENUToGPS(sp.gps(), oriWGSBL, oriAlt, oriCol);
        gps.ENUToGPS(sp.result(), impWGSBL, impAlt, impCol);
        if (kml_original && *kml_original) gps.createKML(kml_original, oriWGSBL, oriAlt, 0, oriCol);
        if (kml_calibrated && *kml_calibrated) gps.createKML(kml_calibrated, impWGSBL, impAlt, 1, impCol);
        const auto t4 = clk::now();
        if (seconds) {
            seconds[0] = sec(t0, t1);
            seconds[1] = sec(t1, t2);
            seconds[2] = sec(t2, t3);
            seconds[3] = sec(t3, t4);
            seconds[4] = sec(t0, t4);
        }
        if (n_out) {
            n_out[0] = (int)longs.size();
            n_out[1] = (int)shorts.size();
            n_out[2] = (int)sp.gps().size();
            n_out[3] = (int)sp.result().size();
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "gpscal_host_pipeline_sweeps: %s\n", e.what());
        return 1;
    }
    return 0;
}

// Reads one bag's PointCloud2 topic into caller buffers (tests / tools).  Returns 0, -2 when a
// buffer is too small (nmsgs / npts then hold the required sizes), 1 on a malformed file.
extern "C" int gpscal_host_read_bag(const char *path, const char *topic, float *xyz, int cap_pts, int *sweep_off,
                                    double *stamps, int cap_msgs, int *nmsgs, int *npts)
{
    gpscal_host::CloudSeries S;
    std::string err;
    if (!gpscal_host::read_bag_clouds(path, topic, S, err)) {
        fprintf(stderr, "gpscal_host_read_bag: %s\n", err.c_str());
        return 1;
    }
    if (S.sweep_off.empty()) S.sweep_off.push_back(0);
    *nmsgs = (int)S.stamps.size();
    *npts = S.sweep_off.back();
    if (*nmsgs > cap_msgs || *npts > cap_pts) return -2;
    std::copy(S.xyz.begin(), S.xyz.end(), xyz);
    std::copy(S.sweep_off.begin(), S.sweep_off.end(), sweep_off);
    std::copy(S.stamps.begin(), S.stamps.end(), stamps);
    return 0;
}
