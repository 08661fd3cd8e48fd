// pipeline.cc -- the whole track pipeline (what the long and short nodes plus the KML
// writer do between the end of SLAM and the files on disk) as one in-process C call, so
// benchmarks and tests can time it without process start-up.  Same steps as gpscal_run.
#include <chrono>
#include <cstdio>
#include <stdexcept>

#include "gps_process.h"
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
