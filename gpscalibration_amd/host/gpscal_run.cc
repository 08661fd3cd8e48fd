// gpscal_run.cc -- ROS-free driver with run.sh's surface: the same parameters
// (run.sh:27-61) as --name value options, the same two passes (long segments -> weights,
// short overlapping segments -> fits -> overlap merge), the same outputs (original and
// calibrated KML).  What roslaunch + the seven nodes do per message at 1 Hz
// (input_data.cpp:251,333) happens here in two batched GPU launches.
//
// Input: rosbag reading and LOAM are not part of this round, so
// the SLAM side enters as a track file, one block per segment exactly as input_data
// publishes them on /slam_track (input_data.cpp:355-363):
//     <track_flag> <n>          0 = long segment, 1 = short segment
//     x y z t                   n lines, COORDXYZT
// or as one continuous pose chain (--pose_chain) that is cut by travelled distance with
// input_data's rule (long 1000 m / short 300 m with 100 m overlap, input_data.cpp:106-116).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <stdexcept>
#include <string>

#include "gps_process.h"
#include "rosbag_reader.h"
#include "track_process.h"

namespace {

typedef std::vector<COORDXYZT> Track;

bool read_tracks(const std::string &path, std::vector<Track> &longs, std::vector<Track> &shorts)
{
    std::ifstream in(path.c_str());
    if (!in.is_open()) return false;
    long flag, n;
    while (in >> flag >> n) {
        Track t(n);
        for (long i = 0; i < n; ++i)
            if (!(in >> t[i].x >> t[i].y >> t[i].z >> t[i].t)) return false;
        (flag == 0 ? longs : shorts).push_back(t);
    }
    return true;
}

bool read_chain(const std::string &path, Track &chain)
{
    std::ifstream in(path.c_str());
    if (!in.is_open()) return false;
    COORDXYZT p;
    while (in >> p.x >> p.y >> p.z >> p.t) chain.push_back(p);
    return !chain.empty();
}

// input_data's segmentation by travelled 3-D distance: a segment ends once the path
// exceeds `dist`; the next one restarts at the last pose within dist - overlap
// (input_data.cpp:106-116,335-343); a tail shorter than dist/3 joins the previous segment
// (input_data.cpp:367-424).  Every segment is re-based to its first pose, as LOAM restarts
// from the origin per segment (laserOdometry.cpp:519-563).
std::vector<Track> cut_segments(const Track &chain, double dist, double overlap)
{
    std::vector<std::pair<size_t, size_t> > spans;
    size_t start = 0;
    while (start + 1 < chain.size()) {
        double acc = 0;
        size_t resume = start, end = chain.size();
        for (size_t i = start + 1; i < chain.size(); ++i) {
            const double dx = chain[i].x - chain[i - 1].x, dy = chain[i].y - chain[i - 1].y, dz = chain[i].z - chain[i - 1].z;
            acc += std::sqrt(dx * dx + dy * dy + dz * dz);
            if (acc <= dist - overlap) resume = i;
            if (acc > dist) {
                end = i + 1;
                break;
            }
        }
        if (end == chain.size() && acc < dist / 3 && !spans.empty()) {
            spans.back().second = chain.size();  // short tail: merged into the previous segment
            break;
        }
        spans.push_back(std::make_pair(start, end));
        if (end == chain.size()) break;
        start = resume > start ? resume : end - 1;
    }
    std::vector<Track> out;
    for (auto &sp : spans) {
        Track t(chain.begin() + sp.first, chain.begin() + sp.second);
        const COORDXYZT o = t[0];
        for (auto &p : t) {
            p.x -= o.x;
            p.y -= o.y;
            p.z = HEIGHT;  // transformMaintenance.cpp:149
        }
        out.push_back(t);
    }
    return out;
}

// SLAM tracks of one replayed cloud series: input_data's replay + segmentation and the four LOAM
// nodes on the device; the tracks come back as /slam_track would carry them.
bool slam_tracks_from_series(const gpscal_host::CloudSeries &C, int nbag, const std::vector<int> &bag_off, double L,
                             double S, double OV, std::vector<Track> &longs, std::vector<Track> &shorts)
{
    const int nsw = (int)C.stamps.size();
    if (nsw < 1) {
        fprintf(stderr, "no lidar sweeps to replay\n");
        return false;
    }
    const int cap_t = 2 * nsw + 8, cap_r = 8 * nsw + 16;
    std::vector<int> flag(cap_t), bag(cap_t), first(cap_t), last(cap_t), toff(cap_t + 1);
    std::vector<double> rows((size_t)cap_r * 4);
    int nt = 0;
    gpscal_host::check(gpscal_input_data_run(gpscal_host::default_ctx(), nbag, C.xyz.data(), C.sweep_off.data(),
                                             bag_off.data(), C.stamps.data(), L, S, OV, cap_t, flag.data(), bag.data(),
                                             first.data(), last.data(), toff.data(), rows.data(), cap_r, &nt, 0, 0),
                       "gpscal_input_data_run");
    for (int k = 0; k < nt; ++k) {
        const COORDXYZT *p = reinterpret_cast<const COORDXYZT *>(rows.data()) + toff[k];
        Track v(p, p + (toff[k + 1] - toff[k]));
        if (v.empty()) continue;
        (flag[k] == 0 ? longs : shorts).push_back(v);
    }
    printf("SLAM Track Calculation Over\n");  // input_data.cpp:442
    return true;
}

// Sweep file (a plain container for tests): "GPSW1\n", int32 nbag, then per bag int32 nsweeps and
// per sweep { double stamp; int32 npoints; float xyz[3*npoints] }.
bool slam_tracks_from_sweeps(const std::string &path, double L, double S, double OV, std::vector<Track> &longs,
                             std::vector<Track> &shorts)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) {
        fprintf(stderr, "open %s error\n", path.c_str());
        return false;
    }
    char magic[6] = {0};
    int nbag = 0;
    bool ok = fread(magic, 1, 6, f) == 6 && memcmp(magic, "GPSW1\n", 6) == 0 && fread(&nbag, 4, 1, f) == 1 && nbag > 0;
    gpscal_host::CloudSeries C;
    C.sweep_off.push_back(0);
    std::vector<int> bag_off(1, 0);
    for (int b = 0; ok && b < nbag; ++b) {
        int ns = 0;
        ok = fread(&ns, 4, 1, f) == 1 && ns >= 0;
        for (int k = 0; ok && k < ns; ++k) {
            double t;
            int n;
            ok = fread(&t, 8, 1, f) == 1 && fread(&n, 4, 1, f) == 1 && n >= 0;
            if (!ok) break;
            const size_t at = C.xyz.size();
            C.xyz.resize(at + (size_t)3 * n);
            ok = n == 0 || fread(C.xyz.data() + at, 12, (size_t)n, f) == (size_t)n;
            C.stamps.push_back(t);
            C.sweep_off.push_back(C.sweep_off.back() + n);
        }
        bag_off.push_back((int)C.stamps.size());
    }
    fclose(f);
    if (!ok || C.stamps.empty()) {
        fprintf(stderr, "%s is not a sweep file\n", path.c_str());
        return false;
    }
    return slam_tracks_from_series(C, nbag, bag_off, L, S, OV, longs, shorts);
}

// run.sh's bag_input_filename: a text file with one bag path per line (input_data.cpp:120-150); the
// bags are replayed one after the other as ONE message sequence (segments may span bags, ID:296-345).
bool slam_tracks_from_bags(const std::string &list, const std::string &topic, double L, double S, double OV,
                           std::vector<Track> &longs, std::vector<Track> &shorts)
{
    std::ifstream in(list.c_str());
    if (!in) {
        fprintf(stderr, "ERROR: open %s error,please check it\n", list.c_str());  // input_data.cpp:127
        return false;
    }
    gpscal_host::CloudSeries C;
    C.sweep_off.push_back(0);
    std::string line, err;
    int nbags = 0;
    while (std::getline(in, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == ' ')) line.pop_back();
        if (line.empty()) continue;
        if (!gpscal_host::read_bag_clouds(line, topic, C, err)) {
            fprintf(stderr, "ERROR: %s\n", err.c_str());
            return false;
        }
        ++nbags;
    }
    if (nbags == 0) {
        fprintf(stderr, "WARN:%s is NULL,please check it.\n", list.c_str());  // input_data.cpp:146
        return false;
    }
    const std::vector<int> bag_off = {0, (int)C.stamps.size()};
    return slam_tracks_from_series(C, 1, bag_off, L, S, OV, longs, shorts);
}

void usage()
{
    fprintf(stderr,
            "usage: gpscal_run --gps_input_filename LOG (--bag_input_filename LIST | --sweeps FILE |\n"
            "                  --slam_track_filename FILE | --pose_chain FILE)\n"
            "       [--gps_original_filename out.kml] [--gps_improved_filename out.kml] [--result_control 1]\n"
            "       [--total_long_distance 1000] [--total_short_distance 300] [--overlap_distance 100]\n"
            "       [--ctm UTM|Gaussion] [--gdt 3|6] [--kml_config src/gpsCalibration/config/kml_config.xml]\n");
}

}  // namespace

int main(int argc, char **argv)
{
    std::map<std::string, std::string> a;
    a["result_control"] = "1";
    a["gps_original_filename"] = "./data/original_gps_file.kml";   // run.sh:30-31
    a["gps_improved_filename"] = "./data/calibration_gps_file.kml";
    a["total_long_distance"] = "1000";                             // run.sh:46-48
    a["total_short_distance"] = "300";
    a["overlap_distance"] = "100";
    a["ctm"] = "UTM";                                              // run.sh:60-61
    a["gdt"] = "3";
    for (int i = 1; i + 1 < argc; i += 2) {
        if (strncmp(argv[i], "--", 2) != 0) {
            usage();
            return -1;
        }
        a[argv[i] + 2] = argv[i + 1];
    }
    if (!a.count("gps_input_filename") ||
        (!a.count("slam_track_filename") && !a.count("pose_chain") && !a.count("sweeps") &&
         !a.count("bag_input_filename"))) {
        usage();
        return -1;
    }
    if (a["ctm"] != "UTM" && a["ctm"] != "Gaussion") {
        fprintf(stderr, "ERROR: ctm=projectmethod(UTM/Gaussion)\n");  // long_distance_track_process.cpp:98-102
        return -1;
    }
    const int gdt = atoi(a["gdt"].c_str());
    if (gdt != 3 && gdt != 6) {
        fprintf(stderr, "ERROR: gdt=bandwidth(3/6)\n");  // long_distance_track_process.cpp:104-108
        return -1;
    }
    try {
        std::vector<Track> longs, shorts;
        if (a.count("slam_track_filename")) {
            if (!read_tracks(a["slam_track_filename"], longs, shorts)) {
                fprintf(stderr, "open %s error\n", a["slam_track_filename"].c_str());
                return -1;
            }
        } else if (a.count("bag_input_filename")) {
            // run.sh's own input: the list of rosbag files with the /velodyne_points clouds
            if (!slam_tracks_from_bags(a["bag_input_filename"], a.count("bag_topic") ? a["bag_topic"] : "velodyne_points",
                                       atof(a["total_long_distance"].c_str()), atof(a["total_short_distance"].c_str()),
                                       atof(a["overlap_distance"].c_str()), longs, shorts))
                return -1;
        } else if (a.count("sweeps")) {
            // raw lidar sweeps (the bags' /velodyne_points): input_data's replay + segmentation and the
            // four LOAM nodes run on the device, the tracks come back as /slam_track would carry them
            if (!slam_tracks_from_sweeps(a["sweeps"], atof(a["total_long_distance"].c_str()),
                                         atof(a["total_short_distance"].c_str()), atof(a["overlap_distance"].c_str()),
                                         longs, shorts))
                return -1;
        } else {
            Track chain;
            if (!read_chain(a["pose_chain"], chain)) {
                fprintf(stderr, "open %s error\n", a["pose_chain"].c_str());
                return -1;
            }
            longs = cut_segments(chain, atof(a["total_long_distance"].c_str()), 0.0);
            shorts = cut_segments(chain, atof(a["total_short_distance"].c_str()), atof(a["overlap_distance"].c_str()));
        }
        GPSPro gps;
        gps.setGPSPath(a["gps_input_filename"]);
        gps.setMethod(a["ctm"]);
        gps.setType(gdt);
        if (a.count("kml_config")) gps.setKMLConfigPath(a["kml_config"]);

        // pass 1: long segments -> whole-run ENU GPS + IRLS weights (the gps_weight topic)
        LongDistanceTrackProcess lp(gps);
        lp.processBatch(longs);
        if (lp.totalTrack().empty()) {
            fprintf(stderr, "WARN: no GPS track,please check it\n");  // long_distance_track_process.cpp:49
            return -1;
        }
        // pass 2: short segments -> weighted fits -> overlap merge
        ShortDistanceTrackProcess sp;
        sp.setGPS(lp.totalTrack());
        sp.processBatch(shorts);

        std::vector<std::pair<double, double> > oriWGSBL, impWGSBL;
        std::vector<double> oriAlt, impAlt;
        std::vector<std::pair<int, std::string> > oriCol, impCol;
        gps.ENUToGPS(sp.gps(), oriWGSBL, oriAlt, oriCol);        // short_distance_track_process.cpp:254-255
        gps.ENUToGPS(sp.result(), impWGSBL, impAlt, impCol);
        printf("oriWGSBL.size() = %zu\n", oriWGSBL.size());
        const int rc = atoi(a["result_control"].c_str());
        if (rc == 2) {  // BAIDU_MAP_FILE, short_distance_track_process.cpp:271-282
            std::vector<std::pair<double, double> > GCJ02, BD09;
            gps.GPSToGCJ(oriWGSBL, GCJ02);
            gps.GCJToBD(GCJ02, BD09);
            gps.createJSON(a["gps_original_filename"], BD09, 0, oriCol);
            GCJ02.clear();
            BD09.clear();
            gps.GPSToGCJ(impWGSBL, GCJ02);
            gps.GCJToBD(GCJ02, BD09);
            gps.createJSON(a["gps_improved_filename"], BD09, 1, impCol);
            return 0;
        }
        if (rc == 3) {  // GAODE_MAP_FILE, short_distance_track_process.cpp:283-291
            std::vector<std::pair<double, double> > GCJ02;
            gps.GPSToGCJ(oriWGSBL, GCJ02);
            gps.createJSON(a["gps_original_filename"], GCJ02, 0, oriCol);
            GCJ02.clear();
            gps.GPSToGCJ(impWGSBL, GCJ02);
            gps.createJSON(a["gps_improved_filename"], GCJ02, 1, impCol);
            return 0;
        }
        if (rc == 4) {  // PUBLISH_MESSAGE, short_distance_track_process.cpp:293-309
            // the node publishes gpsCalibration/IMMessage on /imorpheus_gps; without ROS the same records go to the
            // "improved" file, one "b,l,w" line per IMGPS (INTEGRATION.md shows the five-line publisher)
            const std::vector<IMGPS> track = gps.calibratedGPSMessage(sp.result());
            FILE *f = fopen(a["gps_improved_filename"].c_str(), "w");
            if (!f) {
                fprintf(stderr, "cannot write %s\n", a["gps_improved_filename"].c_str());
                return 1;
            }
            for (size_t i = 0; i < track.size(); ++i) fprintf(f, "%.15g,%.15g,%.15g\n", track[i].b, track[i].l, track[i].w);
            fclose(f);
            printf("==================== Start to publish calibrated gps ====================\n");
            return 0;
        }
        printf("====================  Create original GPS KML  ====================\n");
        gps.createKML(a["gps_original_filename"], oriWGSBL, oriAlt, 0, oriCol);
        printf("==================== Create calibrated GPS KML ====================\n");
        gps.createKML(a["gps_improved_filename"], impWGSBL, impAlt, 1, impCol);
        printf("====================            END            ====================\n");
    } catch (const std::exception &e) {
        fprintf(stderr, "gpscal_run: %s\n", e.what());
        return 1;
    }
    return 0;
}
