// gps_process.h -- the part of the reference's GPSPro (include/gpsCalibration/
// gps_process.h:24-57) that sits on the hot path: GPRMC ingest, GPSToENU, ENUToGPS,
// colour segments, KML.  Text handling stays on the host; projection, interpolation and
// the inverse projection run on the GPU (gpscal_gps_to_enu / gpscal_enu_to_wgs).
// GCJ-02 / BD-09 / JSON outputs are out of scope this round.
#ifndef GPSCAL_HOST_GPS_PROCESS_H
#define GPSCAL_HOST_GPS_PROCESS_H
#include "common.h"

class GPSPro {
public:
    GPSPro();
    int getType();
    void setType(int type);
    std::string getMethod();
    void setMethod(std::string method);
    std::string getGPSPath();
    void setGPSPath(std::string originalGPSPath);
    void setKMLConfigPath(std::string path);  // default: src/gpsCalibration/config/kml_config.xml (gps_process.cc:632)

    // GPS coordinate -> ENU at the SLAM stamps (gps_process.cc:476-521).  Returns an empty
    // vector (the reference exit(0)s) when the log has no fix for the track's time span.
    std::vector<COORDXYZT> GPSToENU(std::vector<COORDXYZT> slamTrack);
    // The same for many segments with ONE device call; out[s] == GPSToENU(tracks[s]).  The log
    // is parsed once and cached (the reference re-opens and re-parses it per segment,
    // gps_process.cc:113-159); every segment still gets its own time window and its own
    // dropout fill, so the numbers are those of per-segment calls.
    std::vector<std::vector<COORDXYZT> > GPSToENUBatch(const std::vector<std::vector<COORDXYZT> > &tracks);
    // ENU -> WGS84 + 50 m colour segments (gps_process.cc:374-386, 600-626, 1010-1058)
    int ENUToGPS(std::vector<COORDXYZTW> enuCoor, std::vector<std::pair<double, double> > &WGSBL,
                 std::vector<double> &altitude, std::vector<std::pair<int, std::string> > &segmentColor);
    // The /imorpheus_gps payload of result_control 4 (short_distance_track_process.cpp:295-309): IMMessage.track
    std::vector<IMGPS> calibratedGPSMessage(const std::vector<COORDXYZTW> &calibrated);
    // KML writer (gps_process.cc:759-847), flag 0 = original track, 1 = calibrated
    int createKML(std::string KMLFileName, std::vector<std::pair<double, double> > WGSBL, std::vector<double> altitude,
                  int flag, std::vector<std::pair<int, std::string> > segmentColor);

    // WGS-84 -> GCJ-02 -> BD-09 and back (gps_process.cc:526-595), pairs {longitude, latitude}
    int GPSToGCJ(std::vector<std::pair<double, double> > vecGpsCoor, std::vector<std::pair<double, double> > &vecGCJ);
    int GCJToBD(std::vector<std::pair<double, double> > vecGCJCoor, std::vector<std::pair<double, double> > &vecBD);
    int BDToGCJ(std::vector<std::pair<double, double> > vecBDCoor, std::vector<std::pair<double, double> > &vecGCJ);
    // map-API JSON writer (gps_process.cc:1210-1250), flag as createKML
    void createJSON(std::string fileName, std::vector<std::pair<double, double> > GPSValue, int flag,
                    std::vector<std::pair<int, std::string> > segmentColor);

    // host-side pieces, exposed for tests
    static int parseGPRMC(const std::string &path, double startTime, double endTime, std::vector<double> &lat,
                          std::vector<double> &lon, std::vector<double> &t);
    static int gpsProcess(std::vector<double> &lat, std::vector<double> &lon, const std::vector<double> &t);
    static std::vector<std::pair<int, std::string> > segment(const std::vector<COORDXYZTW> &enuCoor);

private:
    int type;
    std::string method;
    std::string originalGPSPath;
    std::string kmlConfigPath;
    std::vector<std::string> readKMLParameter();
    // parsed log (all lines, sentinel for 'V'), in file order; stamp 0 for blank lines
    bool logLoaded = false;
    std::vector<double> logLat, logLon, logT;
    std::vector<char> logKeep;  // 0: a $GPGGA line without coordinates, never kept (gps_process.cc:293)
    bool loadLog();
    void window(double startTime, double endTime, std::vector<double> &lat, std::vector<double> &lon,
                std::vector<double> &t) const;
};
#endif
