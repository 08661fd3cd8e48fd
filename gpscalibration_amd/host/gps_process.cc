// gps_process.cc -- GPSPro: NMEA ingest and KML on the host, projections on the GPU.
// Behavioural reference: src/gpsCalibration/src/gps_calibration/gps_process.cc
// (:113-229 ingest, :389-473 dropout fill, :476-521 GPSToENU, :374-386 ENUToGPS,
//  :600-626 + :692-756 colour segments, :629-689 kml_config.xml, :759-847 KML).
#include "gps_process.h"

#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

using gpscal_host::check;
using gpscal_host::default_ctx;

GPSPro::GPSPro() : type(3), method("UTM"), kmlConfigPath("src/gpsCalibration/config/kml_config.xml") {}

int GPSPro::getType() { return type; }
void GPSPro::setType(int t)
{
    if (t != 3 && t != 6) {
        printf("type value is 3 or 6,the default value is 3\n");
        t = 3;
    }
    type = t;
}
std::string GPSPro::getMethod() { return method; }
void GPSPro::setMethod(std::string m)
{
    if (m != "UTM" && m != "Gaussion") {
        printf("method value is \"UTM\" or \"Gaussion\",the default is \"UTM\"\n");
        m = "UTM";
    }
    method = m;
}
std::string GPSPro::getGPSPath() { return originalGPSPath; }
void GPSPro::setGPSPath(std::string p)
{
    originalGPSPath = p;
    logLoaded = false;
}
void GPSPro::setKMLConfigPath(std::string p) { kmlConfigPath = p; }

// ------------------------------------------------------------------ ingest

namespace {
// Fields of one log line with strtok(",") semantics: empty fields collapse.
struct Fields {
    std::vector<const char *> f;
    std::string buf;
    explicit Fields(const std::string &line) : buf(line)
    {
        buf.push_back('\0');
        char *p = &buf[0];
        while (*p) {
            while (*p == ',') ++p;
            if (!*p) break;
            f.push_back(p);
            while (*p && *p != ',') ++p;
            if (*p) *p++ = '\0';
        }
    }
};

double ddmm_to_deg(const char *tok)
{
    const double v = atof(tok);
    // truncation as in gps_process.cc:191,205 (an int cast there); in double here, so that a damaged field
    // (121238894999...9.25) is a wrong coordinate and not an overflow -- identical for every value an int holds
    const double d = std::trunc(v / 100);
    return d + (v - d * 100) / 60.0;
}

enum { SENTENCE_NONE = 0, SENTENCE_RMC = 1, SENTENCE_GGA = 2, SENTENCE_GLL = 3 };

// gps_process.cc:128-152: the second field of the first line names the sentence of the whole log
int sniff_sentence(const Fields &F)
{
    if (F.f.size() < 2) return SENTENCE_NONE;
    if (strcmp(F.f[1], "$GPRMC") == 0) return SENTENCE_RMC;
    if (strcmp(F.f[1], "$GPGGA") == 0) return SENTENCE_GGA;
    if (strcmp(F.f[1], "$GPGLL") == 0) return SENTENCE_GLL;
    return SENTENCE_NONE;
}

// One line of the log (getGPRMCFormat gps_process.cc:161-229, getGPGGAFormat :231-299,
// getGPGLLFormat :300-372), column walk with the reference's early exits.  Returns false when the
// fix must not be kept whatever its stamp (GPGGA without coordinates, :293).
bool parse_fix(int kind, const Fields &F, double &stamp, double &la, double &lo)
{
    stamp = 0;
    la = 90;  // (90,180) marks "no fix" (gps_process.cc:169)
    lo = 180;
    const size_t n = F.f.size();
    if (n >= 1) stamp = atof(F.f[0]);
    if (kind == SENTENCE_RMC) {
        if (n >= 4 && strcmp(F.f[3], "V") == 0) return true;  // :176-179
        if (n >= 5) la = ddmm_to_deg(F.f[4]);
        if (n >= 6 && strcmp(F.f[5], "S") == 0) la = 0 - la;
        if (n >= 7) lo = ddmm_to_deg(F.f[6]);
        if (n >= 8 && strcmp(F.f[7], "W") == 0) lo = 0 - lo;
        return true;
    }
    if (kind == SENTENCE_GGA) {
        if (n >= 4) la = ddmm_to_deg(F.f[3]);
        if (n >= 5) {
            if (strcmp(F.f[4], "N") != 0 && strcmp(F.f[4], "S") != 0) return !(la == 90 || lo == 180);  // :246-249
            if (strcmp(F.f[4], "S") == 0) la = 0 - la;
        }
        if (n >= 6) lo = ddmm_to_deg(F.f[5]);
        if (n >= 7) {
            if (strcmp(F.f[6], "W") != 0 && strcmp(F.f[6], "E") != 0) return !(la == 90 || lo == 180);  // :250-253
            if (strcmp(F.f[6], "W") == 0) lo = 0 - lo;
        }
        return !(la == 90 || lo == 180);  // :293
    }
    // GPGLL: a status V in column 8 ends the walk after every coordinate column (:315-318), and the
    // fix is kept without a sentinel test (:364)
    if (n >= 3) la = ddmm_to_deg(F.f[2]);
    if (n >= 4 && strcmp(F.f[3], "S") == 0) la = 0 - la;
    if (n >= 5) lo = ddmm_to_deg(F.f[4]);
    if (n >= 6 && strcmp(F.f[5], "W") == 0) lo = 0 - lo;
    return true;
}
}  // namespace

int GPSPro::parseGPRMC(const std::string &path, double startTime, double endTime, std::vector<double> &lat,
                       std::vector<double> &lon, std::vector<double> &t)
{
    std::ifstream in(path.c_str());
    if (!in.is_open()) {
        printf("open %s error\n", path.c_str());
        return 1;
    }
    std::string line;
    int kind = -1;
    while (std::getline(in, line)) {
        if (line.size() >= IMSDLEN) break;  // the reference's 512-byte getline would fail here
        Fields F(line);
        if (kind < 0) {
            kind = sniff_sentence(F);  // gps_process.cc:129-154
            if (kind == SENTENCE_NONE) {
                printf("[WARNING] The current version does not support the current GPS format\n");
                return 0;
            }
        }
        double stamp, la, lo;
        const bool keep = parse_fix(kind, F, stamp, la, lo);
        if (keep && (long)stamp >= (long)(startTime - 1) && (long)stamp <= (long)(endTime + 1)) {
            lat.push_back(la);
            lon.push_back(lo);
            t.push_back(stamp);
        }
        if (!(stamp < endTime + 1)) break;
    }
    return 0;
}

int GPSPro::gpsProcess(std::vector<double> &lat, std::vector<double> &lon, const std::vector<double> &t)
{
    // dropout fill, one gap at a time, as gps_process.cc:389-473 walks them
    const int n = (int)lat.size();
    auto bad = [&](int i) { return lat[i] == 90 && lon[i] == 180; };
    int i = 0;
    while (i < n) {
        while (i < n && !bad(i)) ++i;
        if (i >= n) return 0;
        const int begin = i - 1;  // last good fix before the gap, -1 = gap at the start
        while (i < n && bad(i)) ++i;
        const int end = i < n ? i : -2;  // first good fix after the gap
        if (begin == -1) {
            if (end == -2 || end == n - 1) return 1;
            const double dT = t[end + 1] - t[end];
            const double dB = (lat[end + 1] - lat[end]) / dT, dL = (lon[end + 1] - lon[end]) / dT;
            for (int k = end - 1; k > begin; --k) {
                lat[k] = lat[k + 1] - dB * (t[k + 1] - t[k]);
                lon[k] = lon[k + 1] - dL * (t[k + 1] - t[k]);
            }
        } else if (end == -2) {
            if (begin == 0) return 1;
            const double dT = t[begin] - t[begin - 1];
            const double dB = (lat[begin] - lat[begin - 1]) / dT, dL = (lon[begin] - lon[begin - 1]) / dT;
            for (int k = begin + 1; k < n; ++k) {
                lat[k] = lat[k - 1] + dB * (t[k] - t[k - 1]);
                lon[k] = lon[k - 1] + dL * (t[k] - t[k - 1]);
            }
        } else {
            const double dT = t[end] - t[begin];
            const double dB = (lat[end] - lat[begin]) / dT, dL = (lon[end] - lon[begin]) / dT;
            for (int k = begin + 1; k < end; ++k) {
                lat[k] = lat[k - 1] + dB * (t[k] - t[k - 1]);
                lon[k] = lon[k - 1] + dL * (t[k] - t[k - 1]);
            }
        }
    }
    return 0;
}

std::vector<COORDXYZT> GPSPro::GPSToENU(std::vector<COORDXYZT> slamTrack)
{
    std::vector<COORDXYZT> out;
    if (slamTrack.empty()) return out;
    std::vector<double> lat, lon, t;
    if (parseGPRMC(originalGPSPath, slamTrack.front().t, slamTrack.back().t, lat, lon, t) != 0) return out;
    if (t.empty()) {
        printf("WARN: cannot find GPS information corresponding to slam track time,please check GPS original file.\n");
        return out;
    }
    gpsProcess(lat, lon, t);
    out.resize(slamTrack.size());
    int n_out = 0;
    check(gpscal_gps_to_enu(default_ctx(), method == "UTM" ? GPSCAL_METHOD_UTM : GPSCAL_METHOD_GAUSS, type, lat.data(),
                            lon.data(), t.data(), (int)t.size(), &slamTrack[0].x, (int)slamTrack.size(), &out[0].x,
                            &n_out),
          "gpscal_gps_to_enu");
    out.resize(n_out);
    return out;
}

bool GPSPro::loadLog()
{
    if (logLoaded) return true;
    // every line of the log, window test disabled (start = -inf, end = +inf would stop at
    // nothing): parse with the widest window the (long) casts allow
    logLat.clear();
    logLon.clear();
    logT.clear();
    std::ifstream in(originalGPSPath.c_str());
    if (!in.is_open()) {
        printf("open %s error\n", originalGPSPath.c_str());
        return false;
    }
    std::string line;
    int kind = -1;
    logKeep.clear();
    while (std::getline(in, line)) {
        if (line.size() >= IMSDLEN) break;
        Fields F(line);
        if (kind < 0) {
            kind = sniff_sentence(F);
            if (kind == SENTENCE_NONE) {
                printf("[WARNING] The current version does not support the current GPS format\n");
                break;
            }
        }
        double stamp, la, lo;
        const bool keep = parse_fix(kind, F, stamp, la, lo);
        logLat.push_back(la);
        logLon.push_back(lo);
        logT.push_back(stamp);
        logKeep.push_back(keep ? 1 : 0);
    }
    logLoaded = true;
    return true;
}

void GPSPro::window(double startTime, double endTime, std::vector<double> &lat, std::vector<double> &lon,
                    std::vector<double> &t) const
{
    // the scan of getGPRMCFormat (gps_process.cc:167-227) over the cached lines
    for (size_t i = 0; i < logT.size(); ++i) {
        const double stamp = logT[i];
        if (logKeep[i] && (long)stamp >= (long)(startTime - 1) && (long)stamp <= (long)(endTime + 1)) {
            lat.push_back(logLat[i]);
            lon.push_back(logLon[i]);
            t.push_back(stamp);
        }
        if (!(stamp < endTime + 1)) break;
    }
}

std::vector<std::vector<COORDXYZT> > GPSPro::GPSToENUBatch(const std::vector<std::vector<COORDXYZT> > &tracks)
{
    std::vector<std::vector<COORDXYZT> > out(tracks.size());
    if (tracks.empty() || !loadLog()) return out;
    std::vector<double> lat, lon, t;
    std::vector<COORDXYZT> slam;
    std::vector<int> goff(1, 0), soff(1, 0), which;
    for (size_t s = 0; s < tracks.size(); ++s) {
        if (tracks[s].empty()) continue;
        std::vector<double> la, lo, tt;
        window(tracks[s].front().t, tracks[s].back().t, la, lo, tt);
        if (tt.empty()) {
            printf("WARN: cannot find GPS information corresponding to slam track time,please check GPS original file.\n");
            continue;
        }
        gpsProcess(la, lo, tt);
        lat.insert(lat.end(), la.begin(), la.end());
        lon.insert(lon.end(), lo.begin(), lo.end());
        t.insert(t.end(), tt.begin(), tt.end());
        slam.insert(slam.end(), tracks[s].begin(), tracks[s].end());
        goff.push_back((int)t.size());
        soff.push_back((int)slam.size());
        which.push_back((int)s);
    }
    const int nseg = (int)which.size();
    if (nseg == 0) return out;
    std::vector<COORDXYZT> enu(slam.size());
    std::vector<int> kept(nseg, 0);
    check(gpscal_gps_to_enu_batched(default_ctx(), method == "UTM" ? GPSCAL_METHOD_UTM : GPSCAL_METHOD_GAUSS, type,
                                    lat.data(), lon.data(), t.data(), goff.data(), &slam[0].x, soff.data(), nseg,
                                    &enu[0].x, kept.data()),
          "gpscal_gps_to_enu_batched");
    for (int k = 0; k < nseg; ++k)
        out[which[k]].assign(enu.begin() + soff[k], enu.begin() + soff[k] + kept[k]);
    return out;
}

// ---------------------------------------------------------- colour segments

static std::string rgbColor(double w, double distance)
{
    // gps_process.cc:692-756; `a` is a float there
    w = w / distance;
    const double q = w / 0.667;
    w = (1.0 < q) ? 1.0 : q;
    const float a = (float)((1 - w) / 0.25);
    int r = 0, g = 0, b = 0;
    if (a >= 0.0f && a < 5.0f) {  // NaN / out of range leaves the reference's values uninitialised: black here
        const int x = (int)std::floor(a);
        const int y = (int)std::floor(255 * (a - x));
        switch (x) {
        case 0: r = 255; g = y; break;
        case 1: r = 255 - y; g = 255; break;
        case 2: g = 255; b = y; break;
        case 3: g = 255 - y; b = 255; break;
        case 4: b = 255; break;
        }
    }
    char buf[16];
    snprintf(buf, sizeof buf, "%02X%02X%02X", r & 255, g & 255, b & 255);
    return buf;
}

std::vector<std::pair<int, std::string> > GPSPro::segment(const std::vector<COORDXYZTW> &e)
{
    std::vector<std::pair<int, std::string> > out;
    if (e.empty()) return out;
    double dist = 0, wsum = e[0].w;
    for (size_t i = 1; i < e.size(); ++i) {
        const double dx = e[i].x - e[i - 1].x, dy = e[i].y - e[i - 1].y;
        wsum += e[i].w;
        dist += std::sqrt(dx * dx + dy * dy);
        if (dist > 50 /* SEGMENTLEN */ || i == e.size() - 1) {
            out.push_back(std::make_pair((int)i, rgbColor(wsum, dist)));
            dist = 0;
            wsum = 0;
        }
    }
    return out;
}

int GPSPro::ENUToGPS(std::vector<COORDXYZTW> enu, std::vector<std::pair<double, double> > &WGSBL,
                     std::vector<double> &altitude, std::vector<std::pair<int, std::string> > &segmentColor)
{
    if (enu.empty()) return 1;
    segmentColor = segment(enu);
    std::vector<double> ll(enu.size() * 2), alt(enu.size());
    check(gpscal_enu_to_wgs(default_ctx(), method == "UTM" ? GPSCAL_METHOD_UTM : GPSCAL_METHOD_GAUSS, type, &enu[0].x,
                            (int)enu.size(), ll.data(), alt.data()),
          "gpscal_enu_to_wgs");
    for (size_t i = 0; i < enu.size(); ++i) {
        WGSBL.push_back(std::make_pair(ll[2 * i], ll[2 * i + 1]));  // (longitude, latitude), gps_process.cc:1053
        altitude.push_back(alt[i]);
    }
    return 0;
}

std::vector<IMGPS> GPSPro::calibratedGPSMessage(const std::vector<COORDXYZTW> &calibrated)
{
    std::vector<IMGPS> track(calibrated.size());
    if (calibrated.empty()) return track;
    check(gpscal_imgps_message(default_ctx(), method == "UTM" ? GPSCAL_METHOD_UTM : GPSCAL_METHOD_GAUSS, type,
                               &calibrated[0].x, (int)calibrated.size(), &track[0].b),
          "gpscal_imgps_message");
    return track;
}

// --------------------------------------------------------------------- KML

std::vector<std::string> GPSPro::readKMLParameter()
{
    // The six strings of config/kml_config.xml in document order (the reference walks the
    // libxml2 tree, gps_process.cc:629-689); defaults = the shipped file.
    std::vector<std::string> cfg;
    std::ifstream in(kmlConfigPath.c_str());
    if (in.is_open()) {
        std::stringstream ss;
        ss << in.rdbuf();
        const std::string x = ss.str();
        static const char *tags[6] = {"styleid", "Linewidth", "styleUrl", "Lineextrude", "Linetessellate",
                                      "LinealtitudeMode"};
        for (int k = 0; k < 6; ++k) {
            const std::string open = std::string("<") + tags[k] + ">", close = std::string("</") + tags[k] + ">";
            const size_t a = x.find(open), b = x.find(close);
            if (a == std::string::npos || b == std::string::npos || b < a) {
                cfg.clear();
                break;
            }
            cfg.push_back(x.substr(a + open.size(), b - a - open.size()));
        }
    }
    if (cfg.size() != 6) {
        static const char *def[6] = {"GPScolor", "4", "#GPScolor", "1", "1", "absolute"};
        cfg.assign(def, def + 6);
    }
    return cfg;
}

namespace {
// "%.15g" of ofstream::precision(15) (gps_process.cc:769), via std::to_chars: the KML of a
// long run holds millions of numbers and iostream formatting dominated the output stage.
inline void put_num(std::string &o, double v)
{
    char b[40];
    auto r = std::to_chars(b, b + sizeof b, v, std::chars_format::general, IMDP);
    o.append(b, r.ptr);
}
inline void put_coord(std::string &o, double lon, double lat, double alt)
{
    put_num(o, lon);
    o.push_back(',');
    put_num(o, lat);
    o.push_back(',');
    put_num(o, alt);
    o.push_back('\n');
}
}  // namespace

int GPSPro::createKML(std::string name, std::vector<std::pair<double, double> > WGSBL, std::vector<double> altitude,
                      int flag, std::vector<std::pair<int, std::string> > segmentColor)
{
    const std::vector<std::string> cfg = readKMLParameter();
    FILE *fp = fopen(name.c_str(), "w");
    if (!fp) {
        printf("open %s error\n", name.c_str());
        return 1;
    }
    std::string o;
    o.reserve(WGSBL.size() * 48 + segmentColor.size() * 400 + 1024);
    const std::string placemark_head = "<Placemark>\n<styleUrl>" + cfg[2] + "</styleUrl>\n<LineString>\n<extrude>" + cfg[3] +
                                       "</extrude>\n<tessellate>" + cfg[4] + "</tessellate>\n<altitudeMode>" + cfg[5] +
                                       "</altitudeMode>\n<coordinates>\n";
    o += "<?xml version=\"1.0\" encoding=\"UTF-8\"?>\n<kml xmlns=\"http://www.opengis.net/kml/2.2\">\n<Document>\n";
    if (flag == 0) {
        o += "<name>original GPS</name>\n<description>original GPS</description>\n";
        o += "<Style id=\"" + cfg[0] + "\">\n<LineStyle>\n<color>7fFF00FF</color>\n<width>" + cfg[1] +
             "</width>\n</LineStyle>\n<PolyStyle>\n<color>7fFF00FF</color>\n</PolyStyle>\n</Style>\n";
        o += placemark_head;
        for (size_t i = 0; i < WGSBL.size() && i < altitude.size(); ++i)
            put_coord(o, WGSBL[i].first, WGSBL[i].second, altitude[i]);
        o += "</coordinates>\n</LineString></Placemark>\n";
    } else {
        o += "<name>calibrated GPS</name>\n<description>calibrated GPS</description>\n";
        size_t ic = 0;
        for (size_t k = 0; k < segmentColor.size(); ++k) {
            o += "<Style id=\"" + cfg[0] + "\">\n<LineStyle>\n<color>7f" + segmentColor[k].second + "</color>\n<width>" +
                 cfg[1] + "</width>\n</LineStyle>\n<PolyStyle>\n<color>" + segmentColor[k].second +
                 "</color>\n</PolyStyle>\n</Style>\n";
            o += placemark_head;
            // the reference's loop tests its config cursor (== 6) against altitude.size() and
            // stops BEFORE segment end: the last point of the track is never written (gps_process.cc:832)
            for (; ic < (size_t)segmentColor[k].first && 6 < altitude.size(); ++ic)
                put_coord(o, WGSBL[ic].first, WGSBL[ic].second, altitude[ic]);
            o += "</coordinates>\n</LineString></Placemark>\n";
        }
    }
    o += "</Document></kml>\n";
    fwrite(o.data(), 1, o.size(), fp);
    fclose(fp);
    return 0;
}

// ------------------------------------------------------------- GCJ-02 / BD-09 / JSON

namespace {
int mars(const char *who, int (*fn)(gpscal_ctx *, const double *, int, double *),
         const std::vector<std::pair<double, double> > &in, std::vector<std::pair<double, double> > &out)
{
    if (in.empty()) {
        printf("%s data NULL\n", who);  // gps_process.cc:528-532
        return -1;
    }
    static_assert(sizeof(std::pair<double, double>) == 16, "pairs are passed as packed doubles");
    std::vector<std::pair<double, double> > tmp(in.size());
    check(fn(default_ctx(), &in[0].first, (int)in.size(), &tmp[0].first), who);
    out.insert(out.end(), tmp.begin(), tmp.end());  // the reference push_backs onto the caller's vector
    return 0;
}
}  // namespace

int GPSPro::GPSToGCJ(std::vector<std::pair<double, double> > v, std::vector<std::pair<double, double> > &o)
{
    return mars("GPS", gpscal_gps_to_gcj, v, o);
}
int GPSPro::GCJToBD(std::vector<std::pair<double, double> > v, std::vector<std::pair<double, double> > &o)
{
    return mars("GCJ02", gpscal_gcj_to_bd, v, o);
}
int GPSPro::BDToGCJ(std::vector<std::pair<double, double> > v, std::vector<std::pair<double, double> > &o)
{
    return mars("BD09", gpscal_bd_to_gcj, v, o);
}

void GPSPro::createJSON(std::string fileName, std::vector<std::pair<double, double> > GPSValue, int flag,
                        std::vector<std::pair<int, std::string> > segmentColor)
{
    FILE *fp = fopen(fileName.c_str(), "w");
    if (!fp) {
        printf("ERROR: open %s error.\n", fileName.c_str());
        throw std::runtime_error("createJSON: cannot open " + fileName);  // the reference exit(0)s
    }
    // ofstream << double with precision(15) is printf("%.15g")
    size_t index = 0;
    if (flag == 0) {
        fputs("[{\"line\":[", fp);
        for (; index < GPSValue.size(); ++index) fprintf(fp, "[%.15g,%.15g],", GPSValue[index].first, GPSValue[index].second);
        fputs("],\"color\":\"FF00FF\"}]", fp);
    } else {
        fputs("[", fp);
        for (size_t c = 0; c < segmentColor.size(); ++c) {
            fputs("{\"line\":[", fp);
            for (; (long)index <= (long)segmentColor[c].first && index < GPSValue.size(); ++index)
                fprintf(fp, "[%.15g,%.15g],", GPSValue[index].first, GPSValue[index].second);
            fprintf(fp, "],\"color\":\"%s\"},", segmentColor[c].second.c_str());
        }
        fputs("]", fp);
    }
    fclose(fp);
    printf("finished map\n");
}
