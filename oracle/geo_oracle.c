/*
 * geo_oracle.c -- CPU restatement of GPSPro (NMEA ingest, dropout fill,
 * WGS84 <-> local projection, time interpolation, colour segments, KML).
 * TEST INFRASTRUCTURE ONLY (see gpscal_oracle.h).
 *   GP   = src/gpsCalibration/src/gps_calibration/gps_process.cc
 *   CM.h = src/gpsCalibration/include/gpsCalibration/common.h
 * Quirks reproduced as coded: PI truncated to 3.141592653589 (CM.h:17), the
 * A^6 term of the UTM northing sits outside N*tan (GP:899), band number taken
 * from the first fix only (GP:869-877), calibrated KML drops the last point and
 * tests `index` instead of `indexCoor` (GP:832).
 */
#include "gpscal_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define REF_PI 3.141592653589 /* CM.h:17 */
#define SEGMENT_LEN 50        /* GP:2 */
#define LINE_CAP 512          /* IMSDLEN, CM.h:19 */

/* WGSParameter, GP:1111-1118 */
static const double kA = 6378137;
static const double kB = 6356752.314;
static double E1(void) { return sqrt(pow(kA, 2) - pow(kB, 2)) / kA; }
static double E2(void) { return sqrt(pow(kA, 2) - pow(kB, 2)) / kB; }
static double CC(void) { return pow(kA, 2) / kB; }

/* ----------------------------------------------------------------- parse */

/* strtok(",") semantics: runs of delimiters collapse, empty fields vanish. */
static char *next_tok(char **cursor)
{
    char *p = *cursor;
    if (!p) return NULL;
    while (*p == ',') ++p;
    if (*p == '\0') {
        *cursor = NULL;
        return NULL;
    }
    char *start = p;
    while (*p && *p != ',') ++p;
    if (*p) {
        *p = '\0';
        *cursor = p + 1;
    } else {
        *cursor = NULL;
    }
    return start;
}

int orc_parse_gprmc(const char *text, size_t len, double t0, double t1,
                    double *lat, double *lon, double *t, int cap)
{
    int count = 0;
    size_t pos = 0;
    char buf[LINE_CAP];
    while (pos < len) {
        /* ifstream::getline(buf, 512): one line without the '\n' */
        size_t e = pos;
        while (e < len && text[e] != '\n') ++e;
        size_t l = e - pos;
        if (l > LINE_CAP - 1) break; /* getline would set failbit */
        memcpy(buf, text + pos, l);
        buf[l] = '\0';
        pos = e < len ? e + 1 : e;

        int column = 0;
        double latitude = 90, longitude = 180, stamp = 0; /* GP:169-170 */
        char *cur = buf;
        for (char *tok = next_tok(&cur); tok; tok = next_tok(&cur)) {
            ++column;
            if (column == 4 && strcmp("V", tok) == 0) break; /* GP:176-179 */
            switch (column) {
            case 1: stamp = atof(tok); break; /* GP:184-188 */
            case 5: {                         /* GP:189-194 */
                int d = (int)(atof(tok) / 100);
                latitude = d + (atof(tok) - d * 100) / 60.0;
                break;
            }
            case 6: if (strcmp("S", tok) == 0) latitude = 0 - latitude; break;
            case 7: { /* GP:203-208 */
                int d = (int)(atof(tok) / 100);
                longitude = d + (atof(tok) - d * 100) / 60.0;
                break;
            }
            case 8: if (strcmp("W", tok) == 0) longitude = 0 - longitude; break;
            default: break;
            }
        }
        if ((long)stamp >= (long)(t0 - 1) && (long)stamp <= (long)(t1 + 1)) {
            if (count >= cap) return -2; /* GP:222-226 */
            lat[count] = latitude;
            lon[count] = longitude;
            t[count] = stamp;
            ++count;
        }
        if (!(stamp < t1 + 1)) break; /* GP:227 */
    }
    return count;
}


/* GP:231-299 ($GPGGA) and GP:300-372 ($GPGLL): same line loop, other columns.  kind 1 = GPGGA,
 * 2 = GPGLL.  GPGLL pushes fixes with status V as they are (no sentinel test, GP:364). */
static int parse_gga_gll(int kind, const char *text, size_t len, double t0, double t1, double *lat, double *lon,
                         double *t, int cap)
{
    int count = 0;
    size_t pos = 0;
    char buf[LINE_CAP];
    while (pos < len) {
        size_t e = pos;
        while (e < len && text[e] != '\n') ++e;
        size_t l = e - pos;
        if (l > LINE_CAP - 1) break;
        memcpy(buf, text + pos, l);
        buf[l] = '\0';
        pos = e < len ? e + 1 : e;
        int column = 0;
        double latitude = 90, longitude = 180, stamp = 0;
        char *cur = buf;
        const int c_lat = kind == 1 ? 4 : 3;
        for (char *tok = next_tok(&cur); tok; tok = next_tok(&cur)) {
            ++column;
            if (kind == 1) { /* GP:246-253 */
                if (column == 5 && strcmp("N", tok) != 0 && strcmp("S", tok) != 0) break;
                if (column == 7 && strcmp("W", tok) != 0 && strcmp("E", tok) != 0) break;
            } else if (column == 8 && strcmp("V", tok) == 0) { /* GP:315-318 */
                break;
            }
            if (column == 1) stamp = atof(tok);
            else if (column == c_lat) {
                int d = (int)(atof(tok) / 100);
                latitude = d + (atof(tok) - d * 100) / 60.0;
            } else if (column == c_lat + 1) {
                if (strcmp("S", tok) == 0) latitude = 0 - latitude;
            } else if (column == c_lat + 2) {
                int d = (int)(atof(tok) / 100);
                longitude = d + (atof(tok) - d * 100) / 60.0;
            } else if (column == c_lat + 3) {
                if (strcmp("W", tok) == 0) longitude = 0 - longitude;
            }
        }
        int keep = (long)stamp >= (long)(t0 - 1) && (long)stamp <= (long)(t1 + 1);
        if (kind == 1) keep = keep && 90 != latitude && 180 != longitude; /* GP:293 */
        if (keep) {
            if (count >= cap) return -2;
            lat[count] = latitude;
            lon[count] = longitude;
            t[count] = stamp;
            ++count;
        }
        if (!(stamp < t1 + 1)) break;
    }
    return count;
}

int orc_parse_gps_log(const char *text, size_t len, double t0, double t1, double *lat, double *lon, double *t,
                      int cap)
{
    /* GP:128-152: the second comma-separated token of the FIRST line names the sentence */
    char buf[LINE_CAP];
    size_t e = 0;
    while (e < len && text[e] != '\n') ++e;
    if (e > LINE_CAP - 1) return -1;
    memcpy(buf, text, e);
    buf[e] = '\0';
    char *cur = buf;
    char *tok = next_tok(&cur);
    tok = tok ? next_tok(&cur) : NULL;
    if (!tok) return -1; /* the reference dereferences NULL here */
    if (strcmp("$GPRMC", tok) == 0) return orc_parse_gprmc(text, len, t0, t1, lat, lon, t, cap);
    if (strcmp("$GPGGA", tok) == 0) return parse_gga_gll(1, text, len, t0, t1, lat, lon, t, cap);
    if (strcmp("$GPGLL", tok) == 0) return parse_gga_gll(2, text, len, t0, t1, lat, lon, t, cap);
    return 0; /* "[WARNING] The current version does not support the current GPS format" */
}

/* ------------------------------------------------ GCJ-02 / BD-09 (GP:1127-1207) */
#define ORC_PI 3.141592653589 /* common.h:17 */
#define ORC_LONG_AXIS 6378245.0
#define ORC_SHORT_AXIS 6356863.0188
#define ORC_X_PI (3.1415926535897932384626 * 3000.0 / 180.0)

static double mars_lat(double x, double y)
{
    double ret = -100.0 + 2.0 * x + 3.0 * y + 0.2 * y * y + 0.1 * x * y + 0.2 * sqrt(fabs(x));
    ret += (20.0 * sin(6.0 * x * ORC_PI) + 20.0 * sin(2.0 * x * ORC_PI)) * 2.0 / 3.0;
    ret += (20.0 * sin(y * ORC_PI) + 40.0 * sin(y / 3.0 * ORC_PI)) * 2.0 / 3.0;
    ret += (160.0 * sin(y / 12.0 * ORC_PI) + 320 * sin(y * ORC_PI / 30.0)) * 2.0 / 3.0;
    return ret;
}

static double mars_lon(double x, double y)
{
    double ret = 300.0 + x + 2.0 * y + 0.1 * x * x + 0.1 * x * y + 0.1 * sqrt(fabs(x));
    ret += (20.0 * sin(6.0 * x * ORC_PI) + 20.0 * sin(2.0 * x * ORC_PI)) * 2.0 / 3.0;
    ret += (20.0 * sin(x * ORC_PI) + 40.0 * sin(x / 3.0 * ORC_PI)) * 2.0 / 3.0;
    ret += (150.0 * sin(x / 12.0 * ORC_PI) + 300.0 * sin(x / 30.0 * ORC_PI)) * 2.0 / 3.0;
    return ret;
}

/* GPSToGCJ (GP:526-545) on {longitude, latitude} pairs, as ENUToGPS emits them */
void orc_gps_to_gcj(const double *lonlat, int n, double *out)
{
    const double ee = (ORC_LONG_AXIS * ORC_LONG_AXIS - ORC_SHORT_AXIS * ORC_SHORT_AXIS) / (ORC_LONG_AXIS * ORC_LONG_AXIS);
    for (int i = 0; i < n; ++i) {
        const double lon = lonlat[2 * i], lat = lonlat[2 * i + 1];
        double glat, glon;
        if (lon < 72.004 || lon > 137.8347 || lat < 0.8293 || lat > 55.8271) { /* outOfChina, GP:1127-1138 */
            glat = lat;
            glon = lon;
        } else {
            double dLat = mars_lat(lon - 105.0, lat - 35.0), dLon = mars_lon(lon - 105.0, lat - 35.0);
            double radLat = lat / 180.0 * ORC_PI;
            double magic = sin(radLat);
            magic = 1 - ee * magic * magic;
            double sqrtMagic = sqrt(magic);
            dLat = (dLat * 180.0) / ((ORC_LONG_AXIS * (1 - ee)) / (magic * sqrtMagic) * ORC_PI);
            dLon = (dLon * 180.0) / (ORC_LONG_AXIS / sqrtMagic * cos(radLat) * ORC_PI);
            glat = lat + dLat;
            glon = lon + dLon;
        }
        out[2 * i] = glon;
        out[2 * i + 1] = glat;
    }
}

/* GCJToBD / BDToGCJ (GP:551-595, 1183-1207) */
void orc_gcj_to_bd(const double *lonlat, int n, double *out)
{
    for (int i = 0; i < n; ++i) {
        double x = lonlat[2 * i], y = lonlat[2 * i + 1];
        double z = sqrt(x * x + y * y) + 0.00002 * sin(y * ORC_X_PI);
        double theta = atan2(y, x) + 0.000003 * cos(x * ORC_X_PI);
        out[2 * i] = z * cos(theta) + 0.0065;
        out[2 * i + 1] = z * sin(theta) + 0.006;
    }
}

void orc_bd_to_gcj(const double *lonlat, int n, double *out)
{
    for (int i = 0; i < n; ++i) {
        double x = lonlat[2 * i] - 0.0065, y = lonlat[2 * i + 1] - 0.006;
        double z = sqrt(x * x + y * y) - 0.00002 * sin(y * ORC_X_PI);
        double theta = atan2(y, x) - 0.000003 * cos(x * ORC_X_PI);
        out[2 * i] = z * cos(theta);
        out[2 * i + 1] = z * sin(theta);
    }
}

/* createJSON (GP:1210-1250): ofstream with precision(15) = "%.15g" */
long orc_json(char *buf, size_t cap, const double *lonlat, int n, int flag, const int *seg_end, const uint32_t *rgb,
              int nseg)
{
    size_t at = 0;
#define JPUT(...)                                                              \
    do {                                                                       \
        int k_ = snprintf(at < cap ? buf + at : NULL, at < cap ? cap - at : 0, __VA_ARGS__); \
        at += (size_t)k_;                                                      \
    } while (0)
    int index = 0;
    if (flag == 0) {
        JPUT("[{\"line\":[");
        for (; index < n; ++index) JPUT("[%.15g,%.15g],", lonlat[2 * index], lonlat[2 * index + 1]);
        JPUT("],\"color\":\"FF00FF\"}]");
    } else {
        JPUT("[");
        for (int c = 0; c < nseg; ++c) {
            JPUT("{\"line\":[");
            for (; index <= seg_end[c]; ++index) JPUT("[%.15g,%.15g],", lonlat[2 * index], lonlat[2 * index + 1]);
            JPUT("],\"color\":\"%06X\"},", (unsigned)rgb[c]);
        }
        JPUT("]");
    }
#undef JPUT
    return (long)at;
}

/* -------------------------------------------------------------- gap fill */

int orc_gap_fill(double *lat, double *lon, const double *t, int n)
{
    /* GP:389-473 */
    int index = 0;
    while (index < n) {
        int begin = -2, end = -2, flag = 0;
        for (; index < n; ++index) {
            if (90 == lat[index] && 180 == lon[index]) {
                if (flag == 0) {
                    begin = index - 1;
                    flag = 1;
                }
            } else if (flag == 1) {
                end = index;
                break;
            }
        }
        if (begin == -2 && end == -2) return 0;
        if (begin == -1) {
            if (end == -2 || end == n - 1) return 1;
            double dT = t[end + 1] - t[end];
            double dB = (lat[end + 1] - lat[end]) / dT;
            double dL = (lon[end + 1] - lon[end]) / dT;
            for (int i = end - 1; i > begin; --i) {
                lat[i] = lat[i + 1] - dB * (t[i + 1] - t[i]);
                lon[i] = lon[i + 1] - dL * (t[i + 1] - t[i]);
            }
        } else {
            if (begin == 0 && end == -2) return 1;
            if (begin > 0 && end == -2) {
                double dT = t[begin] - t[begin - 1];
                double dB = (lat[begin] - lat[begin - 1]) / dT;
                double dL = (lon[begin] - lon[begin - 1]) / dT;
                for (int i = begin + 1; i < n; ++i) {
                    lat[i] = lat[i - 1] + dB * (t[i] - t[i - 1]);
                    lon[i] = lon[i - 1] + dL * (t[i] - t[i - 1]);
                }
            } else {
                double dT = t[end] - t[begin];
                double dB = (lat[end] - lat[begin]) / dT;
                double dL = (lon[end] - lon[begin]) / dT;
                for (int i = begin + 1; i < end; ++i) {
                    lat[i] = lat[i - 1] + dB * (t[i] - t[i - 1]);
                    lon[i] = lon[i - 1] + dL * (t[i] - t[i - 1]);
                }
            }
        }
    }
    return 0;
}

/* ------------------------------------------------------------ projection */

static double arc_length(double latitude)
{
    /* GP:38-56 */
    double e1 = E1();
    double m0 = kA * (1 - pow(e1, 2));
    double m2 = 3.0 / 2.0 * pow(e1, 2) * m0;
    double m4 = 5.0 / 4.0 * pow(e1, 2) * m2;
    double m6 = 7.0 / 6.0 * pow(e1, 2) * m4;
    double m8 = 9.0 / 8.0 * pow(e1, 2) * m6;
    double a0 = m0 + 1.0 / 2.0 * m2 + 3.0 / 8.0 * m4 + 5.0 / 16.0 * m6 +
                35.0 / 128.0 * m8;
    double a2 = 1.0 / 2.0 * m2 + 1.0 / 2.0 * m4 + 15.0 / 32.0 * m6 +
                7.0 / 16.0 * m8;
    double a4 = 1.0 / 8.0 * m4 + 3.0 / 16.0 * m6 + 7.0 / 32.0 * m8;
    double a6 = 1.0 / 32.0 * m6 + 1.0 / 16.0 * m8;
    double a8 = 1.0 / 128.0 * m8;
    double rB = latitude * REF_PI / 180.0;
    return a0 * rB - a2 / 2.0 * sin(2 * rB) + a4 / 4.0 * sin(4 * rB) -
           a6 / 6.0 * sin(6 * rB) + a8 / 8.0 * sin(8 * rB);
}

int orc_wgs_to_local(int method, int band_type, const double *lat,
                     const double *lon, int n, double *xy)
{
    if (n <= 0) return 1;
    const double e1 = E1(), e2 = E2();
    int band = 0;
    double meridian = 0;
    for (int i = 0; i < n; ++i) {
        if (band_type == 3) { /* GP:867-879 / 966-978 */
            if (band == 0) {
                band = (int)(lon[i] / 3);
                double tmp = lon[i] / 3;
                if (tmp - band > 0.5) band += 1;
            }
            meridian = 3 * band;
        } else if (band_type == 6) { /* GP:880-887 / 979-986 */
            if (band == 0) band = (int)lon[i] / 6 + 1;
            meridian = 6 * band - 6 / 2;
        }
        double x, y;
        if (method == 0) { /* UTM, GP:889-902 */
            double k0 = 0.9996;
            double rB = lat[i] * REF_PI / 180.0;
            double t = tan(rB) * tan(rB);
            double c = pow(e2, 2) * pow(cos(rB), 2);
            double A = (lon[i] - meridian) * REF_PI / 180.0 * cos(rB);
            double N = kA / sqrt(1 - e1 * e1 * sin(rB) * sin(rB));
            double M =
                kA * ((1 - pow(e1, 2) / 4.0 - 3.0 * pow(e1, 4) / 64.0 -
                       5.0 * pow(e1, 6) / 256.0) * rB -
                      (3.0 * pow(e1, 2) / 8.0 + 3.0 * pow(e1, 4) / 32.0 +
                       45.0 * pow(e1, 6) / 1024.0) * sin(2 * rB) +
                      (15.0 * pow(e1, 4) / 256.0 + 45.0 * pow(e1, 6) / 1024.0) *
                          sin(4 * rB) -
                      35.0 * pow(e1, 6) / 3072.0 * sin(6 * rB));
            x = k0 * (M +
                      N * tan(rB) *
                          (A * A / 2.0 +
                           (5 - t + 9 * c + 4 * c * c) * pow(A, 4) / 24.0) +
                      (61 - 58 * t + t * t + 600 * c - 330 * e2 * e2) *
                          pow(A, 6) / 720.0);
            y = k0 * N *
                    (A + (1 - t + c) * pow(A, 3) / 6.0 +
                     (5 - 18 * t + t * t + 72 * c - 58 * e2 * e2) * pow(A, 5) /
                         120.0) +
                500000;
        } else { /* Gauss-Krueger, GP:988-999 */
            double rB = lat[i] * REF_PI / 180.0;
            double t = tan(rB);
            double ng2 = pow(e2, 2) * pow(cos(rB), 2);
            double N = CC() / sqrt(1 + ng2);
            double m = cos(rB) * REF_PI / 180.0 * (lon[i] - meridian);
            double ml = arc_length(lat[i]);
            x = ml + N * t *
                         (1.0 / 2.0 * m * m +
                          1.0 / 24.0 * (5 - t * t + 9 * ng2 + 4 * ng2 * ng2) *
                              pow(m, 4) +
                          1.0 / 720.0 *
                              (61 - 58 * t * t + pow(t, 4) + 270 * ng2 -
                               330 * ng2 * t * t) *
                              pow(m, 6));
            y = N * (m + 1.0 / 6.0 * (1 - t * t + ng2) * pow(m, 3) +
                     1.0 / 120.0 *
                         (5 - 18 * t * t + pow(t, 4) + 14 * ng2 -
                          58 * ng2 * t * t) *
                         pow(m, 5)) +
                500000;
        }
        y += band * 10000000; /* GP:902 / 1001 */
        xy[2 * i + 0] = x;
        xy[2 * i + 1] = y;
    }
    return 0;
}

int orc_local_to_wgs(int method, int band_type, const double *enu, int n,
                     double *lonlat, double *alt)
{
    if (n <= 0) return 1;
    const double e1 = E1(), e2 = E2();
    for (int i = 0; i < n; ++i) {
        int band = (int)(enu[5 * i + 1] / 10000000); /* GP:1023 / 918 */
        double meridian = 0;
        if (band_type == 3) meridian = 3 * band;
        else if (band_type == 6) meridian = 6 * band - 6 / 2;
        double ly = enu[5 * i + 1] - band * 10000000 - 500000;
        double k0 = (method == 0) ? 0.9996 : 1.0;
        double X = (method == 0) ? enu[5 * i + 0] / k0 : enu[5 * i + 0];
        double fi = X / (kA * (1 - pow(e1, 2) / 4 - 3 * pow(e1, 4) / 64 -
                               5 * pow(e1, 6) / 256));
        double e = (1 - kB / kA) / (1 + kB / kA);
        double Bf = fi + (3 * e / 2 - 27 * pow(e, 3) / 32) * sin(2 * fi) +
                    (21 * e * e / 16 - 55 * pow(e, 4) / 32) * sin(4 * fi) +
                    151 * pow(e, 3) / 96 * sin(6 * fi);
        double Nf = kA / sqrt(1 - e1 * e1 * pow(sin(Bf), 2));
        double Rf = kA * (1 - e1 * e1) / pow((1 - e1 * e1 * pow(sin(Bf), 2)), 1.5);
        double Cf = e2 * e2 * cos(Bf) * cos(Bf);
        double Tf = tan(Bf) * tan(Bf);
        double latitude, longitude;
        if (method == 0) { /* GP:1043-1049 */
            double D = ly / (k0 * Nf);
            latitude = Bf - Nf * tan(Bf) / Rf *
                                (D * D / 2 -
                                 (5 + 3 * Tf + 10 * Cf - 4 * Cf * Cf - 9 * e2 * e2) *
                                     pow(D, 4) / 24.0 +
                                 (61 + 90 * Tf + 298 * Cf + 45 * Tf * Tf -
                                  252 * e2 * e2 - 3 * Cf * Cf) *
                                     pow(D, 6) / 720);
            longitude = meridian +
                        (1.0 / cos(Bf) *
                         (D - (1 + 2 * Tf + Cf) * pow(D, 3) / 6.0 +
                          (5 - 2 * Cf + 28 * Tf - 3 * Cf * Cf + 8 * e2 * e2 +
                           24 * Tf * Tf) *
                              pow(D, 5) / 120.0)) *
                            180 / REF_PI;
        } else { /* GP:937-943 */
            double D = ly / (Nf);
            latitude = Bf - Nf * tan(Bf) / Rf *
                                (D * D / 2 -
                                 (5 + 3 * Tf + Cf - 9 * Tf * Cf) * pow(D, 4) / 24 +
                                 (61 + 90 * Tf + 45 * Tf * Tf) * pow(D, 6) / 720);
            longitude = meridian +
                        (1.0 / cos(Bf) *
                         (D - (1 + 2 * Tf + Cf) * pow(D, 3) / 6 +
                          (5 + 28 * Tf + 6 * Cf + 8 * Tf * Cf + 24 * Tf * Tf) *
                              pow(D, 5) / 120)) *
                            180 / REF_PI;
        }
        latitude = latitude * 180 / REF_PI;
        lonlat[2 * i + 0] = longitude; /* GP:1053 (lon first) */
        lonlat[2 * i + 1] = latitude;
        alt[i] = enu[5 * i + 2];
    }
    return 0;
}

/* ---------------------------------------------------------- interpolation */

int orc_interpolate(const double *xy, const double *gt, int ngps,
                    const double *st, int nslam, double *out)
{
    /* GP:85-107 */
    int count = 0;
    for (int s = 0; s < ngps - 1; ++s) {
        double s1 = gt[s], s2 = gt[s + 1], s3 = s2 - s1;
        double x1 = xy[2 * s], x2 = xy[2 * (s + 1)];
        double y1 = xy[2 * s + 1], y2 = xy[2 * (s + 1) + 1];
        for (int r = count; r < nslam; ++r) {
            if (st[r] > s2) break;
            double c1 = (st[r] - s1) / s3;
            double c2 = 1.0 - c1;
            out[2 * count + 0] = c1 * x2 + c2 * x1;
            out[2 * count + 1] = c1 * y2 + c2 * y1;
            ++count;
        }
    }
    return count;
}

int orc_gps_to_enu(int method, int band_type, double *lat, double *lon,
                   const double *gt, int ngps, const double *slam, int nslam,
                   double *enu)
{
    if (ngps <= 0 || nslam <= 0) return -1;           /* GP:491-495 */
    orc_gap_fill(lat, lon, gt, ngps);                  /* GP:496 */
    double *xy = (double *)malloc(sizeof(double) * 2 * (size_t)ngps);
    double *st = (double *)malloc(sizeof(double) * (size_t)nslam);
    double *ixy = (double *)malloc(sizeof(double) * 2 * (size_t)nslam);
    orc_wgs_to_local(method, band_type, lat, lon, ngps, xy); /* GP:498-505 */
    for (int i = 0; i < nslam; ++i) st[i] = slam[4 * i + 3];
    int m = orc_interpolate(xy, gt, ngps, st, nslam, ixy); /* GP:506 */
    for (int i = 0; i < m; ++i) {                          /* GP:510-518 */
        enu[4 * i + 0] = ixy[2 * i + 0];
        enu[4 * i + 1] = ixy[2 * i + 1];
        enu[4 * i + 2] = slam[4 * i + 2];
        enu[4 * i + 3] = st[i];
    }
    free(xy);
    free(st);
    free(ixy);
    return m;
}

/* -------------------------------------------------------- colour segments */

static uint32_t rgb_colour(double w, double distance)
{
    /* GP:692-756.  `a` is a float in the reference. */
    w = w / distance;
    double q = w / 0.667;
    w = (1.0 < q) ? 1.0 : q; /* std::min(q, 1.0) */
    float a = (float)((1 - w) / 0.25);
    if (!(a >= 0.0f && a < 5.0f)) return 0; /* NaN/out of range: reference UB */
    int x = (int)floorf(a);
    int y = (int)floorf(255 * (a - x));
    int r = 0, g = 0, b = 0;
    switch (x) {
    case 0: r = 255; g = y; b = 0; break;
    case 1: r = 255 - y; g = 255; b = 0; break;
    case 2: r = 0; g = 255; b = y; break;
    case 3: r = 0; g = 255 - y; b = 255; break;
    case 4: r = 0; g = 0; b = 255; break;
    }
    return ((uint32_t)(r & 255) << 16) | ((uint32_t)(g & 255) << 8) |
           (uint32_t)(b & 255);
}

int orc_colour_segments(const double *enu, int n, int *seg_end, uint32_t *rgb,
                        int cap)
{
    /* GP:600-626 */
    if (n <= 0) return -1;
    int k = 0;
    double distance = 0, wsum = enu[4];
    for (int i = 1; i < n; ++i) {
        double dx = enu[5 * i + 0] - enu[5 * (i - 1) + 0];
        double dy = enu[5 * i + 1] - enu[5 * (i - 1) + 1];
        wsum += enu[5 * i + 4];
        distance += sqrt(dx * dx + dy * dy);
        if (distance > SEGMENT_LEN || i == n - 1) {
            if (k >= cap) return -2;
            seg_end[k] = i;
            rgb[k] = rgb_colour(wsum, distance);
            ++k;
            distance = 0;
            wsum = 0;
        }
    }
    return k;
}

/* -------------------------------------------------------------------- KML */

struct sbuf {
    char *p;
    size_t cap, len;
};
static void sb_puts(struct sbuf *s, const char *str)
{
    size_t l = strlen(str);
    if (s->p && s->len + l < s->cap) memcpy(s->p + s->len, str, l);
    s->len += l;
}
static void sb_coord(struct sbuf *s, double lon, double lat, double alt)
{
    /* ofstream with precision(15), default float format == %.15g (GP:769) */
    char line[128];
    snprintf(line, sizeof line, "%.15g,%.15g,%.15g\n", lon, lat, alt);
    sb_puts(s, line);
}

long orc_kml(char *buf, size_t cap, const double *lonlat, const double *alt,
             int n, int flag, const int *seg_end, const uint32_t *rgb, int nseg)
{
    /* config/kml_config.xml as shipped: GPScolor,4,#GPScolor,1,1,absolute */
    static const char *cfg[6] = {"GPScolor", "4", "#GPScolor",
                                 "1",        "1", "absolute"};
    struct sbuf s = {buf, cap, 0};
    char tmp[160];
    sb_puts(&s, "<?xml version=\"1.0\" encoding=\"UTF-8\"?>\n");
    sb_puts(&s, "<kml xmlns=\"http://www.opengis.net/kml/2.2\">\n");
    sb_puts(&s, "<Document>\n");
    if (flag == 0) { /* GP:774-801 */
        sb_puts(&s, "<name>original GPS</name>\n");
        sb_puts(&s, "<description>original GPS</description>\n");
        snprintf(tmp, sizeof tmp, "<Style id=\"%s\">\n", cfg[0]);
        sb_puts(&s, tmp);
        sb_puts(&s, "<LineStyle>\n<color>7fFF00FF</color>\n");
        snprintf(tmp, sizeof tmp, "<width>%s</width>\n", cfg[1]);
        sb_puts(&s, tmp);
        sb_puts(&s, "</LineStyle>\n<PolyStyle>\n<color>7fFF00FF</color>\n"
                    "</PolyStyle>\n</Style>\n<Placemark>\n");
        snprintf(tmp, sizeof tmp, "<styleUrl>%s</styleUrl>\n", cfg[2]);
        sb_puts(&s, tmp);
        sb_puts(&s, "<LineString>\n");
        snprintf(tmp, sizeof tmp,
                 "<extrude>%s</extrude>\n<tessellate>%s</tessellate>\n"
                 "<altitudeMode>%s</altitudeMode>\n",
                 cfg[3], cfg[4], cfg[5]);
        sb_puts(&s, tmp);
        sb_puts(&s, "<coordinates>\n");
        for (int i = 0; i < n; ++i)
            sb_coord(&s, lonlat[2 * i], lonlat[2 * i + 1], alt[i]);
        sb_puts(&s, "</coordinates>\n</LineString></Placemark>\n");
    } else { /* GP:805-839 */
        sb_puts(&s, "<name>calibrated GPS</name>\n");
        sb_puts(&s, "<description>calibrated GPS</description>\n");
        int ic = 0;
        for (int k = 0; k < nseg; ++k) {
            char hex[8];
            snprintf(hex, sizeof hex, "%06X", rgb[k] & 0xFFFFFFu);
            snprintf(tmp, sizeof tmp, "<Style id=\"%s\">\n", cfg[0]);
            sb_puts(&s, tmp);
            snprintf(tmp, sizeof tmp, "<LineStyle>\n<color>7f%s</color>\n", hex);
            sb_puts(&s, tmp);
            snprintf(tmp, sizeof tmp, "<width>%s</width>\n", cfg[1]);
            sb_puts(&s, tmp);
            snprintf(tmp, sizeof tmp,
                     "</LineStyle>\n<PolyStyle>\n<color>%s</color>\n"
                     "</PolyStyle>\n</Style>\n<Placemark>\n",
                     hex);
            sb_puts(&s, tmp);
            snprintf(tmp, sizeof tmp, "<styleUrl>%s</styleUrl>\n", cfg[2]);
            sb_puts(&s, tmp);
            sb_puts(&s, "<LineString>\n");
            snprintf(tmp, sizeof tmp,
                     "<extrude>%s</extrude>\n<tessellate>%s</tessellate>\n"
                     "<altitudeMode>%s</altitudeMode>\n",
                     cfg[3], cfg[4], cfg[5]);
            sb_puts(&s, tmp);
            sb_puts(&s, "<coordinates>\n");
            /* GP:832: `index` is 6 here (config cursor), not indexCoor */
            for (; ic < seg_end[k] && 6 < n; ++ic)
                sb_coord(&s, lonlat[2 * ic], lonlat[2 * ic + 1], alt[ic]);
            sb_puts(&s, "</coordinates>\n</LineString></Placemark>\n");
        }
    }
    sb_puts(&s, "</Document></kml>\n");
    if (s.p && s.len < s.cap) s.p[s.len] = '\0';
    return (long)s.len;
}
