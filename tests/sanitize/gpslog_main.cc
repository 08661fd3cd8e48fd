// Test driver for the sanitizer run of the NMEA log parser (gps_process.cc's host-side pieces: parseGPRMC and the
// dropout fill gpsProcess): parses every log named on the command line over its whole time span and prints one line
// per log.  No device call is made.
#include <cstdio>
#include <string>
#include <vector>

#include "gps_process.h"

int main(int argc, char **argv)
{
    for (int k = 1; k < argc; ++k) {
        std::vector<double> lat, lon, t;
        const int rc = GPSPro::parseGPRMC(argv[k], -1e300, 1e300, lat, lon, t);
        int rc2 = 0;
        if (rc == 0 && !t.empty()) rc2 = GPSPro::gpsProcess(lat, lon, t);
        double sum = 0;
        for (size_t i = 0; i < t.size(); ++i) sum += lat[i] + lon[i];
        std::printf("%s %d %d fixes %zu checksum %.9g\n", rc == 0 ? "OK" : "ERR", rc, rc2, t.size(), sum);
    }
    return 0;
}
