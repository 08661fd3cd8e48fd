// track_calibration.h -- same public interface as the reference's
// include/gpsCalibration/track_calibration.h:12-23.  The fit itself is one launch of
// gpscal_track_fit (HIP, float64); no Eigen.
#ifndef GPSCAL_HOST_TRACK_CALIBRATION_H
#define GPSCAL_HOST_TRACK_CALIBRATION_H
#include "common.h"

class trackCalibration {
public:
    // throws std::runtime_error on size mismatch (the reference exit(1)s, track_calibration.cc:46-50)
    trackCalibration(std::vector<COORDXYZT> &SLAMTrackTmp, std::vector<COORDXYZT> &ENUTrackTmp,
                     std::vector<double> weightCoeTmp);
    int doICP();
    int doCalibration(std::vector<COORDXYZT> &calENUTrack);
    // extras the reference keeps private: final 4x4 (row-major) and the rotated track
    const double *transform() const { return T_; }
    const std::vector<double> &rotated() const { return rotated_; }

private:
    std::vector<COORDXYZT> slam_, enu_;
    std::vector<double> w_;
    double T_[16];
    std::vector<double> rotated_;     // n x 3
    std::vector<COORDXYZT> calibrated_;
    bool fitted_ = false;
};
#endif
