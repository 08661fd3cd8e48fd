// track_calibration.cc -- trackCalibration over the C ABI (reference:
// track_calibration.cc:4-37; the fit of :97-201, 366-545, 591-689 is one HIP launch).
#include "track_calibration.h"

#include <stdexcept>

using gpscal_host::check;
using gpscal_host::default_ctx;

trackCalibration::trackCalibration(std::vector<COORDXYZT> &slam, std::vector<COORDXYZT> &enu,
                                   std::vector<double> w)
    : slam_(slam), enu_(enu), w_(std::move(w))
{
    // dataInitial's size check (track_calibration.cc:46-50) -- an exception, not exit(1)
    if (slam_.size() != enu_.size() || slam_.size() != w_.size() || slam_.empty())
        throw std::runtime_error("there's something wrong in icp data no, please check it out");
    for (int k = 0; k < 16; ++k) T_[k] = (k % 5 == 0) ? 1.0 : 0.0;
}

int trackCalibration::doICP()
{
    const int n = (int)slam_.size();
    rotated_.resize((size_t)n * 3);
    calibrated_.resize(n);
    check(gpscal_track_fit(default_ctx(), &slam_[0].x, &enu_[0].x, w_.data(), n, T_, rotated_.data(), &calibrated_[0].x),
          "gpscal_track_fit");
    fitted_ = true;
    return 1;
}

int trackCalibration::doCalibration(std::vector<COORDXYZT> &calENUTrack)
{
    if (!fitted_) doICP();
    // the reference appends (push_back, track_calibration.cc:685)
    calENUTrack.insert(calENUTrack.end(), calibrated_.begin(), calibrated_.end());
    return 1;
}
