// weight_calculation.h -- same class and overloads as the reference's
// include/gpsCalibration/weight_calculation.h:9-19; the arithmetic runs on the GPU
// (gpscal_weights_speed / gpscal_weights_irls).
#ifndef GPSCAL_HOST_WEIGHT_CALCULATION_H
#define GPSCAL_HOST_WEIGHT_CALCULATION_H
#include "common.h"

#define SPEED (2.2)
#define DELTA 0.01

class WeightCoeCal {
public:
    // weight coefficient from slam speed (appends to weightCoe, like the reference's push_back)
    int ICPWeightCoeCal(std::vector<COORDXYZT> &SLAMTrackTmp, std::vector<double> &weightCoe);
    // weight coefficient from the difference between ENU original and rotated SLAM
    int ICPWeightCoeCal(std::vector<COORDXYZT> &SLAMTrackTmp, std::vector<double> &weightCoe,
                        std::vector<COORDXYZT> &ENUOriTMP, std::vector<COORDXYZT> &SLAMRotatedTrackTmp);
};
#endif
