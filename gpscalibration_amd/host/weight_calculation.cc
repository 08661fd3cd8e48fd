// weight_calculation.cc -- WeightCoeCal over the C ABI (reference: weight_calculation.cc:4-78).
#include "weight_calculation.h"

using gpscal_host::check;
using gpscal_host::default_ctx;

int WeightCoeCal::ICPWeightCoeCal(std::vector<COORDXYZT> &slam, std::vector<double> &weightCoe)
{
    const size_t n = slam.size();
    if (n == 0) return 1;
    const size_t base = weightCoe.size();  // the reference push_backs onto whatever is there
    weightCoe.resize(base + n);
    check(gpscal_weights_speed(default_ctx(), &slam[0].x, (int)n, weightCoe.data() + base), "gpscal_weights_speed");
    return 1;
}

int WeightCoeCal::ICPWeightCoeCal(std::vector<COORDXYZT> &slam, std::vector<double> &weightCoe,
                                  std::vector<COORDXYZT> &enu, std::vector<COORDXYZT> &fit)
{
    const size_t n = slam.size();
    if (n == 0) return 1;
    if (enu.size() != n || fit.size() != n) check(GPSCAL_ESIZE, "ICPWeightCoeCal: track sizes differ");
    const size_t base = weightCoe.size();
    weightCoe.resize(base + n);
    check(gpscal_weights_irls(default_ctx(), &slam[0].x, &enu[0].x, &fit[0].x, (int)n, weightCoe.data() + base),
          "gpscal_weights_irls");
    return 1;
}
