// Instantiates the two-steps-per-pass kernel (step2_kernel.hpp) for D3Q27 / BGK / FP32FP32, periodic boxes only: the
// lifetime-packed LDS ring holds 54 population-planes of an (8 x 64) tile = 142 560 B; the boundary-condition form (63) does not fit.
#include "step2_launch.hpp"

namespace xlb {
int launch_step2_d3q27_bgk(const StepLaunch& p) { return launch2<D3Q27, 0, 8, 64, false, true>(p); }
}  // namespace xlb
