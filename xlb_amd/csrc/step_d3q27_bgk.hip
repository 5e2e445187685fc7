// Instantiates the fused step kernel for D3Q27 / BGK (all precision policies).

#include "step_launch.hpp"

namespace xlb {
int launch_step_d3q27_bgk(const StepLaunch& p) { return launch_step<D3Q27, XLBHIP_BGK>(p); }
}  // namespace xlb
