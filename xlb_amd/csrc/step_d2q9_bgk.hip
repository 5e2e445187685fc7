// Instantiates the fused step kernel for D2Q9 / BGK (all precision policies).

#include "step_launch.hpp"

namespace xlb {
int launch_step_d2q9_bgk(const StepLaunch& p) { return launch_step<D2Q9, XLBHIP_BGK>(p); }
}  // namespace xlb
