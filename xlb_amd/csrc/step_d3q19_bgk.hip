// Instantiates the fused step kernel for D3Q19 / BGK (all precision policies).
#define XLB_TUNE_VARIANTS 1
#include "step_launch.hpp"

namespace xlb {
int launch_step_d3q19_bgk(const StepLaunch& p) { return launch_step<D3Q19, XLBHIP_BGK>(p); }
}  // namespace xlb
