// Instantiates the fused step kernel for D3Q27 / KBC with the tolerance-graded fast collision (cell.hpp: kbc_fast)
// for the fp64-compute precision policies — BASELINE configs[4].
#include "step_launch.hpp"

namespace xlb {
int launch_step_d3q27_kbc_fast64(const StepLaunch& p) { return launch_step_f64<D3Q27, XLBHIP_KBC | COLL_FAST>(p); }
}  // namespace xlb
