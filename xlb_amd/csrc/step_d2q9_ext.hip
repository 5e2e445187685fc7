// Instantiates the fused step kernel for D2Q9 with the extended collisions
// (Smagorinsky LES BGK, exact-difference forcing): SURVEY.md section 8f rank 3.
#include "step_launch.hpp"

namespace xlb {
int launch_step_d2q9_ext(const StepLaunch& p, int coll) {
  switch (coll) {
    case XLBHIP_SMAGORINSKY_LES_BGK: return launch_step_ext<D2Q9, XLBHIP_SMAGORINSKY_LES_BGK>(p);
    case XLBHIP_SMAGORINSKY_LES_BGK | COLL_FORCED: return launch_step_ext<D2Q9, XLBHIP_SMAGORINSKY_LES_BGK | COLL_FORCED>(p);
    case XLBHIP_BGK | COLL_FORCED: return launch_step_ext<D2Q9, XLBHIP_BGK | COLL_FORCED>(p);
    case XLBHIP_KBC | COLL_FORCED: return launch_step_ext<D2Q9, XLBHIP_KBC | COLL_FORCED>(p);
  }
  XLB_FAIL("collision variant %d not built for D2Q9", coll);
}
}  // namespace xlb
