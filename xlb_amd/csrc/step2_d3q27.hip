// Instantiates the two-steps-per-pass kernel (step2_kernel.hpp) for D3Q27 with fp32 store: the lifetime-packed LDS ring holds 54
// population-planes of an (8 x 64) tile = 142 560 B; the boundary-condition form (63 planes, BGK) runs (8 x 48) tiles = 126 000 B.
// BGK in fp32; KBC in fp32 and in fp64 (BASELINE configs[4]: FP64FP32 — f(t+1) sits in LDS in the store type, as it would in memory).
#include "step2_launch.hpp"

namespace xlb {
// BGK: (8 x 64) tiles without boundary conditions; with the basic ones (round 3) the 63-plane BC ring needs (8 x 48) tiles: 126 KB + meta words
int launch_step2_d3q27_bgk(const StepLaunch& p) {
  if (p.has_bc) {
    XLB_REQUIRE(p.tile_tz == 48 && !p.fast_bgk, "two-step kernel: D3Q27 BGK with boundary conditions runs (8 x 48) tiles, bit-exact body");
    return launch2f<D3Q27, 1, 8, 48, false, true, false>(p);
  }
  return launch2<D3Q27, 0, 8, 64, false, true>(p);
}
// fast: the tolerance-graded fp64 collision (cell.hpp kbc_fast; exact_math = 0, the default)
int launch_step2_d3q27_kbc(const StepLaunch& p) {
  if (p.compute_dtype == XLBHIP_F32) return launch2f<D3Q27, 0, 8, 48, false, true, false, float, XLBHIP_KBC>(p);
  XLB_REQUIRE(p.fast_math, "two-step kernel: the bit-exact fp64 KBC collision is not built (step2_eligible)");
  return launch2f<D3Q27, 0, 8, 48, false, true, false, double, XLBHIP_KBC | COLL_FAST | (XLB_KBC_GAMMA32 ? COLL_G32 : 0)>(p);  // (fp32 store: gamma reduction in fp32)
}
}  // namespace xlb
