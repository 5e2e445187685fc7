// The strip-buffer instantiations of the two-steps-per-pass kernel (step2_kernel.hpp, "Strip buffers") for D3Q19 / BGK / FP32FP32,
// (8 x 64) tiles, bit-exact body: phase A's halo columns come from the source field's strips (STRIPS & 1), phase B writes the
// destination's (STRIPS & 2).  A pass whose source has no valid strips (a single step or an upload wrote it) only writes them; on
// slab-decomposed fields the two edge launches — whose phase A pulls from ghost planes, which carry no strips — only write, too.
#include "step2_launch.hpp"

namespace xlb {

int launch_step2_d3q19_bgk_strips(const StepLaunch& p) {
  XLB_REQUIRE(p.tile_ty == 8 && p.tile_tz == 64 && !p.fast_bgk && (p.strips == 2 || p.strips == 3 || p.strips == 4), "strip buffers: (8 x 64) tiles, bit-exact body");
  if (p.strips == 4) {  // row-aligned lanes in both bodies of the BC kernel, no strips ("fuse2_rowmap": measurement option)
    XLB_REQUIRE(p.has_bc, "fuse2_rowmap: the BC-free kernel has row-aligned lanes anyway");
    return p.halo ? launch2f<D3Q19, 1, 8, 64, true, true, false, float, XLBHIP_BGK, 4>(p) : launch2f<D3Q19, 1, 8, 64, false, true, false, float, XLBHIP_BGK, 4>(p);
  }
  if (p.halo) {
    if (p.strips == 3) return p.has_bc ? launch2f<D3Q19, 1, 8, 64, true, true, false, float, XLBHIP_BGK, 3>(p) : launch2f<D3Q19, 0, 8, 64, true, true, false, float, XLBHIP_BGK, 3>(p);
    return p.has_bc ? launch2f<D3Q19, 1, 8, 64, true, true, false, float, XLBHIP_BGK, 2>(p) : launch2f<D3Q19, 0, 8, 64, true, true, false, float, XLBHIP_BGK, 2>(p);
  }
  if (p.strips == 3) return p.has_bc ? launch2f<D3Q19, 1, 8, 64, false, true, false, float, XLBHIP_BGK, 3>(p) : launch2f<D3Q19, 0, 8, 64, false, true, false, float, XLBHIP_BGK, 3>(p);
  return p.has_bc ? launch2f<D3Q19, 1, 8, 64, false, true, false, float, XLBHIP_BGK, 2>(p) : launch2f<D3Q19, 0, 8, 64, false, true, false, float, XLBHIP_BGK, 2>(p);
}

}  // namespace xlb
