// The strip-buffer instantiations of the two-steps-per-pass kernel (step2_kernel.hpp, "Strip buffers") for D3Q19 / BGK / FP32FP32,
// (8 x 64) tiles, bit-exact body: phase A's halo columns come from the source field's strips (STRIPS & 1), phase B writes the
// destination's (STRIPS & 2).  Slab-decomposed fields: the interior launch reads and writes them, the two edge launches — whose
// phase A pulls from ghost planes, which carry no strips — only write.
#include "step2_launch.hpp"

namespace xlb {

int launch_step2_d3q19_bgk_strips(const StepLaunch& p) {
  XLB_REQUIRE(p.tile_ty == 8 && p.tile_tz == 64 && !p.fast_bgk && (p.strips == 2 || p.strips == 3), "strip buffers: (8 x 64) tiles, bit-exact body");
  if (p.halo) {
    if (p.strips == 3) return p.has_bc ? launch2f<D3Q19, 1, 8, 64, true, true, false, float, XLBHIP_BGK, 3>(p) : launch2f<D3Q19, 0, 8, 64, true, true, false, float, XLBHIP_BGK, 3>(p);
    return p.has_bc ? launch2f<D3Q19, 1, 8, 64, true, true, false, float, XLBHIP_BGK, 2>(p) : launch2f<D3Q19, 0, 8, 64, true, true, false, float, XLBHIP_BGK, 2>(p);
  }
  XLB_REQUIRE(p.strips == 3, "strip buffers: fields without ghost planes read and write them");
  return p.has_bc ? launch2f<D3Q19, 1, 8, 64, false, true, false, float, XLBHIP_BGK, 3>(p) : launch2f<D3Q19, 0, 8, 64, false, true, false, float, XLBHIP_BGK, 3>(p);
}

// strips of the planes [x_begin, x_begin + x_count) of a field from the field itself (after anything but the two-step kernel wrote it)
int build_strips(const StepLaunch& p, const void* field, void* strips, int x_begin, int x_count) {
  XLB_REQUIRE(field && strips && p.nz % 64 == 0 && x_count >= 1, "build_strips: bad arguments");
  unsigned czp = 0, czm = 0;
  for (int l = 0; l < D3Q19::Q; ++l) {
    czp |= (D3Q19::c(2, l) == 1 ? 1u : 0u) << l;
    czm |= (D3Q19::c(2, l) == -1 ? 1u : 0u) << l;
  }
  const size_t ghost = (size_t)p.halo * p.ny * p.nz;
  const size_t n = (size_t)D3Q19::Q * x_count * (p.nz / 64) * p.ny;
  const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, (size_t)1 << 20);
  hipLaunchKernelGGL(k_build_strips<float>, dim3(blocks), dim3(256), 0, p.stream, static_cast<const float*>(field) + ghost,
                     static_cast<float*>(strips) + (ghost >> 5), p.plane_stride, (int)D3Q19::Q, x_begin, x_count, p.ny, p.nz, p.tile_oz, czp, czm);
  XLB_HIP(hipGetLastError());
  return 0;
}

}  // namespace xlb
