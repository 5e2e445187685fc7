// Instantiates the two-steps-per-pass kernel (step2_kernel.hpp) for D3Q19 / BGK / FP32FP32.
#include "step2_kernel.hpp"
#include "step_launch.hpp"

namespace xlb {

bool step2_eligible(const StepLaunch& p, int lattice, int collision) {
  // do-nothing BCs would need a second redirected-load form (own cell, same population): not built, single-step kernel instead
  for (int i = 0; i < p.n_bc && i < 8; ++i)
    if (((p.kinds_packed >> (4 * i)) & 0xfu) == XLBHIP_BC_DO_NOTHING) return false;
  return lattice == XLBHIP_D3Q19 && collision == XLBHIP_BGK && p.compute_dtype == XLBHIP_F32 && p.store_dtype == XLBHIP_F32 &&
         p.halo == 0 && p.has_bc <= 1 && p.n_bc <= MAX_FAST_BCS &&
         p.plane_stride >= (size_t)p.nx * p.ny * p.nz + 64 /* idle-wave stores land in the padding */ && p.ny % 8 == 0 && p.nz % 64 == 0 && p.nx >= 1;
}

template <int HASBC, int TY, int TZ>
static int launch2(const StepLaunch& p) {
  StepArgs<float, float> a;
  a.src = static_cast<const float*>(p.src);
  a.dst = static_cast<float*>(p.dst);
  a.bc = p.bc;
  a.miss = p.miss;
  a.meta = p.meta;
  a.tile_order = (TY == 8 && TZ == 64) ? p.tile_order : nullptr;
  a.x_segments = (p.x_segments > 1 && p.nx >= 8 * p.x_segments) ? p.x_segments : 1;
  a.bc_kind = p.tab_kind;
  a.bc_values = static_cast<const float*>(p.tab_values);
  a.ids_packed = p.ids_packed;
  a.kinds_packed = p.kinds_packed;
  a.n_bc = p.n_bc;
  a.plane_stride = p.plane_stride;
  a.nx = p.nx;
  a.ny = p.ny;
  a.nz = p.nz;
  a.halo = 0;
  a.x_begin = 0;
  a.nzq = p.nz;
  a.omega = static_cast<float>(p.omega);
  a.extra.force[0] = a.extra.force[1] = a.extra.force[2] = 0.0;
  a.extra.smag_cs = p.smag_cs;
  const unsigned tiles = (unsigned)(p.ny / TY) * (unsigned)(p.nz / TZ);
  a.xcd_swizzle = (p.xcd_swizzle && tiles % 8u == 0u) ? 1 : 0;
  hipLaunchKernelGGL((k_step2<D3Q19, float, float, XLBHIP_BGK, HASBC, TY, TZ>), dim3(tiles * (unsigned)a.x_segments), dim3(S2Geom<TY, TZ>::THREADS), 0, p.stream, a);
  XLB_HIP(hipGetLastError());
  return 0;
}

// f(t) in p.src -> f(t+2) in p.dst
// block_tz selects the tile: 0 / 64 -> 8 x 64 (one block per CU), 32 -> 8 x 32 (two blocks per CU)
int launch_step2_d3q19_bgk(const StepLaunch& p) {
  if (p.block_tz == 32) return p.has_bc ? launch2<1, 8, 32>(p) : launch2<0, 8, 32>(p);
  if (p.block_tz == 16) return p.has_bc ? launch2<1, 16, 16>(p) : launch2<0, 16, 16>(p);
  return p.has_bc ? launch2<1, 8, 64>(p) : launch2<0, 8, 64>(p);
}

}  // namespace xlb
