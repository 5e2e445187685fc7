// Instantiates the two-steps-per-pass kernel (step2_kernel.hpp) for D3Q19 / BGK / FP32FP32.
#include "step2_kernel.hpp"
#include "step_launch.hpp"

namespace xlb {

bool step2_eligible(const StepLaunch& p, int lattice, int collision) {
  // do-nothing BCs would need a second redirected-load form (own cell, same population): not built; like the Zou-He
  // family they are fine on the x end planes, which the two-step kernel leaves to the single-step kernel (edge_ext)
  for (int i = 0; i < p.n_bc && i < 8; ++i)
    if (((p.kinds_packed >> (4 * i)) & 0xfu) == XLBHIP_BC_DO_NOTHING && !p.edge_ext) return false;
  return lattice == XLBHIP_D3Q19 && collision == XLBHIP_BGK && p.compute_dtype == XLBHIP_F32 && p.store_dtype == XLBHIP_F32 &&
         (p.halo == 0 || p.halo == 2) && (p.has_bc <= 1 || (p.edge_ext && p.halo == 0 && p.nx >= 16)) && p.n_bc <= MAX_FAST_BCS &&
         p.ny % 8 == 0 && p.nz % 64 == 0 && p.nx >= 4;
}

template <int HASBC, int TY, int TZ, bool SLAB>
static int launch2(const StepLaunch& p) {
  StepArgs<float, float> a;
  // SLAB: pointers advanced to interior plane 0 (the kernel addresses the ghost planes with negative indices)
  const size_t ghost = (size_t)p.halo * p.ny * p.nz;
  a.src = static_cast<const float*>(p.src) + ghost;
  a.dst = static_cast<float*>(p.dst) + ghost;
  a.bc = p.bc;
  a.miss = p.miss;
  a.meta = p.meta ? p.meta + ghost : nullptr;
  a.tile_order = p.tile_order;
  a.x_segments = (p.x_segments > 1 && p.x_count >= 8 * p.x_segments) ? p.x_segments : 1;
  a.bc_kind = p.tab_kind;
  a.bc_values = static_cast<const float*>(p.tab_values);
  a.prof_keys = nullptr;  // (profile BCs are Zou-He / Regularized: single-step kernel)
  a.prof_vals = nullptr;
  a.n_prof = 0;
  a.ids_packed = p.ids_packed;
  a.kinds_packed = p.kinds_packed;
  a.n_bc = p.n_bc;
  a.plane_stride = p.plane_stride;
  a.nx = p.nx;
  a.ny = p.ny;
  a.nz = p.nz;
  a.halo = p.halo;
  a.x_begin = p.x_begin;
  a.x_count = p.x_count;
  a.nzq = p.nz;
  a.omega = static_cast<float>(p.omega);
  a.extra.force[0] = a.extra.force[1] = a.extra.force[2] = 0.0;
  a.extra.smag_cs = p.smag_cs;
  const unsigned tiles = (unsigned)(p.ny / TY) * (unsigned)(p.nz / TZ);
  a.xcd_swizzle = (p.xcd_swizzle && tiles % 8u == 0u) ? 1 : 0;
  hipLaunchKernelGGL((k_step2<D3Q19, float, float, XLBHIP_BGK, HASBC, TY, TZ, SLAB>), dim3(tiles * (unsigned)a.x_segments), dim3(S2Geom<TY, TZ>::THREADS), 0, p.stream, a);
  XLB_HIP(hipGetLastError());
  return 0;
}

// f(t) in p.src -> f(t+2) in p.dst; (8 x 64) tiles, one block per CU (8 x 32 and 16 x 16 tiles with two blocks per CU
// were measured slower: profiles/r01/sweeps.md)
int launch_step2_d3q19_bgk(const StepLaunch& p) {
  if (p.halo) return p.has_bc ? launch2<1, 8, 64, true>(p) : launch2<0, 8, 64, true>(p);
  return p.has_bc ? launch2<1, 8, 64, false>(p) : launch2<0, 8, 64, false>(p);
}

}  // namespace xlb
