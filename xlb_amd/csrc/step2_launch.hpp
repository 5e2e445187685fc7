// Launcher of the two-steps-per-pass kernel, shared by its per-lattice translation units (step2_*.hip).
#pragma once
#include "step2_kernel.hpp"
#include "step_launch.hpp"

namespace xlb {

#ifndef XLB_STEP2_PACKED_DEFAULT
#define XLB_STEP2_PACKED_DEFAULT true
#endif
// effective launch geometry (shared by the kernel launch and the clean-flag pass: both must map blocks alike)
inline int step2_eff_segments(const StepLaunch& p) { return (p.x_segments > 1 && p.x_count >= 8 * p.x_segments) ? p.x_segments : 1; }
// thin end segments need >= 3 segments and inner segments of at least 8 planes
inline int step2_eff_cap(const StepLaunch& p) {
  const int n = step2_eff_segments(p);
  return (p.x_cap > 0 && n >= 3 && p.x_count - 2 * p.x_cap >= 8 * (n - 2)) ? p.x_cap : 0;
}
inline int step2_eff_swizzle(const StepLaunch& p, unsigned tiles) { return (p.xcd_swizzle && tiles % 8u == 0u) ? 1 : 0; }

template <class L, int HASBC, int TY, int TZ, bool SLAB, bool PACKED, bool FAST, class T = float, int COLL = XLBHIP_BGK, int STRIPS = 0>
static int launch2f(const StepLaunch& p);

// fast_bgk = 1 (opt-in): the tolerance-graded fast BGK body (cell.hpp: bgk_fast).  Measured: 20 % fewer VALU instructions buy
// 2-4 % (profiles/r02/step2_sweeps.txt) — the kernel is bound by its real memory traffic, not by VALU issue — so the
// bit-exact body stays the default.
template <class L, int HASBC, int TY, int TZ, bool SLAB, bool PACKED = XLB_STEP2_PACKED_DEFAULT>
static int launch2(const StepLaunch& p) {
  return p.fast_bgk ? launch2f<L, HASBC, TY, TZ, SLAB, PACKED, true>(p) : launch2f<L, HASBC, TY, TZ, SLAB, PACKED, false>(p);
}

// T / COLL: compute type and collision (cell.hpp collide<>) of the instantiation; the store type is always fp32
template <class L, int HASBC, int TY, int TZ, bool SLAB, bool PACKED, bool FAST, class T, int COLL, int STRIPS>
static int launch2f(const StepLaunch& p) {
  static_assert(HASBC == 0 || (sizeof(T) == 4 && COLL == XLBHIP_BGK), "boundary-condition tables of the two-step kernel are fp32 / BGK");
  StepArgs<T, float> a;
  // SLAB: pointers advanced to interior plane 0 (the kernel addresses the ghost planes with negative indices)
  const size_t ghost = (size_t)p.halo * p.ny * p.nz;
  a.src = static_cast<const float*>(p.src) + ghost;
  a.dst = static_cast<float*>(p.dst) + ghost;
  a.bc = p.bc;
  a.miss = p.miss;
  a.meta = p.meta ? p.meta + ghost : nullptr;
  a.tile_order = p.tile_order;
  a.clean = (HASBC != 0) ? p.clean : nullptr;
  // (the strip buffer mirrors the field at 1 / 32: its interior plane 0 sits ghost / 32 elements in)
  a.strips_src = (STRIPS & 1) ? static_cast<const float*>(p.strips_src) + (ghost >> 5) : nullptr;
  a.strips_dst = (STRIPS & 2) ? static_cast<float*>(p.strips_dst) + (ghost >> 5) : nullptr;
  if constexpr (STRIPS != 0) {
    XLB_REQUIRE(((STRIPS & 1) == 0 || p.strips_src) && ((STRIPS & 2) == 0 || p.strips_dst), "two-step kernel: strip buffers missing");
    XLB_REQUIRE(p.plane_stride % 32 == 0 && ((size_t)p.ny * p.nz) % 32 == 0, "two-step kernel: strip buffers need 32-element aligned planes");
  }
  a.x_segments = step2_eff_segments(p);
  a.x_cap = step2_eff_cap(p);
  a.tile_oy = p.tile_oy;
  a.tile_oz = p.tile_oz;
  a.bc_kind = p.tab_kind;
  a.bc_values = static_cast<const T*>(p.tab_values);
  a.prof_keys = nullptr;  // (profile BCs are Zou-He / Regularized: single-step kernel)
  a.prof_vals = nullptr;
  a.n_prof = 0;
  a.dist_keys = nullptr;
  a.dist_vals = nullptr;
  a.n_dist = 0;
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
  a.omega = static_cast<T>(p.omega);
  a.extra.force[0] = a.extra.force[1] = a.extra.force[2] = 0.0;
  a.extra.smag_cs = p.smag_cs;
  const unsigned tiles = (unsigned)(p.ny / TY) * (unsigned)(p.nz / TZ);
  a.xcd_swizzle = step2_eff_swizzle(p, tiles);
  hipLaunchKernelGGL((k_step2<L, T, float, COLL, HASBC, TY, TZ, SLAB, PACKED, FAST, STRIPS>), dim3(tiles * (unsigned)a.x_segments), dim3(S2Geom<L, HASBC, TY, TZ, PACKED>::THREADS), 0, p.stream, a);
  XLB_HIP(hipGetLastError());
  return 0;
}

}  // namespace xlb
