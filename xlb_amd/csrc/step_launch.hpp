// Type-erased launch descriptor + the per-(lattice, collision) dispatcher that each
// step_*.hip translation unit instantiates.
#pragma once
#include "common.hpp"
#include "step_kernel.hpp"

namespace xlb {

struct StepLaunch {
  const void* src;
  void* dst;
  const uint8_t* bc;
  const uint32_t* miss;
  const uint32_t* meta;  // two-step kernel only
  const uint32_t* tile_order;  // two-step kernel only
  const uint8_t* clean;        // two-step kernel only: per-block "no boundary cells" flags for THIS launch geometry, or nullptr
  const void* strips_src;      // two-step kernel only: strip buffers of the source / destination field (storage plane 0), or nullptr
  void* strips_dst;
  int strips;                  // two-step kernel only: 0 = none, 3 = phase A reads + phase B writes them, 2 = phase B writes them only
  int x_segments;              // two-step kernel only
  int x_cap;                   // two-step kernel only: thin first / last segment (planes), 0 = uniform cuts
  int tile_oy, tile_oz;        // two-step kernel only: periodic origin shift of the tiling (0 .. tile - 1)
  int tile_ty, tile_tz;        // two-step kernel only: tile of the (y, z) plane a block owns
  const uint8_t* tab_kind;
  const void* tab_values;  // compute dtype [256][27]
  const uint32_t* prof_keys;  // profile table of Zou-He / Regularized BCs (sorted storage cell indices), or nullptr
  const void* prof_vals;      // compute dtype [n_prof][3]
  int n_prof;
  const uint32_t* dist_keys;  // wall-distance table of HybridBC cells (sorted storage cell indices), or nullptr
  const float* dist_vals;     // [n_dist][q]
  int n_dist;
  unsigned long long ids_packed;
  unsigned kinds_packed;
  int n_bc;
  size_t plane_stride;
  int nx, ny, nz, halo;
  int x_begin, x_count;
  double omega;
  double force[3];
  double smag_cs;
  int compute_dtype, store_dtype;
  int vec;     // requested cells per thread (1, 2, 4); must divide nz
  int has_bc;  // 0: no BCs; 1: basic kinds; 2: + Zou-He / Regularized
  int edge_ext;  // has_bc == 2 but every extended-kind cell sits in plane 0 or nx - 1 (two-step kernel: interior planes only)
  int flags;   // bit 0: non-temporal stores
  int block_threads;  // 0 = default (256)
  int block_tz;       // threads along z per block, 0 = as many as fit
  int xcd_swizzle;
  int fast_bgk;   // two-step kernel only: 1 = tolerance-graded fast BGK body (opt-in)
  int fast_math;  // 1: tolerance-graded fast collision where one is built (fp64 KBC: cell.hpp kbc_fast); 0: bit-exact builds only
  hipStream_t stream;
};

// which (T, S, VEC) combinations exist: fp32 compute -> VEC in {1, 2, 4}; fp64 compute -> {1, 2}
inline int pick_vec(int compute_dtype, int nz, int requested, int collision) {
  // Measured on MI355X: one cell per thread wins everywhere — D3Q19 fp32 at 512^3 (profiles/r01/sweeps.md: 44-60 VGPRs ->
  // 8 waves/SIMD; 75.9 % of the HBM peak vs 74.5 % for VEC=2 and 72.9 % for VEC=4) and fp64 KBC at 384^3
  // (profiles/r02/d3q27_kbc_384_sweep.txt: 122 VGPRs / 4 waves vs 164 / 3; 0.684 vs 0.618 of peak), so "auto" (0) means 1.
  // Wider variants stay selectable through the "vec" option.
  const int vmax = (compute_dtype == XLBHIP_F32) ? 4 : 2;
  (void)collision;
  int v = requested > 0 ? requested : 1;
  if (v > vmax) v = vmax;
  if (v == 3) v = 2;
  if (nz % v != 0) v = 1;
  return v;
}

template <class L, class T, class S, int VEC, int COLL, int HASBC, int FLAGS>
int launch_typed(const StepLaunch& p) {
  StepArgs<T, S> a;
  a.src = static_cast<const S*>(p.src);
  a.dst = static_cast<S*>(p.dst);
  a.bc = p.bc;
  a.miss = p.miss;
  a.meta = nullptr;
  a.clean = nullptr;
  a.tile_order = nullptr;
  a.x_segments = 1;
  a.x_cap = 0;
  a.tile_oy = a.tile_oz = 0;
  a.bc_kind = p.tab_kind;
  a.bc_values = static_cast<const T*>(p.tab_values);
  a.prof_keys = p.prof_keys;
  a.prof_vals = static_cast<const T*>(p.prof_vals);
  a.n_prof = p.n_prof;
  a.dist_keys = p.dist_keys;
  a.dist_vals = p.dist_vals;
  a.n_dist = p.n_dist;
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
  a.nzq = p.nz / VEC;
  a.omega = static_cast<T>(p.omega);
  a.extra.force[0] = p.force[0];
  a.extra.force[1] = p.force[1];
  a.extra.force[2] = p.force[2];
  a.extra.smag_cs = p.smag_cs;
  const int threads = p.block_threads > 0 ? p.block_threads : 256;
  // threads along z: split a row into equal chunks of whole waves (nz = 384 -> 2 x 192, not 256 + 128)
  int tz = a.nzq;
  if (tz > threads) {
    const int chunks = (a.nzq + threads - 1) / threads;
    tz = (a.nzq + chunks - 1) / chunks;
    tz = (tz + 63) / 64 * 64;
    if (tz > threads) tz = threads;
  }
  if (p.block_tz > 0 && p.block_tz < tz) tz = p.block_tz;
  int ty = threads / tz;
  if (ty < 1) ty = 1;
  if (ty > p.ny) ty = p.ny;
  dim3 block(tz, ty, 1);
  dim3 grid((a.nzq + tz - 1) / tz, (p.ny + ty - 1) / ty, p.x_count);
  XLB_REQUIRE(threads <= (VEC == 1 ? XLB_LB1 : 256), "block_threads %d exceeds the kernel's launch bound", threads);
  a.xcd_swizzle = (p.xcd_swizzle && grid.x > 1 && grid.y % 8 == 0) ? 1 : 0;
  XLB_REQUIRE(grid.y <= 65535 && grid.z <= 65535, "grid too large: ny/ty=%u x_count=%u", grid.y, grid.z);
  hipLaunchKernelGGL((k_step<L, T, S, VEC, COLL, HASBC, FLAGS>), grid, block, 0, p.stream, a);
  XLB_HIP(hipGetLastError());
  return 0;
}

template <class L, class T, class S, int VEC, int COLL, int HASBC>
int launch_flags(const StepLaunch& p) {
#ifdef XLB_TUNE_VARIANTS
  switch (p.flags & 7) {
    case 0: return launch_typed<L, T, S, VEC, COLL, HASBC, 0>(p);
    case 1: return launch_typed<L, T, S, VEC, COLL, HASBC, 1>(p);
    case 3: return launch_typed<L, T, S, VEC, COLL, HASBC, 3>(p);
    case 7: return launch_typed<L, T, S, VEC, COLL, HASBC, 7>(p);
    default: XLB_FAIL("flag combination %d not instantiated", p.flags);
  }
#else
  return (p.flags & 2) ? launch_typed<L, T, S, VEC, COLL, HASBC, 3>(p) : launch_typed<L, T, S, VEC, COLL, HASBC, 1>(p);
#endif
}

template <class L, class T, class S, int VEC, int COLL>
int launch_vec(const StepLaunch& p) {
  if (p.has_bc == 2) return launch_flags<L, T, S, VEC, COLL, 2>(p);
  if (p.has_bc == 1) return launch_flags<L, T, S, VEC, COLL, 1>(p);
  return launch_flags<L, T, S, VEC, COLL, 0>(p);
}

template <class L, class T, class S, int COLL>
int launch_policy(const StepLaunch& p) {
  const int v = pick_vec(p.compute_dtype, p.nz, p.vec, COLL & 3);
  if constexpr (sizeof(T) == 4) {
    if (v == 4) return launch_vec<L, T, S, 4, COLL>(p);
  }
  if (v == 2) return launch_vec<L, T, S, 2, COLL>(p);
  return launch_vec<L, T, S, 1, COLL>(p);
}

template <class L, int COLL>
int launch_step(const StepLaunch& p) {
  const int c = p.compute_dtype, s = p.store_dtype;
  if (c == XLBHIP_F32 && s == XLBHIP_F32) return launch_policy<L, float, float, COLL>(p);
  if (c == XLBHIP_F32 && s == XLBHIP_F16) return launch_policy<L, float, _Float16, COLL>(p);
  if (c == XLBHIP_F64 && s == XLBHIP_F64) return launch_policy<L, double, double, COLL>(p);
  if (c == XLBHIP_F64 && s == XLBHIP_F32) return launch_policy<L, double, float, COLL>(p);
  if (c == XLBHIP_F64 && s == XLBHIP_F16) return launch_policy<L, double, _Float16, COLL>(p);
  XLB_FAIL("unsupported precision policy compute=%d store=%d", c, s);
}

// Extended collisions (Smagorinsky LES, exact-difference forcing): three policies only, to bound build time
template <class L, int COLL>
int launch_step_ext(const StepLaunch& p) {
  const int c = p.compute_dtype, s = p.store_dtype;
  if (c == XLBHIP_F32 && s == XLBHIP_F32) return launch_policy<L, float, float, COLL>(p);
  if (c == XLBHIP_F64 && s == XLBHIP_F64) return launch_policy<L, double, double, COLL>(p);
  if (c == XLBHIP_F64 && s == XLBHIP_F32) return launch_policy<L, double, float, COLL>(p);
  XLB_FAIL("SmagorinskyLESBGK / forced collisions are built for FP32FP32, FP64FP64 and FP64FP32 only (got compute=%d store=%d)", c, s);
}

// fp64-compute policies only (the fast KBC variant: cell.hpp kbc_fast)
template <class L, int COLL>
int launch_step_f64(const StepLaunch& p) {
  const int c = p.compute_dtype, s = p.store_dtype;
  // (a store type narrower than the compute type: the gamma reduction of the fast KBC collision runs in fp32, cell.hpp COLL_G32)
  constexpr int NARROW = COLL | (XLB_KBC_GAMMA32 ? COLL_G32 : 0);
  if (c == XLBHIP_F64 && s == XLBHIP_F64) return launch_policy<L, double, double, COLL>(p);
  if (c == XLBHIP_F64 && s == XLBHIP_F32) return launch_policy<L, double, float, NARROW>(p);
  if (c == XLBHIP_F64 && s == XLBHIP_F16) return launch_policy<L, double, _Float16, NARROW>(p);
  XLB_FAIL("fast fp64 variant asked for compute=%d store=%d", c, s);
}

// defined one per translation unit (step_<lattice>_<collision>.hip)
int launch_step_d3q27_kbc_fast64(const StepLaunch& p);
// two steps per pass (step2_kernel.hpp): f(t) in src -> f(t+2) in dst
bool step2_eligible(const StepLaunch& p, int lattice, int collision);
int launch_step2_d3q19_bgk(const StepLaunch& p);
int launch_step2_d3q19_bgk_strips(const StepLaunch& p);  // p.strips != 0 (step2_d3q19_strips.hip)

int launch_step2_d3q27_bgk(const StepLaunch& p);
int launch_step2_d3q27_kbc(const StepLaunch& p);
int step2_build_clean(const StepLaunch& p, uint8_t* out);
int step2_items(const StepLaunch& p);
int launch_step_d2q9_ext(const StepLaunch& p, int coll);
int launch_step_d3q19_ext(const StepLaunch& p, int coll);
int launch_step_d3q27_ext(const StepLaunch& p, int coll);
int launch_step_d2q9_bgk(const StepLaunch& p);
int launch_step_d2q9_kbc(const StepLaunch& p);
int launch_step_d3q19_bgk(const StepLaunch& p);
int launch_step_d3q27_bgk(const StepLaunch& p);
int launch_step_d3q27_kbc(const StepLaunch& p);

}  // namespace xlb
