// Instantiates the two-steps-per-pass kernel (step2_kernel.hpp) for D3Q19 / BGK / FP32FP32 and holds the eligibility rule.
#include "step2_launch.hpp"

namespace xlb {

// the (TY x TZ) tile a launch uses: fuse2_tile 0 -> 8 x 64 (default), 2 -> 16 x 32
bool step2_eligible(const StepLaunch& p, int lattice, int collision) {
  // do-nothing BCs would need a second redirected-load form (own cell, same population): not built; like the Zou-He
  // family they are fine on the x end planes, which the two-step kernel leaves to the single-step kernel (edge_ext)
  for (int i = 0; i < p.n_bc && i < 8; ++i)
    if (((p.kinds_packed >> (4 * i)) & 0xfu) == XLBHIP_BC_DO_NOTHING && !p.edge_ext) return false;
  if (!(p.store_dtype == XLBHIP_F32 && p.nx >= 4 && p.ny % p.tile_ty == 0 && p.nz % p.tile_tz == 0)) return false;
  if (lattice == XLBHIP_D3Q27) {  // without ghost planes only; BGK also with the basic boundary conditions, on (8 x 48) tiles (63 population-planes)
    if (!(p.halo == 0 && p.tile_ty == 8)) return false;
    if (collision == XLBHIP_BGK)
      return p.compute_dtype == XLBHIP_F32 && (p.has_bc == 0 ? p.tile_tz == 64 : (p.has_bc == 1 && p.tile_tz == 48 && p.n_bc <= MAX_FAST_BCS));
    if (p.has_bc != 0) return false;
    // KBC: fp32 and fp64 compute, (8 x 48) tiles (api.hip make_launch)
    // (the bit-exact fp64 collision needs 940 B of scratch there — 12 ms per step: it stays on the single-step kernel)
    return collision == XLBHIP_KBC && (p.compute_dtype == XLBHIP_F32 || (p.compute_dtype == XLBHIP_F64 && p.fast_math)) && p.tile_tz == 48;
  }
  return collision == XLBHIP_BGK && p.compute_dtype == XLBHIP_F32 && lattice == XLBHIP_D3Q19 && (p.halo == 0 || p.halo == 2) && (p.has_bc <= 1 || (p.edge_ext && p.halo == 0 && p.nx >= 16)) &&
         p.n_bc <= MAX_FAST_BCS;
}

// f(t) in p.src -> f(t+2) in p.dst; (8 x 64) tiles, one block per CU (8 x 32 and 16 x 16 tiles with two blocks per CU
// were measured slower: profiles/r01/sweeps.md; 16 x 32: profiles/r02/sweeps.md)
int launch_step2_d3q19_bgk(const StepLaunch& p) {
#ifdef XLB_TUNE_VARIANTS
  // (16 x 32) tiles: 8 % SLOWER than (8 x 64) on the periodic box and the cavity at 512^3 (profiles/r02/sweeps.md); kept
  // compilable for re-measurement, not built by default
  if (p.tile_ty == 16 && p.tile_tz == 32) {
    if (p.halo) return p.has_bc ? launch2<D3Q19, 1, 16, 32, true>(p) : launch2<D3Q19, 0, 16, 32, true>(p);
    return p.has_bc ? launch2<D3Q19, 1, 16, 32, false>(p) : launch2<D3Q19, 0, 16, 32, false>(p);
  }
  // smaller tiles on the lifetime-packed ring, several blocks per CU (periodic boxes without ghost planes only)
  if (p.tile_ty == 8 && p.tile_tz == 32 && !p.halo && !p.has_bc) return launch2<D3Q19, 0, 8, 32, false, true>(p);
  if (p.tile_ty == 4 && p.tile_tz == 64 && !p.halo && !p.has_bc) return launch2<D3Q19, 0, 4, 64, false, true>(p);
#else
  XLB_REQUIRE(p.tile_ty == 8 && p.tile_tz == 64, "two-step kernel: only the (8 x 64) tile is built (fuse2_tile=2 needs -DXLB_TUNE_VARIANTS)");
#endif
  if (p.halo) return p.has_bc ? launch2<D3Q19, 1, 8, 64, true>(p) : launch2<D3Q19, 0, 8, 64, true>(p);
  return p.has_bc ? launch2<D3Q19, 1, 8, 64, false>(p) : launch2<D3Q19, 0, 8, 64, false>(p);
}

// per-block "no boundary cell in this work item" flags for the launch geometry of p (n = tiles x effective segments bytes)
int step2_build_clean(const StepLaunch& p, uint8_t* out) {
  XLB_REQUIRE(p.meta && out && p.tile_ty == 8 && (p.tile_tz == 64 || (p.tile_tz == 48 && p.halo == 0)), "clean flags: (8 x 64) / (8 x 48) tiles with meta words only");
  const size_t ghost = (size_t)p.halo * p.ny * p.nz;
  const unsigned tiles = (unsigned)(p.ny / 8) * (unsigned)(p.nz / p.tile_tz);
  const int segs = step2_eff_segments(p), swz = step2_eff_swizzle(p, tiles);
  if (p.tile_tz == 48)
    hipLaunchKernelGGL((k_step2_clean<8, 48, false>), dim3(tiles * (unsigned)segs), dim3(256), 0, p.stream, p.meta, p.tile_order, swz, segs, step2_eff_cap(p), p.x_begin,
                       p.x_count, p.nx, p.ny, p.nz, p.tile_oy, p.tile_oz, out);
  else if (p.halo)
    hipLaunchKernelGGL((k_step2_clean<8, 64, true>), dim3(tiles * (unsigned)segs), dim3(256), 0, p.stream, p.meta + ghost, p.tile_order, swz, segs, step2_eff_cap(p), p.x_begin,
                       p.x_count, p.nx, p.ny, p.nz, p.tile_oy, p.tile_oz, out);
  else
    hipLaunchKernelGGL((k_step2_clean<8, 64, false>), dim3(tiles * (unsigned)segs), dim3(256), 0, p.stream, p.meta, p.tile_order, swz, segs, step2_eff_cap(p), p.x_begin,
                       p.x_count, p.nx, p.ny, p.nz, p.tile_oy, p.tile_oz, out);
  XLB_HIP(hipGetLastError());
  return 0;
}
int step2_items(const StepLaunch& p) { return (p.ny / p.tile_ty) * (p.nz / p.tile_tz) * step2_eff_segments(p); }

}  // namespace xlb

#ifdef XLB_STEP2_TRACE
// debug builds only (tools/step2_phase_trace.py): the phase stamps of the last launch
extern "C" int xlbhip_debug_step2_trace(unsigned long long* out, int n) {
  constexpr int N = xlb::TRACE_PLANES * xlb::TRACE_WAVES * xlb::TRACE_EVENTS;
  if (n < N) return N;
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(xlb::g_step2_trace), sizeof(unsigned long long) * N) == hipSuccess ? 0 : -1;
}
#endif
