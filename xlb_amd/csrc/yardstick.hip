// Measurement only: a copy of a population field with the LAUNCH SHAPE of the two-step kernel — bench.py's second yardstick
// (roofline.pattern_copy_ms / frac_of_pattern_copy).  Not part of any step; bench.py's kernel_source_hash leaves this file out.
//
// One 704-thread block per CU (150 KB of dynamic LDS keep a second one away, as k_step2's ring of f(t+1) does), (8 x 64) tiles of the
// (y, z) plane marching along x, wave j < 8 pulling row j of its tile from the Q populations of a plane and storing the previous
// plane's row, the plane's stores and pulls separated by a workgroup barrier as in k_step2 (step2_kernel.hpp).  No arithmetic, no LDS
// traffic: what the memory system gives such a launch (profiles/r03/cu_read_rate.md; tools/cu_read_rate.hip is the stand-alone form
// with the variants — unsynchronised, other tile shapes, interleaved populations, two blocks per CU).
#include <type_traits>

#include "common.hpp"

namespace xlb {

template <int Q>
__global__ void __launch_bounds__(704) k_copy_tiles(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t plane_stride, int planes, int ny, int nz) {
  extern __shared__ uint32_t copy_tiles_lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned tiles_z = (unsigned)nz / 64u;
  const int ty0 = (int)(blockIdx.x / tiles_z) * 8, tz0 = (int)(blockIdx.x % tiles_z) * 64;
  const bool active = wave < 8;
  const size_t cell = (size_t)(ty0 + (wave & 7)) * nz + tz0 + lane, pc = (size_t)ny * nz;
  uint32_t v[Q];
  if (threadIdx.x == 0xffffff) copy_tiles_lds[0] = 0;  // (never: keeps the allocation)
  if (active) {
#pragma unroll
    for (int l = 0; l < Q; ++l) v[l] = src[(size_t)l * plane_stride + cell];
  }
  for (int x = 1; x <= planes; ++x) {
    if (active) {
#pragma unroll
      for (int l = 0; l < Q; ++l) __builtin_nontemporal_store(v[l], dst + (size_t)l * plane_stride + (size_t)(x - 1) * pc + cell);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (active && x < planes) {
#pragma unroll
      for (int l = 0; l < Q; ++l) v[l] = src[(size_t)l * plane_stride + (size_t)x * pc + cell];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

}  // namespace xlb

using namespace xlb;

extern "C" int xlbhip_field_copy_tiles(xlbhip_field* dst, const xlbhip_field* src) {
  XLB_REQUIRE(dst && src, "null field");
  XLB_REQUIRE(dst->alloc_bytes == src->alloc_bytes && dst != src && dst->plane_stride == src->plane_stride && dst->nx == src->nx && dst->ny == src->ny &&
                  dst->nz == src->nz && dst->halo == src->halo && dst->card == src->card,
              "field_copy_tiles: layouts differ");
  XLB_REQUIRE(dtype_size(src->dtype) == 4 && dtype_size(dst->dtype) == 4, "field_copy_tiles: 4-byte elements only");
  XLB_REQUIRE(src->ny % 8 == 0 && src->nz % 64 == 0, "field_copy_tiles: (8 x 64) tiles need ny %% 8 == 0 and nz %% 64 == 0");
  XLB_REQUIRE(src->card == 9 || src->card == 19 || src->card == 27, "field_copy_tiles: 9, 19 or 27 populations");
  if (int rc = xlbhip_field_touch(dst)) return rc;  // (new contents: caches keyed on the old ones are stale)
  hipStream_t st = dst->ctx->stream;
  const unsigned blocks = (unsigned)(src->ny / 8) * (unsigned)(src->nz / 64);
  const int planes = src->nx + 2 * src->halo;
  const size_t lds = 150 * 1024;
  auto go = [&](auto qc) -> int {
    constexpr int Q = decltype(qc)::value;
    XLB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_copy_tiles<Q>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k_copy_tiles<Q>, dim3(blocks), dim3(704), lds, st, static_cast<const uint32_t*>(src->data), static_cast<uint32_t*>(dst->data),
                       src->plane_stride, planes, src->ny, src->nz);
    XLB_HIP(hipGetLastError());
    return 0;
  };
  if (src->card == 9) return go(std::integral_constant<int, 9>{});
  if (src->card == 19) return go(std::integral_constant<int, 19>{});
  return go(std::integral_constant<int, 27>{});
}
