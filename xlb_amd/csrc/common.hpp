// Shared host-side definitions for libxlbhip (context, field, error channel).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <string>

#include "../../include/xlbhip.h"

namespace xlb {

void set_error(const char* fmt, ...);

#define XLB_FAIL(...)            \
  do {                           \
    ::xlb::set_error(__VA_ARGS__); \
    return 1;                    \
  } while (0)

#define XLB_HIP(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) XLB_FAIL("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

#define XLB_REQUIRE(cond, ...) \
  do {                         \
    if (!(cond)) XLB_FAIL(__VA_ARGS__); \
  } while (0)

inline size_t dtype_size(int dt) {
  switch (dt) {
    case XLBHIP_F64: return 8;
    case XLBHIP_F32: return 4;
    case XLBHIP_F16: return 2;
    case XLBHIP_U8: return 1;
    case XLBHIP_BOOL: return 1;
    case XLBHIP_MISSING: return 4;
  }
  return 0;
}

struct Comm;  // comm.cpp

}  // namespace xlb

struct xlbhip_ctx {
  int device = 0;
  hipStream_t stream = nullptr;       // compute stream
  hipStream_t comm_stream = nullptr;  // halo exchange stream
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  hipEvent_t ev_edge = nullptr, ev_halo = nullptr;
  // telemetry of the slab protocol: how long the compute stream sat at the halo event after the interior launch
  // (a lost overlap shows up here).  A ring of timing-event pairs, harvested when a slot comes round again.
  static const int WAIT_RING = 32;
  hipEvent_t ev_w0[WAIT_RING] = {}, ev_w1[WAIT_RING] = {};
  bool wait_used[WAIT_RING] = {};
  int wait_head = 0;
  double halo_wait_ms = 0.0;
  int64_t halo_waits = 0;
  std::map<std::string, int64_t> opts;
  xlb::Comm* comm = nullptr;
  int compute_units = 0;
};

// Device layout of a field (DESIGN.md "Data layout in HBM"):
//   element (l, x, y, z) lives at data[l * plane_stride + ((x + halo) * ny + y) * nz + z]
// plane_stride >= (nx + 2 halo) * ny * nz, padded so that population planes do not alias
// in the HBM channel hash.  XLBHIP_MISSING fields store ONE plane of u32 bit-sets.
struct xlbhip_field {
  xlbhip_ctx* ctx = nullptr;
  int card = 0, nx = 0, ny = 0, nz = 0, halo = 0, dtype = 0;
  size_t plane_stride = 0;  // elements
  size_t planes = 0;        // stored planes (card, or 1 for XLBHIP_MISSING)
  size_t alloc_bytes = 0;
  void* base = nullptr;  // hipMalloc pointer
  void* data = nullptr;  // base + guard
  uint64_t version = 0;  // bumped by every C-ABI call that writes the field (caches keyed on a mask's contents)
  // strip buffer of a population field (two-step kernel, step2_kernel.hpp "Strip buffers"): 1 / 32 of the field, allocated on first
  // use; valid for the contents `strips_version` and the z origin `strips_oz` of the tiling only
  void* strips = nullptr;
  uint64_t strips_version = 0;
  int strips_oz = -1;
  size_t cells() const { return (size_t)nx * ny * nz; }
  size_t cells_with_halo() const { return (size_t)(nx + 2 * halo) * ny * nz; }
};

namespace xlb {
int64_t opt(const xlbhip_ctx* c, const char* key, int64_t dflt);
int lattice_q(int lattice);
int lattice_d(int lattice);
}  // namespace xlb
