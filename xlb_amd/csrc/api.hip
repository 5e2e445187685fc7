// libxlbhip: C-ABI entry points (context, fields, whole-field operators, masker, stepper).
// See include/xlbhip.h for the contract and the reference methods each call replaces.
#include <algorithm>
#include <array>
#include <cstring>
#include <map>
#include <utility>
#include <vector>

#include "comm.hpp"
#include "common.hpp"
#include "ops_kernels.hpp"
#include "step_launch.hpp"

namespace xlb {

static thread_local std::string g_err;

void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

int64_t opt(const xlbhip_ctx* c, const char* key, int64_t dflt) {
  auto it = c->opts.find(key);
  return it == c->opts.end() ? dflt : it->second;
}

int lattice_q(int lattice) { return lattice == XLBHIP_D2Q9 ? 9 : (lattice == XLBHIP_D3Q19 ? 19 : (lattice == XLBHIP_D3Q27 ? 27 : 0)); }
int lattice_d(int lattice) { return lattice == XLBHIP_D2Q9 ? 2 : 3; }

static const size_t GUARD_BYTES = 256;

// contents version of a field: globally unique, so a cache keyed on (field address, version) cannot be fooled by a
// new field that recycles a freed one's address
static uint64_t g_version = 0;
static void touch(xlbhip_field* f) { f->version = ++g_version; }

static FieldView view(const xlbhip_field* f) {
  FieldView v;
  v.data = f ? f->data : nullptr;
  v.plane_stride = f ? f->plane_stride : 0;
  v.dtype = f ? f->dtype : 0;
  v.halo = f ? f->halo : 0;
  return v;
}
static Dims dims(const xlbhip_field* f) { return Dims{f->nx, f->ny, f->nz}; }
static bool same_grid(const xlbhip_field* a, const xlbhip_field* b) { return a->nx == b->nx && a->ny == b->ny && a->nz == b->nz; }
static bool is_float(int dt) { return dt == XLBHIP_F64 || dt == XLBHIP_F32 || dt == XLBHIP_F16; }
static unsigned blocks_for(size_t n, int threads = 256) { return (unsigned)((n + threads - 1) / threads); }
// for the grid-stride kernels (k_copy, k_fill): HIP refuses launches of 2^32 threads or more
static unsigned blocks_capped(size_t n, int threads = 256) { return (unsigned)std::min<size_t>((n + threads - 1) / threads, (size_t)1 << 23); }

template <class F>
static int by_lattice(int lattice, F&& f) {
  switch (lattice) {
    case XLBHIP_D2Q9: return f(D2Q9{});
    case XLBHIP_D3Q19: return f(D3Q19{});
    case XLBHIP_D3Q27: return f(D3Q27{});
  }
  XLB_FAIL("unknown lattice id %d", lattice);
}

template <class L>
static void fill_lattice(int* d, int* q, int32_t* c, double* w, int32_t* op, int32_t* ccv) {
  *d = L::D;
  *q = L::Q;
  for (int l = 0; l < L::Q; ++l) {
    for (int a = 0; a < 3; ++a) c[a * L::Q + l] = L::c(a, l);
    w[l] = L::w(l);
    op[l] = opp<L>(l);
    for (int k = 0; k < 6; ++k) ccv[l * 6 + k] = k < n_pi<L>() ? cc<L>(l, k) : 0;
  }
}

}  // namespace xlb

using namespace xlb;

template <class L, class T>
static int collide_launch(xlbhip_ctx* c, int coll, const xlbhip_field* f, const xlbhip_field* feq, xlbhip_field* fo, double omega) {
  const size_t n = f->cells();
  const T cs = (T)((double)opt(c, "smagorinsky_coef_e6", 170000) * 1e-6);
  if (coll == XLBHIP_BGK) {
    hipLaunchKernelGGL((k_collide<L, T, XLBHIP_BGK>), blocks_for(n), 256, 0, c->stream, view(f), view(feq), view(fo), dims(f), (T)omega, cs);
  } else if (coll == XLBHIP_SMAGORINSKY_LES_BGK) {
    hipLaunchKernelGGL((k_collide<L, T, XLBHIP_SMAGORINSKY_LES_BGK>), blocks_for(n), 256, 0, c->stream, view(f), view(feq), view(fo), dims(f),
                       (T)omega, cs);
  } else {
    if constexpr (L::ID == XLBHIP_D3Q19) {
      XLB_FAIL("Velocity set not supported: D3Q19 has no KBC (reference kbc.py:65-66)");
    } else {
      hipLaunchKernelGGL((k_collide<L, T, XLBHIP_KBC>), blocks_for(n), 256, 0, c->stream, view(f), view(feq), view(fo), dims(f), (T)omega, cs);
    }
  }
  XLB_HIP(hipGetLastError());
  return 0;
}

template <int MODE>
static int velocity_gradient_launch(xlbhip_ctx* c, const xlbhip_field* u, const xlbhip_field* bcm, xlbhip_field* out_a, xlbhip_field* out_b,
                                    const char* what) {
  XLB_REQUIRE(c && u && bcm && out_a && out_b, "%s: null argument", what);
  XLB_REQUIRE(u->card == 3 && (u->dtype == XLBHIP_F32 || u->dtype == XLBHIP_F64), "%s: u must be a (3, nx, ny, nz) fp32 / fp64 field", what);
  XLB_REQUIRE(u->halo == 0, "%s: fields with ghost planes are not supported (single-rank post-processing, like the reference)", what);
  XLB_REQUIRE(bcm->dtype == XLBHIP_U8 && bcm->card == 1 && same_grid(bcm, u) && bcm->halo == 0, "%s: bad bc_mask field", what);
  XLB_REQUIRE(out_a->card == (MODE == 0 ? 3 : 1) && out_b->card == 1 && is_float(out_a->dtype) && is_float(out_b->dtype) &&
                  same_grid(out_a, u) && same_grid(out_b, u) && out_a->halo == 0 && out_b->halo == 0,
              "%s: bad output fields", what);
  const size_t n = u->cells();
  if (u->dtype == XLBHIP_F32)
    hipLaunchKernelGGL((k_velocity_gradient<float, MODE>), blocks_for(n), 256, 0, c->stream, view(u), view(bcm), view(out_a), view(out_b), dims(u));
  else
    hipLaunchKernelGGL((k_velocity_gradient<double, MODE>), blocks_for(n), 256, 0, c->stream, view(u), view(bcm), view(out_a), view(out_b), dims(u));
  XLB_HIP(hipGetLastError());
  return 0;
}

extern "C" {

const char* xlbhip_last_error(void) { return g_err.c_str(); }

int xlbhip_create(int device, xlbhip_ctx** out) {
  XLB_REQUIRE(out, "out is null");
  int n = 0;
  XLB_HIP(hipGetDeviceCount(&n));
  XLB_REQUIRE(n > 0, "no HIP device visible");
  XLB_REQUIRE(device >= 0 && device < n, "device %d out of range (have %d)", device, n);
  XLB_HIP(hipSetDevice(device));
  xlbhip_ctx* c = new xlbhip_ctx();
  c->device = device;
  XLB_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  // the halo stream outranks the compute stream: its small kernels / copies must not queue behind the interior launch
  int prio_low = 0, prio_high = 0;
  XLB_HIP(hipDeviceGetStreamPriorityRange(&prio_low, &prio_high));
  XLB_HIP(hipStreamCreateWithPriority(&c->comm_stream, hipStreamNonBlocking, prio_high));
  XLB_HIP(hipEventCreate(&c->ev_a));
  XLB_HIP(hipEventCreate(&c->ev_b));
  XLB_HIP(hipEventCreateWithFlags(&c->ev_edge, hipEventDisableTiming));
  XLB_HIP(hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
  hipDeviceProp_t p;
  XLB_HIP(hipGetDeviceProperties(&p, device));
  c->compute_units = p.multiProcessorCount;
  // defaults of the tuning knobs
  c->opts["vec"] = 0;              // cells per thread: 0 = auto (1), or 1 / 2 / 4
  c->opts["nt_store"] = 1;         // non-temporal stores in the fused kernel
  c->opts["nt_load"] = 1;          // 1: nt loads for the c_z == 0 directions (+1.8 %), 3: for all (slower; D3Q19 BGK tuning variants only)
  c->opts["plane_pad_bytes"] = 4352;  // de-alias the q population planes
  c->opts["block_threads"] = 256;
  c->opts["xcd_swizzle"] = 0;      // blocks of one row on one XCD (see step_kernel.hpp)
  c->opts["block_tz"] = 0;         // threads along z per block (0 = a whole row when it fits)
  c->opts["overlap"] = 1;          // halo exchange overlapped with the interior kernel
  c->opts["smagorinsky_coef_e6"] = 170000;  // Smagorinsky constant x 1e6 for the STAND-ALONE collision operator (0.17)
  c->opts["fuse2"] = 1;            // xlbhip_run: two steps per pass (step2_kernel.hpp): 0 never, 1 where eligible and the grid fills the chip, 2 wherever eligible
  c->opts["fuse2_xseg"] = 0;       // x segments per tile column in the two-step kernel (0 = auto: 4, fewer for short domains)
  c->opts["fuse2_lpt"] = 1;        // two-step kernel with BCs: hull tiles first (longest-processing-time-first dispatch)
  c->opts["fuse2_xcd"] = 1;        // compact tile patch per XCD in the two-step kernel
  c->opts["fuse2_shift"] = 1;      // two-step kernel with BCs: tiling shifted by half a tile (both walls of an axis in ONE wrapping tile row)
  c->opts["fuse2_xcap"] = 8;       // with fuse2_clean: planes of the thin first / last x-segment (0 = uniform cuts)
  c->opts["fuse2_clean"] = 1;      // two-step kernel with BCs: work items without boundary cells run the BC-free body (same launch)
  c->opts["fuse2_tile"] = 0;       // tile of the two-step kernel: 0 = 8 x 64, 2 = 16 x 32 (D3Q19)
  c->opts["fuse2_cus"] = 0;        // CUs the chip-filling rule of fuse2 = 1 assumes (0 = the device's; tests of the rule)
  c->opts["fuse2_strips"] = 1;     // two-step kernel (D3Q19): halo columns of phase A from the fields' strip buffers (step2_kernel.hpp): 0 never,
                                   // 1 = for steppers with boundary conditions (where they pay), 2 = always
  c->opts["fuse2_rowmap"] = 0;     // measurement: 1 = the BC kernel's bodies with row-aligned lanes and NO strip buffers (fuse2_strips must be 0)
  c->opts["fast_bgk"] = 0;         // two-step kernel: 1 = tolerance-graded fast BGK body (rounding-level differences; +2-4 %)
  c->opts["exact_math"] = 0;       // 1: bit-exact builds only (fp64 KBC otherwise uses the tolerance-graded fast collision, cell.hpp kbc_fast)
  c->opts["external_halo"] = 0;    // 1: the caller fills the ghost planes before every step (host-staged transports, tests)
  c->opts["halo_telemetry"] = 1;   // slab runs: time the compute stream's wait for the halo event (xlbhip_comm_stats)
  c->opts["ipc_copy"] = 1;         // ipc transport: 1 = one pull kernel per exchange (8 blocks per plane, no LDS: hidden behind the interior launch),
                                   // 0 = plane-sized hipMemcpyAsync pulls (copy engines; ~40 us per call on a shared device: profiles/r03/ipc_transport.md)
  c->opts["ipc_timeout_ms"] = 180000;  // ipc transport: bound of every device-side / host-side wait for a neighbour (ranks reach their first
                                       // exchange as unevenly as their set-up takes: callers should put a barrier in front of the first run)
  c->opts["halo_skip"] = 0;        // 1: MEASUREMENT ONLY — the slab protocol's launches without moving any ghost plane (wrong results)
  c->opts["comm_self_test"] = 0;   // 1: a one-rank RCCL communicator also runs the all-reduce of comm_all_min (tests)
  *out = c;
  return 0;
}

int xlbhip_destroy(xlbhip_ctx* c) {
  if (!c) return 0;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  xlbhip_comm_destroy(c);
  (void)hipEventDestroy(c->ev_a);
  (void)hipEventDestroy(c->ev_b);
  (void)hipEventDestroy(c->ev_edge);
  (void)hipEventDestroy(c->ev_halo);
  for (int i = 0; i < xlbhip_ctx::WAIT_RING; ++i) {
    if (c->ev_w0[i]) (void)hipEventDestroy(c->ev_w0[i]);
    if (c->ev_w1[i]) (void)hipEventDestroy(c->ev_w1[i]);
  }
  (void)hipStreamDestroy(c->stream);
  (void)hipStreamDestroy(c->comm_stream);
  delete c;
  return 0;
}

int xlbhip_sync(xlbhip_ctx* c) {
  XLB_REQUIRE(c, "ctx is null");
  XLB_HIP(hipStreamSynchronize(c->comm_stream));
  XLB_HIP(hipStreamSynchronize(c->stream));
  return comm_check(c);
}

int xlbhip_device_info(xlbhip_ctx* c, char* name, int name_len, int* cus, uint64_t* hbm) {
  XLB_REQUIRE(c, "ctx is null");
  hipDeviceProp_t p;
  XLB_HIP(hipGetDeviceProperties(&p, c->device));
  if (name && name_len > 0) {
    snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName);
  }
  if (cus) *cus = p.multiProcessorCount;
  if (hbm) *hbm = p.totalGlobalMem;
  return 0;
}

int xlbhip_set_option(xlbhip_ctx* c, const char* key, int64_t value) {
  XLB_REQUIRE(c && key, "null argument");
  auto it = c->opts.find(key);
  XLB_REQUIRE(it != c->opts.end(), "unknown option '%s'", key);
  it->second = value;
  return 0;
}
int xlbhip_get_option(xlbhip_ctx* c, const char* key, int64_t* value) {
  XLB_REQUIRE(c && key && value, "null argument");
  auto it = c->opts.find(key);
  XLB_REQUIRE(it != c->opts.end(), "unknown option '%s'", key);
  *value = it->second;
  return 0;
}

int xlbhip_lattice_info(int lattice, int* d, int* q, int32_t* c, double* w, int32_t* op, int32_t* ccv) {
  XLB_REQUIRE(d && q && c && w && op && ccv, "null output");
  return by_lattice(lattice, [&](auto L) {
    fill_lattice<decltype(L)>(d, q, c, w, op, ccv);
    return 0;
  });
}

// ---- fields ---------------------------------------------------------------------------
int xlbhip_field_create(xlbhip_ctx* c, int card, int nx, int ny, int nz, int dtype, int halo, double fill, xlbhip_field** out) {
  XLB_REQUIRE(c && out, "null argument");
  XLB_REQUIRE(card >= 1 && nx >= 1 && ny >= 1 && nz >= 1, "bad field shape (%d,%d,%d,%d)", card, nx, ny, nz);
  XLB_REQUIRE(halo >= 0 && halo <= 2, "halo must be 0, 1 or 2");
  XLB_REQUIRE(dtype_size(dtype) > 0, "bad dtype %d", dtype);
  XLB_REQUIRE(dtype != XLBHIP_MISSING || card <= 32, "missing_mask cardinality %d > 32", card);
  XLB_HIP(hipSetDevice(c->device));
  xlbhip_field* f = new xlbhip_field();
  f->ctx = c;
  f->card = card;
  f->nx = nx;
  f->ny = ny;
  f->nz = nz;
  f->halo = halo;
  f->dtype = dtype;
  f->planes = dtype == XLBHIP_MISSING ? 1 : card;
  const size_t es = dtype_size(dtype);
  size_t stride = f->cells_with_halo();
  if (f->planes > 1) {
    // pad the plane stride (multiple of 16 B keeps vector alignment) so that the q planes
    // of one cell do not land on the same HBM channel when the extent is a power of two
    size_t pad = (size_t)opt(c, "plane_pad_bytes", 0) / es;
    stride += pad;
  }
  const size_t align_elems = 256 / es;  // keep every plane 256-B aligned
  stride = (stride + align_elems - 1) / align_elems * align_elems;
  f->plane_stride = stride;
  f->alloc_bytes = f->planes * stride * es + 2 * GUARD_BYTES;
  hipError_t e = hipMalloc(&f->base, f->alloc_bytes);
  if (e != hipSuccess) {
    delete f;
    XLB_FAIL("hipMalloc(%zu bytes) failed: %s", f->alloc_bytes, hipGetErrorString(e));
  }
  f->data = static_cast<char*>(f->base) + GUARD_BYTES;
  touch(f);  // a fresh contents version: no cache entry of a freed field at this address can match
  *out = f;
  if (dtype == XLBHIP_MISSING) {
    XLB_REQUIRE(fill == 0.0, "missing_mask fill must be 0");
  }
  return xlbhip_field_fill(f, fill);
}

int xlbhip_field_destroy(xlbhip_field* f) {
  if (!f) return 0;
  (void)hipSetDevice(f->ctx->device);
  (void)hipStreamSynchronize(f->ctx->stream);
  (void)hipStreamSynchronize(f->ctx->comm_stream);
  comm_forget_buffer(f->ctx, f->base);
  if (f->strips) (void)hipFree(f->strips);
  (void)hipFree(f->base);
  delete f;
  return 0;
}

int xlbhip_field_fill(xlbhip_field* f, double v) {
  XLB_REQUIRE(f, "field is null");
  touch(f);
  hipStream_t st = f->ctx->stream;
  if (v == 0.0) {
    XLB_HIP(hipMemsetAsync(f->base, 0, f->alloc_bytes, st));
    return 0;
  }
  const size_t n = f->planes * f->plane_stride;
  switch (f->dtype) {
    case XLBHIP_F64: hipLaunchKernelGGL(k_fill<double>, blocks_capped(n), 256, 0, st, (double*)f->data, n, v); break;
    case XLBHIP_F32: hipLaunchKernelGGL(k_fill<float>, blocks_capped(n), 256, 0, st, (float*)f->data, n, (float)v); break;
    case XLBHIP_F16: hipLaunchKernelGGL(k_fill<_Float16>, blocks_capped(n), 256, 0, st, (_Float16*)f->data, n, (_Float16)v); break;
    case XLBHIP_U8: hipLaunchKernelGGL(k_fill<uint8_t>, blocks_capped(n), 256, 0, st, (uint8_t*)f->data, n, (uint8_t)v); break;
    case XLBHIP_BOOL: hipLaunchKernelGGL(k_fill<uint8_t>, blocks_capped(n), 256, 0, st, (uint8_t*)f->data, n, (uint8_t)(v != 0.0)); break;
    default: XLB_FAIL("cannot fill dtype %d with a non-zero value", f->dtype);
  }
  XLB_HIP(hipGetLastError());
  return 0;
}

int xlbhip_field_copy(xlbhip_field* dst, const xlbhip_field* src) {
  XLB_REQUIRE(dst && src, "null field");
  XLB_REQUIRE(dst->dtype == src->dtype && dst->card == src->card && same_grid(dst, src) && dst->halo == src->halo &&
                  dst->plane_stride == src->plane_stride,
              "field_copy: layouts differ");
  touch(dst);
  XLB_HIP(hipMemcpyAsync(dst->base, src->base, src->alloc_bytes, hipMemcpyDeviceToDevice, dst->ctx->stream));
  return 0;
}

int xlbhip_field_copy_kernel(xlbhip_field* dst, const xlbhip_field* src, int bytes_per_lane) {
  XLB_REQUIRE(dst && src, "null field");
  XLB_REQUIRE(dst->alloc_bytes == src->alloc_bytes && dst != src, "field_copy_kernel: layouts differ");
  XLB_REQUIRE(bytes_per_lane == 4 || bytes_per_lane == 16, "bytes_per_lane must be 4 or 16");
  const size_t bytes = src->planes * src->plane_stride * dtype_size(src->dtype);
  XLB_REQUIRE(bytes % 16 == 0, "field size not a multiple of 16 bytes");
  touch(dst);
  hipStream_t st = dst->ctx->stream;
  if (bytes_per_lane == 4) {
    const size_t n = bytes / 4;
    hipLaunchKernelGGL(k_copy<uint32_t>, blocks_capped(n), 256, 0, st, (const uint32_t*)src->data, (uint32_t*)dst->data, n);
  } else {
    const size_t n = bytes / 16;
    hipLaunchKernelGGL(k_copy<u32x4>, blocks_capped(n), 256, 0, st, (const u32x4*)src->data, (u32x4*)dst->data, n);
  }
  XLB_HIP(hipGetLastError());
  return 0;
}

int xlbhip_field_touch(xlbhip_field* f) {
  XLB_REQUIRE(f, "field is null");
  touch(f);
  return 0;
}

int xlbhip_mem_info(xlbhip_ctx* c, uint64_t* free_bytes, uint64_t* total_bytes) {
  XLB_REQUIRE(c, "ctx is null");
  XLB_HIP(hipSetDevice(c->device));
  size_t fr = 0, tot = 0;
  XLB_HIP(hipMemGetInfo(&fr, &tot));
  if (free_bytes) *free_bytes = fr;
  if (total_bytes) *total_bytes = tot;
  return 0;
}

int xlbhip_field_info(const xlbhip_field* f, int* card, int* nx, int* ny, int* nz, int* dtype, int* halo, uint64_t* ps, void** ptr) {
  XLB_REQUIRE(f, "field is null");
  if (card) *card = f->card;
  if (nx) *nx = f->nx;
  if (ny) *ny = f->ny;
  if (nz) *nz = f->nz;
  if (dtype) *dtype = f->dtype;
  if (halo) *halo = f->halo;
  if (ps) *ps = f->plane_stride;
  if (ptr) *ptr = f->data;
  return 0;
}

int xlbhip_field_upload(xlbhip_field* f, const void* host, size_t bytes) {
  XLB_REQUIRE(f && host, "null argument");
  touch(f);
  hipStream_t st = f->ctx->stream;
  const size_t n = f->cells();
  if (f->dtype == XLBHIP_MISSING) {
    XLB_REQUIRE(bytes == n * f->card, "upload size %zu != %zu", bytes, n * f->card);
    uint8_t* tmp = nullptr;
    XLB_HIP(hipMalloc(&tmp, bytes));
    XLB_HIP(hipMemcpyAsync(tmp, host, bytes, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(k_pack_missing, blocks_for(n), 256, 0, st, tmp, (uint32_t*)f->data, f->card, dims(f), f->halo);
    XLB_HIP(hipGetLastError());
    XLB_HIP(hipStreamSynchronize(st));
    XLB_HIP(hipFree(tmp));
    return 0;
  }
  const size_t es = dtype_size(f->dtype);
  XLB_REQUIRE(bytes == n * f->card * es, "upload size %zu != %zu", bytes, n * f->card * es);
  for (int l = 0; l < f->card; ++l) {
    char* d = static_cast<char*>(f->data) + ((size_t)l * f->plane_stride + (size_t)f->halo * f->ny * f->nz) * es;
    XLB_HIP(hipMemcpyAsync(d, static_cast<const char*>(host) + (size_t)l * n * es, n * es, hipMemcpyHostToDevice, st));
  }
  XLB_HIP(hipStreamSynchronize(st));  // the host buffer may be reused by the caller
  return 0;
}

int xlbhip_field_download(const xlbhip_field* f, void* host, size_t bytes) {
  XLB_REQUIRE(f && host, "null argument");
  hipStream_t st = f->ctx->stream;
  const size_t n = f->cells();
  XLB_HIP(hipStreamSynchronize(f->ctx->comm_stream));
  if (f->dtype == XLBHIP_MISSING) {
    XLB_REQUIRE(bytes == n * f->card, "download size %zu != %zu", bytes, n * f->card);
    uint8_t* tmp = nullptr;
    XLB_HIP(hipMalloc(&tmp, bytes));
    hipLaunchKernelGGL(k_unpack_missing, blocks_for(n), 256, 0, st, (const uint32_t*)f->data, tmp, f->card, dims(f), f->halo);
    XLB_HIP(hipGetLastError());
    XLB_HIP(hipMemcpyAsync(host, tmp, bytes, hipMemcpyDeviceToHost, st));
    XLB_HIP(hipStreamSynchronize(st));
    XLB_HIP(hipFree(tmp));
    return 0;
  }
  const size_t es = dtype_size(f->dtype);
  XLB_REQUIRE(bytes == n * f->card * es, "download size %zu != %zu", bytes, n * f->card * es);
  for (int l = 0; l < f->card; ++l) {
    const char* s = static_cast<const char*>(f->data) + ((size_t)l * f->plane_stride + (size_t)f->halo * f->ny * f->nz) * es;
    XLB_HIP(hipMemcpyAsync(static_cast<char*>(host) + (size_t)l * n * es, s, n * es, hipMemcpyDeviceToHost, st));
  }
  XLB_HIP(hipStreamSynchronize(st));
  return 0;
}

int xlbhip_field_plane_download(const xlbhip_field* f, int population, int storage_plane, void* host, size_t bytes) {
  XLB_REQUIRE(f && host, "null argument");
  // (the bit-packed missing_mask has ONE device plane of uint32 bit-sets: population 0)
  XLB_REQUIRE(population >= 0 && population < f->planes && storage_plane >= 0 && storage_plane < f->nx + 2 * f->halo,
              "plane (%d, %d) out of range", population, storage_plane);
  const size_t es = dtype_size(f->dtype), plane = (size_t)f->ny * f->nz;
  XLB_REQUIRE(bytes == plane * es, "plane size %zu != %zu", bytes, plane * es);
  const char* s = static_cast<const char*>(f->data) + ((size_t)population * f->plane_stride + (size_t)storage_plane * plane) * es;
  XLB_HIP(hipMemcpyAsync(host, s, bytes, hipMemcpyDeviceToHost, f->ctx->stream));
  XLB_HIP(hipStreamSynchronize(f->ctx->stream));
  return 0;
}

int xlbhip_field_plane_upload(xlbhip_field* f, int population, int storage_plane, const void* host, size_t bytes) {
  XLB_REQUIRE(f && host, "null argument");
  // (the bit-packed missing_mask has ONE device plane of uint32 bit-sets: population 0)
  XLB_REQUIRE(population >= 0 && population < f->planes && storage_plane >= 0 && storage_plane < f->nx + 2 * f->halo,
              "plane (%d, %d) out of range", population, storage_plane);
  const size_t es = dtype_size(f->dtype), plane = (size_t)f->ny * f->nz;
  XLB_REQUIRE(bytes == plane * es, "plane size %zu != %zu", bytes, plane * es);
  touch(f);
  char* d = static_cast<char*>(f->data) + ((size_t)population * f->plane_stride + (size_t)storage_plane * plane) * es;
  XLB_HIP(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, f->ctx->stream));
  XLB_HIP(hipStreamSynchronize(f->ctx->stream));
  return 0;
}

// ---- whole-field operators --------------------------------------------------------------
#define XLB_CHECK_POP(f, lattice, what)                                                                     \
  XLB_REQUIRE((f) && is_float((f)->dtype) && (f)->card == lattice_q(lattice), "%s: expected a %d-population float field", \
              what, lattice_q(lattice))

int xlbhip_stream(xlbhip_ctx* c, int lattice, const xlbhip_field* src, xlbhip_field* dst) {
  XLB_REQUIRE(c, "ctx is null");
  XLB_REQUIRE(src && dst && src->card == lattice_q(lattice) && dst->card == src->card && same_grid(src, dst),
              "stream: field shapes do not match the lattice");
  XLB_REQUIRE(src->dtype != XLBHIP_MISSING && dst->dtype == src->dtype, "stream: dtype mismatch");
  XLB_REQUIRE(src != dst, "stream: f_0 and f_1 must be different fields");
  touch(dst);
  const size_t n = src->cells();
  return by_lattice(lattice, [&](auto L) {
    hipLaunchKernelGGL(k_stream<decltype(L)>, blocks_for(n), 256, 0, c->stream, view(src), view(dst), dims(src));
    XLB_HIP(hipGetLastError());
    return 0;
  });
}

int xlbhip_equilibrium(xlbhip_ctx* c, int lattice, int cdt, const xlbhip_field* rho, const xlbhip_field* u, xlbhip_field* f) {
  XLB_REQUIRE(c, "ctx is null");
  XLB_CHECK_POP(f, lattice, "equilibrium");
  XLB_REQUIRE(rho && u && rho->card == 1 && u->card == lattice_d(lattice) && same_grid(rho, f) && same_grid(u, f) &&
                  is_float(rho->dtype) && is_float(u->dtype),
              "equilibrium: rho must be (1,...) and u (d,...) float fields on f's grid");
  XLB_REQUIRE(cdt == XLBHIP_F32 || cdt == XLBHIP_F64, "bad compute dtype %d", cdt);
  touch(f);
  const size_t n = f->cells();
  return by_lattice(lattice, [&](auto L) {
    using LL = decltype(L);
    if (cdt == XLBHIP_F32)
      hipLaunchKernelGGL((k_equilibrium<LL, float>), blocks_for(n), 256, 0, c->stream, view(rho), view(u), view(f), dims(f));
    else
      hipLaunchKernelGGL((k_equilibrium<LL, double>), blocks_for(n), 256, 0, c->stream, view(rho), view(u), view(f), dims(f));
    XLB_HIP(hipGetLastError());
    return 0;
  });
}

int xlbhip_macroscopic(xlbhip_ctx* c, int lattice, int cdt, const xlbhip_field* f, xlbhip_field* rho, xlbhip_field* u) {
  XLB_REQUIRE(c, "ctx is null");
  XLB_CHECK_POP(f, lattice, "macroscopic");
  XLB_REQUIRE(!rho || (rho->card == 1 && same_grid(rho, f) && is_float(rho->dtype)), "macroscopic: bad rho field");
  XLB_REQUIRE(!u || (u->card == lattice_d(lattice) && same_grid(u, f) && is_float(u->dtype)), "macroscopic: bad u field");
  XLB_REQUIRE(cdt == XLBHIP_F32 || cdt == XLBHIP_F64, "bad compute dtype %d", cdt);
  const size_t n = f->cells();
  return by_lattice(lattice, [&](auto L) {
    using LL = decltype(L);
    if (cdt == XLBHIP_F32)
      hipLaunchKernelGGL((k_macroscopic<LL, float>), blocks_for(n), 256, 0, c->stream, view(f), view(rho), view(u), dims(f));
    else
      hipLaunchKernelGGL((k_macroscopic<LL, double>), blocks_for(n), 256, 0, c->stream, view(f), view(rho), view(u), dims(f));
    XLB_HIP(hipGetLastError());
    return 0;
  });
}

int xlbhip_momentum_transfer(xlbhip_ctx* c, int lattice, int cdt, const xlbhip_bc_desc* bc, const xlbhip_field* f_0, const xlbhip_field* bcm,
                             const xlbhip_field* miss, double force_out[3]) {
  XLB_REQUIRE(c && bc && force_out, "null argument");
  XLB_CHECK_POP(f_0, lattice, "momentum_transfer(f_0)");
  XLB_REQUIRE(bcm && bcm->dtype == XLBHIP_U8 && bcm->card == 1 && same_grid(bcm, f_0) && bcm->halo == f_0->halo, "momentum_transfer: bad bc_mask field");
  XLB_REQUIRE(miss && miss->dtype == XLBHIP_MISSING && same_grid(miss, f_0) && miss->halo == f_0->halo, "momentum_transfer: needs the missing_mask field");
  XLB_REQUIRE(bc->kind == XLBHIP_BC_HALFWAY_BB || bc->kind == XLBHIP_BC_FULLWAY_BB,
              "momentum_transfer: the no-slip BC must be a halfway or fullway bounce-back (kind %d given)", bc->kind);
  XLB_REQUIRE(cdt == XLBHIP_F32 || cdt == XLBHIP_F64, "bad compute dtype %d", cdt);
  BcValues vals;
  std::memcpy(vals.v, bc->values, sizeof(vals.v));
  double* dforce = nullptr;
  XLB_HIP(hipMalloc(&dforce, 3 * sizeof(double)));
  XLB_HIP(hipMemsetAsync(dforce, 0, 3 * sizeof(double), c->stream));
  const size_t n = f_0->cells();
  const int wall = bc->kind == XLBHIP_BC_HALFWAY_BB ? 1 : 0;
  int rc = by_lattice(lattice, [&](auto L) {
    using LL = decltype(L);
    if (cdt == XLBHIP_F32)
      hipLaunchKernelGGL((k_momentum_transfer<LL, float>), blocks_for(n), 256, 0, c->stream, view(f_0), view(bcm), view(miss), dims(f_0), bc->id, vals,
                         wall, dforce);
    else
      hipLaunchKernelGGL((k_momentum_transfer<LL, double>), blocks_for(n), 256, 0, c->stream, view(f_0), view(bcm), view(miss), dims(f_0), bc->id, vals,
                         wall, dforce);
    XLB_HIP(hipGetLastError());
    return 0;
  });
  if (rc == 0) {
    hipError_t e = hipMemcpyAsync(force_out, dforce, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
      (void)hipFree(dforce);
      XLB_FAIL("momentum_transfer: %s", hipGetErrorString(e));
    }
  }
  (void)hipFree(dforce);
  return rc;
}

int xlbhip_grid_to_point(xlbhip_ctx* c, const xlbhip_field* grid, int64_t n, const float* points, void* values) {
  XLB_REQUIRE(c && grid && (n == 0 || (points && values)), "grid_to_point: null argument");
  XLB_REQUIRE((grid->dtype == XLBHIP_F32 || grid->dtype == XLBHIP_F64) && grid->halo == 0, "grid_to_point: fp32 / fp64 field without ghost planes");
  if (n == 0) return 0;
  // every point needs its surrounding cube inside the field (the reference does not check; an out-of-range point reads
  // out of bounds there)
  for (int64_t i = 0; i < n; ++i)
    for (int a = 0; a < 3; ++a) {
      const float p = points[3 * i + a];
      const int ext = a == 0 ? grid->nx : (a == 1 ? grid->ny : grid->nz);
      XLB_REQUIRE(p >= 0.0f && (int)p + 1 <= ext - 1, "grid_to_point: point %lld (%g along axis %d) needs cells outside the field", (long long)i,
                  (double)p, a);
    }
  const size_t es = dtype_size(grid->dtype);
  float* dp = nullptr;
  void* dv = nullptr;
  XLB_HIP(hipMalloc(&dp, (size_t)n * 3 * sizeof(float)));
  hipError_t e = hipMalloc(&dv, (size_t)n * es);
  if (e != hipSuccess) {
    (void)hipFree(dp);
    XLB_FAIL("grid_to_point: %s", hipGetErrorString(e));
  }
  e = hipMemcpyAsync(dp, points, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) {
    if (grid->dtype == XLBHIP_F32)
      hipLaunchKernelGGL((k_grid_to_point<float>), blocks_for((size_t)n), 256, 0, c->stream, view(grid), dims(grid), dp, static_cast<float*>(dv), n);
    else
      hipLaunchKernelGGL((k_grid_to_point<double>), blocks_for((size_t)n), 256, 0, c->stream, view(grid), dims(grid), dp, static_cast<double*>(dv), n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(values, dv, (size_t)n * es, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(dp);
  (void)hipFree(dv);
  XLB_REQUIRE(e == hipSuccess, "grid_to_point: %s", hipGetErrorString(e));
  return 0;
}

int xlbhip_vorticity(xlbhip_ctx* c, const xlbhip_field* u, const xlbhip_field* bcm, xlbhip_field* vorticity, xlbhip_field* magnitude) {
  return velocity_gradient_launch<0>(c, u, bcm, vorticity, magnitude, "vorticity");
}

int xlbhip_q_criterion(xlbhip_ctx* c, const xlbhip_field* u, const xlbhip_field* bcm, xlbhip_field* norm_mu, xlbhip_field* q) {
  return velocity_gradient_launch<1>(c, u, bcm, norm_mu, q, "q_criterion");
}

int xlbhip_second_moment(xlbhip_ctx* c, int lattice, int cdt, const xlbhip_field* f, xlbhip_field* pi) {
  XLB_REQUIRE(c, "ctx is null");
  XLB_CHECK_POP(f, lattice, "second_moment");
  const int nt = lattice_d(lattice) * (lattice_d(lattice) + 1) / 2;
  XLB_REQUIRE(pi && pi->card == nt && same_grid(pi, f) && is_float(pi->dtype), "second_moment: pi must be a (%d,...) float field", nt);
  XLB_REQUIRE(cdt == XLBHIP_F32 || cdt == XLBHIP_F64, "bad compute dtype %d", cdt);
  const size_t n = f->cells();
  return by_lattice(lattice, [&](auto L) {
    using LL = decltype(L);
    if (cdt == XLBHIP_F32)
      hipLaunchKernelGGL((k_second_moment<LL, float>), blocks_for(n), 256, 0, c->stream, view(f), view(pi), dims(f));
    else
      hipLaunchKernelGGL((k_second_moment<LL, double>), blocks_for(n), 256, 0, c->stream, view(f), view(pi), dims(f));
    XLB_HIP(hipGetLastError());
    return 0;
  });
}

int xlbhip_collide(xlbhip_ctx* c, int lattice, int coll, int cdt, const xlbhip_field* f, const xlbhip_field* feq, xlbhip_field* fo,
                   double omega) {
  XLB_REQUIRE(c, "ctx is null");
  XLB_CHECK_POP(f, lattice, "collide(f)");
  XLB_CHECK_POP(feq, lattice, "collide(feq)");
  XLB_CHECK_POP(fo, lattice, "collide(fout)");
  XLB_REQUIRE(same_grid(f, feq) && same_grid(f, fo), "collide: grids differ");
  XLB_REQUIRE(coll == XLBHIP_BGK || coll == XLBHIP_KBC || coll == XLBHIP_SMAGORINSKY_LES_BGK, "unknown collision %d", coll);
  XLB_REQUIRE(cdt == XLBHIP_F32 || cdt == XLBHIP_F64, "bad compute dtype %d", cdt);
  touch(fo);
  return by_lattice(lattice, [&](auto L) {
    using LL = decltype(L);
    return cdt == XLBHIP_F32 ? collide_launch<LL, float>(c, coll, f, feq, fo, omega) : collide_launch<LL, double>(c, coll, f, feq, fo, omega);
  });
}

int xlbhip_apply_bc(xlbhip_ctx* c, int lattice, int cdt, const xlbhip_bc_desc* bc, const xlbhip_field* f_pre, xlbhip_field* f_post,
                    const xlbhip_field* bcm, const xlbhip_field* miss) {
  return xlbhip_apply_bc_profile(c, lattice, cdt, bc, f_pre, f_post, bcm, miss, 0, nullptr, nullptr);
}

int xlbhip_apply_bc_profile(xlbhip_ctx* c, int lattice, int cdt, const xlbhip_bc_desc* bc, const xlbhip_field* f_pre, xlbhip_field* f_post,
                            const xlbhip_field* bcm, const xlbhip_field* miss, int64_t n_prof, const uint32_t* storage_cells,
                            const double* values) {
  XLB_REQUIRE(c && bc, "null argument");
  XLB_CHECK_POP(f_pre, lattice, "bc(f_pre)");
  XLB_CHECK_POP(f_post, lattice, "bc(f_post)");
  XLB_REQUIRE(bcm && bcm->dtype == XLBHIP_U8 && bcm->card == 1 && same_grid(bcm, f_post), "bc: bad bc_mask field");
  XLB_REQUIRE(same_grid(f_pre, f_post), "bc: grids differ");
  touch(f_post);
  XLB_REQUIRE(bc->id >= 1 && bc->id <= 255, "bc id %d out of range", bc->id);
  XLB_REQUIRE(bc->kind >= XLBHIP_BC_EQUILIBRIUM && bc->kind <= XLBHIP_BC_HYBRID_NEQ_REGULARIZED, "unknown bc kind %d (wall-velocity tables live in the stepper)", bc->kind);
  XLB_REQUIRE(bc->kind < XLBHIP_BC_HYBRID_BB_REGULARIZED || (lattice_d(lattice) == 3 && bc->values[4] == 0.0),
              "HybridBC as a stand-alone operator: 3-D lattices, without mesh distances (those live in the stepper: xlbhip_stepper_set_bc_distances)");
  if (bc->kind == XLBHIP_BC_HALFWAY_BB || bc->kind >= XLBHIP_BC_ZOUHE_VELOCITY)
    XLB_REQUIRE(miss && miss->dtype == XLBHIP_MISSING && same_grid(miss, f_post), "bc: this boundary condition needs a missing_mask field");
  XLB_REQUIRE(cdt == XLBHIP_F32 || cdt == XLBHIP_F64, "bad compute dtype %d", cdt);
  BcValues vals;
  std::memcpy(vals.v, bc->values, sizeof(vals.v));
  const size_t n = f_post->cells();
  // per-cell prescribed values (sorted by storage cell): uploaded for this call only — the stand-alone operator is
  // not on the hot path
  uint32_t* dk = nullptr;
  double* dv = nullptr;
  if (n_prof > 0) {
    XLB_REQUIRE(storage_cells && values, "null profile table");
    XLB_REQUIRE(bc->kind >= XLBHIP_BC_ZOUHE_VELOCITY && bc->kind <= XLBHIP_BC_REGULARIZED_PRESSURE, "profiles belong to Zou-He / Regularized BCs");
    std::vector<std::pair<uint32_t, int64_t>> order((size_t)n_prof);
    for (int64_t i = 0; i < n_prof; ++i) order[(size_t)i] = {storage_cells[i], i};
    std::sort(order.begin(), order.end());
    std::vector<uint32_t> keys((size_t)n_prof);
    std::vector<double> v((size_t)n_prof * 3);
    for (int64_t i = 0; i < n_prof; ++i) {
      keys[(size_t)i] = order[(size_t)i].first;
      for (int a = 0; a < 3; ++a) v[(size_t)i * 3 + a] = values[order[(size_t)i].second * 3 + a];
    }
    XLB_HIP(hipMalloc(&dk, keys.size() * sizeof(uint32_t)));
    XLB_HIP(hipMalloc(&dv, v.size() * sizeof(double)));
    XLB_HIP(hipMemcpyAsync(dk, keys.data(), keys.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    XLB_HIP(hipMemcpyAsync(dv, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    XLB_HIP(hipStreamSynchronize(c->stream));  // the host vectors die with this scope
  }
  const int np = (int)n_prof;
  int rc = by_lattice(lattice, [&](auto L) {
    using LL = decltype(L);
    if (cdt == XLBHIP_F32)
      hipLaunchKernelGGL((k_apply_bc<LL, float>), blocks_for(n), 256, 0, c->stream, bc->id, bc->kind, vals, view(f_pre), view(f_post),
                         view(bcm), view(miss), dims(f_post), dk, dv, np);
    else
      hipLaunchKernelGGL((k_apply_bc<LL, double>), blocks_for(n), 256, 0, c->stream, bc->id, bc->kind, vals, view(f_pre), view(f_post),
                         view(bcm), view(miss), dims(f_post), dk, dv, np);
    XLB_HIP(hipGetLastError());
    return 0;
  });
  if (dk) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(dk);
    (void)hipFree(dv);
  }
  return rc;
}

// ---- masker -------------------------------------------------------------------------------
int xlbhip_build_masks(xlbhip_ctx* c, int lattice, int n_bc, const int32_t* ids, const int32_t* const* tag_idx, const int64_t* tag_count,
                       const int32_t* const* solid_idx, const int64_t* solid_count, const int32_t gshape[3], int x_offset,
                       xlbhip_field* bcm, xlbhip_field* miss) {
  XLB_REQUIRE(c && bcm && miss && gshape, "null argument");
  XLB_REQUIRE(bcm->dtype == XLBHIP_U8 && bcm->card == 1, "bc_mask must be a (1,...) uint8 field");
  XLB_REQUIRE(miss->dtype == XLBHIP_MISSING && miss->card == lattice_q(lattice), "missing_mask must be a (q,...) missing field");
  XLB_REQUIRE(same_grid(bcm, miss) && bcm->halo == miss->halo, "masks live on different grids");
  XLB_REQUIRE(gshape[1] == bcm->ny && gshape[2] == bcm->nz && x_offset >= 0 && x_offset + bcm->nx <= gshape[0],
              "slab (offset %d, nx %d) does not fit global shape (%d,%d,%d)", x_offset, bcm->nx, gshape[0], gshape[1], gshape[2]);
  XLB_REQUIRE(n_bc == 0 || (ids && tag_idx && tag_count), "null bc arrays");
  touch(bcm);
  touch(miss);
  hipStream_t st = c->stream;
  const Dims d = dims(bcm);
  const size_t plane = (size_t)d.ny * d.nz;
  // solid scratch with one ghost plane per side
  uint8_t* solid = nullptr;
  const size_t solid_bytes = (size_t)(d.nx + 2) * plane;
  XLB_HIP(hipMalloc(&solid, solid_bytes));
  if (hipError_t e_ = hipMemsetAsync(solid, 0, solid_bytes, st); e_ != hipSuccess) {
    (void)hipFree(solid);
    XLB_FAIL("hipMemsetAsync failed: %s", hipGetErrorString(e_));
  }
  std::vector<int32_t*> tmp;
  auto cleanup = [&]() {
    (void)hipStreamSynchronize(st);
    for (auto p : tmp) (void)hipFree(p);
    (void)hipFree(solid);
  };
  // every failure below releases the temporaries (the stream is drained first: copies may still read them)
#define XLB_MASK_HIP(expr)                                                                                     \
  do {                                                                                                         \
    hipError_t e_ = (expr);                                                                                    \
    if (e_ != hipSuccess) {                                                                                    \
      cleanup();                                                                                               \
      XLB_FAIL("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__);                     \
    }                                                                                                          \
  } while (0)
  uint8_t* bc_base = static_cast<uint8_t*>(bcm->data) + (size_t)bcm->halo * plane;  // interior plane 0
  for (int i = 0; i < n_bc; ++i) {
    if (ids[i] < 1 || ids[i] > 255) {
      cleanup();
      XLB_FAIL("bc id %d out of range 1..255", ids[i]);
    }
    if (solid_idx && solid_idx[i] && solid_count && solid_count[i] > 0) {
      const int64_t n = solid_count[i];
      int32_t* dv = nullptr;
      XLB_MASK_HIP(hipMalloc(&dv, (size_t)n * 3 * sizeof(int32_t)));
      tmp.push_back(dv);
      XLB_MASK_HIP(hipMemcpyAsync(dv, solid_idx[i], (size_t)n * 3 * sizeof(int32_t), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_scatter_u8, blocks_for((size_t)n), 256, 0, st, solid, dv, n, (uint8_t)1, x_offset - 1, x_offset + d.nx + 1,
                         d.ny, d.nz);
    }
    if (tag_count[i] > 0) {
      const int64_t n = tag_count[i];
      int32_t* dv = nullptr;
      XLB_MASK_HIP(hipMalloc(&dv, (size_t)n * 3 * sizeof(int32_t)));
      tmp.push_back(dv);
      XLB_MASK_HIP(hipMemcpyAsync(dv, tag_idx[i], (size_t)n * 3 * sizeof(int32_t), hipMemcpyHostToDevice, st));
      hipLaunchKernelGGL(k_scatter_u8, blocks_for((size_t)n), 256, 0, st, bc_base, dv, n, (uint8_t)ids[i], x_offset, x_offset + d.nx, d.ny,
                         d.nz);
    }
  }
  // stream the (old | solid | outside) marks: missing'[l, x] = marks[l, x - c_l]
  uint32_t* old = nullptr;
  const size_t mbytes = miss->cells_with_halo() * sizeof(uint32_t);
  XLB_MASK_HIP(hipMalloc(&old, mbytes));
  tmp.push_back(reinterpret_cast<int32_t*>(old));
  XLB_MASK_HIP(hipMemcpyAsync(old, miss->data, mbytes, hipMemcpyDeviceToDevice, st));
  const size_t n = bcm->cells();
  int rc = by_lattice(lattice, [&](auto L) {
    hipLaunchKernelGGL(k_missing<decltype(L)>, blocks_for(n), 256, 0, st, (uint32_t*)miss->data, old, solid, d, miss->halo, gshape[0],
                       x_offset);
    XLB_HIP(hipGetLastError());
    return 0;
  });
  cleanup();
  return rc;
#undef XLB_MASK_HIP
}

}  // extern "C"

extern "C" int xlbhip_mesh_mask_aabb(xlbhip_ctx* c, int lattice, int bc_id, int64_t n_triangles, const float* vertices, xlbhip_field* bcm,
                                     xlbhip_field* miss) {
  XLB_REQUIRE(c && bcm && miss && (n_triangles == 0 || vertices), "mesh masker: null argument");
  XLB_REQUIRE(lattice == XLBHIP_D3Q19 || lattice == XLBHIP_D3Q27, "MeshBoundaryMasker is only implemented for 3D velocity sets!");
  XLB_REQUIRE(bcm->dtype == XLBHIP_U8 && bcm->card == 1 && bcm->halo == 0, "mesh masker: bc_mask must be a (1, nx, ny, nz) uint8 field without ghost planes");
  XLB_REQUIRE(miss->dtype == XLBHIP_MISSING && miss->card == lattice_q(lattice) && same_grid(miss, bcm) && miss->halo == 0, "mesh masker: bad missing_mask field");
  XLB_REQUIRE(bc_id >= 1 && bc_id <= 254, "bc id %d out of range 1..254", bc_id);
  touch(bcm);
  touch(miss);
  // the mesh must lie inside the domain (mesh_boundary_masker.py:196-201)
  for (int64_t i = 0; i < n_triangles * 3; ++i)
    for (int a = 0; a < 3; ++a) {
      const float p = vertices[3 * i + a];
      const int ext = a == 0 ? bcm->nx : (a == 1 ? bcm->ny : bcm->nz);
      XLB_REQUIRE(p >= 0.0f && p < (float)ext, "Mesh extents exceed domain dimensions (%d,%d,%d). The mesh must be fully contained within the domain.",
                  bcm->nx, bcm->ny, bcm->nz);
    }
  hipStream_t st = c->stream;
  const size_t cells = bcm->cells();
  uint8_t* solid = nullptr;
  float* dv = nullptr;
  XLB_HIP(hipMalloc(&solid, cells));
  hipError_t e = hipMemsetAsync(solid, 0, cells, st);
  if (e == hipSuccess && n_triangles > 0) {
    e = hipMalloc(&dv, (size_t)n_triangles * 9 * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(dv, vertices, (size_t)n_triangles * 9 * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(k_mesh_solid, blocks_for((size_t)n_triangles), 256, 0, st, dv, n_triangles, solid, bcm->nx, bcm->ny, bcm->nz);
      e = hipGetLastError();
    }
  }
  int rc = 0;
  if (e == hipSuccess)
    rc = by_lattice(lattice, [&](auto L) {
      hipLaunchKernelGGL(k_mesh_classify<decltype(L)>, blocks_for(cells), 256, 0, st, solid, static_cast<uint8_t*>(bcm->data),
                         static_cast<uint32_t*>(miss->data), dims(bcm), bc_id);
      XLB_HIP(hipGetLastError());
      return 0;
    });
  (void)hipStreamSynchronize(st);
  (void)hipFree(solid);
  if (dv) (void)hipFree(dv);
  XLB_REQUIRE(e == hipSuccess, "mesh masker: %s", hipGetErrorString(e));
  return rc;
}

extern "C" int xlbhip_mesh_mask_ray(xlbhip_ctx* c, int lattice, int bc_id, int64_t n_triangles, const float* vertices, xlbhip_field* bcm,
                                    xlbhip_field* miss) {
  XLB_REQUIRE(c && bcm && miss && (n_triangles == 0 || vertices), "mesh masker: null argument");
  XLB_REQUIRE(lattice == XLBHIP_D3Q19 || lattice == XLBHIP_D3Q27, "MeshBoundaryMasker is only implemented for 3D velocity sets!");
  XLB_REQUIRE(bcm->dtype == XLBHIP_U8 && bcm->card == 1 && bcm->halo == 0, "mesh masker: bc_mask must be a (1, nx, ny, nz) uint8 field without ghost planes");
  XLB_REQUIRE(miss->dtype == XLBHIP_MISSING && miss->card == lattice_q(lattice) && same_grid(miss, bcm) && miss->halo == 0, "mesh masker: bad missing_mask field");
  XLB_REQUIRE(bc_id >= 1 && bc_id <= 254, "bc id %d out of range 1..254", bc_id);
  touch(bcm);
  touch(miss);
  for (int64_t i = 0; i < n_triangles * 3; ++i)
    for (int a = 0; a < 3; ++a) {
      const float p = vertices[3 * i + a];
      const int ext = a == 0 ? bcm->nx : (a == 1 ? bcm->ny : bcm->nz);
      XLB_REQUIRE(p >= 0.0f && p < (float)ext, "Mesh extents exceed domain dimensions (%d,%d,%d). The mesh must be fully contained within the domain.",
                  bcm->nx, bcm->ny, bcm->nz);
    }
  hipStream_t st = c->stream;
  float* dv = nullptr;
  hipError_t e = hipSuccess;
  int rc = 0;
  if (n_triangles > 0) {
    XLB_HIP(hipMalloc(&dv, (size_t)n_triangles * 9 * sizeof(float)));
    e = hipMemcpyAsync(dv, vertices, (size_t)n_triangles * 9 * sizeof(float), hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
      rc = by_lattice(lattice, [&](auto L) {
        hipLaunchKernelGGL(k_mesh_ray<decltype(L)>, blocks_for((size_t)n_triangles), 256, 0, st, dv, n_triangles, static_cast<uint8_t*>(bcm->data),
                           static_cast<uint32_t*>(miss->data), dims(bcm), bc_id);
        XLB_HIP(hipGetLastError());
        return 0;
      });
  }
  if (e == hipSuccess && rc == 0)
    rc = by_lattice(lattice, [&](auto L) {
      hipLaunchKernelGGL(k_mesh_resolve<decltype(L)>, blocks_for(bcm->cells()), 256, 0, st, static_cast<const uint8_t*>(bcm->data),
                         static_cast<uint32_t*>(miss->data), dims(bcm), bc_id);
      XLB_HIP(hipGetLastError());
      return 0;
    });
  (void)hipStreamSynchronize(st);
  if (dv) (void)hipFree(dv);
  XLB_REQUIRE(e == hipSuccess, "mesh masker: %s", hipGetErrorString(e));
  return rc;
}

// RAII scratch of the mesh maskers: device buffers released (after the stream drained) on every exit path
namespace {
struct DeviceScratch {
  hipStream_t st;
  std::vector<void*> ptrs;
  explicit DeviceScratch(hipStream_t s) : st(s) {}
  ~DeviceScratch() {
    (void)hipStreamSynchronize(st);
    for (void* p : ptrs) (void)hipFree(p);
  }
  template <class P>
  hipError_t alloc(P** out, size_t bytes) {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipSuccess) ptrs.push_back(p);
    *out = static_cast<P*>(p);
    return e;
  }
};
}  // namespace

// Mesh voxelisation, all methods (boundary_masker/{aabb,ray,winding,aabb_close}.py), with the wall distances.
extern "C" int xlbhip_mesh_mask(xlbhip_ctx* c, int lattice, int method, int bc_id, int64_t n_triangles, const float* vertices, int close_voxels,
                                xlbhip_field* bcm, xlbhip_field* miss, xlbhip_field* dist) {
  if (method == XLBHIP_MESH_AABB && !dist) return xlbhip_mesh_mask_aabb(c, lattice, bc_id, n_triangles, vertices, bcm, miss);
  if (method == XLBHIP_MESH_RAY && !dist) return xlbhip_mesh_mask_ray(c, lattice, bc_id, n_triangles, vertices, bcm, miss);
  XLB_REQUIRE(c && bcm && miss && (n_triangles == 0 || vertices), "mesh masker: null argument");
  XLB_REQUIRE(method == XLBHIP_MESH_RAY || method == XLBHIP_MESH_WINDING || method == XLBHIP_MESH_AABB_CLOSE,
              "mesh masker: method %d has no wall distances (RAY, WINDING, AABB_CLOSE do)", method);
  XLB_REQUIRE(lattice == XLBHIP_D3Q19 || lattice == XLBHIP_D3Q27, "MeshBoundaryMasker is only implemented for 3D velocity sets!");
  XLB_REQUIRE(bcm->dtype == XLBHIP_U8 && bcm->card == 1 && bcm->halo == 0, "mesh masker: bc_mask must be a (1, nx, ny, nz) uint8 field without ghost planes");
  XLB_REQUIRE(miss->dtype == XLBHIP_MISSING && miss->card == lattice_q(lattice) && same_grid(miss, bcm) && miss->halo == 0, "mesh masker: bad missing_mask field");
  XLB_REQUIRE(bc_id >= 1 && bc_id <= 254, "bc id %d out of range 1..254", bc_id);
  XLB_REQUIRE(!dist || (dist->dtype == XLBHIP_F32 && dist->card == lattice_q(lattice) && same_grid(dist, bcm) && dist->halo == 0),
              "mesh masker: distances must be a (q, nx, ny, nz) fp32 field on the masks' grid");
  XLB_REQUIRE(method != XLBHIP_MESH_AABB_CLOSE || (close_voxels >= 1 && close_voxels <= 8), "AABB_CLOSE: close_voxels must be 1..8 (got %d)", close_voxels);
  float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  for (int64_t i = 0; i < n_triangles * 3; ++i)
    for (int a = 0; a < 3; ++a) {
      const float p = vertices[3 * i + a];
      const int ext = a == 0 ? bcm->nx : (a == 1 ? bcm->ny : bcm->nz);
      XLB_REQUIRE(p >= 0.0f && p < (float)ext, "Mesh extents exceed domain dimensions (%d,%d,%d). The mesh must be fully contained within the domain.",
                  bcm->nx, bcm->ny, bcm->nz);
      lo[a] = i == 0 ? p : std::min(lo[a], p);
      hi[a] = i == 0 ? p : std::max(hi[a], p);
    }
  touch(bcm);
  touch(miss);
  if (dist) touch(dist);
  if (n_triangles == 0) return 0;
  hipStream_t st = c->stream;
  DeviceScratch scratch(st);
  const Dims d = dims(bcm);
  const size_t cells = bcm->cells();
  const int q = lattice_q(lattice);
  float* dv = nullptr;
  XLB_HIP(scratch.alloc(&dv, (size_t)n_triangles * 9 * sizeof(float)));
  XLB_HIP(hipMemcpyAsync(dv, vertices, (size_t)n_triangles * 9 * sizeof(float), hipMemcpyHostToDevice, st));
  unsigned* tbuf = nullptr;  // closest ray parameter per (link, voxel), +inf = none
  if (dist || method == XLBHIP_MESH_WINDING) {
    XLB_HIP(scratch.alloc(&tbuf, (size_t)q * cells * sizeof(unsigned)));
    hipLaunchKernelGGL(k_fill<unsigned>, blocks_capped((size_t)q * cells), 256, 0, st, tbuf, (size_t)q * cells, T_NONE);
  }
  uint8_t* bcp = static_cast<uint8_t*>(bcm->data);
  uint32_t* mp = static_cast<uint32_t*>(miss->data);
  const FieldView dview = view(dist);
  return by_lattice(lattice, [&](auto L) {
    using LL = decltype(L);
    if (method == XLBHIP_MESH_RAY) {
      hipLaunchKernelGGL(k_mesh_ray_dist<LL>, blocks_for((size_t)n_triangles), 256, 0, st, dv, n_triangles, bcp, mp, tbuf, d, bc_id);
      hipLaunchKernelGGL(k_mesh_weights_ray<LL>, blocks_for(cells), 256, 0, st, tbuf, dview, d);
    } else if (method == XLBHIP_MESH_WINDING) {
      uint8_t* solid = nullptr;
      XLB_HIP(scratch.alloc(&solid, cells));
      XLB_HIP(hipMemsetAsync(solid, 0, cells, st));
      int b0[3], nb[3];
      const int ext[3] = {d.nx, d.ny, d.nz};
      for (int a = 0; a < 3; ++a) {
        b0[a] = std::max(0, (int)floorf(lo[a]) - 1);
        nb[a] = std::min(ext[a] - 1, (int)floorf(hi[a]) + 1) - b0[a] + 1;
      }
      hipLaunchKernelGGL(k_mesh_winding, blocks_for((size_t)nb[0] * nb[1] * nb[2]), 256, 0, st, dv, n_triangles, solid, d, b0[0], b0[1], b0[2], nb[0],
                         nb[1], nb[2]);
      hipLaunchKernelGGL(k_mesh_winding_rays<LL>, blocks_for((size_t)n_triangles), 256, 0, st, dv, n_triangles, solid, tbuf, d);
      hipLaunchKernelGGL(k_mesh_winding_tag<LL>, blocks_for(cells), 256, 0, st, solid, tbuf, bcp, mp, dview, d, bc_id);
    } else {
      const int h = close_voxels, pad = 2 * h;
      const int px = d.nx + 2 * pad, py = d.ny + 2 * pad, pz = d.nz + 2 * pad;
      const size_t pcells = (size_t)px * py * pz;
      uint8_t *pa = nullptr, *pb = nullptr, *solid = nullptr;
      XLB_HIP(scratch.alloc(&pa, pcells));
      XLB_HIP(scratch.alloc(&pb, pcells));
      XLB_HIP(scratch.alloc(&solid, cells));
      XLB_HIP(hipMemsetAsync(pa, 0, pcells, st));
      hipLaunchKernelGGL(k_mesh_solid_padded, blocks_for((size_t)n_triangles), 256, 0, st, dv, n_triangles, pa, px, py, pz, pad);
      hipLaunchKernelGGL(k_morph, blocks_for(pcells), 256, 0, st, pa, pb, px, py, pz, h, 1);
      hipLaunchKernelGGL(k_morph, blocks_for(pcells), 256, 0, st, pb, pa, px, py, pz, h, 0);
      hipLaunchKernelGGL(k_crop, blocks_for(cells), 256, 0, st, pa, solid, d, py, pz, pad);
      hipLaunchKernelGGL(k_mesh_classify<LL>, blocks_for(cells), 256, 0, st, solid, bcp, mp, d, bc_id);
      if (dist) {
        hipLaunchKernelGGL(k_mesh_close_rays<LL>, blocks_for((size_t)n_triangles), 256, 0, st, dv, n_triangles, solid, bcp, tbuf, d, bc_id);
        hipLaunchKernelGGL(k_mesh_weights_close<LL>, blocks_for(cells), 256, 0, st, solid, bcp, tbuf, dview, d, bc_id);
      }
    }
    if (method != XLBHIP_MESH_AABB_CLOSE)  // (k_mesh_classify resolves the out-of-box directions itself)
      hipLaunchKernelGGL(k_mesh_resolve<LL>, blocks_for(cells), 256, 0, st, bcp, mp, d, bc_id);
    XLB_HIP(hipGetLastError());
    return 0;
  });
}

extern "C" int xlbhip_field_gather(const xlbhip_field* f, int64_t n, const uint32_t* cells, void* out, size_t bytes) {
  XLB_REQUIRE(f && (n == 0 || (cells && out)), "field_gather: null argument");
  XLB_REQUIRE(f->dtype != XLBHIP_MISSING, "field_gather: not for the bit-packed missing_mask");
  const size_t es = dtype_size(f->dtype);
  XLB_REQUIRE(bytes == (size_t)n * f->card * es, "field_gather: output size %zu != %zu", bytes, (size_t)n * f->card * es);
  if (n == 0) return 0;
  for (int64_t i = 0; i < n; ++i) XLB_REQUIRE(cells[i] < f->cells(), "field_gather: cell %u outside the field", cells[i]);
  hipStream_t st = f->ctx->stream;
  DeviceScratch scratch(st);
  uint32_t* dc = nullptr;
  char* dout = nullptr;
  XLB_HIP(scratch.alloc(&dc, (size_t)n * sizeof(uint32_t)));
  XLB_HIP(scratch.alloc(&dout, bytes));
  XLB_HIP(hipMemcpyAsync(dc, cells, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  const size_t ghost = (size_t)f->halo * f->ny * f->nz;
  switch (es) {
    case 8: hipLaunchKernelGGL(k_gather<uint64_t>, blocks_for((size_t)n), 256, 0, st, (const uint64_t*)f->data, f->plane_stride, ghost, f->card, dc, n, (uint64_t*)dout); break;
    case 4: hipLaunchKernelGGL(k_gather<uint32_t>, blocks_for((size_t)n), 256, 0, st, (const uint32_t*)f->data, f->plane_stride, ghost, f->card, dc, n, (uint32_t*)dout); break;
    case 2: hipLaunchKernelGGL(k_gather<uint16_t>, blocks_for((size_t)n), 256, 0, st, (const uint16_t*)f->data, f->plane_stride, ghost, f->card, dc, n, (uint16_t*)dout); break;
    default: hipLaunchKernelGGL(k_gather<uint8_t>, blocks_for((size_t)n), 256, 0, st, (const uint8_t*)f->data, f->plane_stride, ghost, f->card, dc, n, (uint8_t*)dout); break;
  }
  XLB_HIP(hipGetLastError());
  XLB_HIP(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, st));
  XLB_HIP(hipStreamSynchronize(st));
  return 0;
}

// ---- stepper --------------------------------------------------------------------------------
struct xlbhip_stepper {
  xlbhip_ctx* ctx = nullptr;
  int lattice = 0, collision = 0, cdt = 0, sdt = 0;
  int n_bc = 0;
  bool needs_missing = false;
  bool extended_bcs = false;
  bool has_outflow = false;  // ExtrapolationOutflowBC present: k_outflow_aux runs after every step
  // two-step kernel with extended BCs on the x end planes only (inlet / outlet): the end planes go through the
  // single-step kernel twice via this third population field
  xlbhip_field* scratch = nullptr;
  bool edge_ext_ok = false;
  bool has_edge_kinds = false;  // kinds the two-step kernel does not evaluate itself: Zou-He family, outflow, do-nothing
  // result of the last "are all such cells in the x end planes" scan, valid for this (bc_mask field, contents version)
  const xlbhip_field* scan_field = nullptr;
  uint64_t scan_version = 0;
  int scan_flag = 1;
  uint32_t* tile_order = nullptr;  // two-step kernel: block -> (8 x 64) tile, hull tiles first
  int order_ty = 0, order_tz = 0, order_mode = -1;
  uint32_t* meta = nullptr;  // two-step kernel: id | missing << 8, rebuilt by every xlbhip_run that fuses
  size_t meta_cells = 0;
  // two-step kernel: per launch geometry (x_begin, x_count, segments, tile order?, swizzle) the per-block "no boundary
  // cell" flags; dropped whenever the meta words are rebuilt
  std::map<std::array<int, 8>, uint8_t*> clean_cache;
  // the masks (address + contents version) the meta words were built from: xlbhip_step2 called per pair (the Python
  // stepper pairing reference-style calls) must not rebuild them every time
  const xlbhip_field* meta_bc = nullptr;
  const xlbhip_field* meta_miss = nullptr;
  uint64_t meta_bc_version = 0, meta_miss_version = 0;
  unsigned long long meta_opts = 0;
  bool forced = false;
  double force[3] = {0, 0, 0};
  double smag_cs = 0.17;
  uint8_t* tab_kind = nullptr;  // device [256]
  unsigned long long ids_packed = 0;
  unsigned kinds_packed = 0;
  unsigned moving_mask = 0;     // slots (first 8 BCs) whose halfway wall has a non-zero moving-wall term
  void* tab_values = nullptr;   // device [256][27] compute dtype
  // per-cell prescribed values of Zou-He / Regularized BCs built with a profile: host map (storage cell -> 3 values)
  // and its sorted device image
  std::map<uint32_t, std::array<double, 3>> prof_host;
  uint32_t* prof_keys = nullptr;
  void* prof_vals = nullptr;  // compute dtype [n_prof][3]
  int n_prof = 0;
  // wall-distance weights of HybridBC cells (mesh maskers): host map (storage cell -> q weights) and its sorted device image
  std::map<uint32_t, std::array<float, 27>> dist_host;
  uint32_t* dist_keys = nullptr;
  float* dist_vals = nullptr;  // [n_dist][q]
  int n_dist = 0;
};

namespace xlb {

static int launch_any(const xlbhip_stepper* s, const StepLaunch& p) {
  if (s->forced || s->collision == XLBHIP_SMAGORINSKY_LES_BGK) {
    const int coll = s->collision | (s->forced ? COLL_FORCED : 0);
    if (s->lattice == XLBHIP_D2Q9) return launch_step_d2q9_ext(p, coll);
    if (s->lattice == XLBHIP_D3Q19) return launch_step_d3q19_ext(p, coll);
    return launch_step_d3q27_ext(p, coll);
  }
  if (s->lattice == XLBHIP_D2Q9) return s->collision == XLBHIP_BGK ? launch_step_d2q9_bgk(p) : launch_step_d2q9_kbc(p);
  if (s->lattice == XLBHIP_D3Q19) return launch_step_d3q19_bgk(p);
  if (s->collision == XLBHIP_BGK) return launch_step_d3q27_bgk(p);
  return (p.fast_math && p.compute_dtype == XLBHIP_F64) ? launch_step_d3q27_kbc_fast64(p) : launch_step_d3q27_kbc(p);
}

static int check_step_fields(const xlbhip_stepper* s, const xlbhip_field* a, const xlbhip_field* b, const xlbhip_field* bcm,
                             const xlbhip_field* miss) {
  XLB_REQUIRE(s && a && b, "null argument");
  XLB_REQUIRE(a != b, "f_0 and f_1 must be different fields (double buffering)");
  const int q = lattice_q(s->lattice);
  XLB_REQUIRE(a->card == q && b->card == q, "population fields must have cardinality %d", q);
  XLB_REQUIRE(a->dtype == s->sdt && b->dtype == s->sdt, "population fields must have the stepper's store dtype %d", s->sdt);
  XLB_REQUIRE(same_grid(a, b) && a->halo == b->halo && a->plane_stride == b->plane_stride, "f_0 and f_1 layouts differ");
  if (s->n_bc > 0) {
    XLB_REQUIRE(bcm, "this stepper has boundary conditions: bc_mask is required");
  }
  if (bcm) {
    XLB_REQUIRE(bcm->dtype == XLBHIP_U8 && bcm->card == 1 && same_grid(bcm, a) && bcm->halo == a->halo, "bad bc_mask field");
  }
  if (s->needs_missing) {
    XLB_REQUIRE(miss && miss->dtype == XLBHIP_MISSING && same_grid(miss, a) && miss->halo == a->halo,
                "halfway bounce-back needs a missing_mask field on the same grid");
  }
  return 0;
}

// tile of the two-step kernel: (8 x 64); the (16 x 32) form (fuse2_tile = 2) exists in -DXLB_TUNE_VARIANTS builds only
static int fuse2_tile_ty(const xlbhip_ctx* c) {
#ifdef XLB_TUNE_VARIANTS
  if (opt(c, "fuse2_tile", 0) == 2) return 16;
  if (opt(c, "fuse2_tile", 0) == 4) return 4;
#endif
  (void)c;
  return 8;
}
static int fuse2_tile_tz(const xlbhip_ctx* c) {
#ifdef XLB_TUNE_VARIANTS
  if (opt(c, "fuse2_tile", 0) == 2 || opt(c, "fuse2_tile", 0) == 3) return 32;
#endif
  (void)c;
  return 64;
}

static StepLaunch make_launch(xlbhip_stepper* s, const xlbhip_field* src, xlbhip_field* dst, const xlbhip_field* bcm, const xlbhip_field* miss,
                              double omega) {
  xlbhip_ctx* c = s->ctx;
  StepLaunch p;
  p.src = src->data;
  p.dst = dst->data;
  p.bc = (s->n_bc > 0 && bcm) ? static_cast<const uint8_t*>(bcm->data) : nullptr;
  p.miss = miss ? static_cast<const uint32_t*>(miss->data) : nullptr;
  p.meta = nullptr;
  p.clean = nullptr;
  p.strips_src = nullptr;
  p.strips_dst = nullptr;
  p.strips = 0;
  p.tile_order = nullptr;
  p.x_segments = 1;
  p.x_cap = 0;
  p.tile_oy = p.tile_oz = 0;
  p.tile_ty = fuse2_tile_ty(c);
  p.tile_tz = fuse2_tile_tz(c);
  // D3Q27 KBC: (8 x 48) tiles — 8 waves per block, i.e. 2 per SIMD and 256 VGPRs for the collision (the (8 x 64) tile's 11 waves
  // leave 168: 3.8 KB of scratch in fp64); the grown tile is 500 cells for 384 outputs, the same ratio as (8 x 64)
  // D3Q27 BGK with boundary conditions (round 3): the BC ring — 63 population-planes — fits the LDS on the same (8 x 48) tile: 126 KB
  if (s->lattice == XLBHIP_D3Q27 && (s->collision == XLBHIP_KBC || (s->n_bc > 0 && bcm))) p.tile_tz = 48;
  p.tab_kind = s->tab_kind;
  p.ids_packed = s->ids_packed;
  p.kinds_packed = s->kinds_packed;
  p.n_bc = s->n_bc;
  p.tab_values = s->tab_values;
  p.prof_keys = s->prof_keys;
  p.prof_vals = s->prof_vals;
  p.n_prof = s->n_prof;
  p.dist_keys = s->dist_keys;
  p.dist_vals = s->dist_vals;
  p.n_dist = s->n_dist;
  p.plane_stride = src->plane_stride;
  p.nx = src->nx;
  p.ny = src->ny;
  p.nz = src->nz;
  p.halo = src->halo;
  p.omega = omega;
  p.force[0] = s->force[0];
  p.force[1] = s->force[1];
  p.force[2] = s->force[2];
  p.smag_cs = s->smag_cs;
  p.compute_dtype = s->cdt;
  p.store_dtype = s->sdt;
  p.vec = (int)opt(c, "vec", 0);
  p.has_bc = p.bc != nullptr ? (s->extended_bcs ? 2 : 1) : 0;
  p.edge_ext = s->edge_ext_ok ? 1 : 0;
  p.flags = (opt(c, "nt_store", 1) ? 1 : 0) | (int)(opt(c, "nt_load", 0) << 1);
  p.block_threads = (int)opt(c, "block_threads", 256);
  p.block_tz = (int)opt(c, "block_tz", 0);
  p.xcd_swizzle = (int)opt(c, "xcd_swizzle", 0);
  p.fast_math = opt(c, "exact_math", 0) ? 0 : 1;
  p.fast_bgk = (opt(c, "fast_bgk", 0) && !opt(c, "exact_math", 0)) ? 1 : 0;
  p.stream = c->stream;
  p.x_begin = 0;
  p.x_count = src->nx;
  return p;
}

// x segments per tile column of the two-step kernel.  One block per CU marches a segment, so the launch runs in
// ceil(tiles * n / CUs) rounds of (planes per segment + 3 warm-up planes): pick the n that minimises that product
// (320^3: 4 -> 8 segments = 3.1 -> 6.25 rounds, -9 %; 256^3: 2 segments = exactly one round).  With halfway walls the
// hull tiles are the expensive ones and finer items balance them better: take the most segments of >= 32 planes
// (measured at 256^3 ... 512^3: profiles/r01/sweeps.md).
static long fill_cus(const xlbhip_ctx* c) {
  const int64_t o = opt(c, "fuse2_cus", 0);
  return o > 0 ? (long)o : (c->compute_units > 0 ? c->compute_units : 256);
}

static int fuse2_segments(const xlbhip_stepper* s, const StepLaunch& p) {
  const int64_t xseg = opt(s->ctx, "fuse2_xseg", 0);
  if (xseg > 0) {
    int n = (int)xseg;
    while (n > 1 && p.x_count / n < 8) n /= 2;
    return n;
  }
  const long tiles = (long)(p.ny / p.tile_ty) * (p.nz / p.tile_tz), cus = fill_cus(s->ctx);
  // with boundary conditions finer items win twice: the expensive hull tiles balance better (halfway walls), and work
  // items free of boundary cells — most segments of an interior tile column — run the BC-free body (fuse2_clean):
  // take the most segments of >= 32 planes (measured at 256^3 ... 512^3: profiles/r01/sweeps.md, profiles/r02/step2_sweeps.txt)
  const bool finest = s->needs_missing || (p.has_bc && opt(s->ctx, "fuse2_clean", 1));
  int best = 1;
  long best_cost = -1;
  for (int n = 1; n <= 8; n *= 2) {
    if (n > 1 && p.x_count / n < 32) break;
    const long cost = ((tiles * n + cus - 1) / cus) * (p.x_count / n + 3);
    if (best_cost < 0 || cost < best_cost || finest) {
      best = n;
      best_cost = cost;
    }
  }
  return best;
}

// assemble_auxiliary_data of the ExtrapolationOutflowBC cells after a step src -> dst (nse_stepper.py:270-272)
static int outflow_aux(xlbhip_stepper* s, const xlbhip_field* src, xlbhip_field* dst, const xlbhip_field* bcm, const xlbhip_field* miss) {
  if (!s->has_outflow) return 0;
  xlbhip_ctx* c = s->ctx;
  const size_t n = dst->cells();
  return by_lattice(s->lattice, [&](auto L) {
    using LL = decltype(L);
    if (s->cdt == XLBHIP_F32)
      hipLaunchKernelGGL((k_outflow_aux<LL, float>), blocks_for(n), 256, 0, c->stream, view(src), view(dst), view(bcm), view(miss), dims(dst),
                         s->tab_kind, static_cast<const float*>(s->tab_values), s->prof_keys, static_cast<const float*>(s->prof_vals), s->n_prof);
    else
      hipLaunchKernelGGL((k_outflow_aux<LL, double>), blocks_for(n), 256, 0, c->stream, view(src), view(dst), view(bcm), view(miss), dims(dst),
                         s->tab_kind, static_cast<const double*>(s->tab_values), s->prof_keys, static_cast<const double*>(s->prof_vals), s->n_prof);
    XLB_HIP(hipGetLastError());
    return 0;
  });
}

static int launch_any(const xlbhip_stepper* s, const StepLaunch& p);
static void drop_clean_cache(xlbhip_stepper* s) {
  for (auto& kv : s->clean_cache) (void)hipFree(kv.second);
  s->clean_cache.clear();
}

static int launch_step2(xlbhip_stepper* s, StepLaunch p) {
  if (s->lattice == XLBHIP_D3Q27 && s->collision == XLBHIP_KBC) return launch_step2_d3q27_kbc(p);
  p.clean = nullptr;
  if (p.has_bc && p.meta && opt(s->ctx, "fuse2_clean", 1)) {
    // everything the block -> (tile, x-segment) mapping depends on: the flags say "no boundary cell in THIS block's item"
    const std::array<int, 8> key = {p.x_begin, p.x_count, p.x_segments * 64 + p.x_cap, (p.tile_order ? 1 : 0) + 2 * p.tile_oy + 1024 * p.tile_oz,
                                    p.xcd_swizzle, p.tile_order ? s->order_mode : -1, p.tile_ty, p.tile_tz};
    auto it = s->clean_cache.find(key);
    if (it == s->clean_cache.end()) {
      uint8_t* flags = nullptr;
      XLB_HIP(hipMalloc(&flags, (size_t)step2_items(p)));
      if (int rc = step2_build_clean(p, flags)) {
        (void)hipFree(flags);
        return rc;
      }
      it = s->clean_cache.emplace(key, flags).first;
    }
    p.clean = it->second;
  }
  if (s->lattice == XLBHIP_D3Q27) return launch_step2_d3q27_bgk(p);
  return p.strips ? launch_step2_d3q19_bgk_strips(p) : launch_step2_d3q19_bgk(p);
}

// strip buffer of a population field (1 / 32 of it): allocated on first use; false (and no error) when there is no memory for it
static bool ensure_strips(xlbhip_field* f) {
  if (f->strips) return true;
  const size_t bytes = f->planes * f->plane_stride * dtype_size(f->dtype) / 32 + 512;
  if (hipMalloc(&f->strips, bytes) != hipSuccess) {
    f->strips = nullptr;
    (void)hipGetLastError();
    return false;
  }
  f->strips_version = 0;
  f->strips_oz = -1;
  return true;
}

// Pair of steps for a stepper whose Zou-He / Regularized / outflow cells all sit in the planes x = 0 and x = nx - 1
// (inlet / outlet faces): the two-step kernel updates the planes 2 .. nx-3, whose two-step cone never evaluates such a
// cell (its f(t+1) on the planes 1 and nx-2 only PULLS from the end planes), and the four end planes go through the
// single-step kernel twice with a third population field holding their f(t+1).
static int step_twice_edge_ext(xlbhip_stepper* s, StepLaunch p, const xlbhip_field* src, xlbhip_field* dst, const xlbhip_field* bcm,
                               const xlbhip_field* miss, double omega) {
  XLB_REQUIRE(s->scratch && s->scratch->plane_stride == src->plane_stride, "scratch field missing (can_fuse2 allocates it)");
  const int nx = src->nx;
  p.x_begin = 2;
  p.x_count = nx - 4;
  p.x_segments = fuse2_segments(s, p);
  if (int rc = launch_step2_d3q19_bgk(p)) return rc;
  // end planes, step 1: f(t+1) on the planes nx-3 .. nx-1 and 0 .. 2 -> scratch
  StepLaunch q = make_launch(s, src, s->scratch, bcm, miss, omega);
  q.x_begin = nx - 3;
  q.x_count = 3;
  if (int rc = launch_any(s, q)) return rc;
  q.x_begin = 0;
  if (int rc = launch_any(s, q)) return rc;
  if (int rc = outflow_aux(s, src, s->scratch, bcm, miss)) return rc;
  // step 2: f(t+2) on the planes nx-2, nx-1, 0, 1 -> dst
  StepLaunch r = make_launch(s, s->scratch, dst, bcm, miss, omega);
  r.x_begin = nx - 2;
  r.x_count = 2;
  if (int rc = launch_any(s, r)) return rc;
  r.x_begin = 0;
  if (int rc = launch_any(s, r)) return rc;
  return outflow_aux(s, s->scratch, dst, bcm, miss);
}

// the compute stream waits for the halo exchange; with the telemetry on, the wait is bracketed by two timing events
static int harvest_wait(xlbhip_ctx* c, int slot) {
  if (!c->wait_used[slot]) return 0;
  float ms = 0.f;
  XLB_HIP(hipEventSynchronize(c->ev_w1[slot]));
  XLB_HIP(hipEventElapsedTime(&ms, c->ev_w0[slot], c->ev_w1[slot]));
  c->halo_wait_ms += ms;
  c->halo_waits += 1;
  c->wait_used[slot] = false;
  return 0;
}

static int wait_for_halo(xlbhip_ctx* c) {
  if (!opt(c, "halo_telemetry", 1)) {
    XLB_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0));
    return 0;
  }
  const int slot = c->wait_head;
  c->wait_head = (c->wait_head + 1) % xlbhip_ctx::WAIT_RING;
  if (int rc = harvest_wait(c, slot)) return rc;  // (32 exchanges old: long complete)
  if (!c->ev_w0[slot]) {
    XLB_HIP(hipEventCreate(&c->ev_w0[slot]));
    XLB_HIP(hipEventCreate(&c->ev_w1[slot]));
  }
  XLB_HIP(hipEventRecord(c->ev_w0[slot], c->stream));
  XLB_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0));
  XLB_HIP(hipEventRecord(c->ev_w1[slot], c->stream));
  c->wait_used[slot] = true;
  return 0;
}

// two steps in one pass (a -> scratch-free: src -> dst holds f(t+2)); caller checked eligibility
static int step_twice(xlbhip_stepper* s, const xlbhip_field* src, xlbhip_field* dst, const xlbhip_field* bcm, const xlbhip_field* miss,
                      double omega) {
  StepLaunch p = make_launch(s, src, dst, bcm, miss, omega);
  p.meta = s->meta;
  // hull tiles first only pays when they are much more expensive than fluid tiles (halfway walls: redirected
  // loads); with fullway / equilibrium boundaries the XCD-compact order is faster (fuse2_lpt: 0 never, 1 auto, 2 always)
  const int64_t lpt = opt(s->ctx, "fuse2_lpt", 1);
  const bool clean_on = p.has_bc && opt(s->ctx, "fuse2_clean", 1) != 0;
  p.tile_order = (p.has_bc && (lpt >= 2 || (lpt == 1 && (s->needs_missing || clean_on)))) ? s->tile_order : nullptr;
  p.x_segments = fuse2_segments(s, p);
  p.x_cap = clean_on ? (int)opt(s->ctx, "fuse2_xcap", 8) : 0;
  if (p.has_bc && opt(s->ctx, "fuse2_shift", 1)) {  // half-tile shift: both walls of an axis in one (wrapping) tile row
    p.tile_oy = p.tile_ty / 2;
    p.tile_oz = p.tile_tz / 2;
  }
  p.xcd_swizzle = (int)opt(s->ctx, "fuse2_xcd", 1);
  xlbhip_ctx* c = s->ctx;
  p.strips = 0;
  p.strips_src = nullptr;
  p.strips_dst = nullptr;
  touch(dst);  // (new contents: whatever was cached on the old ones — its strip buffer — is stale)
  if (s->edge_ext_ok) return step_twice_edge_ext(s, p, src, dst, bcm, miss, omega);
  // strip buffers (step2_kernel.hpp): phase A's halo columns come from src's strips, phase B writes dst's.  D3Q19, the
  // bit-exact body, (8 x 64) tiles; a field whose strips are not those of its current contents gets them rebuilt first.
  xlbhip_field* srcw = const_cast<xlbhip_field*>(src);
  const bool native_slab = src->halo > 0 && !opt(c, "external_halo", 0);
  // With boundary conditions only: there they buy 1-3 % (cavity 512^3, interleaved A/B: halfway 2.353 -> 2.334, fullway 2.220 -> 2.149
  // ms/step); the BC-free kernel is faster with row-aligned lanes alone (2.12 against 2.18 with strips, 2.29 before: profiles/r03/step2_strips.md).
  // fuse2_strips = 2 forces them for every D3Q19 stepper.
  const int64_t strips_opt = opt(c, "fuse2_strips", 1);
  const bool strips = (strips_opt == 2 || (strips_opt == 1 && p.has_bc)) && s->lattice == XLBHIP_D3Q19 && !p.fast_bgk && p.tile_ty == 8 && p.tile_tz == 64 &&
                      (src->halo == 0 || native_slab) && src->nx >= 8 && ensure_strips(srcw) && ensure_strips(dst);
  // q writes dst's strips, and reads src's when they are those of src's current contents (all interior planes).  After anything but
  // a strip-writing pass wrote src — a single step, an upload — the first pass only WRITES strips (no separate rebuild pass: at 512^3
  // that would cost 1.5 ms, a third of a pair, inside e.g. the driver's 20-step timed region after its 5 warm-up steps).
  auto read_strips = [&](StepLaunch& q) -> int {
    const bool valid = srcw->strips_version == srcw->version && srcw->strips_oz == q.tile_oz;
    q.strips = valid ? 3 : 2;
    q.strips_src = valid ? srcw->strips : nullptr;
    q.strips_dst = dst->strips;
    return 0;
  };
  const bool rowmap_only = !strips && opt(c, "fuse2_rowmap", 0) != 0 && p.has_bc && s->lattice == XLBHIP_D3Q19 && !p.fast_bgk && p.tile_ty == 8 && p.tile_tz == 64;
  if (rowmap_only) p.strips = 4;
  auto dst_strips_done = [&]() {  // every interior plane of dst was written by strip-writing launches
    dst->strips_version = dst->version;
    dst->strips_oz = p.tile_oz;
  };
  if (src->halo == 0 || opt(c, "external_halo", 0)) {
    if (strips) {
      if (int rc = read_strips(p)) return rc;
      if (int rc = launch_step2(s, p)) return rc;
      dst_strips_done();
      return 0;
    }
    return launch_step2(s, p);
  }
  // slab protocol for a PAIR of steps: the two ghost planes per side of src are refilled on the comm stream
  // (comm.cpp, depth 2) while the planes whose two-step cone stays inside the slab are updated; the two edge
  // plane pairs follow (each warms its own 3-plane window up from the fresh ghosts).
  const bool overlap = opt(c, "overlap", 1) != 0 && src->nx >= 16;
  XLB_HIP(hipEventRecord(c->ev_edge, c->stream));  // src complete (previous pair)
  StepLaunch whole = p;
  if (strips) {  // launches whose phase A pulls from ghost planes (no strips there) only WRITE strips
    whole.strips = 2;
    whole.strips_dst = dst->strips;
  }
  if (overlap) {
    // the interior launch goes out BEFORE the exchange is enqueued: posting an exchange costs host time (dozens of copy /
    // send calls, some of which the runtime may only accept once earlier work of the communication stream has finished —
    // measured with the ipc transport: 1 ms of exposed wait per pair when the launch came second) and the device must
    // already have the interior to work on meanwhile
    p.x_begin = 2;
    p.x_count = src->nx - 4;
    p.x_segments = fuse2_segments(s, p);
    if (strips)
      if (int rc = read_strips(p)) return rc;
    if (int rc = launch_step2(s, p)) return rc;
  }
  XLB_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_edge, 0));
  if (int rc = halo_exchange_on(c, s->lattice, srcw, c->comm_stream, 2)) return rc;
  XLB_HIP(hipEventRecord(c->ev_halo, c->comm_stream));
  if (!overlap) {
    XLB_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0));
    if (int rc = launch_step2(s, whole)) return rc;
    if (strips) dst_strips_done();
    return 0;
  }
  if (int rc = wait_for_halo(c)) return rc;
  StepLaunch edge = whole;
  edge.x_segments = 1;
  edge.x_count = 2;
  edge.x_begin = 0;
  if (int rc = launch_step2(s, edge)) return rc;
  edge.x_begin = src->nx - 2;
  if (int rc = launch_step2(s, edge)) return rc;
  if (strips) dst_strips_done();
  return 0;
}

static bool can_fuse2(xlbhip_stepper* s, const xlbhip_field* src, xlbhip_field* dst, const xlbhip_field* bcm, const xlbhip_field* miss) {
  const int64_t mode = opt(s->ctx, "fuse2", 1);
  if (mode == 0 || s->forced) return false;
  s->edge_ext_ok = false;
  if (s->has_edge_kinds) {
    // Zou-He / Regularized / outflow / do-nothing cells: fine when they all sit in the two x end planes (scan of bc_mask, 1 B / cell)
    if (!bcm || src->halo != 0 || src->nx < 16 || s->lattice != XLBHIP_D3Q19) return false;
    xlbhip_ctx* c = s->ctx;
    if (s->scan_field != bcm || s->scan_version != bcm->version) {  // one scan per (stepper, bc_mask contents), not per run
      int* dflag = nullptr;
      int flag = 1;
      if (hipMalloc(&dflag, sizeof(int)) != hipSuccess) return false;
      (void)hipMemsetAsync(dflag, 0, sizeof(int), c->stream);
      hipLaunchKernelGGL(k_ext_interior_scan, blocks_for(bcm->cells()), 256, 0, c->stream, view(bcm), s->tab_kind, dims(bcm), dflag);
      if (hipMemcpyAsync(&flag, dflag, sizeof(int), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
        flag = 1;
      (void)hipFree(dflag);
      s->scan_field = bcm;
      s->scan_version = bcm->version;
      s->scan_flag = flag;
    }
    if (s->scan_flag != 0) return false;
    // the end planes need a third population field; without the memory for it the stepper stays on single steps
    if (!s->scratch || s->scratch->nx != src->nx || s->scratch->ny != src->ny || s->scratch->nz != src->nz || s->scratch->dtype != src->dtype) {
      if (s->scratch) xlbhip_field_destroy(s->scratch);
      s->scratch = nullptr;
      if (xlbhip_field_create(c, src->card, src->nx, src->ny, src->nz, src->dtype, src->halo, 0.0, &s->scratch) != 0 ||
          s->scratch->plane_stride != src->plane_stride) {
        if (s->scratch) xlbhip_field_destroy(s->scratch);
        s->scratch = nullptr;
        (void)hipGetLastError();
        return false;
      }
    }
    s->edge_ext_ok = true;
  }
  StepLaunch p = make_launch(s, src, dst, bcm, miss, 1.0);
  if (!step2_eligible(p, s->lattice, s->collision)) return false;
  if (mode == 1) {
    // D3Q27 KBC pairs on request only (fuse2 = 2): ~1100 (fp32) / 810 (fast fp64) VALU instructions per cell and 8 waves per CU make
    // the two-step form issue-bound — 384^3: 2.41 (FP64FP32) / 2.52 (FP32FP32) ms per step against 2.21 / 2.16 of the HBM-bound
    // single-step kernel (profiles/r02/d3q27_kbc_two_step.txt)
    // Round 3: the gamma reduction in fp32 (cell.hpp COLL_G32) speeds the fp64 / fp32-store pairs up by 5.6 % — and single steps by 3.1 % —
    // on a non-trivial state: pairs 2.28 against 2.23 ms per step at 384^3, still behind (profiles/r03/kbc_gamma32.md; on a uniform
    // f = w the pairs look 3 % FASTER than single steps: identical operands in every lane, higher clocks — not a state to time on).
    if (s->lattice == XLBHIP_D3Q27 && s->collision == XLBHIP_KBC) return false;
    // D3Q27 BGK with boundary conditions (round 3, (8 x 48) tiles, 192 VGPRs): bit-exact, but 2.59-3.16 against 2.15-2.19 ms per step on
    // the 384^3 cavity (0.48-0.59 against 0.70-0.71 of the roofline, profiles/r03/d3q27_walls_two_step.md): on request only
    if (s->lattice == XLBHIP_D3Q27 && s->collision == XLBHIP_BGK && p.has_bc) return false;
    // one block per CU marches an (8 x 64) tile column segment: the work items must fill the chip in whole
    // rounds (128^3 = 32 tiles x 4 segments would leave half of the 256 CUs idle)
    const long items = (long)(p.ny / p.tile_ty) * (p.nz / p.tile_tz) * fuse2_segments(s, p), cus = fill_cus(s->ctx);
    const long rounds = (items + cus - 1) / cus;
    if (items * 100 < rounds * cus * 85) return false;
    // halfway walls make the hull tiles ~1.5x as expensive as fluid tiles; when most tiles are hull tiles two single
    // steps are faster (256^3, 53 % hull tiles: fused 41.7 vs 38.2 GLUPS; thinner domains lose)
    if (s->needs_missing) {
      const long tys = p.ny / p.tile_ty, tzs = p.nz / p.tile_tz;
      // (with the half-tile shift both walls of an axis share one tile row: tys + tzs - 1 hull tiles)
      const long hull = opt(s->ctx, "fuse2_shift", 1) ? std::min(tys * tzs, tys + tzs - 1)
                                                      : tys * tzs - (tys > 2 ? tys - 2 : 0) * (tzs > 2 ? tzs - 2 : 0);
      if (hull * 100 > tys * tzs * 60) return false;
    }
  }
  return true;
}

// per-run tables of the two-step kernel: the meta words (bc kind | slot | missing bits per cell, ghost planes
// included) and the hull-first tile order
static int prepare_fuse2(xlbhip_stepper* s, const xlbhip_field* bcm, const xlbhip_field* miss) {
  if (!(s->n_bc > 0 && bcm)) return 0;
  xlbhip_ctx* c = s->ctx;
  const size_t cells = bcm->cells_with_halo();
  if (s->meta_cells != cells) {
    if (s->meta) {
      comm_forget_buffer(c, s->meta);
      XLB_HIP(hipFree(s->meta));
    }
    s->meta = nullptr;
    XLB_HIP(hipMalloc(&s->meta, cells * sizeof(uint32_t)));
    s->meta_cells = cells;
    s->meta_bc = nullptr;  // (contents gone: rebuild below)
  }
  // (the tile of THIS stepper: make_launch — D3Q27 with boundary conditions marches (8 x 48) tiles)
  const int tile_tz = s->lattice == XLBHIP_D3Q27 ? 48 : fuse2_tile_tz(c);
  const int tys = bcm->ny / fuse2_tile_ty(c), tzs = bcm->nz / tile_tz;
  const bool shifted = opt(c, "fuse2_shift", 1) != 0;
  const int order_mode = (opt(c, "fuse2_lpt", 1) == 3 ? 3 : 0) + (shifted ? 8 : 0);
  if (s->order_ty != tys || s->order_tz != tzs || s->order_mode != order_mode) {
    s->order_mode = order_mode;
    // hull tiles first (the expensive ones when there are walls), then the interior; both lists are dealt so that every
    // XCD (block i runs on XCD i % 8) works on a CONTIGUOUS run of tiles — neighbours share their halo rows / lines
    // through that XCD's L2 (fuse2_lpt = 3: the hull in row-major order as in round 1, for A/B)
    std::vector<uint32_t> hull, inner;
    if (shifted) {
      // half-tile shift: the walls of the y / z faces sit in the LAST tile row / column (the ones that wrap around)
      for (int tz = 0; tz < tzs; ++tz) hull.push_back((uint32_t)((tys - 1) * tzs + tz));
      for (int ty = tys - 2; ty >= 0; --ty) hull.push_back((uint32_t)(ty * tzs + tzs - 1));
      for (int ty = 0; ty < tys - 1; ++ty)
        for (int tz = 0; tz < tzs - 1; ++tz) inner.push_back((uint32_t)(ty * tzs + tz));
    } else if (opt(c, "fuse2_lpt", 1) == 3) {
      for (int ty = 0; ty < tys; ++ty)
        for (int tz = 0; tz < tzs; ++tz)
          if (ty == 0 || ty == tys - 1 || tz == 0 || tz == tzs - 1) hull.push_back((uint32_t)(ty * tzs + tz));
    } else {
      // walk around the perimeter: consecutive entries are adjacent tiles
      for (int tz = 0; tz < tzs; ++tz) hull.push_back((uint32_t)tz);
      for (int ty = 1; ty < tys - 1; ++ty)
        if (tzs > 1) hull.push_back((uint32_t)(ty * tzs + tzs - 1));
      if (tys > 1)
        for (int tz = tzs - 1; tz >= 0; --tz) hull.push_back((uint32_t)((tys - 1) * tzs + tz));
      for (int ty = tys - 2; ty >= 1; --ty) hull.push_back((uint32_t)(ty * tzs));
    }
    if (!shifted)
      for (int ty = 1; ty < tys - 1; ++ty)
        for (int tz = 1; tz < tzs - 1; ++tz) inner.push_back((uint32_t)(ty * tzs + tz));
    std::vector<uint32_t> order;
    order.reserve((size_t)tys * tzs);
    auto deal = [&](const std::vector<uint32_t>& list, bool chunked) {
      const size_t n = list.size(), per = (n + 7) / 8;
      if (!chunked) {
        order.insert(order.end(), list.begin(), list.end());
        return;
      }
      // chunk k = list[k * per ...]; the slot being filled decides the XCD (slot % 8) and takes the next tile of that
      // XCD's chunk (of the fullest chunk once its own is used up)
      size_t cur[8], end[8];
      for (size_t k = 0; k < 8; ++k) {
        cur[k] = std::min(n, k * per);
        end[k] = std::min(n, (k + 1) * per);
      }
      for (size_t done = 0; done < n; ++done) {
        size_t k = order.size() % 8;
        if (cur[k] == end[k])
          for (size_t m = 0; m < 8; ++m)
            if (end[m] - cur[m] > end[k] - cur[k]) k = m;
        order.push_back(list[cur[k]++]);
      }
    };
    deal(hull, opt(c, "fuse2_lpt", 1) != 3);
    deal(inner, true);
    drop_clean_cache(s);  // the flags were computed for the old block -> tile mapping
    if (s->tile_order) XLB_HIP(hipFree(s->tile_order));
    s->tile_order = nullptr;
    XLB_HIP(hipMalloc(&s->tile_order, order.size() * sizeof(uint32_t)));
    XLB_HIP(hipMemcpy(s->tile_order, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    s->order_ty = tys;
    s->order_tz = tzs;
  }
  const unsigned long long opts_key = (unsigned long long)opt(c, "fuse2_shift", 1) | ((unsigned long long)opt(c, "external_halo", 0) << 1);
  if (s->meta_bc == bcm && s->meta_miss == miss && s->meta_bc_version == bcm->version && s->meta_miss_version == (miss ? miss->version : 0) &&
      s->meta_opts == opts_key)
    return 0;  // meta words, tile order and clean flags are those of these very masks
  s->meta_bc = bcm;
  s->meta_miss = miss;
  s->meta_bc_version = bcm->version;
  s->meta_miss_version = miss ? miss->version : 0;
  s->meta_opts = opts_key;
  drop_clean_cache(s);  // (stream-ordered: the flags' last readers were enqueued before this point and hipFree synchronises)
  hipLaunchKernelGGL(k_build_meta, blocks_for(cells), 256, 0, c->stream, static_cast<const uint8_t*>(bcm->data),
                     miss ? static_cast<const uint32_t*>(miss->data) : nullptr, s->meta, cells, s->ids_packed, s->kinds_packed,
                     s->moving_mask, s->lattice == XLBHIP_D3Q27 ? 1 : 0);
  XLB_HIP(hipGetLastError());
  // slab decomposition: phase A also runs on the ghost planes -1 and nx, so it needs the neighbours' boundary
  // information there.  Host-staged transports (external_halo) fill the ghost planes of the masks themselves.
  if (bcm->halo > 0 && !opt(c, "external_halo", 0))
    return plane_exchange_on(c, s->meta, sizeof(uint32_t), bcm->nx, bcm->ny, bcm->nz, bcm->halo, c->stream);
  return 0;
}

// the step kernel(s) of one step src -> dst, with the slab halo protocol when the fields carry ghost planes
static int step_kernels(xlbhip_stepper* s, const xlbhip_field* src, xlbhip_field* dst, const xlbhip_field* bcm, const xlbhip_field* miss,
                     double omega) {
  xlbhip_ctx* c = s->ctx;
  StepLaunch p = make_launch(s, src, dst, bcm, miss, omega);
  if (src->halo == 0) {
    p.x_begin = 0;
    p.x_count = src->nx;
    return launch_any(s, p);
  }
  if (opt(c, "external_halo", 0)) {
    p.x_begin = 0;
    p.x_count = src->nx;
    return launch_any(s, p);
  }
  // slab protocol: ghosts of src are (re)filled from the ring neighbours on the comm stream while
  // the planes that do not touch a ghost are updated; the two edge planes follow.
  // The launch that runs BEFORE this step's exchange has completed must not write a plane a neighbour may still be pulling
  // (ipc transport: the puller, not the owner, knows when a pull is done; what orders the two is that the owner's edge launches
  // wait for the NEXT exchange, which the neighbour posts after its pulls).  On fields with two ghost planes the previous exchange
  // may have been a fused pair's — planes 0, 1, nx - 2, nx - 1 of `dst` lent out — so their "edge" is two planes wide.
  const int edge = src->halo >= 2 ? 2 : 1;
  const bool overlap = opt(c, "overlap", 1) != 0 && src->nx > 2 * edge;
  XLB_HIP(hipEventRecord(c->ev_edge, c->stream));  // src complete (previous step)
  if (overlap) {  // interior first, then the exchange is posted (see step_twice)
    p.x_begin = edge;
    p.x_count = src->nx - 2 * edge;
    if (int rc = launch_any(s, p)) return rc;
  }
  XLB_HIP(hipStreamWaitEvent(c->comm_stream, c->ev_edge, 0));
  if (int rc = halo_exchange_on(c, s->lattice, const_cast<xlbhip_field*>(src), c->comm_stream)) return rc;
  XLB_HIP(hipEventRecord(c->ev_halo, c->comm_stream));
  if (overlap) {
    if (int rc = wait_for_halo(c)) return rc;
    p.x_begin = 0;
    p.x_count = edge;
    if (int rc = launch_any(s, p)) return rc;
    p.x_begin = src->nx - edge;
    return launch_any(s, p);
  }
  XLB_HIP(hipStreamWaitEvent(c->stream, c->ev_halo, 0));
  p.x_begin = 0;
  p.x_count = src->nx;
  return launch_any(s, p);
}

// one step src -> dst; ExtrapolationOutflowBC cells get their auxiliary data afterwards (nse_stepper.py:270-272)
static int step_once(xlbhip_stepper* s, const xlbhip_field* src, xlbhip_field* dst, const xlbhip_field* bcm, const xlbhip_field* miss,
                     double omega) {
  touch(dst);  // (its strip buffer, if any, no longer matches)
  if (int rc = step_kernels(s, src, dst, bcm, miss, omega)) return rc;
  return outflow_aux(s, src, dst, bcm, miss);
}

}  // namespace xlb

extern "C" {

int xlbhip_stepper_create(xlbhip_ctx* c, int lattice, int collision, int cdt, int sdt, int n_bc, const xlbhip_bc_desc* bcs,
                          xlbhip_stepper** out) {
  XLB_REQUIRE(c && out, "null argument");
  XLB_REQUIRE(lattice_q(lattice) > 0, "unknown lattice %d", lattice);
  XLB_REQUIRE(collision == XLBHIP_BGK || collision == XLBHIP_KBC || collision == XLBHIP_SMAGORINSKY_LES_BGK, "unknown collision %d", collision);
  XLB_REQUIRE(!(collision == XLBHIP_KBC && lattice == XLBHIP_D3Q19), "Velocity set not supported: D3Q19 has no KBC (reference kbc.py:65-66)");
  XLB_REQUIRE(cdt == XLBHIP_F32 || cdt == XLBHIP_F64, "bad compute dtype %d", cdt);
  XLB_REQUIRE(is_float(sdt) && dtype_size(sdt) <= dtype_size(cdt), "bad store dtype %d for compute dtype %d", sdt, cdt);
  XLB_REQUIRE(n_bc == 0 || bcs, "null bc list");
  std::vector<uint8_t> kind(256, 0);
  const int q = lattice_q(lattice);
  std::vector<double> vals(256 * 27, 0.0);
  bool needs_missing = false, extended = false, has_outflow = false, has_edge_kinds = false;
  for (int i = 0; i < n_bc; ++i) {
    const xlbhip_bc_desc& b = bcs[i];
    XLB_REQUIRE(b.id >= 1 && b.id <= 255, "bc id %d out of range 1..255", b.id);
    XLB_REQUIRE(b.kind >= XLBHIP_BC_EQUILIBRIUM && b.kind <= XLBHIP_BC_HALFWAY_BB_PROFILE, "unknown bc kind %d", b.kind);
    XLB_REQUIRE(b.kind < XLBHIP_BC_HYBRID_BB_REGULARIZED || b.kind > XLBHIP_BC_HYBRID_NEQ_REGULARIZED || lattice_d(lattice) == 3,
                "This BC is not implemented in 2D!");  // bc_hybrid.py:119-120
    if (b.kind == XLBHIP_BC_EXTRAPOLATION_OUTFLOW) has_outflow = true;
    if (b.kind >= XLBHIP_BC_ZOUHE_VELOCITY || b.kind == XLBHIP_BC_DO_NOTHING) has_edge_kinds = true;
    XLB_REQUIRE(kind[b.id] == 0, "bc id %d used twice", b.id);
    if (b.kind >= XLBHIP_BC_ZOUHE_VELOCITY) extended = needs_missing = true;
    kind[b.id] = (uint8_t)b.kind;
    for (int l = 0; l < q; ++l) vals[b.id * 27 + l] = b.values[l];
    if (b.kind == XLBHIP_BC_HALFWAY_BB) needs_missing = true;
  }
  unsigned long long ids_packed = 0;
  unsigned kinds_packed = 0;
  unsigned moving_mask = 0;
  for (int i = 0; i < n_bc && i < 8; ++i) {
    ids_packed |= (unsigned long long)(bcs[i].id & 0xff) << (8 * i);
    kinds_packed |= (unsigned)(bcs[i].kind & 0xf) << (4 * i);
    if (bcs[i].kind == XLBHIP_BC_HALFWAY_BB)
      for (int l = 0; l < q; ++l)
        if (bcs[i].values[l] != 0.0) moving_mask |= 1u << i;
  }
  XLB_HIP(hipSetDevice(c->device));
  xlbhip_stepper* s = new xlbhip_stepper();
  s->ctx = c;
  s->lattice = lattice;
  s->collision = collision;
  s->cdt = cdt;
  s->sdt = sdt;
  s->n_bc = n_bc;
  s->needs_missing = needs_missing;
  s->extended_bcs = extended;
  s->has_outflow = has_outflow;
  s->has_edge_kinds = has_edge_kinds;
  s->ids_packed = ids_packed;
  s->kinds_packed = kinds_packed;
  s->moving_mask = moving_mask;
  XLB_HIP(hipMalloc(&s->tab_kind, 256));
  XLB_HIP(hipMemcpy(s->tab_kind, kind.data(), 256, hipMemcpyHostToDevice));
  if (cdt == XLBHIP_F32) {
    std::vector<float> v32(vals.begin(), vals.end());
    XLB_HIP(hipMalloc(&s->tab_values, v32.size() * 4));
    XLB_HIP(hipMemcpy(s->tab_values, v32.data(), v32.size() * 4, hipMemcpyHostToDevice));
  } else {
    XLB_HIP(hipMalloc(&s->tab_values, vals.size() * 8));
    XLB_HIP(hipMemcpy(s->tab_values, vals.data(), vals.size() * 8, hipMemcpyHostToDevice));
  }
  *out = s;
  return 0;
}

int xlbhip_stepper_set_bc_profile(xlbhip_stepper* s, int bc_id, int64_t n, const uint32_t* storage_cells, const double* values) {
  XLB_REQUIRE(s && bc_id >= 1 && bc_id <= 255, "bad argument");
  XLB_REQUIRE(n == 0 || (storage_cells && values), "null table");
  xlbhip_ctx* c = s->ctx;
  XLB_HIP(hipSetDevice(c->device));
  XLB_HIP(hipStreamSynchronize(c->stream));
  for (int64_t i = 0; i < n; ++i) s->prof_host[storage_cells[i]] = {values[3 * i], values[3 * i + 1], values[3 * i + 2]};
  // sorted device image (std::map iterates in key order)
  std::vector<uint32_t> keys;
  std::vector<double> v64;
  keys.reserve(s->prof_host.size());
  v64.reserve(3 * s->prof_host.size());
  for (const auto& kv : s->prof_host) {
    keys.push_back(kv.first);
    v64.insert(v64.end(), kv.second.begin(), kv.second.end());
  }
  if (s->prof_keys) XLB_HIP(hipFree(s->prof_keys));
  if (s->prof_vals) XLB_HIP(hipFree(s->prof_vals));
  s->prof_keys = nullptr;
  s->prof_vals = nullptr;
  s->n_prof = (int)keys.size();
  if (s->n_prof > 0) {
    XLB_HIP(hipMalloc(&s->prof_keys, keys.size() * sizeof(uint32_t)));
    XLB_HIP(hipMemcpy(s->prof_keys, keys.data(), keys.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    if (s->cdt == XLBHIP_F32) {
      std::vector<float> v32(v64.begin(), v64.end());
      XLB_HIP(hipMalloc(&s->prof_vals, v32.size() * 4));
      XLB_HIP(hipMemcpy(s->prof_vals, v32.data(), v32.size() * 4, hipMemcpyHostToDevice));
    } else {
      XLB_HIP(hipMalloc(&s->prof_vals, v64.size() * 8));
      XLB_HIP(hipMemcpy(s->prof_vals, v64.data(), v64.size() * 8, hipMemcpyHostToDevice));
    }
  }
  // flag the BC: its prescribed values come from the table (cell.hpp: PROF_FLAG)
  const size_t es = s->cdt == XLBHIP_F32 ? 4 : 8;
  const float one32 = 1.0f;
  const double one64 = 1.0;
  XLB_HIP(hipMemcpy(static_cast<char*>(s->tab_values) + ((size_t)bc_id * 27 + PROF_FLAG) * es, es == 4 ? (const void*)&one32 : (const void*)&one64, es,
                    hipMemcpyHostToDevice));
  return 0;
}

int xlbhip_stepper_momentum_transfer(xlbhip_stepper* s, int bc_id, const xlbhip_field* f_0, const xlbhip_field* bcm, const xlbhip_field* miss,
                                     double force_out[3]) {
  XLB_REQUIRE(s && force_out && bc_id >= 1 && bc_id <= 255, "bad argument");
  xlbhip_ctx* c = s->ctx;
  XLB_CHECK_POP(f_0, s->lattice, "momentum_transfer(f_0)");
  XLB_REQUIRE(bcm && bcm->dtype == XLBHIP_U8 && bcm->card == 1 && same_grid(bcm, f_0) && bcm->halo == f_0->halo, "momentum_transfer: bad bc_mask field");
  XLB_REQUIRE(miss && miss->dtype == XLBHIP_MISSING && same_grid(miss, f_0) && miss->halo == f_0->halo, "momentum_transfer: needs the missing_mask field");
  XLB_REQUIRE(f_0->halo == 0, "momentum_transfer through the stepper's tables: fields without ghost planes (mesh / profile BCs live on one rank)");
  uint8_t kind = 0;
  XLB_HIP(hipSetDevice(c->device));
  XLB_HIP(hipMemcpy(&kind, s->tab_kind + bc_id, 1, hipMemcpyDeviceToHost));
  XLB_REQUIRE((kind >= XLBHIP_BC_HYBRID_BB_REGULARIZED && kind <= XLBHIP_BC_HYBRID_NEQ_REGULARIZED) || kind == XLBHIP_BC_HALFWAY_BB_PROFILE,
              "momentum_transfer through the stepper: bc %d is of kind %d (HybridBC / profile walls; plain walls use xlbhip_momentum_transfer)", bc_id,
              (int)kind);
  double* dforce = nullptr;
  XLB_HIP(hipMalloc(&dforce, 3 * sizeof(double)));
  XLB_HIP(hipMemsetAsync(dforce, 0, 3 * sizeof(double), c->stream));
  const size_t n = f_0->cells();
  int rc = by_lattice(s->lattice, [&](auto L) {
    using LL = decltype(L);
    if (s->cdt == XLBHIP_F32)
      hipLaunchKernelGGL((k_momentum_transfer_tab<LL, float>), blocks_for(n), 256, 0, c->stream, view(f_0), view(bcm), view(miss), dims(f_0), bc_id,
                         s->tab_kind, static_cast<const float*>(s->tab_values), s->prof_keys, static_cast<const float*>(s->prof_vals), s->n_prof,
                         s->dist_keys, s->dist_vals, s->n_dist, dforce);
    else
      hipLaunchKernelGGL((k_momentum_transfer_tab<LL, double>), blocks_for(n), 256, 0, c->stream, view(f_0), view(bcm), view(miss), dims(f_0), bc_id,
                         s->tab_kind, static_cast<const double*>(s->tab_values), s->prof_keys, static_cast<const double*>(s->prof_vals), s->n_prof,
                         s->dist_keys, s->dist_vals, s->n_dist, dforce);
    XLB_HIP(hipGetLastError());
    return 0;
  });
  if (rc == 0) {
    hipError_t e = hipMemcpyAsync(force_out, dforce, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
      (void)hipFree(dforce);
      XLB_FAIL("momentum_transfer: %s", hipGetErrorString(e));
    }
  }
  (void)hipFree(dforce);
  return rc;
}

int xlbhip_stepper_set_bc_distances(xlbhip_stepper* s, int64_t n, const uint32_t* storage_cells, const float* weights) {
  XLB_REQUIRE(s, "stepper is null");
  XLB_REQUIRE(n == 0 || (storage_cells && weights), "null table");
  xlbhip_ctx* c = s->ctx;
  const int q = lattice_q(s->lattice);
  XLB_HIP(hipSetDevice(c->device));
  XLB_HIP(hipStreamSynchronize(c->stream));
  for (int64_t i = 0; i < n; ++i) {
    std::array<float, 27> w{};
    for (int l = 0; l < q; ++l) w[(size_t)l] = weights[i * q + l];
    s->dist_host[storage_cells[i]] = w;
  }
  std::vector<uint32_t> keys;
  std::vector<float> vals;
  keys.reserve(s->dist_host.size());
  vals.reserve(s->dist_host.size() * (size_t)q);
  for (const auto& kv : s->dist_host) {  // std::map iterates in key order
    keys.push_back(kv.first);
    vals.insert(vals.end(), kv.second.begin(), kv.second.begin() + q);
  }
  if (s->dist_keys) XLB_HIP(hipFree(s->dist_keys));
  if (s->dist_vals) XLB_HIP(hipFree(s->dist_vals));
  s->dist_keys = nullptr;
  s->dist_vals = nullptr;
  s->n_dist = (int)keys.size();
  if (s->n_dist > 0) {
    XLB_HIP(hipMalloc(&s->dist_keys, keys.size() * sizeof(uint32_t)));
    XLB_HIP(hipMemcpy(s->dist_keys, keys.data(), keys.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    XLB_HIP(hipMalloc(&s->dist_vals, vals.size() * sizeof(float)));
    XLB_HIP(hipMemcpy(s->dist_vals, vals.data(), vals.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  return 0;
}

int xlbhip_stepper_set_force(xlbhip_stepper* s, const double* force) {
  XLB_REQUIRE(s, "stepper is null");
  s->forced = force != nullptr;
  for (int a = 0; a < 3; ++a) s->force[a] = force ? force[a] : 0.0;
  return 0;
}

int xlbhip_stepper_set_smagorinsky(xlbhip_stepper* s, double coef) {
  XLB_REQUIRE(s, "stepper is null");
  s->smag_cs = coef;
  return 0;
}

int xlbhip_stepper_destroy(xlbhip_stepper* s) {
  if (!s) return 0;
  (void)hipSetDevice(s->ctx->device);
  (void)hipStreamSynchronize(s->ctx->stream);
  (void)hipFree(s->tab_kind);
  (void)hipFree(s->tab_values);
  if (s->prof_keys) (void)hipFree(s->prof_keys);
  if (s->prof_vals) (void)hipFree(s->prof_vals);
  if (s->dist_keys) (void)hipFree(s->dist_keys);
  if (s->dist_vals) (void)hipFree(s->dist_vals);
  if (s->scratch) xlbhip_field_destroy(s->scratch);
  if (s->meta) {
    comm_forget_buffer(s->ctx, s->meta);
    (void)hipFree(s->meta);
  }
  if (s->tile_order) (void)hipFree(s->tile_order);
  drop_clean_cache(s);
  delete s;
  return 0;
}

int xlbhip_step(xlbhip_stepper* s, const xlbhip_field* src, xlbhip_field* dst, const xlbhip_field* bcm, const xlbhip_field* miss,
                double omega, int64_t timestep) {
  (void)timestep;  // only time-dependent BC profiles use it in the reference (out of scope)
  if (int rc = check_step_fields(s, src, dst, bcm, miss)) return rc;
  return step_once(s, src, dst, bcm, miss, omega);
}

// n steps; `fixed_placement`: the result must land in f_a for even n and in f_b for odd n (xlbhip_run's contract);
// otherwise every pair of steps is fused and *result_in_b reports where the result is (xlbhip_run_any)
static int run_steps(xlbhip_stepper* s, xlbhip_field* a, xlbhip_field* b, const xlbhip_field* bcm, const xlbhip_field* miss, double omega,
                     int64_t n, bool fixed_placement, int* result_in_b) {
  XLB_REQUIRE(n >= 0, "n_steps < 0");
  if (int rc = check_step_fields(s, a, b, bcm, miss)) return rc;
  // With two-step fusion ("fuse2") a pair of steps is ONE pass a -> b: pairs alternate direction (a -> b, b -> a, ...).
  // Under the fixed placement contract a trailing half pair (buffer parity) is fixed up by single steps.
  int64_t i = 0;
  xlbhip_field* cur = a;
  xlbhip_field* oth = b;
  // (a host-staged transport refills the ghosts between calls: it drives pairs through xlbhip_step2 itself)
  const bool caller_fills_ghosts = a->halo > 0 && opt(s->ctx, "external_halo", 0) != 0;
  bool fuse = n >= 2 && !caller_fills_ghosts && can_fuse2(s, a, b, bcm, miss);
  if (n >= 2 && a->halo > 0 && !caller_fills_ghosts && comm_ranks(s->ctx) > 1) {
    // pairs and single steps post different message sets (depth-2 / depth-1 exchange, meta planes): every rank must take
    // the same decision, and uneven slabs may sit on either side of the chip-filling rule -> MIN over the ranks
    int all = 0;
    if (int rc = comm_all_min(s->ctx, fuse ? 1 : 0, &all)) return rc;
    fuse = all != 0;
  }
  if (fuse) {
    if (int rc = prepare_fuse2(s, bcm, miss)) return rc;
    // choose the number of pairs so that the remaining single steps land the result in the right buffer:
    // after P pairs the data sits in (P odd ? b : a); then r = n - 2P single steps flip r more times.
    // P + r must be congruent to n (mod 2)  <=>  P even.  Use the largest even P with 2P <= n.
    int64_t pairs = fixed_placement ? ((n / 2) & ~int64_t(1)) : n / 2;
    for (int64_t k = 0; k < pairs; ++k) {
      if (int rc = step_twice(s, cur, oth, bcm, miss, omega)) return rc;
      xlbhip_field* tmp = cur;
      cur = oth;
      oth = tmp;
    }
    i = 2 * pairs;
  }
  for (; i < n; ++i) {
    if (int rc = step_once(s, cur, oth, bcm, miss, omega)) return rc;
    xlbhip_field* tmp = cur;
    cur = oth;
    oth = tmp;
  }
  if (result_in_b) *result_in_b = cur == b ? 1 : 0;
  return 0;
}

int xlbhip_run(xlbhip_stepper* s, xlbhip_field* a, xlbhip_field* b, const xlbhip_field* bcm, const xlbhip_field* miss, double omega,
               int64_t t0, int64_t n) {
  (void)t0;
  return run_steps(s, a, b, bcm, miss, omega, n, true, nullptr);
}

int xlbhip_run_any(xlbhip_stepper* s, xlbhip_field* a, xlbhip_field* b, const xlbhip_field* bcm, const xlbhip_field* miss, double omega,
                   int64_t t0, int64_t n, int* result_in_b) {
  (void)t0;
  XLB_REQUIRE(result_in_b, "result_in_b is null");
  return run_steps(s, a, b, bcm, miss, omega, n, false, result_in_b);
}

int xlbhip_step2_eligible(xlbhip_stepper* s, const xlbhip_field* src, xlbhip_field* dst, const xlbhip_field* bcm, const xlbhip_field* miss) {
  if (check_step_fields(s, src, dst, bcm, miss)) return 0;
  return can_fuse2(s, src, dst, bcm, miss) ? 1 : 0;
}

int xlbhip_step2(xlbhip_stepper* s, const xlbhip_field* src, xlbhip_field* dst, const xlbhip_field* bcm, const xlbhip_field* miss,
                 double omega, int64_t timestep) {
  (void)timestep;
  if (int rc = check_step_fields(s, src, dst, bcm, miss)) return rc;
  XLB_REQUIRE(can_fuse2(s, src, dst, bcm, miss), "this stepper / field layout has no two-step kernel (see xlbhip_step2_eligible)");
  if (int rc = prepare_fuse2(s, bcm, miss)) return rc;
  return step_twice(s, src, dst, bcm, miss, omega);
}

int xlbhip_run_timed(xlbhip_stepper* s, xlbhip_field* a, xlbhip_field* b, const xlbhip_field* bcm, const xlbhip_field* miss, double omega,
                     int64_t t0, int64_t n, float* ms, int* result_in_b) {
  XLB_REQUIRE(s && ms, "null argument");
  xlbhip_ctx* c = s->ctx;
  XLB_HIP(hipEventRecord(c->ev_a, c->stream));
  // (result_in_b == NULL: xlbhip_run's fixed placement; else as xlbhip_run_any)
  if (int rc = run_steps(s, a, b, bcm, miss, omega, n, result_in_b == nullptr, result_in_b)) return rc;
  XLB_HIP(hipEventRecord(c->ev_b, c->stream));
  XLB_HIP(hipEventSynchronize(c->ev_b));
  XLB_HIP(hipEventElapsedTime(ms, c->ev_a, c->ev_b));
  return 0;
}

int xlbhip_comm_stats(xlbhip_ctx* c, double* halo_wait_ms, int64_t* halo_waits, int reset) {
  XLB_REQUIRE(c, "ctx is null");
  for (int i = 0; i < xlbhip_ctx::WAIT_RING; ++i)
    if (int rc = harvest_wait(c, i)) return rc;
  if (halo_wait_ms) *halo_wait_ms = c->halo_wait_ms;
  if (halo_waits) *halo_waits = c->halo_waits;
  if (reset) {
    c->halo_wait_ms = 0.0;
    c->halo_waits = 0;
  }
  return 0;
}

int xlbhip_halo_exchange(xlbhip_ctx* c, int lattice, xlbhip_field* f) {
  XLB_REQUIRE(c && f, "null argument");
  XLB_REQUIRE(f->halo >= 1, "field has no ghost planes");
  XLB_REQUIRE(f->card == lattice_q(lattice), "field cardinality does not match the lattice");
  return halo_exchange_on(c, lattice, f, c->stream, 1);
}

int xlbhip_halo_exchange_wide(xlbhip_ctx* c, int lattice, xlbhip_field* f) {
  XLB_REQUIRE(c && f, "null argument");
  XLB_REQUIRE(f->halo >= 2, "field has fewer than two ghost planes per side");
  XLB_REQUIRE(f->card == lattice_q(lattice), "field cardinality does not match the lattice");
  return halo_exchange_on(c, lattice, f, c->stream, 2);
}

}  // extern "C"
