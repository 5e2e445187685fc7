// RCCL-backed ring halo exchange.  librccl is dlopen'ed on first use so that single-GPU
// runs never depend on it.  Semantics reference: xlb/distribute/distribute.py:18-48 — after a
// local step the populations that cross the slab faces are swapped with the ring neighbours
// (rightPerm / leftPerm); here they are delivered into ghost planes BEFORE the pull instead.
#include "comm.hpp"

#include <dlfcn.h>
#include <cstring>
#include <rccl/rccl.h>

#include <vector>

#include "lattice.hpp"

namespace xlb {

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi g_rccl;

static int load_rccl() {
  if (g_rccl.lib) return 0;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* lib = nullptr;
  for (const char* n : names) {
    lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
  }
  XLB_REQUIRE(lib, "cannot load librccl: %s", dlerror());
#define XLB_SYM(field, name)                                         \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, name)); \
  XLB_REQUIRE(g_rccl.field, "librccl lacks symbol %s", name)
  XLB_SYM(GetUniqueId, "ncclGetUniqueId");
  XLB_SYM(CommInitRank, "ncclCommInitRank");
  XLB_SYM(CommDestroy, "ncclCommDestroy");
  XLB_SYM(GroupStart, "ncclGroupStart");
  XLB_SYM(GroupEnd, "ncclGroupEnd");
  XLB_SYM(Send, "ncclSend");
  XLB_SYM(Recv, "ncclRecv");
  XLB_SYM(GetErrorString, "ncclGetErrorString");
#undef XLB_SYM
  g_rccl.lib = lib;
  return 0;
}

#define XLB_NCCL(expr)                                                                          \
  do {                                                                                          \
    ncclResult_t r_ = (expr);                                                                   \
    if (r_ != ncclSuccess) XLB_FAIL("%s failed: %s", #expr, g_rccl.GetErrorString(r_));         \
  } while (0)

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, n_ranks = 1, periodic = 1;
};

template <class L>
static void face_sets(std::vector<int>& right, std::vector<int>& left) {
  for (int l = 0; l < L::Q; ++l) {
    if (L::c(0, l) == 1) right.push_back(l);
    if (L::c(0, l) == -1) left.push_back(l);
  }
}

int halo_exchange_on(xlbhip_ctx* c, int lattice, xlbhip_field* f, hipStream_t st) {
  XLB_REQUIRE(f->halo == 1, "halo exchange on a field without ghost planes");
  std::vector<int> right, left;
  if (lattice == XLBHIP_D3Q19)
    face_sets<D3Q19>(right, left);
  else if (lattice == XLBHIP_D3Q27)
    face_sets<D3Q27>(right, left);
  else
    XLB_FAIL("slab decomposition needs a 3-D lattice");
  const size_t es = dtype_size(f->dtype);
  const size_t plane = (size_t)f->ny * f->nz;
  const size_t bytes = plane * es;
  auto ptr = [&](int l, int X) { return static_cast<char*>(f->data) + ((size_t)l * f->plane_stride + (size_t)X * plane) * es; };
  const int nx = f->nx;
  Comm* cm = c->comm;
  if (!cm || !cm->comm) {
    if (cm && !cm->periodic) return 0;
    for (int l : right) XLB_HIP(hipMemcpyAsync(ptr(l, 0), ptr(l, nx), bytes, hipMemcpyDeviceToDevice, st));
    for (int l : left) XLB_HIP(hipMemcpyAsync(ptr(l, nx + 1), ptr(l, 1), bytes, hipMemcpyDeviceToDevice, st));
    return 0;
  }
  const int r = cm->rank, n = cm->n_ranks;
  const int rr = (r + 1) % n, lr = (r + n - 1) % n;
  const bool has_right = cm->periodic || r + 1 < n;
  const bool has_left = cm->periodic || r > 0;
  XLB_NCCL(g_rccl.GroupStart());
  for (int l : right) {
    if (has_right) XLB_NCCL(g_rccl.Send(ptr(l, nx), bytes, ncclInt8, rr, cm->comm, st));
    if (has_left) XLB_NCCL(g_rccl.Recv(ptr(l, 0), bytes, ncclInt8, lr, cm->comm, st));
  }
  for (int l : left) {
    if (has_left) XLB_NCCL(g_rccl.Send(ptr(l, 1), bytes, ncclInt8, lr, cm->comm, st));
    if (has_right) XLB_NCCL(g_rccl.Recv(ptr(l, nx + 1), bytes, ncclInt8, rr, cm->comm, st));
  }
  XLB_NCCL(g_rccl.GroupEnd());
  return 0;
}

}  // namespace xlb

using namespace xlb;

extern "C" {

int xlbhip_comm_unique_id(void* out) {
  XLB_REQUIRE(out, "null output");
  static_assert(sizeof(ncclUniqueId) == XLBHIP_UNIQUE_ID_BYTES, "ncclUniqueId size");
  if (int rc = load_rccl()) return rc;
  ncclUniqueId id;
  XLB_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(out, &id, sizeof(id));
  return 0;
}

int xlbhip_comm_init(xlbhip_ctx* c, int rank, int n_ranks, const void* id_bytes, int periodic_x) {
  XLB_REQUIRE(c, "ctx is null");
  XLB_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "bad rank %d of %d", rank, n_ranks);
  XLB_REQUIRE(!c->comm, "communicator already initialised");
  Comm* cm = new Comm();
  cm->rank = rank;
  cm->n_ranks = n_ranks;
  cm->periodic = periodic_x ? 1 : 0;
  XLB_REQUIRE(n_ranks == 1 || id_bytes, "unique id is null");
  if (id_bytes) {
    if (int rc = load_rccl()) {
      delete cm;
      return rc;
    }
    XLB_HIP(hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof(id));
    ncclResult_t r = g_rccl.CommInitRank(&cm->comm, n_ranks, id, rank);
    if (r != ncclSuccess) {
      delete cm;
      XLB_FAIL("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
    }
  }
  c->comm = cm;
  return 0;
}

int xlbhip_comm_destroy(xlbhip_ctx* c) {
  if (!c || !c->comm) return 0;
  if (c->comm->comm) g_rccl.CommDestroy(c->comm->comm);
  delete c->comm;
  c->comm = nullptr;
  return 0;
}

}  // extern "C"
