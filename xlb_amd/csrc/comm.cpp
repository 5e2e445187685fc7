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
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
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
  XLB_SYM(AllReduce, "ncclAllReduce");
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
  int* flag = nullptr;  // device scratch of comm_all_min
};

template <class L>
static void face_sets(std::vector<int>& right, std::vector<int>& left) {
  for (int l = 0; l < L::Q; ++l) {
    if (L::c(0, l) == 1) right.push_back(l);
    if (L::c(0, l) == -1) left.push_back(l);
  }
}

// one plane-sized message of the ring: dir = +1 travels to the right neighbour (and the matching receive comes
// from the left one), dir = -1 the other way
struct HaloMsg {
  const char* send;
  char* recv;
  int dir;
};

static int run_messages(xlbhip_ctx* c, const std::vector<HaloMsg>& msgs, size_t bytes, hipStream_t st) {
  Comm* cm = c->comm;
  if (!cm || !cm->comm) {
    // single rank without a communicator: the ring neighbour is the field itself (periodic wrap)
    if (cm && !cm->periodic) return 0;
    for (const HaloMsg& m : msgs) XLB_HIP(hipMemcpyAsync(m.recv, m.send, bytes, hipMemcpyDeviceToDevice, st));
    return 0;
  }
  const int r = cm->rank, n = cm->n_ranks;
  const int rr = (r + 1) % n, lr = (r + n - 1) % n;
  const bool has_right = cm->periodic || r + 1 < n;
  const bool has_left = cm->periodic || r > 0;
  if (!has_right && !has_left) return 0;
  XLB_NCCL(g_rccl.GroupStart());
  // a failing call must not leave the group open (later RCCL calls of this thread would be queued into it silently)
  ncclResult_t bad = ncclSuccess;
  auto post = [&](ncclResult_t r) {
    if (bad == ncclSuccess && r != ncclSuccess) bad = r;
    return bad == ncclSuccess;
  };
  for (const HaloMsg& m : msgs) {
    if (m.dir > 0) {
      if (has_right && !post(g_rccl.Send(m.send, bytes, ncclInt8, rr, cm->comm, st))) break;
      if (has_left && !post(g_rccl.Recv(m.recv, bytes, ncclInt8, lr, cm->comm, st))) break;
    } else {
      if (has_left && !post(g_rccl.Send(m.send, bytes, ncclInt8, lr, cm->comm, st))) break;
      if (has_right && !post(g_rccl.Recv(m.recv, bytes, ncclInt8, rr, cm->comm, st))) break;
    }
  }
  const ncclResult_t end = g_rccl.GroupEnd();
  if (bad != ncclSuccess) XLB_FAIL("ncclSend / ncclRecv failed: %s", g_rccl.GetErrorString(bad));
  if (end != ncclSuccess) XLB_FAIL("ncclGroupEnd failed: %s", g_rccl.GetErrorString(end));
  return 0;
}

// MIN over the ranks of the communicator of a small host integer (a decision every rank must take alike, e.g. whether
// xlbhip_run pairs its steps: the two protocols post different message sets).  Blocking; off the per-step path.
int comm_all_min(xlbhip_ctx* c, int value, int* out) {
  Comm* cm = c->comm;
  *out = value;
  if (!cm || !cm->comm || cm->n_ranks <= 1) return 0;
  if (!cm->flag) XLB_HIP(hipMalloc(&cm->flag, 2 * sizeof(int)));
  XLB_HIP(hipMemcpyAsync(cm->flag, &value, sizeof(int), hipMemcpyHostToDevice, c->stream));
  XLB_NCCL(g_rccl.AllReduce(cm->flag, cm->flag + 1, 1, ncclInt32, ncclMin, cm->comm, c->stream));
  XLB_HIP(hipMemcpyAsync(out, cm->flag + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  XLB_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int comm_ranks(const xlbhip_ctx* c) { return (c->comm && c->comm->comm) ? c->comm->n_ranks : 1; }

int halo_exchange_on(xlbhip_ctx* c, int lattice, xlbhip_field* f, hipStream_t st, int depth) {
  XLB_REQUIRE(depth == 1 || depth == 2, "halo depth must be 1 or 2");
  XLB_REQUIRE(f->halo >= depth, "halo exchange of depth %d on a field with %d ghost plane(s)", depth, f->halo);
  XLB_REQUIRE(f->nx >= depth, "slab thinner than the halo");
  std::vector<int> right, left;
  int q = 0;
  if (lattice == XLBHIP_D3Q19) {
    face_sets<D3Q19>(right, left);
    q = D3Q19::Q;
  } else if (lattice == XLBHIP_D3Q27) {
    face_sets<D3Q27>(right, left);
    q = D3Q27::Q;
  } else {
    XLB_FAIL("slab decomposition needs a 3-D lattice");
  }
  const size_t es = dtype_size(f->dtype);
  const size_t plane = (size_t)f->ny * f->nz;
  const int h = f->halo, nx = f->nx;
  // interior plane X (ghosts: -h .. -1 and nx .. nx + h - 1) of population l
  auto ptr = [&](int l, int X) { return static_cast<char*>(f->data) + ((size_t)l * f->plane_stride + (size_t)(X + h) * plane) * es; };
  std::vector<HaloMsg> msgs;
  if (depth == 1) {
    // one step: only the populations that cross the face are pulled from the ghost plane
    for (int l : right) msgs.push_back({ptr(l, nx - 1), ptr(l, -1), +1});
    for (int l : left) msgs.push_back({ptr(l, 0), ptr(l, nx), -1});
  } else {
    // two fused steps: f(t+1) is recomputed on the ghost planes -1 and nx, which takes every population of the
    // neighbour's edge plane (a halfway wall there may redirect any pull to the own cell) and the crossing
    // populations of the plane behind it
    for (int l = 0; l < q; ++l) msgs.push_back({ptr(l, nx - 1), ptr(l, -1), +1});
    for (int l : right) msgs.push_back({ptr(l, nx - 2), ptr(l, -2), +1});
    for (int l = 0; l < q; ++l) msgs.push_back({ptr(l, 0), ptr(l, nx), -1});
    for (int l : left) msgs.push_back({ptr(l, 1), ptr(l, nx + 1), -1});
  }
  return run_messages(c, msgs, plane * es, st);
}

// ghost planes -1 and nx of a one-component per-cell array with `halo` ghost planes (the two-step kernel's meta words)
int plane_exchange_on(xlbhip_ctx* c, void* base, size_t elem_bytes, int nx, int ny, int nz, int halo, hipStream_t st) {
  XLB_REQUIRE(halo >= 1 && nx >= 1, "plane exchange needs ghost planes");
  const size_t bytes = (size_t)ny * nz * elem_bytes;
  auto ptr = [&](int X) { return static_cast<char*>(base) + (size_t)(X + halo) * bytes; };
  std::vector<HaloMsg> msgs = {{ptr(nx - 1), ptr(-1), +1}, {ptr(0), ptr(nx), -1}};
  return run_messages(c, msgs, bytes, st);
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
  if (c->comm->flag) (void)hipFree(c->comm->flag);
  delete c->comm;
  c->comm = nullptr;
  return 0;
}

}  // extern "C"
