// Slab halo exchange along the slowest spatial axis, two device-side transports:
//   * RCCL ncclSend / ncclRecv (librccl is dlopen'ed on first use: single-GPU runs never depend on it);
//   * "ipc": every rank exports its fields with hipIpcGetMemHandle, the neighbours map them once, and an exchange is a
//     list of plane-sized hipMemcpyAsync PULLS on the communication stream — copy engines, no compute unit, no LDS —
//     ordered across processes by sequence counters in a small host shared-memory segment that tiny kernels post and poll.
// Semantics reference: xlb/distribute/distribute.py:18-48 — after a local step the populations that cross the slab faces
// are swapped with the ring neighbours (rightPerm / leftPerm); here they are delivered into ghost planes BEFORE the pull.
#include "comm.hpp"

#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <cerrno>
#include <cstring>
#include <map>
#include <string>
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

// ---- the shared-memory control block of the ipc transport (one file in /dev/shm per job, unlinked once every rank has it) ----
static const uint32_t IPC_MAGIC = 0x584c4249u;  // "XLBI"
static const int IPC_MAX_BUFS = 96;             // exported buffers per rank over the life of the communicator

// polled / posted by kernels through the host-registered mapping; one cache line per rank
struct alignas(64) IpcFlags {
  uint32_t ready;  // exchange sequence number up to which this rank's source planes are complete
  uint32_t error;  // non-zero: a wait of this rank timed out (the sequence number it waited for)
  uint32_t pad[14];
};

struct IpcSlot {
  uint32_t published;  // 1 once the record below is complete (release store)
  uint32_t pad;
  hipIpcMemHandle_t handle;
  uint64_t bytes;         // size of the exported allocation
  uint64_t data_offset;   // byte offset of element (population 0, storage plane 0) from the allocation base
  uint64_t plane_stride;  // elements between populations
  int32_t nx, ny, nz, halo, card, elem_bytes;
};

struct IpcRank {
  IpcFlags flags;
  uint32_t attached;     // this rank mapped the segment
  uint32_t finished;     // this rank is tearing its communicator down (peers stop trusting its memory afterwards)
  uint32_t vote_seq;     // host-side all-min: latest vote posted
  int32_t vote_val[2];   // double-buffered by vote_seq parity
  uint32_t pad[11];
  IpcSlot slots[IPC_MAX_BUFS];
};

struct IpcHeader {
  uint32_t magic, n_ranks;
  uint32_t pad[14];
};

struct PeerBuf {
  void* base = nullptr;  // hipIpcOpenMemHandle mapping of the peer's allocation
  IpcSlot rec;
};

struct Comm {
  ncclComm_t comm = nullptr;
  int rank = 0, n_ranks = 1, periodic = 1;
  int* flag = nullptr;  // device scratch of comm_all_min (RCCL)
  // ipc transport
  bool ipc = false;
  char* shm = nullptr;      // host mapping of the control block
  char* shm_dev = nullptr;  // the same bytes as the device sees them (hipHostRegister)
  size_t shm_bytes = 0;
  uint32_t seq = 0;         // exchanges posted so far (every rank counts the same ones: SPMD)
  uint32_t vote_seq = 0;
  int next_slot = 0;
  std::map<const void*, int> slot_of;              // local allocation base -> slot
  std::map<std::pair<int, int>, PeerBuf> peer_bufs;  // (peer rank, slot) -> mapping
  double timeout_s = 60.0;
  uint64_t wall_khz = 100000;
  IpcHeader* header() const { return reinterpret_cast<IpcHeader*>(shm); }
  IpcRank* rank_block(int r) const { return reinterpret_cast<IpcRank*>(shm + sizeof(IpcHeader)) + r; }
  // device address of a member of the control block
  template <class T>
  T* dev(T* host_member) const {
    return reinterpret_cast<T*>(shm_dev + (reinterpret_cast<char*>(host_member) - shm));
  }
};

template <class L>
static void face_sets(std::vector<int>& right, std::vector<int>& left) {
  for (int l = 0; l < L::Q; ++l) {
    if (L::c(0, l) == 1) right.push_back(l);
    if (L::c(0, l) == -1) left.push_back(l);
  }
}

// One plane-sized message of the ring.  dir = +1: plane (nx - 1 - depth) of population l travels to the right neighbour
// and lands in its ghost plane (-1 - depth); dir = -1: plane `depth` travels left into the ghost plane (nx + depth).
struct HaloMsg {
  int l, depth, dir;
};

// where the planes of one exported / local buffer live
struct BufLayout {
  char* data = nullptr;  // element (population 0, storage plane 0)
  size_t plane_stride = 0, plane = 0, es = 0;
  int nx = 0, halo = 0;
  char* ptr(int l, int X) const { return data + ((size_t)l * plane_stride + (size_t)(X + halo) * plane) * es; }
  const char* send_ptr(const HaloMsg& m) const { return m.dir > 0 ? ptr(m.l, nx - 1 - m.depth) : ptr(m.l, m.depth); }
  char* recv_ptr(const HaloMsg& m) const { return m.dir > 0 ? ptr(m.l, -1 - m.depth) : ptr(m.l, nx + m.depth); }
};

// ---- ipc transport: kernels -----------------------------------------------------------------------------------
__global__ void k_ipc_post(uint32_t* flag, uint32_t seq) {
  __threadfence_system();
  __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// one lane polls the neighbours' counters (host memory: always coherent) until both reached `seq`; the loop ends after
// `timeout_ticks` of the constant-rate wall clock whatever happens, and says so in *err (read by the host at the next sync)
__global__ void k_ipc_wait(const uint32_t* a, const uint32_t* b, uint32_t seq, unsigned long long timeout_ticks, uint32_t* err) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = (unsigned long long)wall_clock64();
  for (;;) {
    const bool ok_a = !a || (int32_t)(__hip_atomic_load(a, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) >= 0;
    const bool ok_b = !b || (int32_t)(__hip_atomic_load(b, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) >= 0;
    if (ok_a && ok_b) break;
    if ((unsigned long long)wall_clock64() - t0 > timeout_ticks) {
      __hip_atomic_store(err, seq ? seq : 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
    __builtin_amdgcn_s_sleep(64);
  }
  __threadfence_system();
}

// Alternative to the copy calls (option "ipc_copy" = 1): ONE launch pulls every plane of an exchange through the IPC mappings,
// 16 bytes per lane; blockIdx.y = message.  Few blocks: it runs beside the interior launch on wave slots that kernel leaves free.
struct IpcPullTable {
  static const int MAX = 56;  // D3Q27, depth 2: 2 x (27 + 9) messages would not fit: the caller splits
  const void* src[MAX];
  void* dst[MAX];
};

__global__ void __launch_bounds__(256) k_ipc_pull(IpcPullTable tab, size_t bytes) {
  const char* s = static_cast<const char*>(tab.src[blockIdx.y]);
  char* d = static_cast<char*>(tab.dst[blockIdx.y]);
  const size_t n16 = bytes / 16, stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride)
    reinterpret_cast<uint4*>(d)[i] = reinterpret_cast<const uint4*>(s)[i];
  for (size_t i = n16 * 16 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < bytes; i += stride) d[i] = s[i];
}

static double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

static void nap() {
  timespec ts = {0, 50000};  // 50 us
  nanosleep(&ts, nullptr);
}

// export a local allocation (once) and return its slot; every rank exports its buffers in the same order (SPMD), so
// slot k of a neighbour is the buffer that plays the same role there
static int ipc_slot_for(Comm* cm, void* base, size_t bytes, const BufLayout& lay, int ny, int nz, int card, int* out) {
  auto it = cm->slot_of.find(base);
  if (it != cm->slot_of.end()) {
    *out = it->second;
    return 0;
  }
  XLB_REQUIRE(cm->next_slot < IPC_MAX_BUFS, "ipc transport: more than %d exported buffers", IPC_MAX_BUFS);
  const int slot = cm->next_slot++;
  IpcSlot* s = &cm->rank_block(cm->rank)->slots[slot];
  XLB_HIP(hipIpcGetMemHandle(&s->handle, base));
  s->bytes = bytes;
  s->data_offset = (uint64_t)(lay.data - static_cast<char*>(base));
  s->plane_stride = lay.plane_stride;
  s->nx = lay.nx;
  s->ny = ny;
  s->nz = nz;
  s->halo = lay.halo;
  s->card = card;
  s->elem_bytes = (int32_t)lay.es;
  __atomic_store_n(&s->published, 1u, __ATOMIC_RELEASE);
  cm->slot_of[base] = slot;
  *out = slot;
  return 0;
}

// -1: error (message set)
static int ipc_peer_buf(Comm* cm, int peer, int slot, int ny, int nz, int card, size_t es, BufLayout* out) {
  auto key = std::make_pair(peer, slot);
  auto it = cm->peer_bufs.find(key);
  if (it == cm->peer_bufs.end()) {
    IpcSlot* s = &cm->rank_block(peer)->slots[slot];
    const double t0 = now_s();
    while (!__atomic_load_n(&s->published, __ATOMIC_ACQUIRE)) {
      XLB_REQUIRE(!__atomic_load_n(&cm->rank_block(peer)->finished, __ATOMIC_ACQUIRE), "ipc transport: rank %d left the job", peer);
      XLB_REQUIRE(now_s() - t0 < cm->timeout_s, "ipc transport: rank %d did not export buffer %d within %.0f s (ranks must exchange the same fields in the same order)",
                  peer, slot, cm->timeout_s);
      nap();
    }
    PeerBuf pb;
    pb.rec = *s;
    XLB_REQUIRE(pb.rec.ny == ny && pb.rec.nz == nz && pb.rec.card == card && (size_t)pb.rec.elem_bytes == es,
                "ipc transport: buffer %d of rank %d is (%d, ., %d, %d) x %d B, this rank's is (%d, ., %d, %d) x %zu B — the ranks exchange different fields",
                slot, peer, pb.rec.card, pb.rec.ny, pb.rec.nz, pb.rec.elem_bytes, card, ny, nz, es);
    XLB_HIP(hipIpcOpenMemHandle(&pb.base, pb.rec.handle, hipIpcMemLazyEnablePeerAccess));
    it = cm->peer_bufs.emplace(key, pb).first;
  }
  const PeerBuf& pb = it->second;
  out->data = static_cast<char*>(pb.base) + pb.rec.data_offset;
  out->plane_stride = pb.rec.plane_stride;
  out->plane = (size_t)ny * nz;
  out->es = es;
  out->nx = pb.rec.nx;
  out->halo = pb.rec.halo;
  return 0;
}

static int ipc_check_error(Comm* cm) {
  const uint32_t e = __atomic_load_n(&cm->rank_block(cm->rank)->flags.error, __ATOMIC_ACQUIRE);
  XLB_REQUIRE(e == 0, "ipc transport: rank %d waited more than %.0f s for its neighbours at exchange %u (a peer died or the ranks diverged); "
              "the ghost planes of that exchange are undefined", cm->rank, cm->timeout_s, e);
  return 0;
}

// local: the buffer the exchange is about (alloc_base / alloc_bytes: its allocation, what hipIpcGetMemHandle exports)
static int run_messages(xlbhip_ctx* c, const BufLayout& local, void* alloc_base, size_t alloc_bytes, int ny, int nz, int card,
                        const std::vector<HaloMsg>& msgs, hipStream_t st) {
  Comm* cm = c->comm;
  const size_t bytes = local.plane * local.es;
  if (opt(c, "halo_skip", 0)) return 0;  // measurement only: the protocol's launches without any exchange (results are wrong)
  if (!cm || (!cm->comm && !cm->ipc)) {
    // single rank without a communicator: the ring neighbour is the field itself (periodic wrap)
    if (cm && !cm->periodic) return 0;
    for (const HaloMsg& m : msgs) XLB_HIP(hipMemcpyAsync(local.recv_ptr(m), local.send_ptr(m), bytes, hipMemcpyDeviceToDevice, st));
    return 0;
  }
  const int r = cm->rank, n = cm->n_ranks;
  const int rr = (r + 1) % n, lr = (r + n - 1) % n;
  const bool has_right = cm->periodic || r + 1 < n;
  const bool has_left = cm->periodic || r > 0;
  if (!has_right && !has_left) return 0;
  if (cm->ipc) {
    if (int rc = ipc_check_error(cm)) return rc;
    const uint32_t seq = ++cm->seq;
    int slot = -1;
    if (int rc = ipc_slot_for(cm, alloc_base, alloc_bytes, local, ny, nz, card, &slot)) return rc;
    // 1. everything enqueued on `st` so far (the caller made it wait for the step that produced this buffer) is done:
    //    my planes may be pulled
    hipLaunchKernelGGL(k_ipc_post, 1, 1, 0, st, cm->dev(&cm->rank_block(r)->flags.ready), seq);
    XLB_HIP(hipGetLastError());
    BufLayout left_buf, right_buf;
    if (has_left)
      if (int rc = ipc_peer_buf(cm, lr, slot, ny, nz, card, local.es, &left_buf)) return rc;
    if (has_right)
      if (int rc = ipc_peer_buf(cm, rr, slot, ny, nz, card, local.es, &right_buf)) return rc;
    // 2. wait (on the device, bounded) until the neighbours said the same.  A neighbour that posted `seq` has also
    //    finished every pull of exchange seq - 1 from this rank (it posts after the step that consumed them), so no
    //    further acknowledgement is needed before this rank overwrites the planes it lent
    const unsigned long long ticks = (unsigned long long)(cm->timeout_s * 1e3 * (double)cm->wall_khz);
    hipLaunchKernelGGL(k_ipc_wait, 1, 64, 0, st, has_left ? cm->dev(&cm->rank_block(lr)->flags.ready) : nullptr,
                       has_right ? cm->dev(&cm->rank_block(rr)->flags.ready) : nullptr, seq, ticks, cm->dev(&cm->rank_block(r)->flags.error));
    XLB_HIP(hipGetLastError());
    // 3. pull: what the left neighbour would have sent right lands in my left ghosts, and vice versa
    if (opt(c, "ipc_copy", 0) == 1 && (reinterpret_cast<uintptr_t>(local.data) % 16 == 0) && bytes % 16 == 0) {
      IpcPullTable tab;
      int n_tab = 0;
      auto flush = [&]() {
        if (n_tab == 0) return 0;
        hipLaunchKernelGGL(k_ipc_pull, dim3(8, n_tab), 256, 0, st, tab, bytes);
        XLB_HIP(hipGetLastError());
        n_tab = 0;
        return 0;
      };
      for (const HaloMsg& m : msgs) {
        const bool from_left = m.dir > 0 && has_left, from_right = m.dir < 0 && has_right;
        if (!from_left && !from_right) continue;
        tab.src[n_tab] = from_left ? left_buf.send_ptr(m) : right_buf.send_ptr(m);
        tab.dst[n_tab] = local.recv_ptr(m);
        if (++n_tab == IpcPullTable::MAX)
          if (int rc = flush()) return rc;
      }
      return flush();
    }
    for (const HaloMsg& m : msgs) {
      if (m.dir > 0 && has_left) XLB_HIP(hipMemcpyAsync(local.recv_ptr(m), left_buf.send_ptr(m), bytes, hipMemcpyDefault, st));
      if (m.dir < 0 && has_right) XLB_HIP(hipMemcpyAsync(local.recv_ptr(m), right_buf.send_ptr(m), bytes, hipMemcpyDefault, st));
    }
    return 0;
  }
  XLB_NCCL(g_rccl.GroupStart());
  // a failing call must not leave the group open (later RCCL calls of this thread would be queued into it silently)
  ncclResult_t bad = ncclSuccess;
  auto post = [&](ncclResult_t r) {
    if (bad == ncclSuccess && r != ncclSuccess) bad = r;
    return bad == ncclSuccess;
  };
  for (const HaloMsg& m : msgs) {
    if (m.dir > 0) {
      if (has_right && !post(g_rccl.Send(local.send_ptr(m), bytes, ncclInt8, rr, cm->comm, st))) break;
      if (has_left && !post(g_rccl.Recv(local.recv_ptr(m), bytes, ncclInt8, lr, cm->comm, st))) break;
    } else {
      if (has_left && !post(g_rccl.Send(local.send_ptr(m), bytes, ncclInt8, lr, cm->comm, st))) break;
      if (has_right && !post(g_rccl.Recv(local.recv_ptr(m), bytes, ncclInt8, rr, cm->comm, st))) break;
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
  const bool self_test = cm && cm->comm && cm->n_ranks == 1 && opt(c, "comm_self_test", 0) != 0;  // exercises the RCCL call with one rank
  if (!cm || (!cm->comm && !cm->ipc) || (cm->n_ranks <= 1 && !self_test)) return 0;
  if (cm->ipc) {
    // host-side vote through the control block; two value slots by parity (a rank can be at most one vote ahead)
    const uint32_t v = ++cm->vote_seq;
    IpcRank* me = cm->rank_block(cm->rank);
    me->vote_val[v & 1] = value;
    __atomic_store_n(&me->vote_seq, v, __ATOMIC_RELEASE);
    int m = value;
    const double t0 = now_s();
    for (int r = 0; r < cm->n_ranks; ++r) {
      IpcRank* o = cm->rank_block(r);
      while ((int32_t)(__atomic_load_n(&o->vote_seq, __ATOMIC_ACQUIRE) - v) < 0) {
        XLB_REQUIRE(!__atomic_load_n(&o->finished, __ATOMIC_ACQUIRE), "ipc transport: rank %d left the job during a vote", r);
        XLB_REQUIRE(now_s() - t0 < cm->timeout_s, "ipc transport: rank %d did not vote within %.0f s", r, cm->timeout_s);
        nap();
      }
      const int ov = o->vote_val[v & 1];
      if (ov < m) m = ov;
    }
    *out = m;
    return 0;
  }
  if (!cm->flag) XLB_HIP(hipMalloc(&cm->flag, 2 * sizeof(int)));
  XLB_HIP(hipMemcpyAsync(cm->flag, &value, sizeof(int), hipMemcpyHostToDevice, c->stream));
  XLB_NCCL(g_rccl.AllReduce(cm->flag, cm->flag + 1, 1, ncclInt32, ncclMin, cm->comm, c->stream));
  XLB_HIP(hipMemcpyAsync(out, cm->flag + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
  XLB_HIP(hipStreamSynchronize(c->stream));
  return 0;
}

int comm_ranks(const xlbhip_ctx* c) { return (c->comm && (c->comm->comm || c->comm->ipc)) ? c->comm->n_ranks : 1; }

int comm_check(xlbhip_ctx* c) {
  if (c->comm && c->comm->ipc) return ipc_check_error(c->comm);
  return 0;
}

void comm_forget_buffer(xlbhip_ctx* c, const void* alloc_base) {
  // the allocation is about to be freed: a later one at the same address is a different buffer (a new slot).  Mappings
  // the neighbours hold stay valid until they close them (comm destroy).
  if (c && c->comm && c->comm->ipc) c->comm->slot_of.erase(alloc_base);
}

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
  BufLayout lay;
  lay.data = static_cast<char*>(f->data);
  lay.plane_stride = f->plane_stride;
  lay.plane = (size_t)f->ny * f->nz;
  lay.es = dtype_size(f->dtype);
  lay.nx = f->nx;
  lay.halo = f->halo;
  std::vector<HaloMsg> msgs;
  if (depth == 1) {
    // one step: only the populations that cross the face are pulled from the ghost plane
    for (int l : right) msgs.push_back({l, 0, +1});
    for (int l : left) msgs.push_back({l, 0, -1});
  } else {
    // two fused steps: f(t+1) is recomputed on the ghost planes -1 and nx, which takes every population of the
    // neighbour's edge plane (a halfway wall there may redirect any pull to the own cell) and the crossing
    // populations of the plane behind it
    for (int l = 0; l < q; ++l) msgs.push_back({l, 0, +1});
    for (int l : right) msgs.push_back({l, 1, +1});
    for (int l = 0; l < q; ++l) msgs.push_back({l, 0, -1});
    for (int l : left) msgs.push_back({l, 1, -1});
  }
  return run_messages(c, lay, f->base, f->alloc_bytes, f->ny, f->nz, f->card, msgs, st);
}

// ghost planes -1 and nx of a one-component per-cell array with `halo` ghost planes (the two-step kernel's meta words);
// `base` is the allocation itself
int plane_exchange_on(xlbhip_ctx* c, void* base, size_t elem_bytes, int nx, int ny, int nz, int halo, hipStream_t st) {
  XLB_REQUIRE(halo >= 1 && nx >= 1, "plane exchange needs ghost planes");
  BufLayout lay;
  lay.data = static_cast<char*>(base);
  lay.plane_stride = 0;
  lay.plane = (size_t)ny * nz;
  lay.es = elem_bytes;
  lay.nx = nx;
  lay.halo = halo;
  std::vector<HaloMsg> msgs = {{0, 0, +1}, {0, 0, -1}};
  return run_messages(c, lay, base, (size_t)(nx + 2 * halo) * ny * nz * elem_bytes, ny, nz, 1, msgs, st);
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
  if (n_ranks > 1 && !id_bytes) {
    delete cm;
    XLB_FAIL("unique id is null");
  }
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

int xlbhip_comm_init_ipc(xlbhip_ctx* c, int rank, int n_ranks, const char* token, int periodic_x) {
  XLB_REQUIRE(c && token && token[0], "null argument");
  XLB_REQUIRE(n_ranks >= 2 && rank >= 0 && rank < n_ranks, "ipc transport: bad rank %d of %d (it needs at least two ranks)", rank, n_ranks);
  XLB_REQUIRE(!c->comm, "communicator already initialised");
  for (const char* p = token; *p; ++p)
    XLB_REQUIRE((*p >= '0' && *p <= '9') || (*p >= 'a' && *p <= 'z') || (*p >= 'A' && *p <= 'Z') || *p == '-' || *p == '_',
                "ipc transport: the job token may hold letters, digits, '-' and '_' only");
  XLB_HIP(hipSetDevice(c->device));
  const std::string path = std::string("/dev/shm/xlbhip-ipc-") + token;
  const size_t page = (size_t)sysconf(_SC_PAGESIZE);
  size_t bytes = sizeof(IpcHeader) + (size_t)n_ranks * sizeof(IpcRank);
  bytes = (bytes + page - 1) / page * page;
  const double timeout_s = (double)opt(c, "ipc_timeout_ms", 180000) * 1e-3;
  int fd = -1;
  const double t0 = now_s();
  if (rank == 0) {
    // the name carries a fresh per-job nonce (the host side draws it): the file must not exist, and is no link
    fd = open(path.c_str(), O_RDWR | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
    XLB_REQUIRE(fd >= 0, "ipc transport: cannot create %s: %s", path.c_str(), strerror(errno));
    if (ftruncate(fd, (off_t)bytes) != 0) {
      const int e = errno;
      close(fd);
      unlink(path.c_str());
      XLB_FAIL("ipc transport: ftruncate(%s, %zu): %s", path.c_str(), bytes, strerror(e));
    }
  } else {
    for (;;) {
      fd = open(path.c_str(), O_RDWR | O_NOFOLLOW | O_CLOEXEC);
      if (fd >= 0) {
        struct stat sb;
        if (fstat(fd, &sb) == 0 && (size_t)sb.st_size >= bytes) break;
        close(fd);
        fd = -1;
      }
      XLB_REQUIRE(now_s() - t0 < timeout_s, "ipc transport: rank 0's control block %s did not appear within %.0f s", path.c_str(), timeout_s);
      nap();
    }
  }
  void* map = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (map == MAP_FAILED) {
    if (rank == 0) unlink(path.c_str());
    XLB_FAIL("ipc transport: mmap of the control block failed: %s", strerror(errno));
  }
  Comm* cm = new Comm();
  cm->rank = rank;
  cm->n_ranks = n_ranks;
  cm->periodic = periodic_x ? 1 : 0;
  cm->ipc = true;
  cm->shm = static_cast<char*>(map);
  cm->shm_bytes = bytes;
  cm->timeout_s = timeout_s;
  auto fail = [&](const char* what, hipError_t e) {
    if (cm->shm_dev) (void)hipHostUnregister(cm->shm);
    munmap(cm->shm, cm->shm_bytes);
    if (rank == 0) unlink(path.c_str());
    delete cm;
    set_error("ipc transport: %s failed: %s", what, hipGetErrorString(e));
    return 1;
  };
  if (rank == 0) {  // (a fresh tmpfs file is zero-filled)
    cm->header()->n_ranks = (uint32_t)n_ranks;
    __atomic_store_n(&cm->header()->magic, IPC_MAGIC, __ATOMIC_RELEASE);
  } else {
    while (__atomic_load_n(&cm->header()->magic, __ATOMIC_ACQUIRE) != IPC_MAGIC) {
      if (now_s() - t0 > timeout_s) return fail("waiting for rank 0's control block", hipErrorNotReady);
      nap();
    }
    if ((int)cm->header()->n_ranks != n_ranks) return fail("control block of a different job", hipErrorInvalidValue);
  }
  // the counters are polled and posted by kernels: register the mapping with the device
  hipError_t e = hipHostRegister(cm->shm, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
  if (e != hipSuccess) return fail("hipHostRegister of the control block", e);
  void* dev = nullptr;
  e = hipHostGetDevicePointer(&dev, cm->shm, 0);
  if (e != hipSuccess) {
    cm->shm_dev = cm->shm;  // (so that fail() unregisters)
    return fail("hipHostGetDevicePointer of the control block", e);
  }
  cm->shm_dev = static_cast<char*>(dev);
  int khz = 0;
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, c->device) == hipSuccess && khz > 0) cm->wall_khz = (uint64_t)khz;
  __atomic_store_n(&cm->rank_block(rank)->attached, 1u, __ATOMIC_RELEASE);
  // everybody has the segment: its name can go (rank 0), so a crashed job leaves nothing behind in /dev/shm
  for (int r = 0; r < n_ranks; ++r) {
    while (!__atomic_load_n(&cm->rank_block(r)->attached, __ATOMIC_ACQUIRE)) {
      if (now_s() - t0 > timeout_s) {
        set_error("ipc transport: rank %d did not attach to the control block within %.0f s", r, timeout_s);
        (void)hipHostUnregister(cm->shm);
        munmap(cm->shm, cm->shm_bytes);
        if (rank == 0) unlink(path.c_str());
        delete cm;
        return 1;
      }
      nap();
    }
  }
  if (rank == 0) unlink(path.c_str());
  c->comm = cm;
  return 0;
}

int xlbhip_comm_destroy(xlbhip_ctx* c) {
  if (!c || !c->comm) return 0;
  Comm* cm = c->comm;
  if (cm->comm) g_rccl.CommDestroy(cm->comm);
  if (cm->flag) (void)hipFree(cm->flag);
  if (cm->ipc) {
    // my pulls are done (the caller drained the device); say so, and give the neighbours a moment to finish theirs
    // before the memory they map goes away with this process's fields
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    __atomic_store_n(&cm->rank_block(cm->rank)->finished, 1u, __ATOMIC_RELEASE);
    const double t0 = now_s();
    for (int r = 0; r < cm->n_ranks; ++r)
      while (!__atomic_load_n(&cm->rank_block(r)->finished, __ATOMIC_ACQUIRE) && now_s() - t0 < 5.0) nap();
    for (auto& kv : cm->peer_bufs) (void)hipIpcCloseMemHandle(kv.second.base);
    (void)hipHostUnregister(cm->shm);
    munmap(cm->shm, cm->shm_bytes);
  }
  delete cm;
  c->comm = nullptr;
  return 0;
}

}  // extern "C"
