// Two LBM steps per pass over memory ("temporal fusion"): f(t) -> f(t+2) with the intermediate
// f(t+1) living only in LDS.
//
// A block owns a (TY x TZ) tile of the (y, z) plane and marches along (a segment of) x.  For every x it
//   phase B: computes f(t+2) on plane x for the tile proper by pulling from the three LDS slots
//            (planes x-1, x, x+1) and stores it;
//   phase A: computes f(t+1) on plane x+2 for the tile grown by one cell in y and z
//            ((TY+2) x (TZ+2) cells, periodic images included) from pulls of f(t) in global memory that were
//            issued one plane earlier — exactly the single-step kernel's work — writes the q populations into
//            the LDS slot plane x-1 just vacated, and issues the pulls of plane x+3.
// HBM traffic per two steps: ~(1 + halo overlap) reads + 1 write of every population instead of
// 2 + 2.  Arithmetic is the same per-cell code (cell.hpp) in the same order, so the result is
// bit-identical to two single steps (tests/test_gpu_stepper.py::test_two_step_fusion_*).
//
// Restrictions (the launcher falls back to the single-step kernel otherwise — step2_eligible): fp32 store (f(t+1) sits in LDS in the
// store type); D3Q19 BGK fp32 with basic boundary conditions, D3Q27 BGK fp32 and (on request) KBC fp32 / fast fp64 without them;
// ny % TY == 0, nz % TZ == 0; fields without ghost planes (x wraps here) or — D3Q19 — with
// TWO ghost planes per side (slab decomposition: phase A also computes f(t+1) on the ghost planes -1 and nx from
// the neighbours' f(t) — all populations of their edge plane and the inward-moving ones of the plane behind it).
#pragma once
#include "step_kernel.hpp"

#ifndef XLB_STEP2_ALIGN
#define XLB_STEP2_ALIGN 256
#endif
#ifndef XLB_PIN_CLEAN
#define XLB_PIN_CLEAN true
#endif
#ifndef XLB_PIN_BC
#define XLB_PIN_BC true
#endif
#ifndef XLB_STEP2_SLACK
#define XLB_STEP2_SLACK 1  // the clean work items of a BC kernel run on the slack ring (one barrier per plane): cavity 512^3 -3.5 %
#endif
#ifndef XLB_STEP2_SLACK_PLAIN
#define XLB_STEP2_SLACK_PLAIN 0  // ... the kernel without boundary conditions does not: periodic 512^3 +3.5 % with it (profiles/r02/step2_sweeps.txt)
#endif
#ifndef XLB_STEP2_ROWMAP
#define XLB_STEP2_ROWMAP 0
#endif
#ifndef XLB_STEP2_ROWMAP_PLAIN
#define XLB_STEP2_ROWMAP_PLAIN 1
#endif
#ifndef XLB_STEP2_ROWMAP_CLEAN
#define XLB_STEP2_ROWMAP_CLEAN 0  // the BC-free body of the clean work items inside a BC kernel (slack ring): measurement builds
#endif
#ifndef XLB_STEP2_STAGE
#define XLB_STEP2_STAGE 0
#endif
#ifndef XLB_STEP2_MAX_BLOCKS
#define XLB_STEP2_MAX_BLOCKS 2
#endif
#ifndef XLB_STEP2_PLAIN_GMAX
#define XLB_STEP2_PLAIN_GMAX 1  // pair-group width of the kernel without boundary conditions
#endif
#ifndef XLB_STEP2_CLEAN_GMAX
#define XLB_STEP2_CLEAN_GMAX 3  // pair-group width of the BC-free body inside the BC kernel (1 makes hipcc park the pairs in scratch there)
#endif

namespace xlb {

// -DXLB_STEP2_TRACE=1|2 (tools/step2_phase_trace.py, never in the shipped build): shader-clock stamps of the steady-state loop's phases,
// per wave, for one block and a few planes.  2 additionally drains vmcnt at the top of finish_a, which separates "waiting for the pulls"
// from the collision (and perturbs the pipeline: the stores drain too).
#ifdef XLB_STEP2_TRACE
#ifndef XLB_STEP2_TRACE_BLOCK
#define XLB_STEP2_TRACE_BLOCK 300  // (cavity 512^3, 8 x-segments of 512 tile columns, hull tiles first: 522 = a hull tile's second segment, 812 = a clean item)
#endif
constexpr int TRACE_PLANES = 6, TRACE_EVENTS = 8, TRACE_WAVES = 11, TRACE_FIRST = 40, TRACE_BLOCK = XLB_STEP2_TRACE_BLOCK;
static __device__ unsigned long long g_step2_trace[TRACE_PLANES * TRACE_WAVES * TRACE_EVENTS];
__device__ __forceinline__ void trace_stamp(int d, int e) {
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long tm;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tm)::"memory");
  if (blockIdx.x == TRACE_BLOCK && d >= TRACE_FIRST && d < TRACE_FIRST + TRACE_PLANES && (threadIdx.x & 63) == 0)
    g_step2_trace[((d - TRACE_FIRST) * TRACE_WAVES + (threadIdx.x >> 6)) * TRACE_EVENTS + e] = tm;
  __builtin_amdgcn_sched_barrier(0);
}
#define XLB_TRACE(d, e) trace_stamp(d, e)
#else
#define XLB_TRACE(d, e)
#endif

// populations by c_z class (strip buffers)
template <class L>
constexpr int n_cz0() {
  int n = 0;
  for (int l = 0; l < L::Q; ++l) n += L::c(2, l) == 0 ? 1 : 0;
  return n;
}
template <class L>
constexpr int nth_cz(int cz, int k) {  // k-th population (index order) with c_z == cz
  for (int l = 0; l < L::Q; ++l)
    if (L::c(2, l) == cz && k-- == 0) return l;
  return 0;
}

// LDS ring of f(t+1), packed by population LIFETIME.  Plane p of f(t+1) is produced by phase A of iteration p - 2 and
// consumed by phase B of plane p + c_x (the pull f_l(x) <- f_l(x - c_x)): its c_x = -1 populations one iteration later,
// the c_x = 0 ones two, the c_x = +1 ones three iterations later.  So population l needs life(l) = 1 / 2 / 3 plane
// buffers instead of 3: D3Q19 5 + 18 + 15 = 38 population-planes instead of 57, D3Q27 54 instead of 81 (which is what
// makes a D3Q27 instantiation fit the 160 KiB at all).  Plane q (counted from x_lo - 1) of population l lives in buffer
// q % life(l).  With boundary conditions a halfway wall in phase B of plane x also reads the OWN cell's opposite
// populations of plane x — every group must still be there then, so the c_x = -1 group gets a second buffer (43 / 63).
// (Serving those reads by letting phase A pre-write the value into the entry phase B pulls from was built and dropped:
// two adjacent solid cells need two different values in the same entry.)
// PACKED = false keeps three whole planes ([plane % 3][population][cell], round 1's layout).  With pointer-per-buffer addressing the two
// forms run alike for D3Q19 (periodic 512^3 2.28 vs 2.29-2.32 ms/step, cavity 2.40-2.42 vs 2.41-2.44: profiles/r02/step2_sweeps.txt); the
// packed one is the default because D3Q27 needs it and the room it leaves pays for the slack ring of the clean work items.
// SLACK: one more buffer per population, so that phase A of plane x + 2 may WRITE while phase B of plane x still READS (no buffer
// phase B(x) reads — planes x - 1 / x / x + 1 of the c_x = +1 / 0 / -1 groups — is the one that receives plane x + 2): the barrier
// between the two phases goes away, every wave runs its phase B and its phase A back to back and the waves without output cells
// (3 of 11) start phase A at once.  2 + 3 + 4 buffers: D3Q19 5 x 2 + 9 x 3 + 5 x 4 = 57 population-planes again.
template <class L, int HASBC, bool PACKED, bool SLACK = false>
struct S2Ring {
  static_assert(!SLACK || (PACKED && HASBC == 0), "the slack ring is the lifetime-packed one, BC-free body");
  // layout: group-major — [c_x = -1 group: life x n_m planes][c_x = 0: 2 x n_z][c_x = +1: 3 x n_p], a plane buffer of a
  // group is contiguous, so that one VGPR base per group + an immediate offset per population addresses everything
  static constexpr int group(int l) { return L::c(0, l) + 1; }  // 0: c_x = -1, 1: c_x = 0, 2: c_x = +1
  static constexpr int gcount(int g) {
    int n = 0;
    for (int m = 0; m < L::Q; ++m) n += group(m) == g ? 1 : 0;
    return n;
  }
  static constexpr int gidx(int l) {  // position of l inside its group
    int n = 0;
    for (int m = 0; m < l; ++m) n += group(m) == group(l) ? 1 : 0;
    return n;
  }
  static constexpr int glife(int g) { return !PACKED ? 3 : (g == 0 ? (HASBC != 0 ? 2 : 1) : (g == 1 ? 2 : 3)) + (SLACK ? 1 : 0); }
  // buffer of plane q (counted from x_lo - 1) in a group of `life` buffers
  template <int life>
  static __device__ __forceinline__ int buf(int q) {
    static_assert(life >= 1 && life <= 4, "ring life");
    return life == 1 ? 0 : (life == 2 ? (q & 1) : (life == 3 ? q % 3 : (q & 3)));
  }
  static constexpr int gbase(int g) {  // first population-plane of group g
    int b = 0;
    for (int k = 0; k < g; ++k) b += glife(k) * gcount(k);
    return b;
  }
  static constexpr int PLANES = gbase(3);  // population-planes held in LDS
};

// tile geometry of one instantiation
template <class L, int HASBC, int TY, int TZ, bool PACKED>
struct S2Geom {
  static constexpr int EY = TY + 2, EZ = TZ + 2;
  static constexpr int NE = EY * EZ;                     // cells of f(t+1) per plane slot (grown tile)
  static constexpr int NB = TY * TZ;                     // output cells per plane
  static constexpr int THREADS = (NE + 63) / 64 * 64;    // whole waves covering the grown tile
  // STAGE: phase B hands its results to the memory pipeline as 16-byte stores — each wave writes the 19 values of its 64 cells (one z
  // row of the tile) to a wave-private LDS area and reads them back four cells of one population per lane: 5 store instructions per
  // wave and plane instead of 19 (the vector-memory pipeline is paid per instruction).  Needs the packed ring's room: 19 x 64 x 4 B per wave.
  static constexpr int EXTRA_BYTES = HASBC != 0 ? 3 * NE * 4 + 1024 : 8;  // meta words + BC constants
  // the BC-free body (the whole kernel, or the clean work items of a BC kernel) runs on the slack ring where that fits the LDS
  static constexpr int PLANES_BC = S2Ring<L, HASBC, PACKED>::PLANES;
  static constexpr int PLANES_SLACK = S2Ring<L, 0, true, true>::PLANES;
  static constexpr bool SLACK = XLB_STEP2_SLACK != 0 && PACKED && PLANES_SLACK * NE * 4 + EXTRA_BYTES <= 160 * 1024;
  static constexpr bool SLACK_USED = SLACK && (HASBC != 0 || XLB_STEP2_SLACK_PLAIN != 0);
  static constexpr int RING_PLANES = SLACK_USED && PLANES_SLACK > PLANES_BC ? PLANES_SLACK : PLANES_BC;
  static constexpr bool STAGE = XLB_STEP2_STAGE != 0 && PACKED && TZ == 64 &&
                                RING_PLANES * NE * 4 + NB * L::Q * 4 + EXTRA_BYTES <= 160 * 1024;  // (D3Q27: no room)
  static constexpr int STAGE_ELEMS = STAGE ? (NB / 64) * L::Q * 64 : 0;
  static constexpr int N_STORES = STAGE ? (L::Q + 3) / 4 : L::Q;  // vector-memory stores per phase-B thread and plane
  static constexpr int LDS_BYTES = RING_PLANES * NE * 4 + STAGE_ELEMS * 4 + EXTRA_BYTES;  // ring + staging + meta words + BC constants
  static_assert(LDS_BYTES <= 160 * 1024, "tile does not fit the LDS");
  // blocks per CU the LDS admits (one for every tile built by default; XLB_STEP2_MAX_BLOCKS bounds the tuning variants)
  // -> waves per SIMD the register allocation must admit
  static constexpr int BLOCKS_PER_CU = (160 * 1024) / LDS_BYTES < XLB_STEP2_MAX_BLOCKS ? (160 * 1024) / LDS_BYTES : XLB_STEP2_MAX_BLOCKS;
  static constexpr int WAVES_PER_SIMD = (BLOCKS_PER_CU * (THREADS / 64) + 3) / 4;
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0), i.e. every
// global load AND store in flight, which would serialise the software pipeline (prefetched pulls of the
// next plane, asynchronous stores of the finished one) at each of the two barriers per plane.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// meta word (built per run by k_build_meta): kind | slot << 4 | missing << 8
// Two layouts of the 32 bits (k_build_meta, ops_kernels.hpp, writes them):
//   up to 24 populations (D3Q19): kind in bits 0-3, slot in 4-7, the missing bit-set from bit 8;
//   D3Q27 ("wide"): kind in bits 0-2, slot in 3-5, missing bits 1 .. 26 from bit 6.  Bit 0 — the rest population — is dropped:
//   it is only ever set at solid cells given by interior indices, where a halfway wall "redirects" f_0 to the own cell's f_0.
// The halfway wall WITH a moving-wall term is a kind of its own inside the word (15, wide: 5); the public kinds it can hold end at 3.
template <class L>
struct S2Meta {
  static constexpr bool WIDE = L::Q > 24;
  static constexpr unsigned KBITS = WIDE ? 3 : 4, MSHIFT = 2 * KBITS;
  static constexpr unsigned K_HWM = WIDE ? 5u : 15u;
  static __device__ __forceinline__ unsigned kind(unsigned w) { return w & ((1u << KBITS) - 1u); }
  static __device__ __forceinline__ unsigned slot(unsigned w) { return (w >> KBITS) & ((1u << KBITS) - 1u); }
  static __device__ __forceinline__ unsigned missing(unsigned w) { return WIDE ? (w >> MSHIFT) << 1 : w >> MSHIFT; }
};

// Union over the wave of the missing bits of its halfway-wall lanes, as a UNIFORM (scalar) word: 0 for a fluid
// wave; for a z-face wave 5 of 19 bits.  One scalar test of it separates fluid waves (straight-line code) from
// boundary waves, whose per-population work is branch-free selects.  OR-reduction by DPP row shifts (lane 15 of
// every row ends up with its row's union) + 4 readlanes: 11 instructions.  Must be called with ALL lanes active
// (v_readlane ignores EXEC: an inactive lane 15/31/47/63 would deliver a stale register): idle lanes pass m = 0.
__device__ __forceinline__ unsigned wave_or(unsigned v) {
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);  // row_shr:1
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);  // row_shr:2
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true);  // row_shr:4
  v |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true);  // row_shr:8
  return (unsigned)__builtin_amdgcn_readlane((int)v, 15) | (unsigned)__builtin_amdgcn_readlane((int)v, 31) |
         (unsigned)__builtin_amdgcn_readlane((int)v, 47) | (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// slot (0..7) of a bc id in the packed kernel arguments; ids beyond the first 8 make the stepper ineligible
template <class T, class S>
__device__ __forceinline__ unsigned slot_of(const StepArgs<T, S>& a, unsigned id) {
  unsigned slot = 0;
#pragma unroll
  for (int s = 0; s < MAX_FAST_BCS; ++s) {
    const unsigned sid = (unsigned)(a.ids_packed >> (8 * s)) & 0xffu;
    slot = (id == sid) ? (unsigned)s : slot;
  }
  return slot;
}

// SLAB: the fields carry two ghost planes per side and the launcher passes src / dst / meta advanced to interior
// plane 0, so that plane indices -2 .. nx + 1 address the ghosts directly and x never wraps.  (A compile-time
// switch: the kernel sits at the SGPR limit — 19 uniform row bases per plane — and a run-time halo offset pushed
// it into scratch spills.)
// planes [x_lo, x_hi) of segment `seg` of a tile column.  Uniform cuts, or — x_cap > 0, at least 3 segments — thin first and
// last segments of x_cap planes: with walls on the x faces only those two segments of an interior tile column contain
// boundary cells, and everything else of the column runs the BC-free body (k_step2 below).
__device__ __forceinline__ void step2_seg_range(int x_begin, int x_count, int n_seg, int x_cap, int seg, int& x_lo, int& x_hi) {
  if (x_cap > 0 && n_seg >= 3) {
    const int inner = x_count - 2 * x_cap, m = n_seg - 2;
    x_lo = seg == 0 ? x_begin : x_begin + x_cap + (int)(((long)inner * (seg - 1)) / m);
    x_hi = seg == 0 ? x_begin + x_cap : (seg == n_seg - 1 ? x_begin + x_count : x_begin + x_cap + (int)(((long)inner * seg) / m));
    if (seg == n_seg - 1) x_lo = x_begin + x_count - x_cap;
  } else {
    x_lo = x_begin + (int)(((long)x_count * seg) / n_seg);
    x_hi = x_begin + (int)(((long)x_count * (seg + 1)) / n_seg);
  }
}

// The body of the kernel for one block; the LDS arrays belong to the __global__ wrapper below (so that the wrapper can
// run either boundary-condition form of the body in the same allocation).
// STRIPS (D3Q19 / (TY x 64) tiles): bit 0 = phase A reads the halo columns of its grown tile from the source field's STRIP buffer,
// bit 1 = phase B also writes the destination field's strips.  See "Strip buffers" below.
template <class L, class T, class S, int COLL, int HASBC, int TY, int TZ, bool SLAB, bool PACKED, int GMAX, bool PIN, bool FAST, bool SLACK = false, int STRIPS = 0>
__device__ __forceinline__ void step2_body(const StepArgs<T, S>& a, S* lds, unsigned* ldsmeta, T* bcval, S* strip_stage = nullptr) {
  using G = S2Geom<L, HASBC, TY, TZ, PACKED>;
  using R = S2Ring<L, HASBC, PACKED, SLACK>;
  using M = S2Meta<L>;
  constexpr int Q = L::Q, NE = G::NE, EZ = G::EZ;
  constexpr unsigned ES = sizeof(S);
#if defined(XLB_STRIPS_NOREAD)  // measurement builds (tools/r03_call4.sh): one half of the mechanism at a time
  constexpr bool SR = false, SW = (STRIPS & 2) != 0;
#elif defined(XLB_STRIPS_NOWRITE)
  constexpr bool SR = (STRIPS & 1) != 0, SW = false;
#else
  constexpr bool SR = (STRIPS & 1) != 0, SW = (STRIPS & 2) != 0;
#endif
  static_assert(STRIPS == 0 || STRIPS == 2 || STRIPS == 3 || STRIPS == 4, "STRIPS: 0, 2 (write), 3 (read + write) or 4 (row-aligned lanes only)");
  static_assert(STRIPS == 0 || (TZ == 64 && sizeof(S) == 4), "strip buffers: (TY x 64) tiles, 4-byte store type");
  // f(t+1) lives in LDS in the STORE type: the single-step kernel rounds it to that type on its way through memory, so the
  // ring holds exactly what two single steps would have stored, whatever the compute type (fp64 KBC: FP64FP32)
  static_assert(sizeof(S) == 4, "k_step2 keeps f(t+1) as 4-byte store values in LDS");
  // first cell of the buffer that holds population l of plane q (q counted from x_lo - 1; uniform)
  auto ring = [&](auto lc, int q) __attribute__((always_inline)) -> int {
    constexpr int l = decltype(lc)::value;
    if constexpr (!PACKED) {
      return (q % 3) * (Q * NE) + l * NE;
    } else {
      constexpr int g = R::group(l), life = R::glife(g);
      return (R::gbase(g) + R::template buf<life>(q) * R::gcount(g)) * NE + R::gidx(l) * NE;
    }
  };

  // tile of this block: from the launch's order table when there is one (hull tiles first — with boundary
  // conditions they are the expensive ones, and a CU that starts with one should get a cheap one next), else
  // a compact patch of tiles per XCD (blocks b and b + 8 share an XCD)
  const unsigned tiles_z = (unsigned)a.nz / TZ, tiles_y = (unsigned)a.ny / TY, n_tiles = tiles_y * tiles_z;
  // a tile column may be cut into x segments (finer work items -> better balance when hull tiles are expensive);
  // each segment warms its own 3-plane window up, so the result does not depend on the cut
  const unsigned seg = blockIdx.x / n_tiles, slot_in_seg = blockIdx.x % n_tiles;
  unsigned tile = slot_in_seg;
  if (a.tile_order) {
    tile = a.tile_order[slot_in_seg];
  } else if (a.xcd_swizzle) {
    const unsigned per_xcd = n_tiles / 8u;  // launcher guarantees divisibility
    tile = (slot_in_seg % 8u) * per_xcd + slot_in_seg / 8u;
  }
  const int n_seg = a.x_segments > 0 ? a.x_segments : 1;
  int x_lo, x_hi;
  step2_seg_range(a.x_begin, a.x_count, n_seg, a.x_cap, (int)seg, x_lo, x_hi);
  // The tiling may be shifted by (tile_oy, tile_oz) cells, periodically: with walls on the y / z faces a shift of half a
  // tile puts BOTH walls of an axis into the one tile row that wraps around the box, halving the number of hull tiles
  // (the expensive ones: every wave of a z-face tile is a boundary wave) — 71 instead of 140 at 512^2.
  const int ty0 = (int)(tile / tiles_z) * TY + a.tile_oy, tz0 = (int)(tile % tiles_z) * TZ + a.tile_oz;
  const int t = threadIdx.x;
  const int nx = a.nx, ny = a.ny, nz = a.nz;
  const size_t plane_cells = (size_t)ny * nz;

  if constexpr (HASBC != 0) {
    if (t < MAX_FAST_BCS * 32) {
      const unsigned s_ = (unsigned)t / 32u, l_ = (unsigned)t % 32u;
      const unsigned id_ = (unsigned)(a.ids_packed >> (8 * s_)) & 0xffu;
      bcval[t] = (l_ < 27u && id_ != 0u) ? a.bc_values[id_ * 27u + l_] : T(0);
    }
  }

  // IMPORTANT for the software pipeline: every global load / store of the steady-state loop is issued
  // UNCONDITIONALLY and in straight-line order by every thread.  Divergent code around VMEM makes hipcc's
  // s_waitcnt insertion assume the smallest possible number of younger operations at the joins, i.e. it
  // falls back to vmcnt(0), which drains the prefetched pulls and the asynchronous stores every plane.
  // Idle lanes therefore shadow a valid cell (loads) or aim at the padding behind the plane (stores).

  // ---- phase-A cell of this thread (grown tile, periodic images) ----
  const bool act_a = t < NE;
  const int ta = act_a ? t : NE - 1;
  // row-aligned lanes ((TY x 64) tiles): wave j < EY pulls the 64 interior cells of grown row j (for c_z = 0 one aligned 256-byte
  // piece per pull instead of the tail of one row + the head of the next), the last wave the two halo columns of all rows
  // Row-aligned lanes are the default of the stand-alone BC-free kernel (two barriers per plane): periodic 512^3 2.28 -> 2.11-2.16
  // ms/step in round 3's A/B (round 2 measured -2...3 %); the bodies of the BC kernel lose 0-3 % with it and keep the dense mapping.
  // (STRIPS & 4: row-aligned lanes for this body WITHOUT strip buffers — the run-time A/B of api.hip's "fuse2_rowmap" option)
  constexpr bool ROWMAP = (XLB_STEP2_ROWMAP != 0 || SR || (STRIPS & 4) != 0 || (XLB_STEP2_ROWMAP_PLAIN != 0 && HASBC == 0 && !SLACK) ||
                           (XLB_STEP2_ROWMAP_CLEAN != 0 && HASBC == 0 && SLACK)) && TZ == 64;
  // strip buffers: the last wave holds the 2 x EY halo-column cells of the grown tile (ROWMAP) and pulls for them from the
  // strips — the same instructions as every other wave, with the strip buffer's geometry in place of the field's
  const bool halo_wave = SR && __builtin_amdgcn_readfirstlane(t) >= G::EY * 64;
  const unsigned tile_zc = tile % tiles_z;  // tile column: boundary tile_zc is its left edge, tile_zc + 1 (periodic) its right edge
  const int ja = !ROWMAP ? ta / EZ : (ta < G::EY * 64 ? ta / 64 : (ta - G::EY * 64) / 2);
  const int ka = !ROWMAP ? ta % EZ : (ta < G::EY * 64 ? 1 + ta % 64 : ((ta & 1) ? EZ - 1 : 0));
  const int slot_a = ja * EZ + ka;  // my cell inside a grown-tile slot
  int ya = ty0 - 1 + ja, za = tz0 - 1 + ka;
  ya = ya < 0 ? ya + ny : (ya >= ny ? ya - ny : ya);
  za = za < 0 ? za + nz : (za >= nz ? za - nz : za);
  unsigned Yb[3], Zb[3];  // byte offsets of the y / z neighbours (index c + 1 -> source = coordinate - c)
  Yb[0] = (unsigned)((ya + 1 == ny) ? 0 : ya + 1) * (unsigned)nz * ES;
  Yb[1] = (unsigned)ya * (unsigned)nz * ES;
  Yb[2] = (unsigned)((ya == 0) ? ny - 1 : ya - 1) * (unsigned)nz * ES;
  Zb[0] = (unsigned)((za + 1 == nz) ? 0 : za + 1) * ES;
  Zb[1] = (unsigned)za * ES;
  Zb[2] = (unsigned)((za == 0) ? nz - 1 : za - 1) * ES;
  if constexpr (SR) {
    if (halo_wave) {
      // S_l[x][b][y][j] = f_l(x, y, z_b - 1 - c_z(l) + j): what the halo cell at z_b - 1 (j = 0: left halo column of the tile
      // whose left edge is boundary b) or at z_b (j = 1: right halo column of the tile whose right edge it is) pulls, whatever c_z
      const unsigned b = ka == 0 ? tile_zc : (tile_zc + 1 == tiles_z ? 0u : tile_zc + 1);
      const unsigned zoff = (b * (unsigned)ny * 2u + (ka == 0 ? 0u : 1u)) * ES;
      Yb[0] = (unsigned)((ya + 1 == ny) ? 0 : ya + 1) * 2u * ES;
      Yb[1] = (unsigned)ya * 2u * ES;
      Yb[2] = (unsigned)((ya == 0) ? ny - 1 : ya - 1) * 2u * ES;
      Zb[0] = Zb[1] = Zb[2] = zoff;
    }
  }
#if defined(XLB_STEP2_WHATIF) && XLB_STEP2_WHATIF != 0
  // MEASUREMENT ONLY (wrong results; never in the shipped build — tools/r03_whatif.sh): what would phase A's pulls cost if the
  // halo columns (bit 0) / halo rows (bit 1) of the grown tile came for free?  Sources outside the tile proper are clamped onto
  // its edge cells, so a block touches no sector that belongs to a neighbouring tile: the upper bound of what compact strip
  // buffers (bit 0) or perfect L2 sharing between neighbouring tiles (bits 0 + 1) could save.
  {
    auto zsrc = [&](int cz) {
      int kk = ka - cz;  // grown z index of the source: -1 .. TZ + 2; the tile proper is 1 .. TZ
      if (XLB_STEP2_WHATIF & 1) kk = kk < 1 ? 1 : (kk > TZ ? TZ : kk);
      int z = tz0 - 1 + kk;
      z = z < 0 ? z + nz : (z >= nz ? z - nz : z);
      return (unsigned)z * ES;
    };
    auto ysrc = [&](int cy) {
      int jj = ja - cy;
      if (XLB_STEP2_WHATIF & 2) jj = jj < 1 ? 1 : (jj > TY ? TY : jj);
      int y = ty0 - 1 + jj;
      y = y < 0 ? y + ny : (y >= ny ? y - ny : y);
      return (unsigned)y * (unsigned)nz * ES;
    };
    for (int c = -1; c <= 1; ++c) {
      Zb[c + 1] = zsrc(c);
      Yb[c + 1] = ysrc(c);
    }
  }
#endif
  const unsigned cell_a = (unsigned)ya * (unsigned)nz + (unsigned)za;

  // ---- phase-B cell of this thread (tile proper); waves beyond the tile shadow a valid cell ----
  const bool act_b = t < G::NB;
  const int tb = act_b ? t : t % G::NB;
  const int jb = tb / TZ, kb = tb % TZ;
  int yb = ty0 + jb, zb = tz0 + kb;
  yb = yb >= ny ? yb - ny : yb;
  zb = zb >= nz ? zb - nz : zb;
  const unsigned cell_b = (unsigned)yb * (unsigned)nz + (unsigned)zb;
  const int ctr_b = (jb + 1) * EZ + (kb + 1);  // my cell inside a grown-tile slot
  // staged stores (S2Geom::STAGE): lane i of a phase-B wave stores cells 4 (i % 16) .. + 3 of the wave's row for population 4 g + i / 16;
  // byte offset of that piece from population 4 g's row of the plane (the tile's z shift is a multiple of 4: a piece never straddles the wrap)
  // (the last group holds Q % 4 populations: its idle lanes repeat the last one — same bytes to the same address, no branch around the store)
  size_t stage_off = 0, stage_off_last = 0;
  constexpr int SUB_LAST = (Q - 1) % 4;
  const int sub = (t >> 4) & 3, sub_last = sub < SUB_LAST ? sub : SUB_LAST;
  if constexpr (G::STAGE) {
    int zc = tz0 + 4 * (t & 15);
    zc = zc >= nz ? zc - nz : zc;
    const size_t cell_off = (size_t)yb * (unsigned)nz + (unsigned)zc;
    stage_off = ((size_t)sub * a.plane_stride + cell_off) * ES;
    stage_off_last = ((size_t)sub_last * a.plane_stride + cell_off) * ES;
  }

  // logical plane p -> the plane phase A works on: without ghost planes x is periodic; with them planes past x_hi
  // are prefetches whose results are discarded (clamped, so that they stay inside the allocation)
  auto wrapx = [&](int p) __attribute__((always_inline)) {
    if constexpr (SLAB) {
      return p > x_hi ? x_hi : p;
    } else {
      p %= nx;
      return p < 0 ? p + nx : p;
    }
  };
  const ptrdiff_t pc = (ptrdiff_t)plane_cells;
  // where phase A pulls from
  const S* pull_base = a.src;
  size_t pull_stride = a.plane_stride;
  ptrdiff_t pull_pc = pc;
  if constexpr (SR) {
    if (halo_wave) {
      pull_base = a.strips_src;
      pull_stride = a.plane_stride >> 5;
      pull_pc = pc >> 5;
    }
  }
  auto meta_load = [&](int plane, unsigned cell) __attribute__((always_inline)) -> unsigned { return (a.meta + (ptrdiff_t)plane * pc)[cell]; };

  auto finish = [&](T(&f)[Q], bool fullway) __attribute__((always_inline)) {
    if (!fullway) {
      if constexpr (COLL == XLBHIP_BGK && sizeof(T) == 4 && FAST)
        bgk_fast<L>(f, a.omega);  // tolerance-graded (cell.hpp); the default, exact_math=1 selects the bit-exact form below
      else if constexpr (COLL == XLBHIP_BGK && sizeof(T) == 4)
        collide_bgk_packed<L, GMAX, PIN>(f, a.omega);  // same arithmetic, fewer issue slots (cell.hpp); GMAX: measured best 3 / 1 (with / without BCs)
      else
        collide<L, T, COLL>(f, a.omega, a.extra);
    } else {
      static_for<Q>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        constexpr int o = opp<L>(l);
        if constexpr (l < o) {
          const T tmp = f[l];
          f[l] = f[o];
          f[o] = tmp;
        }
      });
    }
  };

  // phase A, first half: issue the Q pulls of plane x from f(t) (kept in registers across phase B).
  // Boundary lanes redirect a pull: a halfway wall reads the own cell's OPPOSITE population for its missing
  // directions — same instruction, same register, nothing dependent later.  Both arms of the wave-uniform
  // branch issue exactly Q loads.
  auto issue_a = [&](int x, S(&raw)[Q], unsigned w, unsigned& mall) __attribute__((always_inline)) {
    mall = 0;
    if constexpr (HASBC != 0) {
      const unsigned kind_ = M::kind(w);
      mall = wave_or((act_a && (kind_ == K_HW || kind_ == M::K_HWM)) ? M::missing(w) : 0u);  // all lanes active here
    }
    if (!act_a) return;  // only the tail of the last wave is idle
    int Xs[3];  // storage planes of the sources (index c_x + 1 -> plane x - c_x)
    Xs[0] = SLAB ? x + 1 : ((x + 1 == nx) ? 0 : x + 1);
    Xs[1] = x;
    Xs[2] = SLAB ? x - 1 : ((x == 0) ? nx - 1 : x - 1);
    if constexpr (HASBC != 0) {
      if (mall != 0u) {
        // Boundary wave (round 3): a halfway-wall lane pulls each of its missing directions from its OWN cell's OPPOSITE population —
        // inside the same pull instruction, through a per-lane 64-bit address (4 VALU per pull, boundary waves only).  Exactly Q loads
        // on either side of this wave-uniform branch, all of them visible to the compiler: its vmcnt model of the loop stays exact
        // WITHOUT round 2's extra inline-asm loads, their hand-counted wait and the no-copy / no-spill invariants they needed —
        // and a hull tile's pull phase issues 19 instructions instead of 24-29 (profiles/r03/step2_redirect.md).
        const unsigned kind = M::kind(w);
        const unsigned m = (kind == K_HW || kind == M::K_HWM) ? M::missing(w) : 0u;
        const unsigned voff = cell_a * ES;
        static_for<Q>([&](auto lc) {
          constexpr int l = decltype(lc)::value;
          constexpr int cx = L::c(0, l), cy = L::c(1, l), cz = L::c(2, l);
          const S* row = pull_base + (size_t)l * pull_stride + (ptrdiff_t)Xs[cx + 1] * pull_pc;         // uniform
          const S* own = a.src + (size_t)opp<L>(l) * a.plane_stride + (ptrdiff_t)x * pc;                // uniform (the field itself, also for the halo wave of a strip build)
          const bool red = ((m >> l) & 1u) != 0u;
          const char* p = reinterpret_cast<const char*>(red ? own : row) + (red ? voff : Yb[cy + 1] + Zb[cz + 1]);
          raw[l] = *reinterpret_cast<const S*>(p);
        });
        return;
      }
    }
    static_for<Q>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      constexpr int cx = L::c(0, l), cy = L::c(1, l), cz = L::c(2, l);
      // uniform; for the halo wave of a strip-buffer build `pull_base` is the strip buffer, which mirrors the field's population /
      // plane structure at 1 / 32 of its size (pull_stride, pull_pc: chosen once per wave, so the per-population arithmetic is the same)
      const S* row = pull_base + (size_t)l * pull_stride + (ptrdiff_t)Xs[cx + 1] * pull_pc;
      raw[l] = ld(row, Yb[cy + 1] + Zb[cz + 1]);
    });
  };
  // boundary kinds that replace whole cells (halfway walls are handled where the pulls are redirected)
  auto bc_regs = [&](T(&f)[Q], unsigned w, bool& fullway) __attribute__((always_inline)) {
    const unsigned kind = M::kind(w);
    if (kind == K_EQ) {
      const T* val = bcval + opaque(M::slot(w) * 32u);
      static_for<Q>([&](auto lc) { f[decltype(lc)::value] = val[decltype(lc)::value]; });
    } else if (kind == K_FW) {
      fullway = true;
    }
  };
  // halfway wall, population l of a lane whose missing bit is set: the redirected pull `got` (own cell, opposite
  // population) + the moving-wall term; the reference adds its 0.0 term for no-slip walls too (bc_halfway_bounce_back.py:116-134).
  // Branch-free over the wave: BRANCHES, not arithmetic, made boundary waves slow (z-face tiles cost 2.25x a fluid tile
  // with per-population scalar branches in four places).  any_moving is wave-uniform.
  auto hw_apply = [&](T(&f)[Q], const T(&got)[Q], unsigned mm, unsigned w, bool any_moving) __attribute__((always_inline)) {
    if (!any_moving) {
      static_for<Q>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        const T cand = got[l] + T(0);
        f[l] = ((mm >> l) & 1u) ? cand : f[l];
      });
    } else {
      const T* val = bcval + M::slot(w) * 32u;
      const bool mov = M::kind(w) == M::K_HWM;
      static_for<Q>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        const T cand = got[l] + (mov ? val[l] : T(0));
        f[l] = ((mm >> l) & 1u) ? cand : f[l];
      });
    }
  };
  // phase A, second half: BCs + collision -> LDS slot (populations and the cell's meta word)
  // q: plane counted from x_lo - 1 (ring buffers q & 1 and q % 3; meta slot q % 3)
  auto finish_a = [&](const S(&raw)[Q], unsigned w, unsigned mall, int q) __attribute__((always_inline)) {
    if (!act_a) return;
    T f[Q];
    static_for<Q>([&](auto lc) { f[decltype(lc)::value] = to_compute<T, S>(raw[decltype(lc)::value]); });
    bool fullway = false;
    if constexpr (HASBC != 0) {
      const unsigned kind = M::kind(w);
      const bool hw = kind == K_HW || kind == M::K_HWM;
      if (mall != 0u) {  // wave-uniform: some halfway-wall lane of this wave had pulls redirected
        // the redirected values arrived with the pulls (issue_a): f[l] already IS the own cell's opposite population where the bit is set
        const unsigned mm = hw ? M::missing(w) : 0u;
        hw_apply(f, f, mm, w, __builtin_amdgcn_ballot_w64(kind == M::K_HWM) != 0ull);
      }
      if (kind != 0u) bc_regs(f, w, fullway);
    }
    finish(f, fullway);
    if (act_a) {
      if constexpr (!PACKED) {
        S* dst = lds + (q % 3) * (Q * NE) + slot_a;
        static_for<Q>([&](auto lc) {
          constexpr int l = decltype(lc)::value;
          dst[l * NE] = to_store<S, T>(f[l]);  // f(t+1) passes through the store precision
        });
      } else {
        // one pointer per group buffer + a compile-time offset per population (pointer form: see phase B)
        S* gb[3];
        static_for<3>([&](auto gc) {
          constexpr int g = decltype(gc)::value, life = R::glife(g);
          gb[g] = lds + (R::gbase(g) + R::template buf<life>(q) * R::gcount(g)) * NE + slot_a;
        });
        static_for<Q>([&](auto lc) {
          constexpr int l = decltype(lc)::value;
          gb[R::group(l)][R::gidx(l) * NE] = to_store<S, T>(f[l]);
        });
      }
      if constexpr (HASBC != 0) ldsmeta[(q % 3) * NE + slot_a] = w;
    }
  };

  // phase B: f(t+2) on plane x from the ring (planes x-1, x, x+1) -> global (Q unconditional stores)
  // d = x - x_lo: plane x is plane q = d + 1 of the ring
  auto phase_b_compute = [&](T(&f)[Q], int d) __attribute__((always_inline)) {
    const int q0 = d + 1;  // plane x; the pull of population l comes from plane q0 - c_x(l)
    unsigned w = 0;
    if constexpr (HASBC != 0) w = ldsmeta[(q0 % 3) * NE + ctr_b];
    unsigned mall = 0, mm = 0;
    if constexpr (HASBC != 0) {
      const unsigned kind = M::kind(w);
      mm = (kind == K_HW || kind == M::K_HWM) ? M::missing(w) : 0u;
      mall = wave_or(mm);  // uniform; all lanes of the wave are active here (NB is a whole number of waves)
    }
    if (mall == 0u) {  // fluid wave (or only fullway / equilibrium lanes): one scalar branch, then straight-line reads
      if constexpr (!PACKED) {
        const S* base[3];  // index c_x + 1 -> plane x - c_x
        base[0] = lds + ((q0 + 1) % 3) * (Q * NE);
        base[1] = lds + (q0 % 3) * (Q * NE);
        base[2] = lds + ((q0 - 1) % 3) * (Q * NE);
        static_for<Q>([&](auto lc) {
          constexpr int l = decltype(lc)::value;
          constexpr int cx = L::c(0, l), cy = L::c(1, l), cz = L::c(2, l);
          f[l] = to_compute<T, S>(base[cx + 1][l * NE + ctr_b - cy * EZ - cz]);
        });
      } else {
        // (pointer + immediate form; indexing one array with run-time buffer offsets — lds[ring(l, q) + ...] — compiles to
        // the same instruction mix but measured 9 % slower on the periodic box: profiles/r02/step2_sweeps.txt)
        const S* gb[3];  // group g = c_x + 1 comes from plane q0 - c_x
        static_for<3>([&](auto gc) {
          constexpr int g = decltype(gc)::value, life = R::glife(g);
          const int qs = q0 - (g - 1);
          gb[g] = lds + (R::gbase(g) + R::template buf<life>(qs) * R::gcount(g)) * NE;
        });
        static_for<Q>([&](auto lc) {
          constexpr int l = decltype(lc)::value;
          constexpr int cy = L::c(1, l), cz = L::c(2, l);
          f[l] = to_compute<T, S>(gb[R::group(l)][R::gidx(l) * NE + ctr_b - cy * EZ - cz]);
        });
      }
    } else {
      // boundary wave: a lane whose missing bit l is set reads its own cell's opposite population instead (select on
      // the LDS index, no branch), then the halfway-wall terms are applied to exactly those populations
      T got[Q];
      static_for<Q>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        constexpr int cx = L::c(0, l), cy = L::c(1, l), cz = L::c(2, l);
        const int idx_pull = ring(lc, q0 - cx) + ctr_b - cy * EZ - cz;
        const int idx_own = ring(std::integral_constant<int, opp<L>(l)>{}, q0) + ctr_b;
        got[l] = to_compute<T, S>(lds[((mm >> l) & 1u) ? idx_own : idx_pull]);
        f[l] = got[l];
      });
      hw_apply(f, got, mm, w, __builtin_amdgcn_ballot_w64(M::kind(w) == M::K_HWM) != 0ull);
    }
    bool fullway = false;
    if constexpr (HASBC != 0) {
      if (M::kind(w) != 0u) bc_regs(f, w, fullway);
    }
    finish(f, fullway);
  };
  // Tried and rejected (profiles/r02/step2_sweeps.txt): pulls running TWO planes ahead (second register buffer, loop unrolled by
  // two; 153 VGPRs) — 0.5 % on the periodic box, 1 % on the cavity: the memory latency behind one phase B is not what the waves wait for.
  // Tried and rejected: issuing the 19 stores from EVERY wave outside the branch (buffer stores, idle waves aimed past
  // the end of the buffer) makes hipcc's vmcnt model exact — with the stores inside `if (act_b)` it waits for the
  // prefetched pulls with vmcnt(19)...vmcnt(1), i.e. drains the wave's own stores every plane — but ran 9 % SLOWER
  // without boundary conditions and the same with them: the kernel is bound by VALU issue, not by these waits.
  auto phase_b = [&](int x, int d) __attribute__((always_inline)) {
    if (!act_b) return;  // whole waves (NB % 64 == 0)
    T f[Q];
    phase_b_compute(f, d);
    if constexpr (G::STAGE) {
      // the wave's 64 cells are one z row of the tile: population-major into the wave's staging area, back as (population, 4 cells)
      S* st = lds + G::RING_PLANES * NE + (t >> 6) * (Q * 64);
      const int lane = t & 63;
      static_for<Q>([&](auto lc) { st[decltype(lc)::value * 64 + lane] = to_store<S, T>(f[decltype(lc)::value]); });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (LDS operations of one wave complete in order; this orders the compiler's view too)
      char* xrow = reinterpret_cast<char*>(a.dst + (ptrdiff_t)x * pc);  // uniform
      static_for<G::N_STORES>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        constexpr bool last = 4 * g + 3 >= Q;
        typedef typename VecOf<S, 4>::aligned V4;
        const V4 v = *reinterpret_cast<const V4*>(st + (4 * g + (last ? sub_last : sub)) * 64 + (lane & 15) * 4);
        __builtin_nontemporal_store(v, reinterpret_cast<V4*>(xrow + (size_t)(4 * g) * a.plane_stride * ES + (last ? stage_off_last : stage_off)));
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the staging area is rewritten by the next plane's phase B
    } else {
      static_for<Q>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        S* drow = a.dst + (size_t)l * a.plane_stride + (ptrdiff_t)x * pc;  // uniform
        S v[1] = {to_store<S, T>(f[l])};
        st_aligned<S, 1, true>(drow, cell_b * ES, v);
      });
    }
    if constexpr (SW) {
      // the two outermost cells of the row on either side feed the strips.  Each of those four lanes parks ALL its Q values in LDS
      // ([edge lane 0..3][row][population]: one exec-masked region, Q plain writes — selecting per population what the strips
      // need cost ~200 instructions per wave and plane, +20 % on the periodic box); the waves without output cells pick the 16
      // values per population the strips hold (c_z = -1: columns 0, 1; c_z = +1: columns TZ - 2, TZ - 1; c_z = 0: columns 0 and
      // TZ - 1) and write them out
      if (kb < 2 || kb >= TZ - 2) {
        // slot k of an edge lane: k < NZ1: its c_z = -1 (left lanes) / c_z = +1 (right lanes) populations in index order, then the c_z = 0 ones
        constexpr int NZ1 = (Q - n_cz0<L>()) / 2, NSLOT = NZ1 + n_cz0<L>();
        const bool left = kb < 2;
        S* stg = strip_stage + (d & 1) * (NSLOT * 4 * TY) + ((left ? kb : kb - (TZ - 4)) * TY + jb) * NSLOT;
        static_for<NZ1>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          stg[k] = to_store<S, T>(left ? f[nth_cz<L>(-1, k)] : f[nth_cz<L>(1, k)]);
        });
        static_for<n_cz0<L>()>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          stg[NZ1 + k] = to_store<S, T>(f[nth_cz<L>(0, k)]);
        });
      }
    }
  };
  // strips of plane x (staged by phase B of ring plane d, visible since the barrier that ended that iteration): the 192
  // lanes without output cells write 16 values of 12 populations per pass
  auto flush_strips = [&](int x, int d) __attribute__((always_inline)) {
    if constexpr (SW) {
      if (act_b) return;
      constexpr unsigned CZP = [] { unsigned m = 0; for (int l = 0; l < Q; ++l) m |= (L::c(2, l) == 1 ? 1u : 0u) << l; return m; }();
      constexpr unsigned CZM = [] { unsigned m = 0; for (int l = 0; l < Q; ++l) m |= (L::c(2, l) == -1 ? 1u : 0u) << l; return m; }();
      const int tl = t - G::NB, g = tl >> 4, i = tl & 15, r = i >> 1, e = i & 1;
      int yy = ty0 + r;
      yy = yy >= ny ? yy - ny : yy;
      const unsigned bL = tile_zc, bR = tile_zc + 1 == tiles_z ? 0u : tile_zc + 1;
      constexpr int GROUPS = (G::THREADS - G::NB) / 16;
      static_assert(GROUPS >= 1, "no idle lanes to write the strips");
#pragma unroll
      for (int rr = 0; rr * GROUPS < Q; ++rr) {
        const int l = rr * GROUPS + g;
        if (l < Q) {
          const int cz = (int)((CZP >> l) & 1u) - (int)((CZM >> l) & 1u);
          const unsigned b = cz < 0 ? bL : (cz > 0 ? bR : (e == 0 ? bL : bR));
          const unsigned j = cz != 0 ? (unsigned)e : (e == 0 ? 1u : 0u);
          // edge lane that produced it: c_z = -1 -> column e (lanes 0, 1); c_z = +1 -> column TZ - 2 + e (lanes 2, 3); c_z = 0 -> column 0 / TZ - 1;
          // slot: position of l among the populations of its c_z class (+ NZ1 for c_z = 0)
          constexpr int NZ1 = (Q - n_cz0<L>()) / 2, NSLOT = NZ1 + n_cz0<L>();
          constexpr unsigned CZ0 = ((1u << Q) - 1u) & ~(CZP | CZM);
          const int lane4 = cz < 0 ? e : (cz > 0 ? 2 + e : (e == 0 ? 0 : 3));
          const unsigned below = (1u << l) - 1u;
          const int k = cz < 0 ? __builtin_popcount(CZM & below) : (cz > 0 ? __builtin_popcount(CZP & below) : NZ1 + __builtin_popcount(CZ0 & below));
          const S v = strip_stage[(d & 1) * (NSLOT * 4 * TY) + (lane4 * TY + r) * NSLOT + k];
          a.strips_dst[(((ptrdiff_t)((size_t)l * a.plane_stride) + (ptrdiff_t)x * pc) >> 5) + (ptrdiff_t)(((size_t)b * (unsigned)ny + (unsigned)yy) * 2u + j)] = v;
        }
      }
    }
  };

  // Plane x_lo + p is plane q = p + 1 of the ring (p = -1: the periodic image / ghost plane below the segment).
  S raw[Q];
  unsigned w_raw = 0, mall_raw = 0;  // meta word / wave union of the plane held in raw
  // prologue: planes -1, 0, 1 straight into the ring (synchronous), then the pulls of plane 2 go in flight.
  for (int p = -1; p <= 1; ++p) {
    if constexpr (HASBC != 0) w_raw = meta_load(wrapx(x_lo + p), cell_a);
    issue_a(wrapx(x_lo + p), raw, w_raw, mall_raw);
    if (p == -1) __syncthreads();  // bcval ready before the first bc_regs
    finish_a(raw, w_raw, mall_raw, p + 1);
  }
  if constexpr (HASBC != 0) w_raw = meta_load(wrapx(x_lo + 2), cell_a);
  issue_a(wrapx(x_lo + 2), raw, w_raw, mall_raw);
  lds_barrier();
  // steady state, branch-free around VMEM: the pulls of plane x + 3 are in flight while phase B of plane x + 1 runs.
  // The last two trips prefetch again plane x_hi (discarded): 2 / nx extra work, no branch.
  for (int x = x_lo; x < x_hi; ++x) {
    const int d = x - x_lo;
    unsigned wa = 0;
    if constexpr (HASBC != 0) wa = meta_load(wrapx(x + 3), cell_a);  // used after phase B: a whole phase of latency cover
    XLB_TRACE(d, 0);
    phase_b(x, d);
    XLB_TRACE(d, 2);
    if constexpr (!SLACK) lds_barrier();  // every reader of the buffers about to be overwritten is done (the slack ring has a spare buffer instead)
    XLB_TRACE(d, 3);
#if defined(XLB_STEP2_TRACE) && XLB_STEP2_TRACE == 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    XLB_TRACE(d, 4);
#endif
    finish_a(raw, w_raw, mall_raw, d + 3);  // plane x + 2
    XLB_TRACE(d, 5);
    w_raw = wa;
    // (KBC: keep the scheduler from hoisting the next plane's pulls above the collision — 27 more live registers there
    // put the fp64 body into scratch)
    if constexpr ((COLL & 3) != XLBHIP_BGK) __builtin_amdgcn_sched_barrier(0);
    issue_a(wrapx(x + 3), raw, w_raw, mall_raw);
    // the strips of plane x - 1 leave here, BEHIND the pulls just issued: the waves that write them (the last ones of the block, the
    // plane's critical path) next wait for vector memory a whole iteration later, when these stores have long been acknowledged.
    // (At the top of the iteration the stores were younger than the pulls finish_a waits for: their acknowledgement latency landed
    // on every plane — periodic 512^3 2.28 -> 2.81 ms/step, profiles/r03/step2_strips.md.)
    if (d > 0) flush_strips(x - 1, d - 1);
    XLB_TRACE(d, 6);
    lds_barrier();  // plane x + 2 visible
    XLB_TRACE(d, 7);
  }
  if (x_hi > x_lo) flush_strips(x_hi - 1, x_hi - 1 - x_lo);
}

// Work items (tile column x x-segment) WITHOUT a single boundary cell in their grown tile and plane range run the body
// compiled without boundary conditions — no meta words, no wave unions, the leaner schedule — inside the same launch:
// `a.clean[blockIdx.x]` (k_step2_clean, rebuilt with the meta words every run) is block-uniform.  On the 512^3 cavity that
// is every segment of an interior tile column but its first and last one, 54 % of the items.  (Round 1 tried the same
// split as separate launches on two streams and lost to launch tails / kernel mixing; one launch has neither.)
template <class L, class T, class S, int COLL, int HASBC, int TY, int TZ, bool SLAB, bool PACKED, bool FAST, int STRIPS = 0>
__global__ void __attribute__((aligned(XLB_STEP2_ALIGN))) __launch_bounds__((S2Geom<L, HASBC, TY, TZ, PACKED>::THREADS), (S2Geom<L, HASBC, TY, TZ, PACKED>::WAVES_PER_SIMD)) k_step2(const StepArgs<T, S> a) {
  using G = S2Geom<L, HASBC, TY, TZ, PACKED>;
  __shared__ S lds[G::RING_PLANES * G::NE + G::STAGE_ELEMS];                          // the ring of f(t+1) (D3Q19, 8x64 tile, three-plane layout: 150 480 B -> one block per CU)
  __shared__ unsigned ldsmeta[HASBC ? 3 * G::NE : 1];           // [plane % 3][cell] kind | slot << 4 | missing << 8 of the f(t+1) cells
  __shared__ T bcval[HASBC ? MAX_FAST_BCS * 32 : 1];            // per-BC constants (feq of EquilibriumBC / moving-wall terms), by slot
  constexpr int STRIP_SLOTS = (L::Q - n_cz0<L>()) / 2 + n_cz0<L>();  // values an edge cell contributes: its c_z = -1 or +1 populations and the c_z = 0 ones
  __shared__ S strip_stage[(STRIPS & 2) ? 2 * STRIP_SLOTS * 4 * TY : 1];   // [ring plane & 1][edge lane][row][slot]: the plane's edge cells on their way to the strips
  static_assert(TY == 8 || (STRIPS & 2) == 0, "strip staging is laid out for 8-row tiles");
  static_assert(G::LDS_BYTES + ((STRIPS & 2) ? 2 * STRIP_SLOTS * 4 * TY * 4 : 0) <= 160 * 1024, "strip staging does not fit the LDS");
  if constexpr (HASBC != 0) {
    if (a.clean != nullptr && a.clean[blockIdx.x] != 0) {
      step2_body<L, T, S, COLL, 0, TY, TZ, SLAB, PACKED, XLB_STEP2_CLEAN_GMAX, XLB_PIN_CLEAN, FAST, G::SLACK, STRIPS>(a, lds, ldsmeta, bcval, strip_stage);
      return;
    }
  }
  step2_body<L, T, S, COLL, HASBC, TY, TZ, SLAB, PACKED, (HASBC != 0 ? 3 : XLB_STEP2_PLAIN_GMAX), (HASBC != 0 ? XLB_PIN_BC : false), FAST, (HASBC == 0 && G::SLACK && XLB_STEP2_SLACK_PLAIN != 0), STRIPS>(a, lds, ldsmeta, bcval, strip_stage);
}

// Strip buffers (round 3).  The grown tile's two halo COLUMNS cost phase A 64-byte sectors for 4-byte values — 5-6 sectors per
// 66-cell row for 4 of payload, in lines that belong to the neighbouring tile (another block, often another XCD's L2): reads
// 1.57x the field on the 512^3 cavity, and a what-if build that gets those columns for free runs 10 % faster
// (profiles/r03/step2_strips.md).  So every population field carries a STRIP buffer, 1 / 32 of its size, same population /
// plane structure:  S_l[x][b][y][j] = f_l(x, y, z_b - 1 - c_z(l) + j),  b = tile boundary (z_b = 64 b + tile_oz), j = 0, 1 —
// for every population exactly the two values the halo cells left and right of boundary b pull from row (x, y).  Phase B
// writes them with the field (two cells per row edge, staged through LDS, written by the waves without output cells: +3 %
// bytes); phase A's last wave — the 2 x EY halo cells — pulls from it: per population and plane 4 sectors instead of ~15.
// Anything else that writes the field invalidates its strips (api.hip: strips_version); the next pass then only writes them.
// clean[b] = 1 when no cell of work item b — grown tile (periodic images included), planes x_lo - 1 .. x_hi + 1 as the kernel
// visits them — carries a boundary condition.  Same block -> (tile, segment) mapping as k_step2; one block per item.
template <int TY, int TZ, bool SLAB>
__global__ void k_step2_clean(const uint32_t* meta /*advanced to interior plane 0 when SLAB*/, const uint32_t* tile_order, int xcd_swizzle, int x_segments,
                              int x_cap, int x_begin, int x_count, int nx, int ny, int nz, int tile_oy, int tile_oz, uint8_t* clean) {
  constexpr int EY = TY + 2, EZ = TZ + 2, NE = EY * EZ;
  const unsigned tiles_z = (unsigned)nz / TZ, tiles_y = (unsigned)ny / TY, n_tiles = tiles_y * tiles_z;
  const unsigned seg = blockIdx.x / n_tiles, slot_in_seg = blockIdx.x % n_tiles;
  unsigned tile = slot_in_seg;
  if (tile_order) {
    tile = tile_order[slot_in_seg];
  } else if (xcd_swizzle) {
    const unsigned per_xcd = n_tiles / 8u;
    tile = (slot_in_seg % 8u) * per_xcd + slot_in_seg / 8u;
  }
  const int n_seg = x_segments > 0 ? x_segments : 1;
  int x_lo, x_hi;
  step2_seg_range(x_begin, x_count, n_seg, x_cap, (int)seg, x_lo, x_hi);
  const int ty0 = (int)(tile / tiles_z) * TY + tile_oy, tz0 = (int)(tile % tiles_z) * TZ + tile_oz;
  const ptrdiff_t pc = (ptrdiff_t)ny * nz;
  int dirty = 0;
  for (int p = x_lo - 1; p <= x_hi + 1; ++p) {
    int xp = p;
    if constexpr (SLAB) {
      xp = p > x_hi ? x_hi : p;
    } else {
      xp %= nx;
      xp = xp < 0 ? xp + nx : xp;
    }
    for (int t = threadIdx.x; t < NE; t += blockDim.x) {
      int y = ty0 - 1 + t / EZ, z = tz0 - 1 + t % EZ;
      y = y < 0 ? y + ny : (y >= ny ? y - ny : y);
      z = z < 0 ? z + nz : (z >= nz ? z - nz : z);
      dirty |= (meta + (ptrdiff_t)xp * pc)[(size_t)y * nz + z] != 0u;
    }
  }
  dirty = __syncthreads_or(dirty);
  if (threadIdx.x == 0) clean[blockIdx.x] = dirty ? 0 : 1;
}

}  // namespace xlb
