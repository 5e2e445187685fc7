// The fused per-timestep kernel: pull-stream -> post-streaming BCs -> rho,u -> feq ->
// collide -> post-collision BCs -> store.  One launch per step.
//
// Replaces the reference's fused Warp kernel (xlb/operator/stepper/nse_stepper.py:427-464)
// with the step ORDER and semantics of the JAX branch (:237-282), which is the parity
// target.  Design:
//   * SoA populations (q, x, y, z), z fastest; a thread owns VEC consecutive z cells
//     and keeps their q populations in registers, so moments/feq/collide need no
//     cross-lane traffic at all.
//   * every global access is `uniform base (SGPR pair) + 32-bit byte offset (VGPR)`: x is
//     block-uniform (blockIdx.z), so the population-plane / x-plane part of each address is
//     scalar arithmetic and the q loads share nine per-thread (y, z) neighbour offsets.
//   * per direction ONE load per thread: aligned for c_z == 0, shifted by one element for
//     c_z = +-1 (dword-aligned vector load when VEC > 1; the single wrapped element at a row
//     end is patched by an exec-masked scalar load).
//   * boundary cells are handled in-register on a slow path taken only by lanes whose
//     bc_mask byte is non-zero; the id -> kind lookup is a scalar compare chain on kernel
//     arguments (no dependent table load), and missing bits (one u32 per cell) and own-cell
//     pre-stream populations are fetched only there.
//   * stores are aligned (vector) stores, optionally non-temporal.
#pragma once
#include "cell.hpp"

#ifndef XLB_LB1
#define XLB_LB1 256  // launch bound of the one-cell-per-thread variants
#endif

namespace xlb {

// kinds as the kernel sees them
enum { K_NONE = 0, K_EQ = XLBHIP_BC_EQUILIBRIUM, K_HW = XLBHIP_BC_HALFWAY_BB, K_FW = XLBHIP_BC_FULLWAY_BB, K_DN = XLBHIP_BC_DO_NOTHING };
constexpr int MAX_FAST_BCS = 8;
template <class T, class S>
struct StepArgs {
  const S* src;
  S* dst;
  const uint8_t* bc;      // (nx+2h, ny, nz) or nullptr
  const uint32_t* miss;   // (nx+2h, ny, nz) bit-sets or nullptr
  const uint32_t* tile_order;  // two-step kernel only: block -> tile, or nullptr
  int x_segments;              // two-step kernel only: x cuts per tile column (>= 1)
  int x_cap;                   // two-step kernel only: > 0: the first and last segment are x_cap planes thin (>= 3 segments)
  int tile_oy, tile_oz;        // two-step kernel only: origin shift of the tiling (periodic), see step2_kernel.hpp
  const uint32_t* meta;   // two-step kernel only: id | missing << 8 per cell (built per run), else nullptr
  const uint8_t* clean;   // two-step kernel only: per block, 1 = no boundary cell in its work item (k_step2_clean), or nullptr
  const S* strips_src;    // two-step kernel with strip buffers (step2_kernel.hpp): the source field's strips / the destination's,
  S* strips_dst;          //   advanced to interior plane 0 like src / dst; else nullptr
  const T* bc_values;     // [256][27]: feq of equilibrium BCs / moving-wall terms
  const uint8_t* bc_kind; // [256] K_*; used only when n_bc > MAX_FAST_BCS
  // per-cell prescribed values of Zou-He / Regularized BCs built with a profile (extended variant only): sorted
  // storage cell indices and 3 values per entry; a BC uses the table when bc_values[id][PROF_FLAG] != 0
  const uint32_t* prof_keys;
  const T* prof_vals;
  int n_prof;
  // wall-distance weights of HybridBC cells (extended variant only): sorted storage cell indices and q floats per entry
  const uint32_t* dist_keys;
  const float* dist_vals;
  int n_dist;
  unsigned long long ids_packed;  // up to 8 bc ids, one per byte
  unsigned kinds_packed;          // their kinds, one per nibble
  int n_bc;
  size_t plane_stride;  // elements
  int nx, ny, nz;       // interior extent of this rank
  int halo;             // ghost planes per side (0: periodic wrap in x done here)
  int x_begin;          // first interior x plane updated by this launch
  int x_count;          // two-step kernel only: planes updated by this launch (cut into x_segments)
  int nzq;              // nz / VEC
  CollideExtra extra;   // force vector / Smagorinsky constant (only read by the variants that use them)
  int xcd_swizzle;      // remap blockIdx so that the blocks of one row share an XCD (needs gridDim.y % 8 == 0)
  T omega;
};

template <class S, int N>
struct VecOf {
  typedef S aligned __attribute__((ext_vector_type(N), aligned(sizeof(S) * N)));
  typedef S shifted __attribute__((ext_vector_type(N), aligned(sizeof(S))));
};

// p = uniform base, boff = per-thread byte offset (32 bit) -> global_load ... v_off, s[base]
template <class S>
__device__ __forceinline__ S ld(const S* base, unsigned boff) {
  return *reinterpret_cast<const S*>(reinterpret_cast<const char*>(base) + boff);
}
template <class S, int VEC, bool NT = false>
__device__ __forceinline__ void ld_aligned(const S* base, unsigned boff, S (&out)[VEC]) {
  if constexpr (VEC == 1) {
    if constexpr (NT)
      out[0] = __builtin_nontemporal_load(reinterpret_cast<const S*>(reinterpret_cast<const char*>(base) + boff));
    else
      out[0] = ld(base, boff);
  } else {
    typedef typename VecOf<S, VEC>::aligned V;
    V v;
    if constexpr (NT)
      v = __builtin_nontemporal_load(reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + boff));
    else
      v = *reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + boff);
#pragma unroll
    for (int k = 0; k < VEC; ++k) out[k] = v[k];
  }
}
template <class S, int VEC>
__device__ __forceinline__ void ld_shifted(const S* base, unsigned boff, S (&out)[VEC]) {
  typedef typename VecOf<S, VEC>::shifted V;
  const V v = *reinterpret_cast<const V*>(reinterpret_cast<const char*>(base) + boff);
#pragma unroll
  for (int k = 0; k < VEC; ++k) out[k] = v[k];
}
template <class S, int VEC, bool NT>
__device__ __forceinline__ void st_aligned(S* base, unsigned boff, const S (&in)[VEC]) {
  char* p = reinterpret_cast<char*>(base) + boff;
  if constexpr (VEC == 1) {
    if constexpr (NT)
      __builtin_nontemporal_store(in[0], reinterpret_cast<S*>(p));
    else
      *reinterpret_cast<S*>(p) = in[0];
  } else {
    typedef typename VecOf<S, VEC>::aligned V;
    V v;
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = in[k];
    if constexpr (NT)
      __builtin_nontemporal_store(v, reinterpret_cast<V*>(p));
    else
      *reinterpret_cast<V*>(p) = v;
  }
}

// Identity the optimiser cannot see through.  The BC branches below issue loads that are
// textually identical across branches (own-cell populations, per-BC constants); LLVM would hoist
// them above the branch and keep ~40 extra VGPRs live in the fluid fast path.
__device__ __forceinline__ unsigned opaque(unsigned v) {
  asm volatile("" : "+v"(v));
  return v;
}
template <class P>
__device__ __forceinline__ const P* opaque(const P* p) {
  asm volatile("" : "+v"(p));
  return p;
}

// scalar compare chain only (callers guarantee n_bc <= MAX_FAST_BCS): contains NO load, so it can sit in a
// software-pipelined loop without making the compiler's vmcnt bookkeeping fall back to vmcnt(0)
template <class T, class S>
__device__ __forceinline__ unsigned kind_fast(const StepArgs<T, S>& a, unsigned id) {
  unsigned kind = K_NONE;
#pragma unroll
  for (int s = 0; s < MAX_FAST_BCS; ++s) {
    const unsigned sid = (unsigned)(a.ids_packed >> (8 * s)) & 0xffu;  // scalar
    const unsigned sk = (a.kinds_packed >> (4 * s)) & 0xfu;            // scalar
    kind = (id == sid) ? sk : kind;
  }
  return kind;
}

template <class T, class S>
__device__ __forceinline__ unsigned kind_of(const StepArgs<T, S>& a, unsigned id) {
  if (a.n_bc > MAX_FAST_BCS) return a.bc_kind[id];
  unsigned kind = K_NONE;
#pragma unroll
  for (int s = 0; s < MAX_FAST_BCS; ++s) {
    const unsigned sid = (unsigned)(a.ids_packed >> (8 * s)) & 0xffu;  // scalar
    const unsigned sk = (a.kinds_packed >> (4 * s)) & 0xfu;            // scalar
    kind = (id == sid) ? sk : kind;
  }
  return kind;
}

// HASBC: 0 = no boundary conditions, 1 = the basic kinds (equilibrium, bounce-back, do-nothing),
// 2 = basic + Zou-He / Regularized (kept out of the basic variant to protect its register budget).
// FLAGS bit 0: non-temporal stores; bit 1: non-temporal loads of the c_z == 0 directions;
// bit 2: non-temporal loads of the z-shifted directions too (VEC == 1)
template <class L, class T, class S, int VEC, int COLL, int HASBC, int FLAGS>
__global__ void __launch_bounds__(VEC == 1 ? XLB_LB1 : 256) k_step(const StepArgs<T, S> a) {
  constexpr int Q = L::Q;
  constexpr unsigned ES = sizeof(S);
  // Blocks are dealt round-robin over the 8 XCDs (b and b + 8 share an L2).  With xcd_swizzle the
  // blockIdx.x that cover ONE row are spaced 8 apart in dispatch order, so the cache line that a
  // z-shifted load shares with the neighbouring block of the same row is served by the same L2.
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if (a.xcd_swizzle) {
    const unsigned gx = gridDim.x;                    // blocks per row
    const unsigned lin = blockIdx.y * gx + blockIdx.x;  // dispatch order inside the x-plane
    const unsigned grp = lin / (8u * gx), r = lin % (8u * gx);
    by = grp * 8u + (r % 8u);
    bx = r / 8u;
  }
  const int zq = bx * blockDim.x + threadIdx.x;
  const int y = by * blockDim.y + threadIdx.y;
  if (zq >= a.nzq || y >= a.ny) return;
  const int x = a.x_begin + blockIdx.z;  // block-uniform
  const int z0 = zq * VEC;
  const int ny = a.ny, nz = a.nz;

  // storage x-plane of the source for c_x = -1, 0, +1 (index c_x + 1); uniform
  int Xs[3];
  if (a.halo) {
    Xs[0] = x + a.halo + 1;
    Xs[1] = x + a.halo;
    Xs[2] = x + a.halo - 1;
  } else {
    Xs[0] = (x + 1 == a.nx) ? 0 : x + 1;
    Xs[1] = x;
    Xs[2] = (x == 0) ? a.nx - 1 : x - 1;
  }
  const size_t plane_cells = (size_t)ny * nz;
  // byte offsets inside an x-plane of the three y-neighbour rows (index c_y + 1)
  unsigned Yb[3];
  Yb[0] = (unsigned)((y + 1 == ny) ? 0 : y + 1) * (unsigned)nz * ES;
  Yb[1] = (unsigned)y * (unsigned)nz * ES;
  Yb[2] = (unsigned)((y == 0) ? ny - 1 : y - 1) * (unsigned)nz * ES;

  const bool at_z_lo = (z0 == 0);
  const bool at_z_hi = (z0 + VEC == nz);
  const unsigned zb = (unsigned)z0 * ES;

  T f[VEC][Q];

  // ---- pull streaming: f[k][l] = src[l, x - cx, y - cy, z0 + k - cz]  (stream.py:57-62) ----
  static_for<Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    constexpr int cx = L::c(0, l), cy = L::c(1, l), cz = L::c(2, l);
    const S* row = a.src + (size_t)l * a.plane_stride + (size_t)Xs[cx + 1] * plane_cells;  // uniform
    const unsigned yb = Yb[cy + 1];
    S v[VEC];
    if constexpr (cz == 0) {
      ld_aligned<S, VEC, (FLAGS & 2) != 0>(row, yb + zb, v);
    } else if constexpr (VEC == 1) {
      int zs = z0 - cz;
      zs = zs < 0 ? nz - 1 : (zs == nz ? 0 : zs);
      if constexpr ((FLAGS & 4) != 0)
        v[0] = __builtin_nontemporal_load(reinterpret_cast<const S*>(reinterpret_cast<const char*>(row) + (yb + (unsigned)zs * ES)));
      else
        v[0] = ld(row, yb + (unsigned)zs * ES);
    } else if constexpr (cz == 1) {
      // need z0-1 .. z0+VEC-2 ; at the row start z0-1 wraps to nz-1
      S t[VEC];
      ld_shifted<S, VEC>(row, yb + zb - (at_z_lo ? 0u : ES), t);
      S wrapv = t[0];
      if (at_z_lo) wrapv = ld(row, yb + (unsigned)(nz - 1) * ES);
      v[0] = at_z_lo ? wrapv : t[0];
#pragma unroll
      for (int k = 1; k < VEC; ++k) v[k] = at_z_lo ? t[k - 1] : t[k];
    } else {
      // cz == -1: need z0+1 .. z0+VEC ; at the row end z0+VEC wraps to 0
      S t[VEC];
      ld_shifted<S, VEC>(row, yb + zb + (at_z_hi ? 0u : ES), t);
      S wrapv = t[VEC - 1];
      if (at_z_hi) wrapv = ld(row, yb);
#pragma unroll
      for (int k = 0; k < VEC - 1; ++k) v[k] = at_z_hi ? t[k + 1] : t[k];
      v[VEC - 1] = at_z_hi ? wrapv : t[VEC - 1];
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) f[k][l] = to_compute<T, S>(v[k]);
  });

  // ---- boundary ids of my VEC cells ----
  const unsigned cell_in_plane = (unsigned)y * (unsigned)nz + (unsigned)z0;  // elements, < 2^30
  unsigned ids[VEC];
  bool any_bc = false;
  if constexpr (HASBC != 0) {
    const uint8_t* bcp = a.bc + (size_t)Xs[1] * plane_cells;  // uniform
    if constexpr (VEC == 4) {
      const unsigned wrd = ld(reinterpret_cast<const unsigned*>(bcp), cell_in_plane);
      any_bc = wrd != 0u;
#pragma unroll
      for (int k = 0; k < 4; ++k) ids[k] = (wrd >> (8 * k)) & 0xffu;
    } else if constexpr (VEC == 2) {
      const unsigned wrd = ld(reinterpret_cast<const unsigned short*>(bcp), cell_in_plane);
      any_bc = wrd != 0u;
      ids[0] = wrd & 0xffu;
      ids[1] = (wrd >> 8) & 0xffu;
    } else {
      ids[0] = ld(bcp, cell_in_plane);
      any_bc = ids[0] != 0u;
    }
  }

#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    bool fullway = false;
    if constexpr (HASBC != 0) {
      if (any_bc && ids[k] != 0u) {
        const unsigned id = ids[k];
        const unsigned kind = kind_of(a, id);
        const unsigned cb = (cell_in_plane + (unsigned)k) * ES;  // own cell, byte offset in the x-plane
        if (kind == K_EQ) {
          // bc_equilibrium.py:75-80: f = feq(rho0, u0)
          const T* val = opaque(a.bc_values + id * 27u);
          static_for<Q>([&](auto lc) {
            constexpr int l = decltype(lc)::value;
            f[k][l] = val[l];
          });
        } else if (kind == K_HW) {
          // bc_halfway_bounce_back.py:124-132: missing & boundary -> f_pre[opp] + moving-wall term
          // (the term is 0.0 for a no-slip wall; the reference adds it in that case too).
          // Rare path: processed in groups of 4 directions so that the temporaries do not
          // raise the register allocation of the whole kernel.
          const unsigned m = ld(a.miss + (size_t)Xs[1] * plane_cells, (cell_in_plane + (unsigned)k) * 4u);
          const T* val = opaque(a.bc_values + id * 27u);
          const unsigned cbh = opaque(cb);
          static_for<(Q + 3) / 4>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            __builtin_amdgcn_sched_barrier(0);
            static_for<4>([&](auto jc) {
              constexpr int l = g * 4 + decltype(jc)::value;
              if constexpr (l < Q) {
                const S* own = a.src + (size_t)opp<L>(l) * a.plane_stride + (size_t)Xs[1] * plane_cells;  // uniform
                const T pre = to_compute<T, S>(ld(own, cbh)) + val[l];
                if ((m >> l) & 1u) f[k][l] = pre;
              }
            });
          });
          __builtin_amdgcn_sched_barrier(0);
        } else if (kind == K_DN) {
          // bc_do_nothing.py:50-54
          const unsigned cbd = opaque(cb);
          static_for<Q>([&](auto lc) {
            constexpr int l = decltype(lc)::value;
            const S* own = a.src + (size_t)l * a.plane_stride + (size_t)Xs[1] * plane_cells;
            f[k][l] = to_compute<T, S>(ld(own, cbd));
          });
        } else if (kind == K_FW) {
          fullway = true;
        } else if constexpr (HASBC == 2) {
          // extended kernel variant only: Zou-He / Regularized inlets and outlets (bc_zouhe.py, bc_regularized.py)
          if (kind == XLBHIP_BC_EXTRAPOLATION_OUTFLOW) {
            // bc_extrapolation_outflow.py:137-145: missing & boundary -> f_pre[opp] (last step's auxiliary data; the
            // new ones are assembled by k_outflow_aux after this kernel)
            const unsigned m = ld(a.miss + (size_t)Xs[1] * plane_cells, opaque((cell_in_plane + (unsigned)k) * 4u));
            const unsigned cbo = opaque(cb);
            static_for<Q>([&](auto lc) {
              constexpr int l = decltype(lc)::value;
              const S* own = a.src + (size_t)opp<L>(l) * a.plane_stride + (size_t)Xs[1] * plane_cells;  // uniform
              if ((m >> l) & 1u) f[k][l] = to_compute<T, S>(ld(own, cbo));
            });
          } else if (kind >= XLBHIP_BC_ZOUHE_VELOCITY && kind <= XLBHIP_BC_REGULARIZED_PRESSURE) {
            const unsigned m = ld(a.miss + (size_t)Xs[1] * plane_cells, opaque((cell_in_plane + (unsigned)k) * 4u));
            const T* val = opaque(a.bc_values + id * 27u);
            if (val[PROF_FLAG] != T(0) && a.n_prof > 0)  // per-cell values (bc_zouhe.py:225-232: the broadcast profile)
              val = a.prof_vals + 3 * (size_t)prof_find(a.prof_keys, a.n_prof, (unsigned)Xs[1] * (unsigned)plane_cells + cell_in_plane + (unsigned)k);
            zouhe_cell<L, T>(f[k], m, val, kind == XLBHIP_BC_ZOUHE_VELOCITY || kind == XLBHIP_BC_REGULARIZED_VELOCITY,
                             kind >= XLBHIP_BC_REGULARIZED_VELOCITY);
          } else if (kind == XLBHIP_BC_HALFWAY_BB_PROFILE) {
            // halfway bounce-back with this cell's wall velocity from the profile table: missing & boundary ->
            // f_pre[opp] + 6 w_l (c_l . u_wall), the components summed in order (helper_functions_bc.py:230-250)
            const unsigned m = ld(a.miss + (size_t)Xs[1] * plane_cells, opaque((cell_in_plane + (unsigned)k) * 4u));
            const unsigned cbo = opaque(cb);
            T uw[3] = {T(0), T(0), T(0)};
            if (a.n_prof > 0) {
              const T* pv = a.prof_vals + 3 * (size_t)prof_find(a.prof_keys, a.n_prof, (unsigned)Xs[1] * (unsigned)plane_cells + cell_in_plane + (unsigned)k);
              uw[0] = pv[0];
              uw[1] = pv[1];
              uw[2] = pv[2];
            }
            static_for<Q>([&](auto lc) {
              constexpr int l = decltype(lc)::value;
              const S* own = a.src + (size_t)opp<L>(l) * a.plane_stride + (size_t)Xs[1] * plane_cells;  // uniform
              T cu = T(0.0);
              static_for<3>([&](auto ac) {
                constexpr int ax = decltype(ac)::value;
                if constexpr (L::c(ax, l) == 1) cu = cu + uw[ax];
                if constexpr (L::c(ax, l) == -1) cu = cu - uw[ax];
              });
              if ((m >> l) & 1u) f[k][l] = to_compute<T, S>(ld(own, cbo)) + cu * (T(6.0) * T(L::w(l)));
            });
          } else if (kind >= XLBHIP_BC_HYBRID_BB_REGULARIZED && kind <= XLBHIP_BC_HYBRID_NEQ_REGULARIZED) {
            if constexpr (L::D == 3) {
              // bc_hybrid.py:254-358: own pre-streaming populations, missing bits, wall velocity and (optionally) the
              // cell's wall-distance weights from the stepper's sparse table
              const unsigned m = ld(a.miss + (size_t)Xs[1] * plane_cells, opaque((cell_in_plane + (unsigned)k) * 4u));
              const T* val = opaque(a.bc_values + id * 27u);
              const unsigned cbo = opaque(cb);
              T pre[Q];
              static_for<Q>([&](auto lc) {
                constexpr int l = decltype(lc)::value;
                const S* own = a.src + (size_t)l * a.plane_stride + (size_t)Xs[1] * plane_cells;  // uniform
                pre[l] = to_compute<T, S>(ld(own, cbo));
              });
              const float* wgt = nullptr;
              const unsigned key = (unsigned)Xs[1] * (unsigned)plane_cells + cell_in_plane + (unsigned)k;
              if (val[4] != T(0) && a.n_dist > 0) {
                const int slot = prof_find(a.dist_keys, a.n_dist, key);
                if (a.dist_keys[slot] == key) wgt = a.dist_vals + (size_t)slot * Q;
              }
              // wall velocity: the BC's constant, or this cell's entry of the profile table (bc_hybrid.py:265: profile(index))
              T uw[5] = {val[0], val[1], val[2], val[3], val[4]};
              if (val[PROF_FLAG] != T(0) && a.n_prof > 0) {
                const T* pv = a.prof_vals + 3 * (size_t)prof_find(a.prof_keys, a.n_prof, key);
                uw[0] = pv[0];
                uw[1] = pv[1];
                uw[2] = pv[2];
              }
              hybrid_cell<L, T>(f[k], pre, m, wgt, uw, (int)kind - XLBHIP_BC_HYBRID_BB_REGULARIZED);
            }
          }
        }
      }
    }
    if (!fullway) {
      collide<L, T, COLL>(f[k], a.omega, a.extra);
    } else {
      // bc_fullway_bounce_back.py:52-56: f_post_collision[l] = f_post_stream[opp l]
      static_for<Q>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        constexpr int o = opp<L>(l);
        if constexpr (l < o) {
          const T t = f[k][l];
          f[k][l] = f[k][o];
          f[k][o] = t;
        }
      });
    }
  }

  // ---- store (cast to store precision, nse_stepper.py:280) ----
  static_for<Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    S* drow = a.dst + (size_t)l * a.plane_stride + (size_t)Xs[1] * plane_cells;  // uniform
    S v[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = to_store<S, T>(f[k][l]);
    st_aligned<S, VEC, (FLAGS & 1) != 0>(drow, Yb[1] + zb, v);
  });
}

}  // namespace xlb
