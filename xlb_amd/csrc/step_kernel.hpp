// The fused per-timestep kernel: pull-stream -> post-streaming BCs -> rho,u -> feq ->
// collide -> post-collision BCs -> store.  One launch per step.
//
// Replaces the reference's fused Warp kernel (xlb/operator/stepper/nse_stepper.py:427-464)
// with the step ORDER and semantics of the JAX branch (:237-282), which is the parity
// target.  Design:
//   * SoA populations (q, x, y, z), z fastest; a thread owns VEC consecutive z cells
//     and keeps their q populations in registers (VEC*q VGPRs), so moments/feq/collide
//     need no cross-lane traffic at all.
//   * per direction ONE vector load per thread: 16-B aligned for c_z == 0, dword-aligned
//     (shifted by one element) for c_z = +-1; the single wrapped element at a row end is
//     patched by an exec-masked scalar load.  x is block-uniform (blockIdx.z), so the
//     plane/x part of every address is scalar (SGPR) arithmetic.
//   * boundary cells are handled in-register on a slow path taken only by threads whose
//     bc_mask word is non-zero; missing bits (one u32 per cell) and own-cell pre-stream
//     populations are fetched only there.
//   * stores are 16-B aligned vector stores, optionally non-temporal.
#pragma once
#include "cell.hpp"

namespace xlb {

template <class T>
struct BcTableDev {
  const uint8_t* kind;  // [256] XLBHIP_BC_* or 0
  const T* values;      // [256][27]
};

template <class T, class S>
struct StepArgs {
  const S* src;
  S* dst;
  const uint8_t* bc;     // (nx+2h, ny, nz) or nullptr
  const uint32_t* miss;  // (nx+2h, ny, nz) bit-sets or nullptr
  BcTableDev<T> tab;
  size_t plane_stride;  // elements
  int nx, ny, nz;       // interior extent of this rank
  int halo;             // ghost planes per side (0: periodic wrap in x done here)
  int x_begin;          // first interior x plane updated by this launch
  int nzq;              // nz / VEC
  T omega;
};

template <class S, int N>
struct VecOf {
  typedef S aligned __attribute__((ext_vector_type(N), aligned(sizeof(S) * N)));
  typedef S shifted __attribute__((ext_vector_type(N), aligned(sizeof(S))));
};
template <class S>
struct VecOf<S, 1> {
  typedef S aligned;
  typedef S shifted;
};

template <class S, int VEC>
__device__ __forceinline__ void load_aligned(const S* p, S (&out)[VEC]) {
  if constexpr (VEC == 1) {
    out[0] = *p;
  } else {
    typename VecOf<S, VEC>::aligned v = *reinterpret_cast<const typename VecOf<S, VEC>::aligned*>(p);
#pragma unroll
    for (int k = 0; k < VEC; ++k) out[k] = v[k];
  }
}
template <class S, int VEC>
__device__ __forceinline__ void load_shifted(const S* p, S (&out)[VEC]) {
  typename VecOf<S, VEC>::shifted v = *reinterpret_cast<const typename VecOf<S, VEC>::shifted*>(p);
#pragma unroll
  for (int k = 0; k < VEC; ++k) out[k] = v[k];
}
template <class S, int VEC, bool NT>
__device__ __forceinline__ void store_aligned(S* p, const S (&in)[VEC]) {
  if constexpr (VEC == 1) {
    if constexpr (NT)
      __builtin_nontemporal_store(in[0], p);
    else
      *p = in[0];
  } else {
    typename VecOf<S, VEC>::aligned v;
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = in[k];
    if constexpr (NT)
      __builtin_nontemporal_store(v, reinterpret_cast<typename VecOf<S, VEC>::aligned*>(p));
    else
      *reinterpret_cast<typename VecOf<S, VEC>::aligned*>(p) = v;
  }
}

// FLAGS bit 0: non-temporal stores
template <class L, class T, class S, int VEC, int COLL, bool HASBC, int FLAGS>
__global__ void __launch_bounds__(256) k_step(const StepArgs<T, S> a) {
  constexpr int Q = L::Q;
  const int zq = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  if (zq >= a.nzq || y >= a.ny) return;
  const int x = a.x_begin + blockIdx.z;  // block-uniform
  const int z0 = zq * VEC;
  const int ny = a.ny, nz = a.nz;

  // storage x-plane indices of the three x neighbours (uniform)
  int Xs[3];  // index by cx + 1 -> source plane for c_x = -1, 0, +1 is x - c_x
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
  // row offsets inside an x-plane for the three y neighbours (per thread, 32 bit)
  unsigned Yo[3];
  Yo[0] = (unsigned)((y + 1 == ny) ? 0 : y + 1) * (unsigned)nz;
  Yo[1] = (unsigned)y * (unsigned)nz;
  Yo[2] = (unsigned)((y == 0) ? ny - 1 : y - 1) * (unsigned)nz;

  const bool at_z_lo = (z0 == 0);
  const bool at_z_hi = (z0 + VEC == nz);

  T f[VEC][Q];

  // ---- pull streaming: f[k][l] = src[l, x - cx, y - cy, z0 + k - cz]  (stream.py:57-62) ----
  static_for<Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    constexpr int cx = L::c(0, l), cy = L::c(1, l), cz = L::c(2, l);
    const S* row = a.src + (size_t)l * a.plane_stride + (size_t)Xs[cx + 1] * plane_cells;  // uniform
    const unsigned yo = Yo[cy + 1];
    S v[VEC];
    if constexpr (cz == 0) {
      load_aligned<S, VEC>(row + (yo + (unsigned)z0), v);
    } else if constexpr (VEC == 1) {
      int zs = z0 - cz;
      zs = zs < 0 ? nz - 1 : (zs == nz ? 0 : zs);
      v[0] = row[yo + (unsigned)zs];
    } else if constexpr (cz == 1) {
      // need z0-1 .. z0+VEC-2 ; at the row start z0-1 wraps to nz-1
      const unsigned off = yo + (unsigned)z0 - (at_z_lo ? 0u : 1u);
      S t[VEC];
      load_shifted<S, VEC>(row + off, t);
      S wrapv = t[0];
      if (at_z_lo) wrapv = row[yo + (unsigned)(nz - 1)];
      v[0] = at_z_lo ? wrapv : t[0];
#pragma unroll
      for (int k = 1; k < VEC; ++k) v[k] = at_z_lo ? t[k - 1] : t[k];
    } else {
      // cz == -1: need z0+1 .. z0+VEC ; at the row end z0+VEC wraps to 0
      const unsigned off = yo + (unsigned)z0 + (at_z_hi ? 0u : 1u);
      S t[VEC];
      load_shifted<S, VEC>(row + off, t);
      S wrapv = t[VEC - 1];
      if (at_z_hi) wrapv = row[yo];
#pragma unroll
      for (int k = 0; k < VEC - 1; ++k) v[k] = at_z_hi ? t[k + 1] : t[k];
      v[VEC - 1] = at_z_hi ? wrapv : t[VEC - 1];
    }
#pragma unroll
    for (int k = 0; k < VEC; ++k) f[k][l] = to_compute<T, S>(v[k]);
  });

  // ---- boundary ids of my VEC cells ----
  const size_t cell0 = (size_t)Xs[1] * plane_cells + Yo[1] + (unsigned)z0;
  unsigned ids[VEC];
  bool any_bc = false;
  if constexpr (HASBC) {
    if constexpr (VEC == 4) {
      const unsigned wrd = *reinterpret_cast<const unsigned*>(a.bc + cell0);
      any_bc = wrd != 0u;
#pragma unroll
      for (int k = 0; k < 4; ++k) ids[k] = (wrd >> (8 * k)) & 0xffu;
    } else if constexpr (VEC == 2) {
      const unsigned wrd = *reinterpret_cast<const unsigned short*>(a.bc + cell0);
      any_bc = wrd != 0u;
      ids[0] = wrd & 0xffu;
      ids[1] = (wrd >> 8) & 0xffu;
    } else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        ids[k] = a.bc[cell0 + k];
        any_bc |= ids[k] != 0u;
      }
    }
  }

  const S* own = a.src + cell0;  // own-cell pre-stream populations: own[l * plane_stride + k]

#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    bool fullway = false;
    if constexpr (HASBC) {
      if (any_bc && ids[k] != 0u) {
        const unsigned id = ids[k];
        const unsigned kind = a.tab.kind[id];
        const T* val = a.tab.values + id * 27u;
        if (kind == XLBHIP_BC_EQUILIBRIUM) {
          // bc_equilibrium.py:75-80: f = feq(rho0, u0)
          static_for<Q>([&](auto lc) {
            constexpr int l = decltype(lc)::value;
            f[k][l] = val[l];
          });
        } else if (kind == XLBHIP_BC_HALFWAY_BB) {
          // bc_halfway_bounce_back.py:124-132: missing & boundary -> f_pre[opp] + moving wall term
          const unsigned m = a.miss[cell0 + k];
          static_for<Q>([&](auto lc) {
            constexpr int l = decltype(lc)::value;
            if ((m >> l) & 1u) f[k][l] = to_compute<T, S>(own[(size_t)opp<L>(l) * a.plane_stride + k]) + val[l];
          });
        } else if (kind == XLBHIP_BC_DO_NOTHING) {
          // bc_do_nothing.py:50-54
          static_for<Q>([&](auto lc) {
            constexpr int l = decltype(lc)::value;
            f[k][l] = to_compute<T, S>(own[(size_t)l * a.plane_stride + k]);
          });
        } else if (kind == XLBHIP_BC_FULLWAY_BB) {
          fullway = true;
        }
      }
    }
    if (!fullway) {
      collide<L, T, COLL>(f[k], a.omega);
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
  S* drow = a.dst + cell0;
  static_for<Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    S v[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = to_store<S, T>(f[k][l]);
    store_aligned<S, VEC, (FLAGS & 1) != 0>(drow + (size_t)l * a.plane_stride, v);
  });
}

}  // namespace xlb
