// Whole-field operator kernels (one thread per cell) and the boundary-masker kernels.
// These mirror the reference's stand-alone operator launches; the
// per-timestep hot path is the fused kernel in step_kernel.hpp.
#pragma once
#include "cell.hpp"

namespace xlb {

// Runtime-typed view of a field on the device.
struct FieldView {
  void* data;
  size_t plane_stride;
  int dtype;
  int halo;
};

struct Dims {
  int nx, ny, nz;
};

template <class T>
__device__ __forceinline__ T load_rt(const FieldView& v, size_t i) {
  switch (v.dtype) {
    case XLBHIP_F64: return static_cast<T>(static_cast<const double*>(v.data)[i]);
    case XLBHIP_F32: return static_cast<T>(static_cast<const float*>(v.data)[i]);
    case XLBHIP_F16: return static_cast<T>(static_cast<const _Float16*>(v.data)[i]);
    default: return static_cast<T>(static_cast<const uint8_t*>(v.data)[i]);
  }
}
template <class T>
__device__ __forceinline__ void store_rt(const FieldView& v, size_t i, T x) {
  switch (v.dtype) {
    case XLBHIP_F64: static_cast<double*>(v.data)[i] = static_cast<double>(x); break;
    case XLBHIP_F32: static_cast<float*>(v.data)[i] = static_cast<float>(x); break;
    case XLBHIP_F16: static_cast<_Float16*>(v.data)[i] = static_cast<_Float16>(x); break;
    default: static_cast<uint8_t*>(v.data)[i] = static_cast<uint8_t>(x); break;
  }
}

__device__ __forceinline__ bool cell_of_thread(const Dims& d, int& x, int& y, int& z) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)d.nx * d.ny * d.nz;
  if (t >= n) return false;
  z = (int)(t % d.nz);
  const size_t r = t / d.nz;
  y = (int)(r % d.ny);
  x = (int)(r / d.ny);
  return true;
}
__device__ __forceinline__ size_t cell_index(const FieldView& v, const Dims& d, int x, int y, int z) {
  return ((size_t)(x + v.halo) * d.ny + y) * d.nz + z;
}
__device__ __forceinline__ int wrap(int i, int n) { return i < 0 ? i + n : (i >= n ? i - n : i); }

// Stream()(f_0, f_1): stream.py:29-62 / :96-110 — periodic pull
template <class L>
__global__ void k_stream(FieldView src, FieldView dst, Dims d) {
  int x, y, z;
  if (!cell_of_thread(d, x, y, z)) return;
  const size_t o = cell_index(dst, d, x, y, z);
  static_for<L::Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    const int xs = src.halo ? x - L::c(0, l) : wrap(x - L::c(0, l), d.nx);
    const int ys = wrap(y - L::c(1, l), d.ny);
    const int zs = wrap(z - L::c(2, l), d.nz);
    const size_t i = (size_t)l * src.plane_stride + cell_index(src, d, xs, ys, zs);
    // pure data movement: go through double, exact for every storage type
    store_rt<double>(dst, (size_t)l * dst.plane_stride + o, load_rt<double>(src, i));
  });
}

// QuadraticEquilibrium()(rho, u, f): quadratic_equilibrium.py:23-30
template <class L, class T>
__global__ void k_equilibrium(FieldView rho, FieldView u, FieldView f, Dims d) {
  int x, y, z;
  if (!cell_of_thread(d, x, y, z)) return;
  const T r = load_rt<T>(rho, cell_index(rho, d, x, y, z));
  T uu[3] = {T(0), T(0), T(0)};
  static_for<L::D>([&](auto ac) {
    constexpr int a = decltype(ac)::value;
    uu[a + 3 - L::D] = load_rt<T>(u, (size_t)a * u.plane_stride + cell_index(u, d, x, y, z));
  });
  T feq[L::Q];
  equilibrium<L, T>(r, uu, feq);
  const size_t o = cell_index(f, d, x, y, z);
  static_for<L::Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    store_rt<T>(f, (size_t)l * f.plane_stride + o, feq[l]);
  });
}

// Macroscopic()(f, rho, u): macroscopic.py:21-26
template <class L, class T>
__global__ void k_macroscopic(FieldView f, FieldView rho, FieldView u, Dims d) {
  int x, y, z;
  if (!cell_of_thread(d, x, y, z)) return;
  T ff[L::Q];
  const size_t i = cell_index(f, d, x, y, z);
  static_for<L::Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    ff[l] = load_rt<T>(f, (size_t)l * f.plane_stride + i);
  });
  T r, uu[3];
  moments<L, T>(ff, r, uu);
  if (rho.data) store_rt<T>(rho, cell_index(rho, d, x, y, z), r);
  if (u.data) {
    static_for<L::D>([&](auto ac) {
      constexpr int a = decltype(ac)::value;
      store_rt<T>(u, (size_t)a * u.plane_stride + cell_index(u, d, x, y, z), uu[a + 3 - L::D]);
    });
  }
}

// SecondMoment()(f, pi): second_moment.py:55
template <class L, class T>
__global__ void k_second_moment(FieldView f, FieldView pi, Dims d) {
  int x, y, z;
  if (!cell_of_thread(d, x, y, z)) return;
  T ff[L::Q];
  const size_t i = cell_index(f, d, x, y, z);
  static_for<L::Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    ff[l] = load_rt<T>(f, (size_t)l * f.plane_stride + i);
  });
  T p[6];
  second_moment<L, T>(ff, p);
  static_for<n_pi<L>()>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    store_rt<T>(pi, (size_t)k * pi.plane_stride + cell_index(pi, d, x, y, z), p[k]);
  });
}

// BGK()/KBC()(f, feq, fout, omega): bgk.py:27-32, kbc.py:40-79
template <class L, class T, int COLL>
__global__ void k_collide(FieldView f, FieldView feq, FieldView fout, Dims d, T omega, T smag_cs) {
  int x, y, z;
  if (!cell_of_thread(d, x, y, z)) return;
  T ff[L::Q], fe[L::Q];
  const size_t i = cell_index(f, d, x, y, z), j = cell_index(feq, d, x, y, z);
  static_for<L::Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    ff[l] = load_rt<T>(f, (size_t)l * f.plane_stride + i);
    fe[l] = load_rt<T>(feq, (size_t)l * feq.plane_stride + j);
  });
  if constexpr (COLL == XLBHIP_BGK)
    bgk<L, T>(ff, fe, omega);
  else if constexpr (COLL == XLBHIP_KBC)
    kbc<L, T>(ff, fe, omega);
  else
    smagorinsky<L, T>(ff, fe, omega, smag_cs);
  const size_t o = cell_index(fout, d, x, y, z);
  static_for<L::Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    store_rt<T>(fout, (size_t)l * fout.plane_stride + o, ff[l]);
  });
}

struct BcValues {
  double v[27];
};

// bc(f_pre, f_post, bc_mask, missing_mask) -> f_post: boundary_condition.py:146-180 with the
// JAX bodies of bc_equilibrium.py:75-80, bc_halfway_bounce_back.py:124-132,
// bc_fullway_bounce_back.py:52-56, bc_do_nothing.py:50-54
template <class L, class T>
__global__ void k_apply_bc(int id, int kind, BcValues vals, FieldView f_pre, FieldView f_post, FieldView bc, FieldView miss,
                           Dims d, const uint32_t* prof_keys, const double* prof_vals, int n_prof) {
  int x, y, z;
  if (!cell_of_thread(d, x, y, z)) return;
  const uint8_t b = static_cast<const uint8_t*>(bc.data)[cell_index(bc, d, x, y, z)];
  if (b != id) return;
  const size_t ip = cell_index(f_pre, d, x, y, z), io = cell_index(f_post, d, x, y, z);
  unsigned m = 0;
  if (kind == XLBHIP_BC_HALFWAY_BB || kind >= XLBHIP_BC_ZOUHE_VELOCITY) m = static_cast<const uint32_t*>(miss.data)[cell_index(miss, d, x, y, z)];
  if (kind >= XLBHIP_BC_ZOUHE_VELOCITY && kind <= XLBHIP_BC_REGULARIZED_PRESSURE) {
    T ff[L::Q], val[3];
    static_for<L::Q>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      ff[l] = load_rt<T>(f_post, (size_t)l * f_post.plane_stride + io);
    });
    const double* pv = vals.v;
    if (n_prof > 0) pv = prof_vals + 3 * (size_t)prof_find(prof_keys, n_prof, (uint32_t)cell_index(bc, d, x, y, z));
    val[0] = static_cast<T>(pv[0]);
    val[1] = static_cast<T>(pv[1]);
    val[2] = static_cast<T>(pv[2]);
    zouhe_cell<L, T>(ff, m, val, kind == XLBHIP_BC_ZOUHE_VELOCITY || kind == XLBHIP_BC_REGULARIZED_VELOCITY,
                     kind >= XLBHIP_BC_REGULARIZED_VELOCITY);
    static_for<L::Q>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      store_rt<T>(f_post, (size_t)l * f_post.plane_stride + io, ff[l]);
    });
    return;
  }
  if (kind >= XLBHIP_BC_HYBRID_BB_REGULARIZED && kind <= XLBHIP_BC_HYBRID_NEQ_REGULARIZED) {
    if constexpr (L::D == 3) {
      T ff[L::Q], pre[L::Q], val[5];
      static_for<L::Q>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        ff[l] = load_rt<T>(f_post, (size_t)l * f_post.plane_stride + io);
        pre[l] = load_rt<T>(f_pre, (size_t)l * f_pre.plane_stride + ip);
      });
      for (int a = 0; a < 5; ++a) val[a] = static_cast<T>(vals.v[a]);
      hybrid_cell<L, T>(ff, pre, m, nullptr, val, kind - XLBHIP_BC_HYBRID_BB_REGULARIZED);
      static_for<L::Q>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        store_rt<T>(f_post, (size_t)l * f_post.plane_stride + io, ff[l]);
      });
    }
    return;
  }
  static_for<L::Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    constexpr int o = opp<L>(l);
    if (kind == XLBHIP_BC_EQUILIBRIUM) {
      store_rt<T>(f_post, (size_t)l * f_post.plane_stride + io, static_cast<T>(vals.v[l]));
    } else if (kind == XLBHIP_BC_HALFWAY_BB) {
      if ((m >> l) & 1u)
        store_rt<T>(f_post, (size_t)l * f_post.plane_stride + io,
                    load_rt<T>(f_pre, (size_t)o * f_pre.plane_stride + ip) + static_cast<T>(vals.v[l]));
    } else if (kind == XLBHIP_BC_EXTRAPOLATION_OUTFLOW) {
      // bc_extrapolation_outflow.py:137-145 (streaming step; the extrapolation itself is k_outflow_aux)
      if ((m >> l) & 1u) store_rt<T>(f_post, (size_t)l * f_post.plane_stride + io, load_rt<T>(f_pre, (size_t)o * f_pre.plane_stride + ip));
    } else if (kind == XLBHIP_BC_FULLWAY_BB) {
      store_rt<T>(f_post, (size_t)l * f_post.plane_stride + io, load_rt<T>(f_pre, (size_t)o * f_pre.plane_stride + ip));
    } else if (kind == XLBHIP_BC_DO_NOTHING) {
      store_rt<T>(f_post, (size_t)l * f_post.plane_stride + io, load_rt<T>(f_pre, (size_t)l * f_pre.plane_stride + ip));
    }
  });
}

// ---- post-processing: Vorticity and QCriterion (postprocess/vorticity.py:30-84, q_criterion.py:36-131) ----
// The reference has these for its kernel backend only: one thread per cell of the box shrunk by one cell per side,
// nothing is written where any of the six face neighbours carries a boundary id, central differences elsewhere.
// MODE 0: out_a = vorticity (3 components), out_b = |vorticity|;  MODE 1: out_a = |vorticity|, out_b = Q.
template <class T, int MODE>
__global__ void k_velocity_gradient(FieldView u, FieldView bc, FieldView out_a, FieldView out_b, Dims d) {
  int x, y, z;
  if (!cell_of_thread(d, x, y, z)) return;
  if (x < 1 || y < 1 || z < 1 || x > d.nx - 2 || y > d.ny - 2 || z > d.nz - 2) return;
  const uint8_t* b = static_cast<const uint8_t*>(bc.data);
  if (b[cell_index(bc, d, x + 1, y, z)] != 0 || b[cell_index(bc, d, x, y + 1, z)] != 0 || b[cell_index(bc, d, x, y, z + 1)] != 0 ||
      b[cell_index(bc, d, x - 1, y, z)] != 0 || b[cell_index(bc, d, x, y - 1, z)] != 0 || b[cell_index(bc, d, x, y, z - 1)] != 0)
    return;
  auto U = [&](int a, int i, int j, int k) { return load_rt<T>(u, (size_t)a * u.plane_stride + cell_index(u, d, i, j, k)); };
  const T two = T(2.0), half = T(0.5);
  const T u_x_dy = (U(0, x, y + 1, z) - U(0, x, y - 1, z)) / two;
  const T u_x_dz = (U(0, x, y, z + 1) - U(0, x, y, z - 1)) / two;
  const T u_y_dx = (U(1, x + 1, y, z) - U(1, x - 1, y, z)) / two;
  const T u_y_dz = (U(1, x, y, z + 1) - U(1, x, y, z - 1)) / two;
  const T u_z_dx = (U(2, x + 1, y, z) - U(2, x - 1, y, z)) / two;
  const T u_z_dy = (U(2, x, y + 1, z) - U(2, x, y - 1, z)) / two;
  const T vx = u_z_dy - u_y_dz, vy = u_x_dz - u_z_dx, vz = u_y_dx - u_x_dy;
  const T mag = sqrt((vx * vx + vy * vy) + vz * vz);
  const size_t oa = cell_index(out_a, d, x, y, z), ob = cell_index(out_b, d, x, y, z);
  if constexpr (MODE == 0) {
    store_rt<T>(out_a, oa, vx);
    store_rt<T>(out_a, out_a.plane_stride + oa, vy);
    store_rt<T>(out_a, 2 * out_a.plane_stride + oa, vz);
    store_rt<T>(out_b, ob, mag);
  } else {
    const T u_x_dx = (U(0, x + 1, y, z) - U(0, x - 1, y, z)) / two;
    const T u_y_dy = (U(1, x, y + 1, z) - U(1, x, y - 1, z)) / two;
    const T u_z_dz = (U(2, x, y, z + 1) - U(2, x, y, z - 1)) / two;
    const T s01 = half * (u_x_dy + u_y_dx), s02 = half * (u_x_dz + u_z_dx), s12 = half * (u_y_dz + u_z_dy);
    // s_dot_s: the nine squares summed in row-major order, as the reference writes them
    T ss = u_x_dx * u_x_dx;
    ss = ss + s01 * s01;
    ss = ss + s02 * s02;
    ss = ss + s01 * s01;
    ss = ss + u_y_dy * u_y_dy;
    ss = ss + s12 * s12;
    ss = ss + s02 * s02;
    ss = ss + s12 * s12;
    ss = ss + u_z_dz * u_z_dz;
    const T o01 = half * (u_x_dy - u_y_dx), o02 = half * (u_x_dz - u_z_dx), o12 = half * (u_y_dz - u_z_dy);
    const T o10 = -o01, o20 = -o02, o21 = -o12;
    // omega_dot_omega: same order; the three 0.0**2 terms add exact zeros to non-negative partial sums
    T oo = T(0.0) + o01 * o01;
    oo = oo + o02 * o02;
    oo = oo + o10 * o10;
    oo = oo + T(0.0);
    oo = oo + o12 * o12;
    oo = oo + o20 * o20;
    oo = oo + o21 * o21;
    oo = oo + T(0.0);
    store_rt<T>(out_a, oa, mag);
    store_rt<T>(out_b, ob, half * (oo - ss));
  }
}

// any Zou-He / Regularized / outflow / do-nothing cell strictly inside the x range?  (two-step kernel with inlet / outlet planes)
__global__ void k_ext_interior_scan(FieldView bc, const uint8_t* kind_tab, Dims d, int* flag) {
  int x, y, z;
  if (!cell_of_thread(d, x, y, z)) return;
  if (x < 1 || x > d.nx - 2) return;
  const unsigned id = static_cast<const uint8_t*>(bc.data)[cell_index(bc, d, x, y, z)];
  if (id != 0u && (kind_tab[id] >= XLBHIP_BC_ZOUHE_VELOCITY || kind_tab[id] == XLBHIP_BC_DO_NOTHING)) *flag = 1;
}

// ---- MomentumTransfer (force/momentum_transfer.py:167-205, JAX): force on the solid behind a no-slip BC ----------
// f0 holds post-collision populations.  At every cell of the BC that is not itself solid (rest direction not
// missing), for every missing direction l:  phi_l = f0[opp l] + f_post_stream[l],  f_post_stream[l] = the BC's
// bounce-back of the own cell (halfway: f0[opp l] + wall term; fullway: f0[opp l]);  F = sum c_{opp l} phi_l.
// One double-precision atomic per block and component (the grid sum's order is unpinned in the reference as well).
template <class L, class T>
__global__ void k_momentum_transfer(FieldView f0, FieldView bc, FieldView miss, Dims d, int id, BcValues vals, int add_wall_term,
                                    double* force /*[3]*/) {
  __shared__ double red[3][256];
  int x, y, z;
  T fx = T(0), fy = T(0), fz = T(0);
  if (cell_of_thread(d, x, y, z)) {
    const uint8_t b = static_cast<const uint8_t*>(bc.data)[cell_index(bc, d, x, y, z)];
    if (b == id) {
      const unsigned m = static_cast<const uint32_t*>(miss.data)[cell_index(miss, d, x, y, z)];
      if ((m & 1u) == 0u) {  // is_edge: boundary & ~missing_mask[0]
        const size_t own = cell_index(f0, d, x, y, z);
        static_for<L::Q>([&](auto lc) {
          constexpr int l = decltype(lc)::value;
          constexpr int o = opp<L>(l);
          if ((m >> l) & 1u) {
            const T fo = load_rt<T>(f0, (size_t)o * f0.plane_stride + own);
            T ps = fo;
            if (add_wall_term) ps = fo + static_cast<T>(vals.v[l]);
            const T phi = fo + ps;
            // force += c[:, opp l] * phi
            if constexpr (L::c(0, o) == 1) fx = fx + phi;
            if constexpr (L::c(0, o) == -1) fx = fx - phi;
            if constexpr (L::c(1, o) == 1) fy = fy + phi;
            if constexpr (L::c(1, o) == -1) fy = fy - phi;
            if constexpr (L::c(2, o) == 1) fz = fz + phi;
            if constexpr (L::c(2, o) == -1) fz = fz - phi;
          }
        });
      }
    }
  }
  const int t = threadIdx.x;
  red[0][t] = (double)fx;
  red[1][t] = (double)fy;
  red[2][t] = (double)fz;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) {
      red[0][t] += red[0][t + s];
      red[1][t] += red[1][t + s];
      red[2][t] += red[2][t + s];
    }
    __syncthreads();
  }
  if (t < 3 && red[t][0] != 0.0) atomicAdd(force + t, red[t][0]);
}

// MomentumTransfer for the BCs whose post-stream populations need the stepper's tables (HybridBC: wall distances, wall-velocity
// profile; halfway bounce-back with a profile) — the kernel backends' path of the reference, force/momentum_transfer.py:225-262 with
// FetchPopulations (:75-92): f_post_stream = the BC applied to (own populations of f_0, populations pulled from f_0).
template <class L, class T>
__global__ void k_momentum_transfer_tab(FieldView f0, FieldView bc, FieldView miss, Dims d, int id, const uint8_t* kind_tab, const T* val_tab,
                                        const uint32_t* prof_keys, const T* prof_vals, int n_prof, const uint32_t* dist_keys,
                                        const float* dist_vals, int n_dist, double* force /*[3]*/) {
  __shared__ double red[3][256];
  constexpr int Q = L::Q;
  int x, y, z;
  T fx = T(0), fy = T(0), fz = T(0);
  if (cell_of_thread(d, x, y, z)) {
    const uint8_t b = static_cast<const uint8_t*>(bc.data)[cell_index(bc, d, x, y, z)];
    if (b == id) {
      const unsigned m = static_cast<const uint32_t*>(miss.data)[cell_index(miss, d, x, y, z)];
      if ((m & 1u) == 0u) {
        const size_t own = cell_index(f0, d, x, y, z);
        const unsigned key = (unsigned)cell_index(bc, d, x, y, z);
        const unsigned kind = kind_tab[id];
        const T* val = val_tab + (unsigned)id * 27u;
        T pre[Q], ps[Q];
        static_for<Q>([&](auto lc) {
          constexpr int l = decltype(lc)::value;
          pre[l] = load_rt<T>(f0, (size_t)l * f0.plane_stride + own);
          const int xs = f0.halo ? x - L::c(0, l) : wrap(x - L::c(0, l), d.nx);
          ps[l] = load_rt<T>(f0, (size_t)l * f0.plane_stride + cell_index(f0, d, xs, wrap(y - L::c(1, l), d.ny), wrap(z - L::c(2, l), d.nz)));
        });
        T uw[5] = {val[0], val[1], val[2], val[3], val[4]};
        if ((kind == XLBHIP_BC_HALFWAY_BB_PROFILE || val[PROF_FLAG] != T(0)) && n_prof > 0) {
          const T* pv = prof_vals + 3 * (size_t)prof_find(prof_keys, n_prof, key);
          uw[0] = pv[0];
          uw[1] = pv[1];
          uw[2] = pv[2];
        }
        if (kind == XLBHIP_BC_HALFWAY_BB_PROFILE) {
          static_for<Q>([&](auto lc) {
            constexpr int l = decltype(lc)::value;
            T cu = T(0.0);
            static_for<3>([&](auto ac) {
              constexpr int ax = decltype(ac)::value;
              if constexpr (L::c(ax, l) == 1) cu = cu + uw[ax];
              if constexpr (L::c(ax, l) == -1) cu = cu - uw[ax];
            });
            if ((m >> l) & 1u) ps[l] = pre[opp<L>(l)] + cu * (T(6.0) * T(L::w(l)));
          });
        } else {
          if constexpr (L::D == 3) {
            const float* wgt = nullptr;
            if (val[4] != T(0) && n_dist > 0) {
              const int slot = prof_find(dist_keys, n_dist, key);
              if (dist_keys[slot] == key) wgt = dist_vals + (size_t)slot * Q;
            }
            hybrid_cell<L, T>(ps, pre, m, wgt, uw, (int)kind - XLBHIP_BC_HYBRID_BB_REGULARIZED);
          }
        }
        static_for<Q>([&](auto lc) {
          constexpr int l = decltype(lc)::value;
          constexpr int o = opp<L>(l);
          if ((m >> l) & 1u) {
            const T phi = pre[o] + ps[l];
            if constexpr (L::c(0, o) == 1) fx = fx + phi;
            if constexpr (L::c(0, o) == -1) fx = fx - phi;
            if constexpr (L::c(1, o) == 1) fy = fy + phi;
            if constexpr (L::c(1, o) == -1) fy = fy - phi;
            if constexpr (L::c(2, o) == 1) fz = fz + phi;
            if constexpr (L::c(2, o) == -1) fz = fz - phi;
          }
        });
      }
    }
  }
  const int t = threadIdx.x;
  red[0][t] = (double)fx;
  red[1][t] = (double)fy;
  red[2][t] = (double)fz;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (t < s) {
      red[0][t] += red[0][t + s];
      red[1][t] += red[1][t + s];
      red[2][t] += red[2][t + s];
    }
    __syncthreads();
  }
  if (t < 3 && red[t][0] != 0.0) atomicAdd(force + t, red[t][0]);
}

// ---- GridToPoint (postprocess/grid_to_point.py:28-94): trilinear interpolation of component 0 at arbitrary points ----
// weights in fp32 from the fp32 point coordinates, as the reference computes them; products and the 8-term sum in its order
template <class T>
__global__ void k_grid_to_point(FieldView g, Dims d, const float* pts /*[n][3]*/, T* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
  const int x = (int)px, y = (int)py, z = (int)pz;
  const float dx = px - (float)x, dy = py - (float)y, dz = pz - (float)z;
  const float one = 1.0f;
  auto G = [&](int a, int b, int c) { return load_rt<T>(g, cell_index(g, d, x + a, y + b, z + c)); };
  T v = T((one - dx) * (one - dy) * (one - dz)) * G(0, 0, 0);
  v = v + T((one - dx) * (one - dy) * dz) * G(0, 0, 1);
  v = v + T((one - dx) * dy * (one - dz)) * G(0, 1, 0);
  v = v + T((one - dx) * dy * dz) * G(0, 1, 1);
  v = v + T(dx * (one - dy) * (one - dz)) * G(1, 0, 0);
  v = v + T(dx * (one - dy) * dz) * G(1, 0, 1);
  v = v + T(dx * dy * (one - dz)) * G(1, 1, 0);
  v = v + T(dx * dy * dz) * G(1, 1, 1);
  out[i] = v;
}

// ---- ExtrapolationOutflowBC: auxiliary data after the collision --------------------------
// Post-stream populations of ONE cell: periodic pull + the STREAMING-step boundary condition of that cell
// (nse_stepper.py:246-257).  Generic slow path: only the outflow cells and the cells behind them run it.
template <class L, class T>
__device__ void post_stream_cell(const FieldView& f0, const FieldView& bc, const FieldView& miss, const Dims& d, const uint8_t* kind_tab,
                                 const T* val_tab, const uint32_t* prof_keys, const T* prof_vals, int n_prof, int x, int y, int z,
                                 T (&f)[L::Q]) {
  static_for<L::Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    const int xs = f0.halo ? x - L::c(0, l) : wrap(x - L::c(0, l), d.nx);
    const int ys = wrap(y - L::c(1, l), d.ny);
    const int zs = wrap(z - L::c(2, l), d.nz);
    f[l] = load_rt<T>(f0, (size_t)l * f0.plane_stride + cell_index(f0, d, xs, ys, zs));
  });
  const unsigned id = static_cast<const uint8_t*>(bc.data)[cell_index(bc, d, x, y, z)];
  if (id == 0u) return;
  const unsigned kind = kind_tab[id];
  const T* val = val_tab + id * 27u;
  const size_t own = cell_index(f0, d, x, y, z);
  unsigned m = 0;
  if (miss.data) m = static_cast<const uint32_t*>(miss.data)[cell_index(miss, d, x, y, z)];
  if (kind == XLBHIP_BC_EQUILIBRIUM) {
    static_for<L::Q>([&](auto lc) { f[decltype(lc)::value] = val[decltype(lc)::value]; });
  } else if (kind == XLBHIP_BC_HALFWAY_BB) {
    static_for<L::Q>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      if ((m >> l) & 1u) f[l] = load_rt<T>(f0, (size_t)opp<L>(l) * f0.plane_stride + own) + val[l];
    });
  } else if (kind == XLBHIP_BC_EXTRAPOLATION_OUTFLOW) {
    static_for<L::Q>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      if ((m >> l) & 1u) f[l] = load_rt<T>(f0, (size_t)opp<L>(l) * f0.plane_stride + own);
    });
  } else if (kind == XLBHIP_BC_DO_NOTHING) {
    static_for<L::Q>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      f[l] = load_rt<T>(f0, (size_t)l * f0.plane_stride + own);
    });
  } else if (kind >= XLBHIP_BC_ZOUHE_VELOCITY && kind <= XLBHIP_BC_REGULARIZED_PRESSURE) {
    if (val[PROF_FLAG] != T(0) && n_prof > 0) val = prof_vals + 3 * (size_t)prof_find(prof_keys, n_prof, (uint32_t)cell_index(bc, d, x, y, z));
    zouhe_cell<L, T>(f, m, val, kind == XLBHIP_BC_ZOUHE_VELOCITY || kind == XLBHIP_BC_REGULARIZED_VELOCITY,
                     kind >= XLBHIP_BC_REGULARIZED_VELOCITY);
  }
}

// assemble_auxiliary_data of ExtrapolationOutflowBC (bc_extrapolation_outflow.py:104-134), run after the step kernel:
// at an outflow cell b with outward normal n, for every direction l whose opposite is missing,
//   f_1[l](b) = cs * f_post_stream[opp l](b - n) + (1 - cs) * f_post_stream[opp l](b)
template <class L, class T>
__global__ void k_outflow_aux(FieldView f0, FieldView f1, FieldView bc, FieldView miss, Dims d, const uint8_t* kind_tab, const T* val_tab,
                              const uint32_t* prof_keys, const T* prof_vals, int n_prof) {
  int x, y, z;
  if (!cell_of_thread(d, x, y, z)) return;
  const unsigned id = static_cast<const uint8_t*>(bc.data)[cell_index(bc, d, x, y, z)];
  if (id == 0u || kind_tab[id] != XLBHIP_BC_EXTRAPOLATION_OUTFLOW) return;
  const T* val = val_tab + id * 27u;
  const int n0 = (int)val[0], n1 = (int)val[1], n2 = (int)val[2];
  const T cs = val[3], omcs = val[4];
  int xn = x - n0;
  if (f0.halo) {
    if (xn < 0 || xn >= d.nx) return;  // the cell behind the face lives on another rank: outflow faces must be domain faces
  } else {
    xn = wrap(xn, d.nx);
  }
  const int yn = wrap(y - n1, d.ny), zn = wrap(z - n2, d.nz);
  T fb[L::Q], fn[L::Q];
  post_stream_cell<L, T>(f0, bc, miss, d, kind_tab, val_tab, prof_keys, prof_vals, n_prof, x, y, z, fb);
  post_stream_cell<L, T>(f0, bc, miss, d, kind_tab, val_tab, prof_keys, prof_vals, n_prof, xn, yn, zn, fn);
  const unsigned m = static_cast<const uint32_t*>(miss.data)[cell_index(miss, d, x, y, z)];
  const size_t o1 = cell_index(f1, d, x, y, z);
  static_for<L::Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    constexpr int o = opp<L>(l);
    if ((m >> o) & 1u) store_rt<T>(f1, (size_t)l * f1.plane_stride + o1, cs * fn[o] + omcs * fb[o]);
  });
}

// ---- masker (indices_boundary_masker.py:73-143, JAX semantics) ---------------------------
// scatter `value` at the listed GLOBAL indices that fall inside this rank's planes
// [x_lo, x_hi) (storage plane = gx - x_lo)
__global__ void k_scatter_u8(uint8_t* out, const int32_t* idx, int64_t n, uint8_t value, int x_lo, int x_hi, int gy,
                             int gz) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = idx[i], y = idx[n + i], z = idx[2 * n + i];
  if (x < x_lo || x >= x_hi || y < 0 || y >= gy || z < 0 || z >= gz) return;
  out[((size_t)(x - x_lo) * gy + y) * gz + z] = value;
}

// missing[l, x] = (x - c_l outside the global domain) | solid[x - c_l] | old_missing[l, x - c_l]
// solid has one ghost plane on each x side; old (bit-sets) uses the field's own halo.
template <class L>
__global__ void k_missing(uint32_t* out, const uint32_t* old, const uint8_t* solid, Dims d, int halo, int gnx, int x_offset) {
  int x, y, z;
  if (!cell_of_thread(d, x, y, z)) return;
  unsigned bits = 0;
  static_for<L::Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    const int xs = x - L::c(0, l), ys = y - L::c(1, l), zs = z - L::c(2, l);
    const int gxs = xs + x_offset;
    bool mis = gxs < 0 || gxs >= gnx || ys < 0 || ys >= d.ny || zs < 0 || zs >= d.nz;
    if (!mis) {
      mis = solid[((size_t)(xs + 1) * d.ny + ys) * d.nz + zs] != 0;
      if (!mis && (halo > 0 || (xs >= 0 && xs < d.nx)))
        mis = (old[((size_t)(xs + halo) * d.ny + ys) * d.nz + zs] >> l) & 1u;
    }
    bits |= (mis ? 1u : 0u) << l;
  });
  out[((size_t)(x + halo) * d.ny + y) * d.nz + z] = bits;
}

// pack / unpack the host (q, nx, ny, nz) u8 view of missing_mask <-> device bit-sets
__global__ void k_pack_missing(const uint8_t* bytes, uint32_t* bits, int q, Dims d, int halo) {
  int x, y, z;
  if (!cell_of_thread(d, x, y, z)) return;
  const size_t n = (size_t)d.nx * d.ny * d.nz;
  const size_t c = ((size_t)x * d.ny + y) * d.nz + z;
  unsigned b = 0;
  for (int l = 0; l < q; ++l) b |= (bytes[(size_t)l * n + c] ? 1u : 0u) << l;
  bits[((size_t)(x + halo) * d.ny + y) * d.nz + z] = b;
}
__global__ void k_unpack_missing(const uint32_t* bits, uint8_t* bytes, int q, Dims d, int halo) {
  int x, y, z;
  if (!cell_of_thread(d, x, y, z)) return;
  const size_t n = (size_t)d.nx * d.ny * d.nz;
  const size_t c = ((size_t)x * d.ny + y) * d.nz + z;
  const unsigned b = bits[((size_t)(x + halo) * d.ny + y) * d.nz + z];
  for (int l = 0; l < q; ++l) bytes[(size_t)l * n + c] = (b >> l) & 1u;
}

// plain streaming copy, W bytes per lane: the bandwidth yardstick and the byte-count
// calibration kernel for the FETCH_SIZE / WRITE_SIZE counters (known traffic: n bytes each way)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <class V>
__global__ void k_copy(const V* __restrict__ src, V* __restrict__ dst, size_t n) {
  // grid-stride: HIP refuses launches of 2^32 threads or more, an 81.6 GB field (4096 x 512 x 512, D3Q19) has 5.1e9 16-byte words
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(src[i], dst + i);
}

// ---- MeshMaskerAABB (boundary_masker/aabb.py:38-100, mesh_boundary_masker.py:62-181): surface voxelisation of a
// triangle mesh with the triangle / unit-box overlap test of Schwarz & Seidel (2010): the box straddles the triangle's
// plane AND its projection overlaps the triangle's projection on the xy, yz and zx planes.  fp32 throughout.
struct TriBox {
  float n[3], d1, d2;       // unit normal, plane offsets of the two critical box corners
  float ne[3][3][2], de[3][3];  // per projection axis-pair [ax] and edge [i]: 2-D edge normal and offset
};
__device__ inline bool tri_box_setup(const float* v /*[3][3]*/, TriBox& t) {
  float e[3][3];
  for (int a = 0; a < 3; ++a) {
    e[0][a] = v[3 + a] - v[a];
    e[1][a] = v[6 + a] - v[3 + a];
    e[2][a] = v[a] - v[6 + a];
  }
  float nx = e[0][1] * (-e[2][2]) - e[0][2] * (-e[2][1]);  // (v1 - v0) x (v2 - v0), v2 - v0 = -e[2]
  float ny = e[0][2] * (-e[2][0]) - e[0][0] * (-e[2][2]);
  float nz = e[0][0] * (-e[2][1]) - e[0][1] * (-e[2][0]);
  const float len = sqrtf(nx * nx + ny * ny + nz * nz);
  if (!(len > 0.0f)) return false;  // degenerate triangle: never intersects (mesh_boundary_masker.py:121)
  t.n[0] = nx / len;
  t.n[1] = ny / len;
  t.n[2] = nz / len;
  float c[3];
  for (int a = 0; a < 3; ++a) c[a] = t.n[a] > 0.0f ? 1.0f : 0.0f;
  t.d1 = t.n[0] * (c[0] - v[0]) + t.n[1] * (c[1] - v[1]) + t.n[2] * (c[2] - v[2]);
  t.d2 = t.n[0] * ((1.0f - c[0]) - v[0]) + t.n[1] * ((1.0f - c[1]) - v[1]) + t.n[2] * ((1.0f - c[2]) - v[2]);
  for (int ax0 = 0; ax0 < 3; ++ax0) {
    const int ax1 = (ax0 + 1) % 3, ax2 = (ax0 + 2) % 3;
    const float sgn = t.n[ax2] < 0.0f ? -1.0f : 1.0f;
    for (int i = 0; i < 3; ++i) {
      const float a0 = -sgn * e[i][ax1], a1 = sgn * e[i][ax0];
      t.ne[ax0][i][0] = a0;
      t.ne[ax0][i][1] = a1;
      t.de[ax0][i] = -(a0 * v[3 * i + ax0] + a1 * v[3 * i + ax1]) + fmaxf(0.0f, a0) + fmaxf(0.0f, a1);
    }
  }
  return true;
}
__device__ inline bool tri_box_overlap(const TriBox& t, float lx, float ly, float lz) {
  const float low[3] = {lx, ly, lz};
  const float np = t.n[0] * lx + t.n[1] * ly + t.n[2] * lz;
  if (!((np + t.d1) * (np + t.d2) <= 0.0f)) return false;
  for (int ax0 = 0; ax0 < 3; ++ax0) {
    const int ax1 = (ax0 + 1) % 3;
    for (int i = 0; i < 3; ++i)
      if (!(t.ne[ax0][i][0] * low[ax0] + t.ne[ax0][i][1] * low[ax1] + t.de[ax0][i] >= 0.0f)) return false;
  }
  return true;
}
// one thread per triangle: every unit voxel [i, i+1]^3 its bounding box touches is tested, hits are marked solid
__global__ void k_mesh_solid(const float* verts /*[n_tri][3][3]*/, int64_t n_tri, uint8_t* solid, int nx, int ny, int nz) {
  const int64_t tix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tix >= n_tri) return;
  const float* v = verts + 9 * tix;
  TriBox t;
  if (!tri_box_setup(v, t)) return;
  int lo[3], hi[3];
  const int ext[3] = {nx, ny, nz};
  for (int a = 0; a < 3; ++a) {
    const float mn = fminf(v[a], fminf(v[3 + a], v[6 + a])), mx = fmaxf(v[a], fmaxf(v[3 + a], v[6 + a]));
    lo[a] = max(0, (int)floorf(mn) - 1);
    hi[a] = min(ext[a] - 1, (int)floorf(mx));
  }
  for (int i = lo[0]; i <= hi[0]; ++i)
    for (int j = lo[1]; j <= hi[1]; ++j)
      for (int k = lo[2]; k <= hi[2]; ++k)
        if (tri_box_overlap(t, (float)i, (float)j, (float)k)) solid[((size_t)i * ny + j) * nz + k] = 1;
}
// per voxel (aabb.py:57-79 + resolve_out_of_bound_kernel): solid voxels get BC_SOLID (255); a fluid voxel with a solid
// neighbour at +c_l gets the BC id and missing[opp l]; voxels of this id also miss every direction pulled from outside the box
template <class L>
__global__ void k_mesh_classify(const uint8_t* solid, uint8_t* bc, uint32_t* miss, Dims d, int id) {
  int x, y, z;
  if (!cell_of_thread(d, x, y, z)) return;
  const size_t c = ((size_t)x * d.ny + y) * d.nz + z;
  if (bc[c] == 255 || solid[c]) {
    bc[c] = 255;
    return;
  }
  unsigned bits = 0;
  static_for<L::Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    if constexpr (l != opp<L>(l)) {  // every direction but the rest one
      const int xn = x + L::c(0, l), yn = y + L::c(1, l), zn = z + L::c(2, l);
      if (xn >= 0 && xn < d.nx && yn >= 0 && yn < d.ny && zn >= 0 && zn < d.nz && solid[((size_t)xn * d.ny + yn) * d.nz + zn]) bits |= 1u << opp<L>(l);
    }
  });
  if (bits != 0u) bc[c] = (uint8_t)id;
  if (bc[c] == id) {
    static_for<L::Q>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      if constexpr (l != opp<L>(l)) {
        const int xp = x - L::c(0, l), yp = y - L::c(1, l), zp = z - L::c(2, l);
        if (xp < 0 || xp >= d.nx || yp < 0 || yp >= d.ny || zp < 0 || zp >= d.nz) bits |= 1u << l;
      }
    });
  }
  if (bits != 0u) miss[c] |= bits;
}

// ---- MeshMaskerRay (boundary_masker/ray.py:38-76): a voxel whose link along c_l (from its centre, length |c_l|) crosses the
// surface gets the BC id and missing[opp l].  One thread per triangle over the voxels around its bounding box; the
// segment / triangle test is Moeller-Trumbore in fp32 (both faces count, like a mesh ray query).
// closest-hit building block: true and *t_out = ray parameter when the segment p + t d, 0 <= t <= max_t, meets the triangle
__device__ inline bool seg_tri_t(const float* v, float px, float py, float pz, float dx, float dy, float dz, float max_t, float* t_out) {
  const float e1x = v[3] - v[0], e1y = v[4] - v[1], e1z = v[5] - v[2];
  const float e2x = v[6] - v[0], e2y = v[7] - v[1], e2z = v[8] - v[2];
  const float pvx = dy * e2z - dz * e2y, pvy = dz * e2x - dx * e2z, pvz = dx * e2y - dy * e2x;
  const float det = (e1x * pvx + e1y * pvy) + e1z * pvz;
  if (fabsf(det) < 1e-12f) return false;
  const float inv = 1.0f / det;
  const float tx = px - v[0], ty = py - v[1], tz = pz - v[2];
  const float u = ((tx * pvx + ty * pvy) + tz * pvz) * inv;
  if (u < 0.0f || u > 1.0f) return false;
  const float qx = ty * e1z - tz * e1y, qy = tz * e1x - tx * e1z, qz = tx * e1y - ty * e1x;
  const float w = ((dx * qx + dy * qy) + dz * qz) * inv;
  if (w < 0.0f || u + w > 1.0f) return false;
  const float t = ((e2x * qx + e2y * qy) + e2z * qz) * inv;
  *t_out = t;
  return t >= 0.0f && t <= max_t;
}
__device__ inline bool seg_tri_hit(const float* v, float px, float py, float pz, float dx, float dy, float dz, float max_t) {
  float t;
  return seg_tri_t(v, px, py, pz, dx, dy, dz, max_t, &t);
}
template <class L, int l>
__device__ __forceinline__ float link_len() {
  constexpr int n2 = L::c(0, l) * L::c(0, l) + L::c(1, l) * L::c(1, l) + L::c(2, l) * L::c(2, l);
  return n2 == 1 ? 1.0f : (n2 == 2 ? 1.41421356237309515f : 1.73205080756887719f);
}
// smallest ray parameter so far of link l of voxel c (non-negative floats order like their bit patterns)
__device__ __forceinline__ void t_min(unsigned* tbuf, size_t cells, int l, size_t c, float t) {
  atomicMin(tbuf + (size_t)l * cells + c, __float_as_uint(t));
}
constexpr unsigned T_NONE = 0x7f800000u;  // +inf: no hit

template <class L>
__global__ void k_mesh_ray(const float* verts, int64_t n_tri, uint8_t* bc, uint32_t* miss, Dims d, int id) {
  const int64_t tix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tix >= n_tri) return;
  const float* v = verts + 9 * tix;
  int lo[3], hi[3];
  const int ext[3] = {d.nx, d.ny, d.nz};
  for (int a = 0; a < 3; ++a) {
    const float mn = fminf(v[a], fminf(v[3 + a], v[6 + a])), mx = fmaxf(v[a], fmaxf(v[3 + a], v[6 + a]));
    lo[a] = max(0, (int)floorf(mn) - 2);  // centre i + 0.5 within one link (<= 1 per axis) of the bounding box
    hi[a] = min(ext[a] - 1, (int)floorf(mx) + 1);
  }
  for (int i = lo[0]; i <= hi[0]; ++i)
    for (int j = lo[1]; j <= hi[1]; ++j)
      for (int k = lo[2]; k <= hi[2]; ++k) {
        const float px = (float)i + 0.5f, py = (float)j + 0.5f, pz = (float)k + 0.5f;
        unsigned bits = 0;
        static_for<L::Q>([&](auto lc) {
          constexpr int l = decltype(lc)::value;
          if constexpr (l != opp<L>(l)) {
            constexpr int n2 = L::c(0, l) * L::c(0, l) + L::c(1, l) * L::c(1, l) + L::c(2, l) * L::c(2, l);
            const float len = n2 == 1 ? 1.0f : (n2 == 2 ? 1.41421356237309515f : 1.73205080756887719f);
            if (seg_tri_hit(v, px, py, pz, (float)L::c(0, l) / len, (float)L::c(1, l) / len, (float)L::c(2, l) / len, len)) bits |= 1u << opp<L>(l);
          }
        });
        if (bits != 0u) {
          const size_t c = ((size_t)i * d.ny + j) * d.nz + k;
          bc[c] = (uint8_t)id;
          atomicOr(miss + c, bits);
        }
      }
}
// resolve_out_of_bound_kernel (mesh_boundary_masker.py:156-176): voxels of this id miss the directions pulled from outside the box
template <class L>
__global__ void k_mesh_resolve(const uint8_t* bc, uint32_t* miss, Dims d, int id) {
  int x, y, z;
  if (!cell_of_thread(d, x, y, z)) return;
  const size_t c = ((size_t)x * d.ny + y) * d.nz + z;
  if (bc[c] != id) return;
  unsigned bits = 0;
  static_for<L::Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    if constexpr (l != opp<L>(l)) {
      const int xp = x - L::c(0, l), yp = y - L::c(1, l), zp = z - L::c(2, l);
      if (xp < 0 || xp >= d.nx || yp < 0 || yp >= d.ny || zp < 0 || zp >= d.nz) bits |= 1u << l;
    }
  });
  if (bits != 0u) miss[c] |= bits;
}

// ---- wall distances of MeshMaskerRay (ray.py:69-76): closest hit per (voxel, link) over all triangles ----
// One thread per triangle over the voxels around its bounding box, like k_mesh_ray; additionally the smallest ray
// parameter of every link goes to tbuf[l][voxel] (atomicMin).  k_mesh_weights turns them into distances[l] = t / |c_l|.
template <class L>
__global__ void k_mesh_ray_dist(const float* verts, int64_t n_tri, uint8_t* bc, uint32_t* miss, unsigned* tbuf, Dims d, int id) {
  const int64_t tix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tix >= n_tri) return;
  const float* v = verts + 9 * tix;
  int lo[3], hi[3];
  const int ext[3] = {d.nx, d.ny, d.nz};
  for (int a = 0; a < 3; ++a) {
    const float mn = fminf(v[a], fminf(v[3 + a], v[6 + a])), mx = fmaxf(v[a], fmaxf(v[3 + a], v[6 + a]));
    lo[a] = max(0, (int)floorf(mn) - 2);
    hi[a] = min(ext[a] - 1, (int)floorf(mx) + 1);
  }
  const size_t cells = (size_t)d.nx * d.ny * d.nz;
  for (int i = lo[0]; i <= hi[0]; ++i)
    for (int j = lo[1]; j <= hi[1]; ++j)
      for (int k = lo[2]; k <= hi[2]; ++k) {
        const float px = (float)i + 0.5f, py = (float)j + 0.5f, pz = (float)k + 0.5f;
        const size_t c = ((size_t)i * d.ny + j) * d.nz + k;
        unsigned bits = 0;
        static_for<L::Q>([&](auto lc) {
          constexpr int l = decltype(lc)::value;
          if constexpr (l != opp<L>(l)) {
            const float len = link_len<L, l>();
            float t;
            if (seg_tri_t(v, px, py, pz, (float)L::c(0, l) / len, (float)L::c(1, l) / len, (float)L::c(2, l) / len, len, &t)) {
              bits |= 1u << opp<L>(l);
              t_min(tbuf, cells, l, c, t);
            }
          }
        });
        if (bits != 0u) {
          bc[c] = (uint8_t)id;
          atomicOr(miss + c, bits);
        }
      }
}
// mode 0 (RAY): distances[l] = t / |c_l| where a hit was recorded
template <class L>
__global__ void k_mesh_weights_ray(const unsigned* tbuf, FieldView dist, Dims d) {
  int x, y, z;
  if (!cell_of_thread(d, x, y, z)) return;
  const size_t cells = (size_t)d.nx * d.ny * d.nz, c = ((size_t)x * d.ny + y) * d.nz + z;
  static_for<L::Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    if constexpr (l != opp<L>(l)) {
      const unsigned tb = tbuf[(size_t)l * cells + c];
      if (tb != T_NONE) static_cast<float*>(dist.data)[(size_t)l * dist.plane_stride + c] = __uint_as_float(tb) / link_len<L, l>();
    }
  });
}

// ---- MeshMaskerWinding (winding.py:46-103) ----
// inside test: generalized winding number of the triangle soup at the voxel centre, exact (sum of the signed solid angles,
// Van Oosterom & Strackee, fp64) instead of Warp's BVH approximation; > 0.5 = inside.  One thread per voxel of the mesh's
// bounding box, every thread walks all triangles (the triangle data is a broadcast read).
__global__ void k_mesh_winding(const float* verts, int64_t n_tri, uint8_t* solid, Dims d, int lo0, int lo1, int lo2, int n0, int n1, int n2) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)n0 * n1 * n2) return;
  const int i = lo0 + (int)(t / ((int64_t)n1 * n2)), j = lo1 + (int)((t / n2) % n1), k = lo2 + (int)(t % n2);
  const double px = i + 0.5, py = j + 0.5, pz = k + 0.5;
  double sum = 0.0;
  for (int64_t f = 0; f < n_tri; ++f) {
    const float* v = verts + 9 * f;
    const double ax = v[0] - px, ay = v[1] - py, az = v[2] - pz;
    const double bx = v[3] - px, by = v[4] - py, bz = v[5] - pz;
    const double cx = v[6] - px, cy = v[7] - py, cz = v[8] - pz;
    const double la = sqrt(ax * ax + ay * ay + az * az), lb = sqrt(bx * bx + by * by + bz * bz), lc = sqrt(cx * cx + cy * cy + cz * cz);
    const double num = ax * (by * cz - bz * cy) + ay * (bz * cx - bx * cz) + az * (bx * cy - by * cx);
    const double den = la * lb * lc + (ax * bx + ay * by + az * bz) * lc + (bx * cx + by * cy + bz * cz) * la + (cx * ax + cy * ay + cz * az) * lb;
    sum += 2.0 * atan2(num, den);
  }
  if (sum / (4.0 * 3.14159265358979323846) > 0.5) solid[((size_t)i * d.ny + j) * d.nz + k] = 1;
}
// rays out of the solid voxels: one thread per triangle over the SOLID voxels around its bounding box
template <class L>
__global__ void k_mesh_winding_rays(const float* verts, int64_t n_tri, const uint8_t* solid, unsigned* tbuf, Dims d) {
  const int64_t tix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tix >= n_tri) return;
  const float* v = verts + 9 * tix;
  int lo[3], hi[3];
  const int ext[3] = {d.nx, d.ny, d.nz};
  for (int a = 0; a < 3; ++a) {
    const float mn = fminf(v[a], fminf(v[3 + a], v[6 + a])), mx = fmaxf(v[a], fmaxf(v[3 + a], v[6 + a]));
    lo[a] = max(0, (int)floorf(mn) - 2);
    hi[a] = min(ext[a] - 1, (int)floorf(mx) + 1);
  }
  const size_t cells = (size_t)d.nx * d.ny * d.nz;
  for (int i = lo[0]; i <= hi[0]; ++i)
    for (int j = lo[1]; j <= hi[1]; ++j)
      for (int k = lo[2]; k <= hi[2]; ++k) {
        const size_t c = ((size_t)i * d.ny + j) * d.nz + k;
        if (!solid[c]) continue;
        const float px = (float)i + 0.5f, py = (float)j + 0.5f, pz = (float)k + 0.5f;
        static_for<L::Q>([&](auto lc) {
          constexpr int l = decltype(lc)::value;
          if constexpr (l != opp<L>(l)) {
            const float len = link_len<L, l>();
            float t;
            if (seg_tri_t(v, px, py, pz, (float)L::c(0, l) / len, (float)L::c(1, l) / len, (float)L::c(2, l) / len, len, &t)) t_min(tbuf, cells, l, c, t);
          }
        });
      }
}
// tags: the fluid neighbour at +c_l of a solid voxel whose ray along c_l met the surface gets the id, missing[l] and
// distances[opp l] = (|c_l| - t) / |c_l|; written from the NEIGHBOUR's thread (one writer per voxel); solid voxels -> BC_SOLID
template <class L>
__global__ void k_mesh_winding_tag(const uint8_t* solid, const unsigned* tbuf, uint8_t* bc, uint32_t* miss, FieldView dist, Dims d, int id) {
  int x, y, z;
  if (!cell_of_thread(d, x, y, z)) return;
  const size_t cells = (size_t)d.nx * d.ny * d.nz, c = ((size_t)x * d.ny + y) * d.nz + z;
  if (solid[c]) {
    bc[c] = 255;
    return;
  }
  unsigned bits = 0;
  static_for<L::Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    if constexpr (l != opp<L>(l)) {
      const int xs = x - L::c(0, l), ys = y - L::c(1, l), zs = z - L::c(2, l);  // the solid voxel whose ray along c_l reaches me
      if (xs >= 0 && xs < d.nx && ys >= 0 && ys < d.ny && zs >= 0 && zs < d.nz) {
        const size_t cs = ((size_t)xs * d.ny + ys) * d.nz + zs;
        if (solid[cs]) {
          const unsigned tb = tbuf[(size_t)l * cells + cs];
          if (tb != T_NONE) {
            bits |= 1u << l;
            if (dist.data) {
              const float len = link_len<L, l>();
              static_cast<float*>(dist.data)[(size_t)opp<L>(l) * dist.plane_stride + c] = (len - __uint_as_float(tb)) / len;
            }
          }
        }
      }
    }
  });
  if (bits != 0u) {
    bc[c] = (uint8_t)id;
    miss[c] |= bits;
  }
}

// ---- MeshMaskerAABBClose (aabb_close.py:67-154, 216-263, 303-344) ----
// AABB voxelisation on the grid padded by `pad` voxels per side (voxel (i, j, k) of the padded grid is the unit cube at
// (i - pad, j - pad, k - pad))
__global__ void k_mesh_solid_padded(const float* verts, int64_t n_tri, uint8_t* solid, int px, int py, int pz, int pad) {
  const int64_t tix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tix >= n_tri) return;
  const float* v = verts + 9 * tix;
  TriBox t;
  if (!tri_box_setup(v, t)) return;
  int lo[3], hi[3];
  const int ext[3] = {px, py, pz};
  for (int a = 0; a < 3; ++a) {
    const float mn = fminf(v[a], fminf(v[3 + a], v[6 + a])), mx = fmaxf(v[a], fmaxf(v[3 + a], v[6 + a]));
    lo[a] = max(0, (int)floorf(mn) - 1 + pad);
    hi[a] = min(ext[a] - 1, (int)floorf(mx) + pad);
  }
  for (int i = lo[0]; i <= hi[0]; ++i)
    for (int j = lo[1]; j <= hi[1]; ++j)
      for (int k = lo[2]; k <= hi[2]; ++k)
        if (tri_box_overlap(t, (float)(i - pad), (float)(j - pad), (float)(k - pad))) solid[((size_t)i * py + j) * pz + k] = 1;
}
// max (dilate) / min (erode) filter over the (2 h + 1)^3 cube; voxels within h of the padded grid's faces are copied
__global__ void k_morph(const uint8_t* in, uint8_t* out, int px, int py, int pz, int h, int dilate) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (int64_t)px * py * pz) return;
  const int i = (int)(t / ((int64_t)py * pz)), j = (int)((t / pz) % py), k = (int)(t % pz);
  if (i < h || i >= px - h || j < h || j >= py - h || k < h || k >= pz - h) {
    out[t] = in[t];
    return;
  }
  uint8_t acc = dilate ? 0 : 1;
  for (int a = -h; a <= h; ++a)
    for (int b = -h; b <= h; ++b)
      for (int c = -h; c <= h; ++c) {
        const uint8_t v = in[((size_t)(i + a) * py + (j + b)) * pz + (k + c)];
        acc = dilate ? (acc | v) : (acc & v);
      }
  out[t] = acc;
}
// crop the padded mask to the domain
__global__ void k_crop(const uint8_t* in, uint8_t* out, Dims d, int py, int pz, int pad) {
  int x, y, z;
  if (!cell_of_thread(d, x, y, z)) return;
  out[((size_t)x * d.ny + y) * d.nz + z] = in[((size_t)(x + pad) * py + (y + pad)) * pz + (z + pad)];
}
// closest hit within 1.5 |c_l| along c_l for the boundary voxels (bc == id) whose neighbour at +c_l is solid
template <class L>
__global__ void k_mesh_close_rays(const float* verts, int64_t n_tri, const uint8_t* solid, const uint8_t* bc, unsigned* tbuf, Dims d, int id) {
  const int64_t tix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (tix >= n_tri) return;
  const float* v = verts + 9 * tix;
  int lo[3], hi[3];
  const int ext[3] = {d.nx, d.ny, d.nz};
  for (int a = 0; a < 3; ++a) {
    const float mn = fminf(v[a], fminf(v[3 + a], v[6 + a])), mx = fmaxf(v[a], fmaxf(v[3 + a], v[6 + a]));
    lo[a] = max(0, (int)floorf(mn) - 3);  // centre within 1.5 links (<= 1.5 per axis) of the bounding box
    hi[a] = min(ext[a] - 1, (int)floorf(mx) + 2);
  }
  const size_t cells = (size_t)d.nx * d.ny * d.nz;
  for (int i = lo[0]; i <= hi[0]; ++i)
    for (int j = lo[1]; j <= hi[1]; ++j)
      for (int k = lo[2]; k <= hi[2]; ++k) {
        const size_t c = ((size_t)i * d.ny + j) * d.nz + k;
        if (bc[c] != id) continue;
        const float px = (float)i + 0.5f, py = (float)j + 0.5f, pz = (float)k + 0.5f;
        static_for<L::Q>([&](auto lc) {
          constexpr int l = decltype(lc)::value;
          if constexpr (l != opp<L>(l)) {
            const int xn = i + L::c(0, l), yn = j + L::c(1, l), zn = k + L::c(2, l);
            if (xn >= 0 && xn < d.nx && yn >= 0 && yn < d.ny && zn >= 0 && zn < d.nz && solid[((size_t)xn * d.ny + yn) * d.nz + zn]) {
              const float len = link_len<L, l>();
              float t;
              if (seg_tri_t(v, px, py, pz, (float)L::c(0, l) / len, (float)L::c(1, l) / len, (float)L::c(2, l) / len, 1.5f * len, &t)) t_min(tbuf, cells, l, c, t);
            }
          }
        });
      }
}
// distances[l] = (t - 0.5 |c_l|) / |c_l| with a hit, 1.0 without, for every link of a boundary voxel that ends in a solid voxel
template <class L>
__global__ void k_mesh_weights_close(const uint8_t* solid, const uint8_t* bc, const unsigned* tbuf, FieldView dist, Dims d, int id) {
  int x, y, z;
  if (!cell_of_thread(d, x, y, z)) return;
  const size_t cells = (size_t)d.nx * d.ny * d.nz, c = ((size_t)x * d.ny + y) * d.nz + z;
  if (bc[c] != id || solid[c]) return;
  static_for<L::Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    if constexpr (l != opp<L>(l)) {
      const int xn = x + L::c(0, l), yn = y + L::c(1, l), zn = z + L::c(2, l);
      if (xn >= 0 && xn < d.nx && yn >= 0 && yn < d.ny && zn >= 0 && zn < d.nz && solid[((size_t)xn * d.ny + yn) * d.nz + zn]) {
        const unsigned tb = tbuf[(size_t)l * cells + c];
        const float len = link_len<L, l>();
        static_cast<float*>(dist.data)[(size_t)l * dist.plane_stride + c] = tb == T_NONE ? 1.0f : (__uint_as_float(tb) - 0.5f * len) / len;
      }
    }
  });
}

// rows of a field at listed interior cells: out[i][l] = field[l][cells[i]] (small device -> host transfers of boundary data)
template <class E>
__global__ void k_gather(const E* data, size_t plane_stride, size_t ghost, int card, const uint32_t* cells, int64_t n, E* out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int l = 0; l < card; ++l) out[i * card + l] = data[(size_t)l * plane_stride + ghost + cells[i]];
}

// meta word of the two-step kernel, resolved once per run so that the kernel never searches an id table (layouts: step2_kernel.hpp
// S2Meta): kind (0 fluid, XLBHIP_BC_* for the basic kinds, or "halfway wall WITH a moving-wall term"), slot of the BC in the
// stepper's packed tables, missing bit-set.  wide = 0: 4 + 4 + up to 24 bits (D3Q19); wide = 1: 3 + 3 + bits 1 .. 26 (D3Q27).
__global__ void k_build_meta(const uint8_t* bc, const uint32_t* miss, uint32_t* meta, size_t n, unsigned long long ids_packed,
                             unsigned kinds_packed, unsigned moving_mask, int wide) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned id = bc[i];
  unsigned w = 0;
  if (id != 0u) {
    unsigned kind = 0, slot = 0;
    for (int s = 0; s < 8; ++s)
      if (((unsigned)(ids_packed >> (8 * s)) & 0xffu) == id) {
        kind = (kinds_packed >> (4 * s)) & 0xfu;
        slot = (unsigned)s;
      }
    const unsigned m = miss ? miss[i] : 0u;
    if (wide) {
      if (kind == XLBHIP_BC_HALFWAY_BB && ((moving_mask >> slot) & 1u)) kind = 5u;  // S2Meta<D3Q27>::K_HWM
      w = (kind & 7u) | (slot << 3) | ((m >> 1) << 6);
    } else {
      if (kind == XLBHIP_BC_HALFWAY_BB && ((moving_mask >> slot) & 1u)) kind = 15u;  // S2Meta<D3Q19>::K_HWM
      w = kind | (slot << 4) | (m << 8);
    }
  }
  meta[i] = w;
}

template <class E>
__global__ void k_fill(E* p, size_t n, E v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;  // grid-stride (see k_copy)
}

}  // namespace xlb
