/*
 * ORACLE DIRECTORY — TEST / MEASUREMENT INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * The OPTIMISED CPU baseline of SURVEY.md 8(d) ("C++17/OpenMP CPU stepper ... -O3 -march=native"):
 * the same LBM step as oracle/lbm_ref.c (reference: xlb/operator/stepper/nse_stepper.py:237-282, BGK
 * bgk.py:27-32, equilibrium quadratic_equilibrium.py:23-30, pull streaming stream.py:29-62, halfway /
 * fullway / equilibrium BCs) written the way a CPU port would be: compile-time lattice, SoA rows,
 * the z loop vectorised (#pragma omp simd), boundary cells on a scalar side path, OpenMP over (x, y)
 * rows.  Built WHERE IT RUNS with `g++ -O3 -march=native -fopenmp` (contraction allowed), so it is NOT
 * bit-exact; tests/test_oracle_c.py checks it against the NumPy oracle to 1e-6.  bench.py times it as
 * the `cpu_baseline` ("kind": "port"); nothing else may load it.
 *
 * The lattice tables come from the constexpr constructions in xlb_amd/csrc/lattice.hpp (the same
 * re-derivation of d3q19.py:19-27 / d3q27.py:19-29 the device code uses; cross-checked against the NumPy
 * oracle in tests/test_capi_symbols.py).
 */
#include <cstddef>
#include <cstdint>
#ifdef _OPENMP
#include <omp.h>
#endif
#if defined(__SSE2__)
#include <xmmintrin.h>
#endif
#define XLB_HD
#include "../xlb_amd/csrc/lattice.hpp"

using namespace xlb;

namespace {

template <class L>
inline void collide_bgk(float (&f)[L::Q], float omega) {
  constexpr int Q = L::Q;
  float rho = f[0];
  static_for<Q - 1>([&](auto lc) { rho += f[decltype(lc)::value + 1]; });
  float u[3] = {0.f, 0.f, 0.f};
  static_for<Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    static_for<3>([&](auto ac) {
      constexpr int a = decltype(ac)::value;
      if constexpr (L::c(a, l) == 1) u[a] += f[l];
      if constexpr (L::c(a, l) == -1) u[a] -= f[l];
    });
  });
  const float inv = 1.0f / rho;
  u[0] *= inv;
  u[1] *= inv;
  u[2] *= inv;
  const float usqr = 1.5f * (u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
  static_for<Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    const float cu = 3.0f * ((float)L::c(0, l) * u[0] + (float)L::c(1, l) * u[1] + (float)L::c(2, l) * u[2]);
    const float feq = rho * (float)L::w(l) * (1.0f + cu * (1.0f + 0.5f * cu) - usqr);
    f[l] -= omega * (f[l] - feq);
  });
}

struct BcTable {
  int kind[256];
  const double* val[256];
};

// one cell with every boundary kind of the basic set (scalar side path)
template <class L>
inline void cell_generic(const float* src, float* dst, const uint8_t* bc, const uint8_t* miss, const BcTable& tab, size_t N, int nx, int ny,
                         int nz, int x, int y, int z, float omega) {
  constexpr int Q = L::Q;
  const size_t cell = ((size_t)x * ny + y) * nz + z;
  float f[Q];
  static_for<Q>([&](auto lc) {
    constexpr int l = decltype(lc)::value;
    int xs = x - L::c(0, l), ys = y - L::c(1, l), zs = z - L::c(2, l);
    xs = xs < 0 ? xs + nx : (xs >= nx ? xs - nx : xs);
    ys = ys < 0 ? ys + ny : (ys >= ny ? ys - ny : ys);
    zs = zs < 0 ? zs + nz : (zs >= nz ? zs - nz : zs);
    f[l] = src[(size_t)l * N + ((size_t)xs * ny + ys) * nz + zs];
  });
  const int id = bc ? bc[cell] : 0;
  const int kind = tab.kind[id];
  if (kind == 1) {
    static_for<Q>([&](auto lc) { f[decltype(lc)::value] = (float)tab.val[id][decltype(lc)::value]; });
  } else if (kind == 2) {
    static_for<Q>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      if (miss[(size_t)l * N + cell]) f[l] = src[(size_t)opp<L>(l) * N + cell] + (float)tab.val[id][l];
    });
  } else if (kind == 4) {
    static_for<Q>([&](auto lc) { f[decltype(lc)::value] = src[(size_t) decltype(lc)::value * N + cell]; });
  }
  if (kind == 3) {
    static_for<Q>([&](auto lc) {
      constexpr int l = decltype(lc)::value;
      dst[(size_t)l * N + cell] = f[opp<L>(l)];
    });
    return;
  }
  collide_bgk<L>(f, omega);
  static_for<Q>([&](auto lc) { dst[(size_t) decltype(lc)::value * N + cell] = f[decltype(lc)::value]; });
}

template <class L>
int step(const float* src, float* dst, const uint8_t* bc, const uint8_t* miss, int nx, int ny, int nz, const BcTable& tab, float omega) {
  constexpr int Q = L::Q;
  const size_t N = (size_t)nx * ny * nz;
#pragma omp parallel
  {
#if defined(__SSE2__)
    // a cavity started at rest is full of denormal velocities far from the lid; x86 handles them in microcode (3-4x slower
    // overall).  Flush-to-zero / denormals-are-zero is what a tuned CPU code (and XLA:CPU) runs with.
    const unsigned csr = _mm_getcsr();
    _mm_setcsr(csr | 0x8040u);
#endif
#pragma omp for collapse(2) schedule(static)
  for (int x = 0; x < nx; ++x) {
    for (int y = 0; y < ny; ++y) {
      const size_t row = ((size_t)x * ny + y) * nz;
      // span of the row without boundary cells (the z ends always take the scalar path: periodic wrap)
      int z0 = 1, z1 = nz - 1;
      if (bc) {
        while (z0 < z1 && bc[row + z0]) ++z0;
        while (z1 > z0 && bc[row + z1 - 1]) --z1;
        for (int z = z0; z < z1; ++z)
          if (bc[row + z]) {  // boundary cells inside the row (solid bodies): whole row on the scalar path
            z0 = z1 = 1;
            break;
          }
      }
      if (z1 <= z0) z0 = z1 = nz > 1 ? 1 : 0;
      for (int z = 0; z < z0 && z < nz; ++z) cell_generic<L>(src, dst, bc, miss, tab, N, nx, ny, nz, x, y, z, omega);
      for (int z = z1 > z0 ? z1 : z0; z < nz; ++z) cell_generic<L>(src, dst, bc, miss, tab, N, nx, ny, nz, x, y, z, omega);
      if (z1 <= z0) continue;
      const float* srow[Q];
      static_for<Q>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
        int xs = x - L::c(0, l), ys = y - L::c(1, l);
        xs = xs < 0 ? xs + nx : (xs >= nx ? xs - nx : xs);
        ys = ys < 0 ? ys + ny : (ys >= ny ? ys - ny : ys);
        srow[l] = src + (size_t)l * N + ((size_t)xs * ny + ys) * nz - L::c(2, l);
      });
      float* drow = dst + row;
#pragma omp simd
      for (int z = z0; z < z1; ++z) {
        float f[Q];
        static_for<Q>([&](auto lc) { f[decltype(lc)::value] = srow[decltype(lc)::value][z]; });
        collide_bgk<L>(f, omega);
        static_for<Q>([&](auto lc) { drow[(size_t) decltype(lc)::value * N + z] = f[decltype(lc)::value]; });
      }
    }
  }
#if defined(__SSE2__)
    _mm_setcsr(csr);
#endif
  }
  return 0;
}

}  // namespace

extern "C" {

/* n steps a <-> b (fp32, BGK); q selects D3Q19 / D3Q27; result in a for even n, else in b.  bc_kinds as in lbm_ref.c
 * (1 equilibrium, 2 halfway, 3 fullway, 4 do-nothing); bc_values [n_bc][27] doubles. */
int lbmfast_run(float* a, float* b, const uint8_t* bc_mask, const uint8_t* missing, int nx, int ny, int nz, int q, int n_bc, const int* bc_ids,
                const int* bc_kinds, const double* bc_values, double omega, int n_steps) {
  BcTable tab;
  for (int i = 0; i < 256; ++i) {
    tab.kind[i] = 0;
    tab.val[i] = nullptr;
  }
  for (int i = 0; i < n_bc; ++i) {
    if (bc_ids[i] < 1 || bc_ids[i] > 255) return 2;
    tab.kind[bc_ids[i]] = bc_kinds[i];
    tab.val[bc_ids[i]] = bc_values + (size_t)i * 27;
  }
  if (n_bc == 0) bc_mask = nullptr;
  for (int i = 0; i < n_steps; ++i) {
    const float* s = (i & 1) ? b : a;
    float* d = (i & 1) ? a : b;
    int rc = q == 19 ? step<D3Q19>(s, d, bc_mask, missing, nx, ny, nz, tab, (float)omega)
                     : (q == 27 ? step<D3Q27>(s, d, bc_mask, missing, nx, ny, nz, tab, (float)omega) : 4);
    if (rc) return rc;
  }
  return 0;
}

int lbmfast_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

}  // extern "C"
