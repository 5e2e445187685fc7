/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C restatement of the reference's per-timestep LBM step (JAX branch,
 * xlb/operator/stepper/nse_stepper.py:237-282 and the operators it composes), written
 * independently of oracle/xlb_numpy.py but with the SAME explicit operation order so that the
 * two agree bit for bit (gcc -O2 -ffp-contract=off, no fast-math).  Uses:
 *   - second, independent checker of the HIP kernels at sizes NumPy is too slow for;
 *   - the CPU baseline timed by bench.py ("kind": "port"), OpenMP over (x, y) rows.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Pinning: see the header of oracle/xlb_numpy.py — the reference ships no golden vectors;
 * this file is checked against the NumPy oracle (itself pinned by the reference's known-answer
 * tests) in tests/test_oracle_c.py.  KBC / halfway bounce-back / the composed stepper are
 * "parity unpinned by the reference".
 *
 * Lattice tables are passed in by the caller (derived in Python from the reference's
 * constructions); only their size Q is compiled in.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  int d, q;
  int c[3][27];   /* internal 3-component form; 2-D sets have c[0][*] = 0 */
  double w[27];
  int opp[27];
  int cc[27][6];
} lattice_t;

#define Q 9
#define T float
#define SUFFIX q9_f32
#include "lbm_ref_body.inc"
#undef T
#undef SUFFIX
#define T double
#define SUFFIX q9_f64
#include "lbm_ref_body.inc"
#undef T
#undef SUFFIX
#undef Q

#define Q 19
#define T float
#define SUFFIX q19_f32
#include "lbm_ref_body.inc"
#undef T
#undef SUFFIX
#define T double
#define SUFFIX q19_f64
#include "lbm_ref_body.inc"
#undef T
#undef SUFFIX
#undef Q

#define Q 27
#define T float
#define SUFFIX q27_f32
#include "lbm_ref_body.inc"
#undef T
#undef SUFFIX
#define T double
#define SUFFIX q27_f64
#include "lbm_ref_body.inc"
#undef T
#undef SUFFIX
#undef Q

/* type-erased entry: is_f64 selects the scalar type (compute == store precision) */
int lbmref_step(const void* src, void* dst, const uint8_t* bc_mask, const uint8_t* missing, int nx, int ny, int nz,
                const lattice_t* L, int n_bc, const int* bc_ids, const int* bc_kinds, const double* bc_values, double omega,
                int collision, int is_f64) {
#define CALL(sfx, TT) \
  return lbmref_step_##sfx((const TT*)src, (TT*)dst, bc_mask, missing, nx, ny, nz, L, n_bc, bc_ids, bc_kinds, bc_values, omega, collision)
  if (L->q == 9) {
    if (is_f64) CALL(q9_f64, double);
    CALL(q9_f32, float);
  }
  if (L->q == 19) {
    if (collision != 0) return 3; /* kbc.py:65-66: D3Q19 unsupported */
    if (is_f64) CALL(q19_f64, double);
    CALL(q19_f32, float);
  }
  if (L->q == 27) {
    if (is_f64) CALL(q27_f64, double);
    CALL(q27_f32, float);
  }
  return 4;
#undef CALL
}

/* n steps with the caller's A/B swap (lid_driven_cavity_2d.py:66-67); result in a if n is even, else b */
int lbmref_run(void* a, void* b, const uint8_t* bc_mask, const uint8_t* missing, int nx, int ny, int nz, const lattice_t* L,
               int n_bc, const int* bc_ids, const int* bc_kinds, const double* bc_values, double omega, int collision,
               int is_f64, int n_steps) {
  for (int i = 0; i < n_steps; ++i) {
    void* s = (i & 1) ? b : a;
    void* d = (i & 1) ? a : b;
    int rc = lbmref_step(s, d, bc_mask, missing, nx, ny, nz, L, n_bc, bc_ids, bc_kinds, bc_values, omega, collision, is_f64);
    if (rc) return rc;
  }
  return 0;
}

int lbmref_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}
