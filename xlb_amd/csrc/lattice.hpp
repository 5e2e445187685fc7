// Lattice descriptors as compile-time functions.
//
// The tables are RE-DERIVED here with the same constructions the reference uses
// (xlb/velocity_set/d2q9.py:18-21, d3q19.py:19-27, d3q27.py:19-29): D3Q27 enumerates
// itertools.product([0,-1,1], repeat=3) (x slowest, z fastest), D3Q19 filters that
// list to |c|_1 <= 2, D2Q9 is the reference's hand-listed order.  Internally every
// lattice is three-component; 2-D sets carry a zero leading component so that a
// (nx, ny) grid is stored as (1, nx, ny) with the fastest axis last.
#pragma once
#include <type_traits>
#include <utility>

#ifndef XLB_HD
#define XLB_HD __host__ __device__
#endif

namespace xlb {

template <int N, class F, int... I>
XLB_HD inline void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
// static_for<N>(f): f(integral_constant<0>) ... f(integral_constant<N-1>), fully unrolled
template <int N, class F>
XLB_HD inline void static_for(F&& f) {
  static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

constexpr int digit_to_c(int dgt) { return dgt == 0 ? 0 : (dgt == 1 ? -1 : 1); }  // product([0,-1,1])
constexpr int iabs(int v) { return v < 0 ? -v : v; }

struct D3Q27 {
  static constexpr int D = 3, Q = 27, ID = 2;
  static constexpr int c(int a, int l) {
    return a == 0 ? digit_to_c(l / 9) : (a == 1 ? digit_to_c((l / 3) % 3) : digit_to_c(l % 3));
  }
  static constexpr double w(int l) {
    int n = iabs(c(0, l)) + iabs(c(1, l)) + iabs(c(2, l));
    return n == 0 ? 8.0 / 27.0 : (n == 1 ? 2.0 / 27.0 : (n == 2 ? 1.0 / 54.0 : 1.0 / 216.0));
  }
};

struct D3Q19 {
  static constexpr int D = 3, Q = 19, ID = 1;
  // index into the D3Q27 enumeration of the l-th direction with |c|_1 <= 2
  static constexpr int idx27(int l) {
    int n = -1;
    for (int m = 0; m < 27; ++m) {
      if (iabs(D3Q27::c(0, m)) + iabs(D3Q27::c(1, m)) + iabs(D3Q27::c(2, m)) <= 2) ++n;
      if (n == l) return m;
    }
    return -1;
  }
  static constexpr int c(int a, int l) { return D3Q27::c(a, idx27(l)); }
  static constexpr double w(int l) {
    int n = iabs(c(0, l)) + iabs(c(1, l)) + iabs(c(2, l));
    return n == 0 ? 1.0 / 3.0 : (n == 1 ? 1.0 / 18.0 : 1.0 / 36.0);
  }
};

struct D2Q9 {
  static constexpr int D = 2, Q = 9, ID = 0;
  // reference order: cx = [0,0,0,1,-1,1,-1,1,-1], cy = [0,1,-1,0,1,-1,0,1,-1]
  static constexpr int cx2(int l) {
    constexpr int t[9] = {0, 0, 0, 1, -1, 1, -1, 1, -1};
    return t[l];
  }
  static constexpr int cy2(int l) {
    constexpr int t[9] = {0, 1, -1, 0, 1, -1, 0, 1, -1};
    return t[l];
  }
  // internal 3-component form: (0, cx, cy)
  static constexpr int c(int a, int l) { return a == 0 ? 0 : (a == 1 ? cx2(l) : cy2(l)); }
  static constexpr double w(int l) {
    int n = iabs(cx2(l)) + iabs(cy2(l));
    return n == 0 ? 4.0 / 9.0 : (n == 1 ? 1.0 / 9.0 : 1.0 / 36.0);
  }
};

// opposite direction: first index with c == -c_l (velocity_set.py:182-195)
template <class L>
constexpr int opp(int l) {
  for (int m = 0; m < L::Q; ++m)
    if (L::c(0, m) == -L::c(0, l) && L::c(1, m) == -L::c(1, l) && L::c(2, m) == -L::c(2, l)) return m;
  return -1;
}

// user-visible component a (0..D-1) of direction l
template <class L>
constexpr int cu(int a, int l) {
  return L::c(a + (3 - L::D), l);
}

// second-moment products in the reference order (velocity_set.py:155-180):
// 3-D: (xx, xy, xz, yy, yz, zz); 2-D: (xx, xy, yy)
template <class L>
constexpr int n_pi() {
  return L::D * (L::D + 1) / 2;
}
template <class L>
constexpr int cc(int l, int k) {
  int n = 0;
  for (int a = 0; a < L::D; ++a)
    for (int b = a; b < L::D; ++b) {
      if (n == k) return cu<L>(a, l) * cu<L>(b, l);
      ++n;
    }
  return 0;
}

}  // namespace xlb
