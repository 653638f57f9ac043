// neo_lbfgs_dir.hpp -- the L-BFGS search direction d = -H g from the stored (s, y) pairs, shared by both forms of the
// run (neo_lbfgs.hpp, neo_lbfgs_sm.hpp): the two-loop recursion (Nocedal 1980) with H0 = I / theta.  2*col dot
// products, each depending on the one before -- on the GPU 2*col dependent LDS reads + wavefront reductions per
// iteration.  This is the form pinned to SciPy's iterates (tests/test_lbfgs_host.py).
//
// (A compact-representation variant -- Byrd, Nocedal, Schnabel 1994: 2*col independent dot products + O(col^2) scalar
//  work -- was built and measured in round 2: 24.3 us per evaluation against 21.7 us at cfg2, slower on the MI355X.  It
//  was a build option until round 3 and is gone from the tree; DESIGN.md section 5 keeps the numbers.)
#pragma once
#include <type_traits>

namespace neo {

// a backend may bring its own form of the recursion (`static constexpr bool kOwnDirection = true` and
// `direction(g, d, col, head, m, theta)`): the all-fp32 device backend walks the pairs two at a time (neo_kernels.hpp)
template <class Backend, class = void>
struct has_own_direction { static constexpr bool value = false; };
template <class Backend>
struct has_own_direction<Backend, std::enable_if_t<Backend::kOwnDirection>> { static constexpr bool value = true; };

// d = -H g.  `scratch` vectors are the caller's (tmp, tmp2 of the run).
template <class Backend>
NEO_HD void lbfgs_direction(Backend &be, const typename Backend::Vec &g, typename Backend::Vec &d,
                            typename Backend::Vec &tmp, typename Backend::Vec &tmp2, int col, int head, int m,
                            double theta) {
  if constexpr (has_own_direction<Backend>::value) {
    be.direction(g, d, col, head, m, theta);
    return;
  }
  if (col == 0) {
    be.neg(d, g);
    return;
  }
  be.copy(d, g);  // d plays q of the two-loop recursion
  for (int k = col - 1; k >= 0; --k) {
    const int slot = head + k < m ? head + k : head + k - m;  // (head + k) % m without the division
    be.hist_get_sy(slot, tmp, tmp2);  // (y issued with the read of s: its latency hides behind the reduction)
    const double a = be.rho_dot(slot, tmp, d);  // rho * s'q
    be.sput(m + slot, a);
    be.axpy(-a, tmp2, d);
  }
  be.scale(d, 1.0 / theta);
  for (int k = 0; k < col; ++k) {
    const int slot = head + k < m ? head + k : head + k - m;
    be.hist_get_sy(slot, tmp2, tmp);
    const double b = be.rho_dot(slot, tmp, d);
    be.axpy(be.sdiff(m + slot, b), tmp2, d);
  }
  be.scale(d, -1.0);
}

}  // namespace neo
