// neo_lbfgs_dir.hpp -- the L-BFGS search direction d = -H g from the stored (s, y) pairs, shared by both forms of the
// run (neo_lbfgs.hpp, neo_lbfgs_sm.hpp).
//
// Two mathematically equal ways to apply H (they agree to round-off, 3e-16 relative; tests/test_lbfgs_host.py runs both
// against SciPy's recorded iterates), chosen by the template parameter COMPACT:
//
//   * false: the two-loop recursion (Nocedal 1980): 2*col dot products, each depending on the one before -- on the GPU
//     2*col dependent LDS reads + wavefront reductions per iteration (470 cycles a step, 3.8 us of a 21.7 us evaluation
//     at cfg2).  This is the form pinned to SciPy: it follows 37 of the 39 recorded reference runs evaluation by
//     evaluation, and every kernel with fp64 sampling (the parity mode, the reference-shaped MinJerkPlanner) uses it.
//   * true: the compact representation (Byrd, Nocedal, Schnabel 1994, eq. 3.1); a build option for the fp32-sampling
//     kernels (-DNEO_COMPACT_DIRECTION=1, off by default: measured slower on MI355X, neo_kernels.hpp optimize_kernel)
//     and exercised on the host (it follows 36 of the 39 recorded runs, the round-off difference tips one more):
//         H = gamma I + [S  gamma Y] [ R^-T (D + gamma Y'Y) R^-1    -R^-T ] [ S'       ]
//                                    [ -R^-1                          0    ] [ gamma Y' ]
//     with R = upper triangle of S'Y (pairs in chronological order), D = its diagonal, gamma = 1 / theta.
//     All 2*col dot products with g are INDEPENDENT (one batched pass over the history), the rest is O(col^2) work on
//     col-vectors (two triangular solves and one product with Y'Y).  S'Y and Y'Y are kept up to date with col + col
//     more independent dot products when a pair is stored.
//
// Backend concept (on top of neo_lbfgs.hpp's):
//   using SVec;                                  a vector with one entry per history slot (m entries)
//   double sv_get(const SVec&, int slot);  void sv_set(SVec&, int slot, double v);
//   void hist_get_sy(int slot, Vec& s, Vec& y);          the pair of a slot in one call
//   void hist_dots(const Vec& v, SVec& ps, SVec& py);     ps[k] = s_k . v, py[k] = y_k . v for every slot k
//   void mat_put_col(int slot, const SVec& sy, const SVec& yy);   SY[i][slot] = sy[i]; YY[i][slot] = YY[slot][i] = yy[i]
//   void sv_init_w(SVec& w, const SVec& u, const SVec& b, double gamma);      w_i = SY[i][i] u_i - gamma b_i
//   void sv_axpy_mat(SVec& u, double coef, int which, int j, int lo, int hi, int head);
//        u_i += coef * M_i for every slot i whose chronological index (i - head) mod m lies in [lo, hi), with
//        M_i = SY[i][j] (which = 0), SY[j][i] (which = 1) or YY[i][j] (which = 2)
//   void hist_combine(Vec& d, const SVec& cs, const SVec& cy, int col, int head);   d += sum_k cs[k] s_k + cy[k] y_k
//   sget(slot) = rho_slot = 1 / (s_slot . y_slot)   (as for the two-loop recursion)
#pragma once

namespace neo {

// d = -H g.  `scratch` vectors are the caller's (tmp, tmp2 of the run).
template <bool COMPACT, class Backend>
NEO_HD void lbfgs_direction(Backend &be, const typename Backend::Vec &g, typename Backend::Vec &d,
                            typename Backend::Vec &tmp, typename Backend::Vec &tmp2, int col, int head, int m,
                            double theta) {
  if (col == 0) {
    be.neg(d, g);
    return;
  }
  if constexpr (!COMPACT) {
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
  } else {
  (void)tmp;
  (void)tmp2;
  using SVec = typename Backend::SVec;
  const double gamma = 1.0 / theta;
  SVec a, b, u, w;
  be.hist_dots(g, a, b);  // a = S'g, b = Y'g: 2*col independent dot products
  // u = R^-1 a: back substitution over the columns, newest pair first
  u = a;
  for (int jj = col - 1; jj >= 0; --jj) {
    const int j = (head + jj) % m;
    const double uj = be.sv_get(u, j) * be.sget(j);  // / R_jj
    be.sv_set(u, j, uj);
    be.sv_axpy_mat(u, -uj, 0, j, 0, jj, head);  // u_i -= SY[i][j] u_j for the older pairs i
  }
  // w = (D + gamma Y'Y) u - gamma b
  be.sv_init_w(w, u, b, gamma);
  for (int jj = 0; jj < col; ++jj) {
    const int j = (head + jj) % m;
    be.sv_axpy_mat(w, gamma * be.sv_get(u, j), 2, j, 0, col, head);
  }
  // t = R^-T w: forward substitution, oldest pair first (w is overwritten)
  for (int jj = 0; jj < col; ++jj) {
    const int j = (head + jj) % m;
    const double tj = be.sv_get(w, j) * be.sget(j);
    be.sv_set(w, j, tj);
    be.sv_axpy_mat(w, -tj, 1, j, jj + 1, col, head);  // t_i -= SY[j][i] t_j for the newer pairs i
  }
  // d = -(gamma g + S t - gamma Y u)
  be.copy(d, g);
  be.scale(d, -gamma);
  be.sv_scale(w, -1.0);
  be.sv_scale(u, gamma);
  be.hist_combine(d, w, u, col, head);
  }
}

// after hist_put(slot, s, y): bring S'Y and Y'Y up to date (compact form only)
template <bool COMPACT, class Backend>
NEO_HD void lbfgs_pair_stored(Backend &be, int slot, const typename Backend::Vec &y) {
  if constexpr (COMPACT) {
    typename Backend::SVec sy, yy;
    be.hist_dots(y, sy, yy);  // s_i . y_new (column `slot` of S'Y), y_i . y_new
    be.mat_put_col(slot, sy, yy);
  } else {
    (void)be;
    (void)slot;
    (void)y;
  }
}

}  // namespace neo
