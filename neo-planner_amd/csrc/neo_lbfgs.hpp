// neo_lbfgs.hpp -- control flow of L-BFGS-B 3.0 for the unconstrained case, as driven by
// scipy.optimize.minimize(method='L-BFGS-B', bounds=None) from plan_once()
// (expert_planner.py:213-225: maxcor 10, maxls 20, tol 1e-4 -> ftol = gtol = 1e-4).
//
// SciPy's numerical core (_lbfgsb setulb/mainlb/lnsrlb) is a third-party dependency whose
// source is not under /root/reference; this restates the published algorithm
// (Byrd, Lu, Nocedal, Zhu 1995; Morales, Nocedal 2011 "L-BFGS-B 3.0") for nbd = 0:
//   * no Cauchy point / subspace step: with every variable free the subspace minimiser
//     is -B^{-1} g, B = theta*I - W M W^T; the same vector comes out of the two-loop
//     recursion over the stored (s, y) pairs with H0 = I / theta, theta = y'y / s'y;
//   * first step length 1/||d|| at iteration 0, else 1 (lnsrlb);
//   * More'-Thuente search, at most `maxls` evaluations; on failure x is restored and,
//     if the memory is not empty, it is dropped and the iteration restarts from -g,
//     otherwise the run ends ("ABNORMAL_TERMINATION_IN_LNSRCH");
//   * stop on max|g| <= gtol or (f_old - f) <= ftol * max(|f_old|, |f|, 1);
//   * the pair is skipped when s'y <= epsmch * (-g_old'd * stp).
// Pinned by tests/test_lbfgs_host.py against SciPy 1.15.3 iterate traces (tests/golden g3_*).
//
// The template runs on whatever `Backend` provides: the HIP kernels give it
// wavefront-cooperative vectors (neo_kernels.hpp), the host test harness plain arrays.
#pragma once
#include "neo_linesearch.hpp"
#include "neo_lbfgs_dir.hpp"

namespace neo {

struct LbfgsOpts {
  double ftol, gtol;
  int maxls, maxiter, maxfun, m;
};

// termination codes = NEO_TRAJ_* of include/neo_planner.h
enum : int {
  TERM_CONVERGED_GRAD = 0,
  TERM_CONVERGED_F = 1,
  TERM_ABNORMAL = 2,
  TERM_MAXITER = 3,
  TERM_NUMERIC_RANGE = 4,
  TERM_NONFINITE = 5
};

struct LbfgsResult {
  double f;
  int nit, nfev, status;
  double costs[4];       // unweighted cost terms at the returned x
  double costs_last[4];  // ... at the last evaluated x (expert_planner.py:233)
};

// Backend concept:
//   using Vec;                                   a length-n vector
//   double dot(const Vec&, const Vec&); double amax(const Vec&);
//   void copy(Vec& dst, const Vec& src); void neg(Vec& dst, const Vec& src);
//   void axpy(double a, const Vec& x, Vec& y);            y += a x
//   void lincomb(Vec& out, const Vec& a, double s, const Vec& b);   out = a + s b
//   void scale(Vec& v, double s);
//   void hist_put(int slot, const Vec& s, const Vec& y); void hist_get_s(int slot, Vec&); hist_get_y
//   void sput(int idx, double v); double sget(int idx);   2*m wave-uniform scalars
//   LineSearch& ls(); double* cost_store();   storage for the search state and 3 x 4 cost terms
//       (on the GPU both live in LDS: wave-uniform data that would otherwise pin ~50 VGPRs)
//   int  eval(const Vec& x, double& f, Vec& g, double* costs4);   0 = ok
//   void note_eval(int nfev, int iter, double stp, double f, const Vec& x, const Vec& g);   diagnostics hook after every
//       counted evaluation: the point just evaluated, its value and gradient (a no-op in the product unless a trace
//       buffer was given: neo_optimize_trace / neo_optimize_trace_xg)
template <class Backend>
NEO_HD void lbfgs_minimize(Backend &be, typename Backend::Vec &x, const LbfgsOpts &o,
                           LbfgsResult &res) {
  using Vec = typename Backend::Vec;
  const double epsmch = 2.220446049250313e-16;
  const double big = 1.0e10;
  Vec g, t, r, d, tmp, tmp2;
  double f = 0.0;
  double *costs = be.cost_store(), *cur = costs + 4, *old = costs + 8;
  int nfev = 0, nit = 0;
  int col = 0, head = 0;  // stored pairs, ring start (oldest)
  double theta = 1.0;
  int iter = 0;  // L-BFGS-B's `iter`: accepted iterations (also gates the first-step rule)

  auto finish = [&](int status) {
    res.f = f;
    res.nit = nit;
    res.nfev = nfev;
    res.status = status;
    for (int k = 0; k < 4; ++k) {
      res.costs[k] = cur[k];
      res.costs_last[k] = costs[k];
    }
  };

  int est = be.eval(x, f, g, costs);
  nfev++;
  be.note_eval(nfev, 0, 0.0, f, x, g);
  for (int k = 0; k < 4; ++k) cur[k] = costs[k];
  if (est != 0) return finish(est);
  if (!(f - f == 0.0)) return finish(TERM_NONFINITE);
  if (be.amax(g) <= o.gtol) return finish(TERM_CONVERGED_GRAD);

  for (;;) {
    // ---- search direction
    lbfgs_direction(be, g, d, tmp, tmp2, col, head, o.m, theta);

    // ---- line search (lnsrlb)
    be.copy(t, x);
    be.copy(r, g);
    const double fold = f;
    for (int k = 0; k < 4; ++k) old[k] = cur[k];
    // (lnsrlb forms |d| every iteration but, without bounds, uses it only for the first step)
    double stp = 1.0;
    if (iter == 0) stp = fmin(1.0 / sqrt(be.dot(d, d)), big);
    double gd = be.dot(g, d);
    const double gdold = gd;
    bool failed = false;
    int term = -1;
    if (gd >= 0.0) {
      failed = true;  // "ascent direction in projection": info = -4
    } else {
      LineSearch &L = be.ls();
      L.ftol = 1.0e-3;
      L.gtol = 0.9;
      L.xtol = 0.1;
      L.stpmin = 0.0;
      L.stpmax = big;
      int task = LS_START;
      int ifun = 0;
      double stp_evaluated = -1.0;
      for (;;) {
        task = dcsrch(L, f, gd, stp, task);
        if (task == LS_CONVERGENCE || task == LS_WARNING) break;
        // LS_FG (an LS_ERROR is treated like FG by lnsrlb's csave test)
        ifun++;
        if (ifun - 1 >= o.maxls) {
          failed = true;
          break;
        }
        if (stp == stp_evaluated) {
          // dcsrch fell back to its best step: x = t + stp*d is bit-identical to the point
          // just evaluated.  SciPy's ScalarFunction serves f and g from its cache in that case
          // and does not count an evaluation; neither do we.
          task = LS_FG;
          continue;
        }
        stp_evaluated = stp;
        be.lincomb(x, t, stp, d);
        est = be.eval(x, f, g, costs);
        nfev++;
        be.note_eval(nfev, iter, stp, f, x, g);
        if (est != 0) {
          term = est;
          break;
        }
        if (!(f - f == 0.0)) {
          term = TERM_NONFINITE;
          break;
        }
        gd = be.dot(g, d);
        task = LS_FG;
      }
    }
    if (term >= 0) {
      for (int k = 0; k < 4; ++k) cur[k] = costs[k];
      return finish(term);
    }
    if (failed) {
      be.copy(x, t);
      be.copy(g, r);
      f = fold;
      for (int k = 0; k < 4; ++k) cur[k] = old[k];
      if (col == 0) return finish(TERM_ABNORMAL);
      col = 0;
      head = 0;
      theta = 1.0;
      continue;  // RESTART_FROM_LNSRCH: same iteration, steepest descent, stp = 1
    }

    // ---- NEW_X
    iter++;
    nit++;
    for (int k = 0; k < 4; ++k) cur[k] = costs[k];
    if (be.amax(g) <= o.gtol) return finish(TERM_CONVERGED_GRAD);
    {
      const double ddum = fmax(fmax(fabs(fold), fabs(f)), 1.0);
      if ((fold - f) <= o.ftol * ddum) return finish(TERM_CONVERGED_F);
    }
    if (nit >= o.maxiter || nfev > o.maxfun) return finish(TERM_MAXITER);

    // ---- update the limited-memory pairs (mainlb + matupd)
    be.lincomb(r, g, -1.0, r);  // r = g - g_old = y
    double dr, ddum;
    if (stp == 1.0) {
      dr = gd - gdold;
      ddum = -gdold;
    } else {
      dr = (gd - gdold) * stp;
      be.scale(d, stp);  // d = s
      ddum = -gdold * stp;
    }
    if (dr <= epsmch * ddum) continue;  // skip the update, keep the old memory
    const double rr = be.dot(r, r);
    int slot;
    if (col < o.m) {
      slot = (head + col) % o.m;
      col++;
    } else {
      slot = head;
      head = (head + 1) % o.m;
    }
    be.hist_put(slot, d, r);
    be.sput(slot, 1.0 / dr);
    theta = rr / dr;
  }
}

}  // namespace neo
