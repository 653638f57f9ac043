// TEST HARNESS ONLY (built by tests/test_lbfgs_host.py with g++, never part of the
// shipped library): runs the product's L-BFGS control flow (csrc/neo_lbfgs.hpp,
// csrc/neo_linesearch.hpp) on plain host arrays with a Python callback as the objective,
// so that it can be compared against SciPy's L-BFGS-B iterate by iterate without a GPU.
#include <cstring>
#include <vector>

#include "neo_lbfgs.hpp"
#include "neo_lbfgs_sm.hpp"

extern "C" {
typedef int (*eval_cb)(const double *x, int n, double *f, double *g, double *costs, void *user);
}

namespace {
struct HostBackend {
  using Vec = std::vector<double>;
  int n, m;
  eval_cb cb;
  void *user;
  std::vector<double> S, Y, scal;
  neo::LineSearch lsearch;
  double cost12[12];
  neo::LineSearch &ls() { return lsearch; }
  double *cost_store() { return cost12; }
  HostBackend(int n_, int m_, eval_cb cb_, void *u)
      : n(n_), m(m_), cb(cb_), user(u), S(size_t(n_) * m_), Y(size_t(n_) * m_), scal(2 * m_) {}
  static void fit(Vec &v, int n) {
    if ((int)v.size() != n) v.assign(n, 0.0);
  }
  double dot(const Vec &a, const Vec &b) {
    double s = 0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
  }
  double amax(const Vec &a) {
    double s = 0;
    for (int i = 0; i < n; ++i) s = fmax(s, fabs(a[i]));
    return s;
  }
  void copy(Vec &d, const Vec &s) { d = s; }
  void neg(Vec &d, const Vec &s) {
    fit(d, n);
    for (int i = 0; i < n; ++i) d[i] = -s[i];
  }
  void axpy(double a, const Vec &x, Vec &y) {
    for (int i = 0; i < n; ++i) y[i] += a * x[i];
  }
  void lincomb(Vec &out, const Vec &a, double s, const Vec &b) {
    fit(out, n);
    for (int i = 0; i < n; ++i) out[i] = a[i] + s * b[i];
  }
  void scale(Vec &v, double s) {
    for (int i = 0; i < n; ++i) v[i] *= s;
  }
  void hist_put(int slot, const Vec &s, const Vec &y) {
    memcpy(&S[size_t(slot) * n], s.data(), n * sizeof(double));
    memcpy(&Y[size_t(slot) * n], y.data(), n * sizeof(double));
  }
  void hist_get_s(int slot, Vec &v) { v.assign(&S[size_t(slot) * n], &S[size_t(slot) * n] + n); }
  void hist_get_y(int slot, Vec &v) { v.assign(&Y[size_t(slot) * n], &Y[size_t(slot) * n] + n); }
  double uni(double v) const { return v; }
  double sdiff(int i, double b) { return sget(i) - b; }
  double rho_dot(int slot, const Vec &a, const Vec &b) { return sget(slot) * dot(a, b); }
  void hist_get_sy(int slot, Vec &s, Vec &y) {
    hist_get_s(slot, s);
    hist_get_y(slot, y);
  }

  void sput(int i, double v) { scal[i] = v; }
  double sget(int i) { return scal[i]; }
  int eval(const Vec &x, double &f, Vec &g, double *costs) {
    fit(g, n);
    return cb(x.data(), n, &f, g.data(), costs, user);
  }
  void note_eval(int, int, double, double, const Vec &, const Vec &) {}
};
}  // namespace

extern "C" {

int lbfgs_host_minimize(int n, double *x, double ftol, double gtol, int maxls, int maxiter,
                        int maxfun, int m, eval_cb cb, void *user, double *f_out, int *nit,
                        int *nfev, int *status, double *costs, double *costs_last) {
  HostBackend be(n, m, cb, user);
  HostBackend::Vec xv(x, x + n);
  neo::LbfgsOpts o{ftol, gtol, maxls, maxiter, maxfun, m};
  neo::LbfgsResult res;
  neo::lbfgs_minimize(be, xv, o, res);
  memcpy(x, xv.data(), n * sizeof(double));
  *f_out = res.f;
  *nit = res.nit;
  *nfev = res.nfev;
  *status = res.status;
  memcpy(costs, res.costs, sizeof(res.costs));
  memcpy(costs_last, res.costs_last, sizeof(res.costs_last));
  return 0;
}

// the same run through the resumable state machine (csrc/neo_lbfgs_sm.hpp): must give identical bits
int lbfgs_host_minimize_sm(int n, double *x, double ftol, double gtol, int maxls, int maxiter,
                           int maxfun, int m, eval_cb cb, void *user, double *f_out, int *nit,
                           int *nfev, int *status, double *costs, double *costs_last) {
  HostBackend be(n, m, cb, user);
  neo::LbfgsOpts o{ftol, gtol, maxls, maxiter, maxfun, m};
  neo::LbfgsMachine<HostBackend> mach(be, o);
  mach.x.assign(x, x + n);
  mach.begin();
  while (mach.need_eval()) {
    const int est = be.eval(mach.x, mach.f, mach.g, mach.costs());
    mach.advance(est);
  }
  neo::LbfgsResult res;
  mach.result(res);
  memcpy(x, mach.x.data(), n * sizeof(double));
  *f_out = res.f;
  *nit = res.nit;
  *nfev = res.nfev;
  *status = res.status;
  memcpy(costs, res.costs, sizeof(res.costs));
  memcpy(costs_last, res.costs_last, sizeof(res.costs_last));
  return 0;
}

// REPLAY of a recorded run (tests/test_gpu_replay.py).  A device run recorded, per counted evaluation k < E, the point
// it evaluated xr[k], the value fr[k], the gradient gr[k] and the cost terms cr[k] (neo_optimize_trace +
// neo_optimize_trace_xg).  This runs the product's L-BFGS-B control flow (the state machine of csrc/neo_lbfgs_sm.hpp)
// in fp64 on the host with those recorded values as its objective: at every evaluation the host's own trial point is
// compared with the device's (x_dev[k] = max |x_host - x_dev| / max(1, max |x_dev|)), then REPLACED by it, so that
// every step is judged on the device's own history and errors do not accumulate (`resync_min_rel` > 0: the host's
// direction is re-derived from the device's trial point as well, see below).  Outputs per evaluation: the host's
// line-search step and iteration counter (to be laid beside the device's trace); per run: nit / nfev / status as the
// host decides them, and `overrun` = 1 when the host asks for an evaluation the device never made.
// last_est: the status the device's last evaluation returned (4 = NUMERIC_RANGE ends the run there).
int lbfgs_host_replay(int n, int E, const double *xr, const double *fr, const double *gr, const double *cr, int last_est,
                      double ftol, double gtol, int maxls, int maxiter, int maxfun, int m, double resync_min_rel, double *x_dev,
                      double *stp_host, int *iter_host, int *nit, int *nfev, int *status, int *overrun) {
  HostBackend be(n, m, nullptr, nullptr);
  neo::LbfgsOpts o{ftol, gtol, maxls, maxiter, maxfun, m};
  neo::LbfgsMachine<HostBackend> mach(be, o);
  mach.x.assign(xr, xr + n);
  mach.g.assign(n, 0.0);
  mach.begin();
  int k = 0;
  *overrun = 0;
  while (mach.need_eval()) {
    if (k >= E) {
      *overrun = 1;
      break;
    }
    const double *xk = xr + (size_t)k * n;
    double dmax = 0.0, xmax = 1.0;
    for (int i = 0; i < n; ++i) {
      dmax = fmax(dmax, fabs(mach.x[i] - xk[i]));
      xmax = fmax(xmax, fabs(xk[i]));
    }
    x_dev[k] = dmax / xmax;
    stp_host[k] = k == 0 ? 0.0 : mach.stp;
    iter_host[k] = mach.iter;
    if (k > 0 && resync_min_rel > 0.0) {
      // the direction the device actually moved along, from its own trial point: d = (x_k - t) / stp.  The pair stored
      // after this line search is then the device's true displacement (s = stp d = x_k - t), so the host's memory is a
      // function of the device's data alone and a rounding difference in one direction cannot feed the next ones.
      // Only while the displacement is numerically there: in a collapsing line search (steps of 1e-15) x_k - t is a few
      // units in the last place of x and says nothing about d.
      double disp = 0.0;
      for (int i = 0; i < n; ++i) disp = fmax(disp, fabs(xk[i] - mach.t[i]));
      if (disp >= resync_min_rel * xmax) {
        const double stp_h = mach.stp;
        for (int i = 0; i < n; ++i) mach.d[i] = (xk[i] - mach.t[i]) / stp_h;
      }
    }
    mach.x.assign(xk, xk + n);
    mach.f = fr[k];
    mach.g.assign(gr + (size_t)k * n, gr + (size_t)(k + 1) * n);
    for (int q = 0; q < 4; ++q) mach.costs()[q] = cr ? cr[(size_t)k * 4 + q] : 0.0;
    mach.advance(k == E - 1 ? last_est : 0);
    ++k;
  }
  neo::LbfgsResult res;
  mach.result(res);
  *nit = res.nit;
  *nfev = res.nfev;
  *status = mach.need_eval() ? -1 : res.status;
  return k;
}

// reverse-communication line search, state kept in a caller-provided 20-double blob
int dcsrch_host(double *state, double f, double g, double *stp, int task, double ftol, double gtol,
                double xtol, double stpmin, double stpmax) {
  static_assert(sizeof(neo::LineSearch) <= 20 * sizeof(double), "state blob too small");
  neo::LineSearch *L = reinterpret_cast<neo::LineSearch *>(state);
  if (task == neo::LS_START) {
    L->ftol = ftol;
    L->gtol = gtol;
    L->xtol = xtol;
    L->stpmin = stpmin;
    L->stpmax = stpmax;
  }
  return neo::dcsrch(*L, f, g, *stp, task);
}

// the same search on fp32 scalars (LineSearchT<float>: what the all-fp32 device kernels instantiate)
int dcsrch_host_f32(float *state, float f, float g, float *stp, int task, float ftol, float gtol, float xtol,
                    float stpmin, float stpmax) {
  static_assert(sizeof(neo::LineSearchT<float>) <= 20 * sizeof(float), "state blob too small");
  neo::LineSearchT<float> *L = reinterpret_cast<neo::LineSearchT<float> *>(state);
  if (task == neo::LS_START) {
    L->ftol = ftol;
    L->gtol = gtol;
    L->xtol = xtol;
    L->stpmin = stpmin;
    L->stpmax = stpmax;
  }
  return neo::dcsrch(*L, f, g, *stp, task);
}
}
