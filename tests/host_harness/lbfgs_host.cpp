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

  // ---- compact-form direction (csrc/neo_lbfgs_dir.hpp): one entry per history slot, S'Y and Y'Y as m x m arrays
  struct SVec {
    double v[16];
  };
  std::vector<double> SYm, YYm;
  double sv_get(const SVec &a, int k) const { return a.v[k]; }
  void sv_set(SVec &a, int k, double x) const { a.v[k] = x; }
  void sv_scale(SVec &a, double s) const {
    for (int k = 0; k < m; ++k) a.v[k] *= s;
  }
  void hist_dots(const Vec &x, SVec &ps, SVec &py) const {
    for (int k = 0; k < m; ++k) {
      double a = 0, b = 0;
      for (int i = 0; i < n; ++i) {
        a += S[size_t(k) * n + i] * x[i];
        b += Y[size_t(k) * n + i] * x[i];
      }
      ps.v[k] = a;
      py.v[k] = b;
    }
  }
  void mat_put_col(int slot, const SVec &sy, const SVec &yy) {
    if (SYm.empty()) {
      SYm.assign(size_t(m) * m, 0.0);
      YYm.assign(size_t(m) * m, 0.0);
    }
    for (int i = 0; i < m; ++i) {
      SYm[size_t(i) * m + slot] = sy.v[i];
      YYm[size_t(i) * m + slot] = yy.v[i];
      YYm[size_t(slot) * m + i] = yy.v[i];
    }
  }
  void sv_init_w(SVec &w, const SVec &u, const SVec &b, double gamma) const {
    for (int i = 0; i < m; ++i) w.v[i] = SYm[size_t(i) * m + i] * u.v[i] - gamma * b.v[i];
  }
  void sv_axpy_mat(SVec &u, double coef, int which, int j, int lo, int hi, int head) const {
    for (int i = 0; i < m; ++i) {
      const int li = (i - head + m) % m;
      if (li < lo || li >= hi) continue;
      const double mv = which == 0 ? SYm[size_t(i) * m + j] : (which == 1 ? SYm[size_t(j) * m + i] : YYm[size_t(i) * m + j]);
      u.v[i] += coef * mv;
    }
  }
  void hist_combine(Vec &d, const SVec &cs, const SVec &cy, int col, int head) const {
    for (int kk = 0; kk < col; ++kk) {
      const int k = (head + kk) % m;
      for (int i = 0; i < n; ++i) d[i] += cs.v[k] * S[size_t(k) * n + i] + cy.v[k] * Y[size_t(k) * n + i];
    }
  }
  void sput(int i, double v) { scal[i] = v; }
  double sget(int i) { return scal[i]; }
  int eval(const Vec &x, double &f, Vec &g, double *costs) {
    fit(g, n);
    return cb(x.data(), n, &f, g.data(), costs, user);
  }
  void note_eval(int, int, double, double) {}
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

// the loop form with the compact-representation direction (what the fp32-sampling kernels run)
int lbfgs_host_minimize_compact(int n, double *x, double ftol, double gtol, int maxls, int maxiter,
                                int maxfun, int m, eval_cb cb, void *user, double *f_out, int *nit,
                                int *nfev, int *status, double *costs, double *costs_last) {
  HostBackend be(n, m, cb, user);
  HostBackend::Vec xv(x, x + n);
  neo::LbfgsOpts o{ftol, gtol, maxls, maxiter, maxfun, m};
  neo::LbfgsResult res;
  neo::lbfgs_minimize<HostBackend, true>(be, xv, o, res);
  memcpy(x, xv.data(), n * sizeof(double));
  *f_out = res.f;
  *nit = res.nit;
  *nfev = res.nfev;
  *status = res.status;
  memcpy(costs, res.costs, sizeof(res.costs));
  memcpy(costs_last, res.costs_last, sizeof(res.costs_last));
  return 0;
}

// ... and through the state machine
int lbfgs_host_minimize_sm_compact(int n, double *x, double ftol, double gtol, int maxls, int maxiter,
                                   int maxfun, int m, eval_cb cb, void *user, double *f_out, int *nit,
                                   int *nfev, int *status, double *costs, double *costs_last) {
  HostBackend be(n, m, cb, user);
  neo::LbfgsOpts o{ftol, gtol, maxls, maxiter, maxfun, m};
  neo::LbfgsMachine<HostBackend, true> mach(be, o);
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
}
