// neo_lbfgs_sm.hpp -- the L-BFGS-B run of neo_lbfgs.hpp (lbfgs_minimize) as a resumable state machine, in the
// reverse-communication style of the original setulb: the caller evaluates the objective, the machine does
// everything between two evaluations.
//
//     LbfgsMachine<Backend> m(be, opts);
//     m.begin();                                   // be's x holds the start point
//     while (m.need_eval()) {
//       est = be.eval(m.x, m.f, m.g, m.costs());   // one cost + gradient evaluation
//       m.advance(est);
//     }
//     m.result(res);
//
// Same arithmetic, same order, same decisions as lbfgs_minimize -- tests/test_lbfgs_host.py runs both on the
// SciPy traces and asks for identical bits.  What it is for: several trajectories sharing one wavefront (lane
// groups) stay in lock step on the expensive part, the evaluation, whatever each of them does in between:
// every round is "evaluate, then advance" for all of them.
#pragma once
#include "neo_lbfgs.hpp"
#ifndef NEO_SM_STAMP  // timing experiments only (NEO_STAMPS builds of the device kernels define it)
#define NEO_SM_STAMP(i)
#endif
#ifndef NEO_MARK
#define NEO_MARK(name)
#endif

namespace neo {

// the type of the run's scalars (f, the step, g . d, theta, the line-search state): double unless the backend names
// another -- the all-fp32 device kernels run them in fp32 (Backend::Scalar = float, neo_kernels.hpp)
template <class B, class = void>
struct backend_scalar { using type = double; };
template <class B>
struct backend_scalar<B, std::void_t<typename B::Scalar>> { using type = typename B::Scalar; };

template <class Backend>
struct LbfgsMachine {
  using Vec = typename Backend::Vec;
  using S = typename backend_scalar<Backend>::type;
  enum : int { PH_FIRST = 0, PH_LS = 1, PH_DONE = 2 };
  enum : int { DO_START_ITER = 0, DO_LS_CONT = 1, DO_SUCCESS = 2, DO_FAILED = 3, DO_RETURN = 4 };

  Backend &be;
  LbfgsOpts o;
  Vec x, g, t, r, d, tmp, tmp2;
  S f = 0, fold = 0, stp = 0, gd = 0, gdold = 0, theta = 1, stp_evaluated = -1;
  int nfev = 0, nit = 0, iter = 0, col = 0, head = 0, task = LS_START, ifun = 0;
  int phase = PH_FIRST, status = -1;

  NEO_HD LbfgsMachine(Backend &b, const LbfgsOpts &opts) : be(b), o(opts) {}

  NEO_HD double *costs() { return be.cost_store(); }
  NEO_HD double *cur() { return be.cost_store() + 4; }
  NEO_HD double *old() { return be.cost_store() + 8; }
  NEO_HD bool need_eval() const { return phase != PH_DONE; }

  NEO_HD void begin() {
    phase = PH_FIRST;
    status = -1;
    nfev = nit = iter = col = head = 0;
    theta = 1;
  }

  NEO_HD void finish(int st) {
    status = st;
    phase = PH_DONE;
  }

  NEO_HD void result(LbfgsResult &res) {
    res.f = f;
    res.nit = nit;
    res.nfev = nfev;
    res.status = status;
    for (int k = 0; k < 4; ++k) {
      res.costs[k] = cur()[k];
      res.costs_last[k] = costs()[k];
    }
  }

  // after be.eval(x, f, g, costs()) returned `est`
  NEO_HD void advance(int est) {
    NEO_MARK("advance_begin");
    const S epsmch = S(2.220446049250313e-16);  // (the fp64 value in the all-fp32 runs too: the same pairs skipped)
    const S big = S(1.0e10);
    int next;
    if (phase == PH_FIRST) {
      nfev++;
      be.note_eval(nfev, 0, 0.0, (double)f, x, g);
      for (int k = 0; k < 4; ++k) cur()[k] = costs()[k];
      if (est != 0) return finish(est);
      if (!(f - f == S(0))) return finish(TERM_NONFINITE);
      if (be.amax(g) <= o.gtol) return finish(TERM_CONVERGED_GRAD);
      phase = PH_LS;
      next = DO_START_ITER;
    } else {
      // an evaluation inside the line search
      nfev++;
      be.note_eval(nfev, iter, (double)stp, (double)f, x, g);
      if (est != 0) {
        for (int k = 0; k < 4; ++k) cur()[k] = costs()[k];
        return finish(est);
      }
      if (!(f - f == S(0))) {
        for (int k = 0; k < 4; ++k) cur()[k] = costs()[k];
        return finish(TERM_NONFINITE);
      }
      gd = (S)be.dot(g, d);
      task = LS_FG;
      next = DO_LS_CONT;
    }

    while (next != DO_RETURN) {
      if (next == DO_START_ITER) {
        // ---- search direction
        NEO_SM_STAMP(0);
        NEO_MARK("dir_begin");
        lbfgs_direction(be, g, d, tmp, tmp2, col, head, o.m, theta);
        NEO_SM_STAMP(1);
        NEO_MARK("dir_end");
        // ---- line search set-up (lnsrlb)
        be.copy(t, x);
        be.copy(r, g);
        fold = f;
        for (int k = 0; k < 4; ++k) old()[k] = cur()[k];
        // (lnsrlb forms |d| every iteration but, without bounds, uses it only for the first step)
        stp = 1;
        if (iter == 0) stp = be.uni(fmin(S(1) / sqrt((S)be.dot(d, d)), big));
        gd = (S)be.dot(g, d);
        gdold = gd;
        if (gd >= S(0)) {
          next = DO_FAILED;  // "ascent direction in projection": info = -4
        } else {
          auto &L = be.ls();
          L.ftol = S(1.0e-3);
          L.gtol = S(0.9);
          L.xtol = S(0.1);
          L.stpmin = S(0);
          L.stpmax = big;
          task = LS_START;
          ifun = 0;
          stp_evaluated = -1;
          next = DO_LS_CONT;
        }
      } else if (next == DO_LS_CONT) {
        next = DO_RETURN;
        for (;;) {
          NEO_MARK("dcsrch_begin");
          task = dcsrch(be.ls(), f, gd, stp, task);
          NEO_MARK("dcsrch_end");
          stp = be.uni(stp);
          if (task == LS_CONVERGENCE || task == LS_WARNING) {
            next = DO_SUCCESS;
            break;
          }
          // LS_FG (an LS_ERROR is treated like FG by lnsrlb's csave test)
          ifun++;
          if (ifun - 1 >= o.maxls) {
            next = DO_FAILED;
            break;
          }
          if (stp == stp_evaluated) {
            // the point just evaluated again: served from SciPy's cache there, not evaluated here
            task = LS_FG;
            continue;
          }
          stp_evaluated = stp;
          be.lincomb(x, t, stp, d);
          break;  // next == DO_RETURN: the caller evaluates x
        }
      } else if (next == DO_FAILED) {
        be.copy(x, t);
        be.copy(g, r);
        f = fold;
        for (int k = 0; k < 4; ++k) cur()[k] = old()[k];
        if (col == 0) return finish(TERM_ABNORMAL);
        col = 0;
        head = 0;
        theta = 1;
        next = DO_START_ITER;  // RESTART_FROM_LNSRCH: same iteration, steepest descent, stp = 1
      } else {  // DO_SUCCESS: NEW_X
        NEO_MARK("newx_begin");
        iter++;
        nit++;
        for (int k = 0; k < 4; ++k) cur()[k] = costs()[k];
        if (be.amax(g) <= o.gtol) return finish(TERM_CONVERGED_GRAD);
        {
          const S ddum = fmax(fmax(fabs(fold), fabs(f)), S(1));
          if ((fold - f) <= (S)o.ftol * ddum) return finish(TERM_CONVERGED_F);
        }
        if (nit >= o.maxiter || nfev > o.maxfun) return finish(TERM_MAXITER);
        // ---- update the limited-memory pairs (mainlb + matupd)
        be.lincomb(r, g, -1.0, r);  // r = g - g_old = y
        S dr, ddum;
        if (stp == S(1)) {
          dr = gd - gdold;
          ddum = -gdold;
        } else {
          dr = (gd - gdold) * stp;
          be.scale(d, stp);  // d = s
          ddum = -gdold * stp;
        }
        next = DO_START_ITER;
        if (dr <= epsmch * ddum) continue;  // skip the update, keep the old memory
        const S rr = (S)be.dot(r, r);
        int slot;
        if (col < o.m) {
          slot = (head + col) % o.m;
          col++;
        } else {
          slot = head;
          head = (head + 1) % o.m;
        }
        be.hist_put(slot, d, r);
        be.sput(slot, S(1) / dr);
        theta = be.uni(rr / dr);
        NEO_MARK("newx_end");
      }
    }
  }
};

}  // namespace neo
