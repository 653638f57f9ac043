// neo_linesearch.hpp -- More'-Thuente line search as used by L-BFGS-B 3.0
// (subroutines dcsrch / dcstep of MINPACK-2, called from lnsrlb).
//
// The reference reaches this arithmetic through scipy.optimize.minimize(
// method='L-BFGS-B') (expert_planner.py:213-225); SciPy 1.15.3 ships a compiled
// translation of the Fortran, whose source is not in the reference tree.  This is a
// restatement of the published algorithm (J. J. More', D. J. Thuente, "Line search
// algorithms with guaranteed sufficient decrease", ACM TOMS 20, 1994; MINPACK-2
// dcsrch/dcstep), pinned by tests/test_lbfgs_host.py against SciPy's iterates.
//
// Scalar, wave-uniform code: every lane of the wavefront that owns a trajectory runs
// it redundantly on identical values, so it needs no cross-lane traffic.
#pragma once

#if defined(__HIPCC__)
#define NEO_HD __host__ __device__ __forceinline__
#else
#define NEO_HD inline
#endif

#include <math.h>

namespace neo {

enum LsTask : int { LS_START = 0, LS_FG = 1, LS_CONVERGENCE = 2, LS_WARNING = 3, LS_ERROR = 4 };

// T: the arithmetic of the search's scalars -- double everywhere but in the all-fp32 device kernels (round 5: there f and g . d
// arrive as fp32 values and the fp64 divisions and square root of dcstep sat on every evaluation's dependent chain)
template <typename T>
struct LineSearchT {
  // parameters (lnsrlb: ftol = 1e-3, gtol = 0.9, xtol = 0.1, stpmin = 0)
  T ftol, gtol, xtol, stpmin, stpmax;
  // saved state between calls
  int brackt, stage;
  T ginit, gtest, gx, gy, finit, fx, fy, stx, sty, stmin, stmax, width, width1;
};

template <typename T>
NEO_HD T ls_max3(T a, T b, T c) { return fmax(fmax(a, b), c); }

// safeguarded cubic/quadratic step; updates the interval [stx, sty] and stp
template <typename T>
struct StepIntervalT {
  T stx, fx, dx, sty, fy, dy, stp;
  int brackt;
};

// (the interval goes in and out BY VALUE: with reference parameters the device compiler kept it in a private-memory
//  array -- the only scratch use of the optimiser kernels)
template <typename T>
NEO_HD StepIntervalT<T> dcstep(const StepIntervalT<T> in, T fp, T dp, T stpmin, T stpmax) {
  T stx = in.stx, fx = in.fx, dx = in.dx, sty = in.sty, fy = in.fy, dy = in.dy, stp = in.stp;
  int brackt = in.brackt;
  T gamma, p, q, r, s, stpc, stpf, stpq, theta;
  const T sgnd = dp * (dx / fabs(dx));
  if (fp > fx) {
    // case 1: higher function value -> minimum bracketed
    theta = T(3.0) * (fx - fp) / (stp - stx) + dx + dp;
    s = ls_max3(fabs(theta), fabs(dx), fabs(dp));
    gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp < stx) gamma = -gamma;
    p = (gamma - dx) + theta;
    q = ((gamma - dx) + gamma) + dp;
    r = p / q;
    stpc = stx + r * (stp - stx);
    stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / T(2.0)) * (stp - stx);
    if (fabs(stpc - stx) < fabs(stpq - stx))
      stpf = stpc;
    else
      stpf = stpc + (stpq - stpc) / T(2.0);
    brackt = 1;
  } else if (sgnd < T(0.0)) {
    // case 2: lower function value, derivatives of opposite sign
    theta = T(3.0) * (fx - fp) / (stp - stx) + dx + dp;
    s = ls_max3(fabs(theta), fabs(dx), fabs(dp));
    gamma = s * sqrt((theta / s) * (theta / s) - (dx / s) * (dp / s));
    if (stp > stx) gamma = -gamma;
    p = (gamma - dp) + theta;
    q = ((gamma - dp) + gamma) + dx;
    r = p / q;
    stpc = stp + r * (stx - stp);
    stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (fabs(stpc - stp) > fabs(stpq - stp))
      stpf = stpc;
    else
      stpf = stpq;
    brackt = 1;
  } else if (fabs(dp) < fabs(dx)) {
    // case 3: lower function value, same sign, derivative magnitude decreases
    theta = T(3.0) * (fx - fp) / (stp - stx) + dx + dp;
    s = ls_max3(fabs(theta), fabs(dx), fabs(dp));
    gamma = s * sqrt(fmax(T(0.0), (theta / s) * (theta / s) - (dx / s) * (dp / s)));
    if (stp > stx) gamma = -gamma;
    p = (gamma - dp) + theta;
    q = (gamma + (dx - dp)) + gamma;
    r = p / q;
    if (r < T(0.0) && gamma != T(0.0))
      stpc = stp + r * (stx - stp);
    else if (stp > stx)
      stpc = stpmax;
    else
      stpc = stpmin;
    stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (brackt) {
      if (fabs(stpc - stp) < fabs(stpq - stp))
        stpf = stpc;
      else
        stpf = stpq;
      if (stp > stx)
        stpf = fmin(stp + T(0.66) * (sty - stp), stpf);
      else
        stpf = fmax(stp + T(0.66) * (sty - stp), stpf);
    } else {
      if (fabs(stpc - stp) > fabs(stpq - stp))
        stpf = stpc;
      else
        stpf = stpq;
      stpf = fmin(stpmax, stpf);
      stpf = fmax(stpmin, stpf);
    }
  } else {
    // case 4: lower function value, same sign, derivative does not decrease
    if (brackt) {
      theta = T(3.0) * (fp - fy) / (sty - stp) + dy + dp;
      s = ls_max3(fabs(theta), fabs(dy), fabs(dp));
      gamma = s * sqrt((theta / s) * (theta / s) - (dy / s) * (dp / s));
      if (stp > sty) gamma = -gamma;
      p = (gamma - dp) + theta;
      q = ((gamma - dp) + gamma) + dy;
      r = p / q;
      stpc = stp + r * (sty - stp);
      stpf = stpc;
    } else if (stp > stx) {
      stpf = stpmax;
    } else {
      stpf = stpmin;
    }
  }
  // update the interval that contains a minimiser
  if (fp > fx) {
    sty = stp;
    fy = fp;
    dy = dp;
  } else {
    if (sgnd < T(0.0)) {
      sty = stx;
      fy = fx;
      dy = dx;
    }
    stx = stp;
    fx = fp;
    dx = dp;
  }
  stp = stpf;
  StepIntervalT<T> out;
  out.stx = stx;
  out.fx = fx;
  out.dx = dx;
  out.sty = sty;
  out.fy = fy;
  out.dy = dy;
  out.stp = stp;
  out.brackt = brackt;
  return out;
}

// one reverse-communication call.  task in: LS_START or LS_FG (f, g evaluated at stp);
// task out: LS_FG (evaluate at the new stp), LS_CONVERGENCE, LS_WARNING or LS_ERROR.
template <typename T>
NEO_HD int dcsrch(LineSearchT<T> &L, T f, T g, T &stp, int task) {
  const T xtrapl = T(1.1), xtrapu = T(4.0);
  if (task == LS_START) {
    if (stp < L.stpmin || stp > L.stpmax || g >= T(0.0)) return LS_ERROR;
    L.brackt = 0;
    L.stage = 1;
    L.finit = f;
    L.ginit = g;
    L.gtest = L.ftol * L.ginit;
    L.width = L.stpmax - L.stpmin;
    L.width1 = L.width / T(0.5);
    L.stx = T(0.0);
    L.fx = L.finit;
    L.gx = L.ginit;
    L.sty = T(0.0);
    L.fy = L.finit;
    L.gy = L.ginit;
    L.stmin = T(0.0);
    L.stmax = stp + xtrapu * stp;
    return LS_FG;
  }
  const T ftest = L.finit + stp * L.gtest;
  if (L.stage == 1 && f <= ftest && g >= T(0.0)) L.stage = 2;

  int out = LS_FG;
  // later tests overwrite earlier ones, convergence overwrites warnings
  if (L.brackt && (stp <= L.stmin || stp >= L.stmax)) out = LS_WARNING;
  if (L.brackt && L.stmax - L.stmin <= L.xtol * L.stmax) out = LS_WARNING;
  if (stp == L.stpmax && f <= ftest && g <= L.gtest) out = LS_WARNING;
  if (stp == L.stpmin && (f > ftest || g >= L.gtest)) out = LS_WARNING;
  if (f <= ftest && fabs(g) <= L.gtol * (-L.ginit)) out = LS_CONVERGENCE;
  if (out != LS_FG) return out;

  // One call of dcstep on local copies (the two call sites of MINPACK-2 -- on the modified function while the
  // sufficient-decrease condition is not yet met, on the function itself otherwise -- made the device compiler keep
  // the interval in a private-memory array selected by pointer); the arithmetic of each case is unchanged.
  {
    const bool modified = L.stage == 1 && f <= L.fx && f > ftest;
    T stx = L.stx, sty = L.sty, fx = L.fx, fy = L.fy, gx = L.gx, gy = L.gy;
    T fp = f, gp = g;
    int brackt = L.brackt;
    if (modified) {
      fp = f - stp * L.gtest;
      fx = L.fx - L.stx * L.gtest;
      fy = L.fy - L.sty * L.gtest;
      gp = g - L.gtest;
      gx = L.gx - L.gtest;
      gy = L.gy - L.gtest;
    }
    StepIntervalT<T> iv;
    iv.stx = stx; iv.fx = fx; iv.dx = gx; iv.sty = sty; iv.fy = fy; iv.dy = gy; iv.stp = stp; iv.brackt = brackt;
    iv = dcstep(iv, fp, gp, L.stmin, L.stmax);
    stx = iv.stx; fx = iv.fx; gx = iv.dx; sty = iv.sty; fy = iv.fy; gy = iv.dy; stp = iv.stp; brackt = iv.brackt;
    if (modified) {
      fx = fx + stx * L.gtest;
      fy = fy + sty * L.gtest;
      gx = gx + L.gtest;
      gy = gy + L.gtest;
    }
    L.stx = stx;
    L.sty = sty;
    L.fx = fx;
    L.fy = fy;
    L.gx = gx;
    L.gy = gy;
    L.brackt = brackt;
  }
  if (L.brackt) {
    if (fabs(L.sty - L.stx) >= T(0.66) * L.width1) stp = L.stx + T(0.5) * (L.sty - L.stx);
    L.width1 = L.width;
    L.width = fabs(L.sty - L.stx);
  }
  if (L.brackt) {
    L.stmin = fmin(L.stx, L.sty);
    L.stmax = fmax(L.stx, L.sty);
  } else {
    L.stmin = stp + xtrapl * (stp - L.stx);
    L.stmax = stp + xtrapu * (stp - L.stx);
  }
  stp = fmax(stp, L.stpmin);
  stp = fmin(stp, L.stpmax);
  if ((L.brackt && (stp <= L.stmin || stp >= L.stmax)) ||
      (L.brackt && L.stmax - L.stmin <= L.xtol * L.stmax))
    stp = L.stx;
  return LS_FG;
}

using LineSearch = LineSearchT<double>;  // (the fp64 modes, the host harness)

}  // namespace neo
