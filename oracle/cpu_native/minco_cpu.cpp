// oracle/cpu_native/minco_cpu.cpp -- TEST INFRASTRUCTURE ONLY (never linked into or called by the product path).
//
// A C++ fp64 restatement, in the REFERENCE'S OWN formulation, of neo-planner's replan inner loop
// (paths relative to /root/reference/src/planner/scripts/):
//
//   * assemble the banded 6M x 6M MINCO system and solve it by LU with partial pivoting
//                                                   (traj_planner/expert_planner.py:261-336)
//   * energy / time / feasibility / collision cost  (expert_planner.py:345-422), per-sample loops as written
//   * their gradients w.r.t. coefficients and durations (expert_planner.py:361-390, 424-466)
//   * adjoint: solve(A^T, grad_C), dW/dq, dW/dT with the stale-T quirk of :528-533, sigmoid chain (:468-537)
//   * nearest-cell 2-D ESDF lookup (map_server/esdf.py:53-82) and the trilinear 3-D lookup that
//     oracle/minco_np.py:Grid3DESDF defines
//   * plan_once (:205-237): the optimiser is either SciPy's own L-BFGS-B driving mc_cost / mc_grad through ctypes
//     (oracle/cpu_native/__init__.py: the checker), or, for timing the "fair CPU" baseline of SURVEY.md 8.d3,
//     the restated L-BFGS-B control flow of csrc/neo_lbfgs.hpp on plain arrays (mc_optimize_batch; that header
//     is pinned against SciPy by tests/test_lbfgs_host.py).
//
// It deliberately shares NO arithmetic with the HIP kernels: those solve a reduced 2x2-block-tridiagonal system
// with a closed-form Hermite map, this one the reference's full banded system.  Used as (i) a second, fast
// oracle for full-size parity tests, (ii) bench.py's `cpu_native` figure, (iii) the parity CONTROL: the same
// optimiser run with the sampling arithmetic rounded to fp32, or with the coefficients perturbed by one ulp,
// which measures how far two faithful implementations of this discontinuous objective drift apart.
//
// Parity status: PINNED -- tests/test_cpu_native.py checks it against the G1 / G3 fixtures captured from the real
// reference (tests/golden/, tools/gen_golden.py) and against oracle/minco_np.py.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "neo_lbfgs.hpp"  // csrc/: control flow only (no device code is compiled here)

extern "C" {

struct mc_params {
  double v_max, T_min, T_max, safe_dis, delta_t;
  double w[4];
  double coll_tol;
  double ftol, gtol;
  int32_t maxls, maxiter, maxfun;
  int32_t stale_T;      // 1 = reproduce expert_planner.py:528-533
  int32_t sample_f32;   // control: sampled terms evaluated in fp32 arithmetic on fp32-rounded coefficients
  int32_t pad_;
  double coeff_eps;     // control: coefficients multiplied by (1 + coeff_eps * u), u in [-1, 1) hashed per entry/eval
  double grad_eps;      // control: gradient entries multiplied by (1 + grad_eps * u) (what a low-precision adjoint does)
};

struct mc_map {
  int32_t kind;  // 0: 2-D nearest-cell reference map, 1: 3-D trilinear field
  int32_t W, H;  // kind 0
  const double *dist, *gx, *gy;
  double res, ox, oy, oz;
  int32_t nx, ny, nz;  // kind 1
  int32_t pad_;
  const float *field;  // [nz][ny][nx]
};
}

namespace {

constexpr double kOob = 10000.0;  // esdf.py:64-65

// ---------------------------------------------------------------- dense-stored banded LU, partial pivoting
struct BandLU {
  int N = 0, kl = 0, ku = 0;
  std::vector<double> a;  // row-major N x N (N <= 384: small), only the band is touched
  std::vector<int> piv;
  double &at(int r, int c) { return a[(size_t)r * N + c]; }
  double at(int r, int c) const { return a[(size_t)r * N + c]; }

  void factor() {
    // bandwidths from the pattern
    kl = ku = 0;
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < N; ++c)
        if (at(r, c) != 0.0) {
          kl = std::max(kl, r - c);
          ku = std::max(ku, c - r);
        }
    piv.assign(N, 0);
    const int kuf = ku + kl;  // fill-in from row swaps
    for (int j = 0; j < N; ++j) {
      const int rmax = std::min(N - 1, j + kl);
      int p = j;
      double best = std::fabs(at(j, j));
      for (int r = j + 1; r <= rmax; ++r)
        if (std::fabs(at(r, j)) > best) {
          best = std::fabs(at(r, j));
          p = r;
        }
      piv[j] = p;
      const int cmax = std::min(N - 1, j + kuf);
      if (p != j)
        for (int c = j; c <= cmax; ++c) std::swap(at(j, c), at(p, c));
      const double d = at(j, j);
      for (int r = j + 1; r <= rmax; ++r) {
        const double l = at(r, j) / d;
        at(r, j) = l;
        if (l != 0.0)
          for (int c = j + 1; c <= cmax; ++c) at(r, c) -= l * at(j, c);
      }
    }
  }
  // solve A X = B, B is N x D row-major (in place)
  void solve(double *B, int D) const {
    const int kuf = ku + kl;
    for (int j = 0; j < N; ++j) {
      const int p = piv[j];
      if (p != j)
        for (int d = 0; d < D; ++d) std::swap(B[(size_t)j * D + d], B[(size_t)p * D + d]);
      const int rmax = std::min(N - 1, j + kl);
      for (int r = j + 1; r <= rmax; ++r) {
        const double l = at(r, j);
        if (l != 0.0)
          for (int d = 0; d < D; ++d) B[(size_t)r * D + d] -= l * B[(size_t)j * D + d];
      }
    }
    for (int j = N - 1; j >= 0; --j) {
      const int cmax = std::min(N - 1, j + kuf);
      for (int d = 0; d < D; ++d) {
        double s = B[(size_t)j * D + d];
        for (int c = j + 1; c <= cmax; ++c) s -= at(j, c) * B[(size_t)c * D + d];
        B[(size_t)j * D + d] = s / at(j, j);
      }
    }
  }
  // solve A^T X = B:  A = P^T L U  ->  A^T = U^T L^T P
  void solve_transposed(double *B, int D) const {
    const int kuf = ku + kl;
    for (int j = 0; j < N; ++j) {  // U^T z = b (forward)
      const int cmin = std::max(0, j - kuf);
      for (int d = 0; d < D; ++d) {
        double s = B[(size_t)j * D + d];
        for (int c = cmin; c < j; ++c) s -= at(c, j) * B[(size_t)c * D + d];
        B[(size_t)j * D + d] = s / at(j, j);
      }
    }
    for (int j = N - 1; j >= 0; --j) {  // L^T and the row swaps, in reverse
      const int rmax = std::min(N - 1, j + kl);
      for (int d = 0; d < D; ++d) {
        double s = B[(size_t)j * D + d];
        for (int r = j + 1; r <= rmax; ++r) s -= at(r, j) * B[(size_t)r * D + d];
        B[(size_t)j * D + d] = s;
      }
      const int p = piv[j];
      if (p != j)
        for (int d = 0; d < D; ++d) std::swap(B[(size_t)j * D + d], B[(size_t)p * D + d]);
    }
  }
};

inline double hash_unit(uint64_t k) {  // deterministic u in [-1, 1)
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return (double)(k >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

struct Planner {
  mc_params p;
  mc_map m;
  int M, D, N;
  std::vector<double> head, tail;  // [3][D]
  std::vector<double> beta;        // [rows][4][6]  (expert_planner.py:250-259)
  int beta_rows = 0;
  // state of the last evaluation (the reference keeps the same on self)
  std::vector<double> ts, tau, coeffs, gradC, gradT, G;
  BandLU lu;
  double costs[4] = {0, 0, 0, 0};
  uint64_t evals = 0;

  Planner(const mc_params &p_, const mc_map &m_, int M_, int D_, const double *h, const double *t)
      : p(p_), m(m_), M(M_), D(D_), N(6 * M_), head(h, h + 3 * D_), tail(t, t + 3 * D_) {
    // np.arange(0, T_max, delta_t): ceil((T_max - 0) / delta_t) rows, t = k * delta_t
    beta_rows = (int)std::ceil(p.T_max / p.delta_t);
    beta.assign((size_t)beta_rows * 24, 0.0);
    for (int k = 0; k < beta_rows; ++k) {
      const double t = (double)k * p.delta_t;
      double *b = &beta[(size_t)k * 24];
      const double t2 = std::pow(t, 2), t3 = std::pow(t, 3), t4 = std::pow(t, 4), t5 = std::pow(t, 5);
      const double r0[6] = {1, t, t2, t3, t4, t5};
      const double r1[6] = {0, 1, 2 * t, 3 * t2, 4 * t3, 5 * t4};
      const double r2[6] = {0, 0, 2, 6 * t, 12 * t2, 20 * t3};
      const double r3[6] = {0, 0, 0, 6, 24 * t, 60 * t2};
      memcpy(b, r0, sizeof r0);
      memcpy(b + 6, r1, sizeof r1);
      memcpy(b + 12, r2, sizeof r2);
      memcpy(b + 18, r3, sizeof r3);
    }
    ts.assign(M, 0.0);
    tau.assign(M, 0.0);
    coeffs.assign((size_t)N * D, 0.0);
    gradC.assign((size_t)N * D, 0.0);
    gradT.assign(M, 0.0);
    G.assign((size_t)N * D, 0.0);
  }

  // ---- map lookups
  bool cell2d(const double *pos, int &row, int &col) const {  // esdf.py:61-62, int() truncates toward zero
    const double fy = (pos[1] - m.oy) / m.res, fx = (pos[0] - m.ox) / m.res;
    if (!(std::fabs(fy) < 1e9) || !(std::fabs(fx) < 1e9)) return false;
    row = (int)fy;
    col = (int)fx;
    return row >= 0 && row < m.H && col >= 0 && col < m.W;
  }
  template <typename Real>
  Real lookup(const Real *pos, Real *g /*[D] or null*/) const {
    if (g)
      for (int d = 0; d < D; ++d) g[d] = Real(0);
    if (m.kind == 0) {
      const double q[2] = {(double)pos[0], (double)pos[1]};
      int row, col;
      if (!cell2d(q, row, col)) return (Real)kOob;
      const size_t i = (size_t)row * m.W + col;
      if (g) {
        g[0] = (Real)m.gx[i];
        g[1] = (Real)m.gy[i];
      }
      return (Real)m.dist[i];
    }
    // oracle/minco_np.py:Grid3DESDF
    const int n[3] = {m.nx, m.ny, m.nz};
    const double org[3] = {m.ox, m.oy, m.oz};
    int i0[3];
    Real fr[3];
    for (int a = 0; a < 3; ++a) {
      Real u = (Real)(((double)pos[a] - org[a]) / m.res);
      if (!(u >= Real(0) && u < (Real)n[a])) return (Real)kOob;
      u -= Real(0.5);
      int i = (int)std::floor(u);
      i = std::min(std::max(i, 0), n[a] - 2);
      i0[a] = i;
      fr[a] = std::min(std::max(u - (Real)i, Real(0)), Real(1));
    }
    const float *d = m.field;
    auto at = [&](int dz, int dy, int dx) -> Real {
      return (Real)d[((size_t)(i0[2] + dz) * m.ny + (i0[1] + dy)) * m.nx + i0[0] + dx];
    };
    const Real fx = fr[0], fy = fr[1], fz = fr[2];
    const Real c000 = at(0, 0, 0), c100 = at(0, 0, 1), c010 = at(0, 1, 0), c110 = at(0, 1, 1);
    const Real c001 = at(1, 0, 0), c101 = at(1, 0, 1), c011 = at(1, 1, 0), c111 = at(1, 1, 1);
    const Real c00 = c000 + fx * (c100 - c000), c10 = c010 + fx * (c110 - c010);
    const Real c01 = c001 + fx * (c101 - c001), c11 = c011 + fx * (c111 - c011);
    const Real c0 = c00 + fy * (c10 - c00), c1 = c01 + fy * (c11 - c01);
    if (g) {
      const Real dx00 = c100 - c000, dx10 = c110 - c010, dx01 = c101 - c001, dx11 = c111 - c011;
      const Real dx0 = dx00 + fy * (dx10 - dx00), dx1 = dx01 + fy * (dx11 - dx01);
      const Real dy0 = c10 - c00, dy1 = c11 - c01;
      const Real r = (Real)m.res;
      g[0] = (dx0 + fz * (dx1 - dx0)) / r;
      g[1] = (dy0 + fz * (dy1 - dy0)) / r;
      g[2] = (c1 - c0) / r;
    }
    return c0 + fz * (c1 - c0);
  }

  // ---- map_tau2T (:477-483); returns false where math.exp raises OverflowError
  bool unpack(const double *x) {
    const int nq = D * (M - 1);
    for (int i = 0; i < M; ++i) {
      tau[i] = x[nq + i];
      if (-tau[i] > 709.782712893384) return false;
      ts[i] = (p.T_max - p.T_min) / (1.0 + std::exp(-tau[i])) + p.T_min;
    }
    return true;
  }

  // ---- get_coeffs (:261-336)
  void get_coeffs(const double *x) {
    lu.N = N;
    lu.a.assign((size_t)N * N, 0.0);
    std::vector<double> &b = coeffs;
    std::fill(b.begin(), b.end(), 0.0);
    for (int k = 0; k < 3; ++k)
      for (int d = 0; d < D; ++d) {
        b[(size_t)k * D + d] = head[(size_t)k * D + d];
        b[(size_t)(N - 3 + k) * D + d] = tail[(size_t)k * D + d];
      }
    lu.at(0, 0) = 1.0;
    lu.at(1, 1) = 1.0;
    lu.at(2, 2) = 2.0;
    for (int i = 0; i < M - 1; ++i) {
      const double T = ts[i], P2 = std::pow(T, 2), P3 = std::pow(T, 3), P4 = std::pow(T, 4), P5 = std::pow(T, 5);
      const int r = 6 * i + 3, c = 6 * i;
      const double pw[6] = {1.0, T, P2, P3, P4, P5};
      for (int k = 0; k < 6; ++k) {
        lu.at(r, c + k) = pw[k];
        lu.at(r + 1, c + k) = pw[k];
      }
      lu.at(r + 1, c + 6) = -1.0;
      const double v[5] = {1.0, 2 * T, 3 * P2, 4 * P3, 5 * P4};
      for (int k = 0; k < 5; ++k) lu.at(r + 2, c + 1 + k) = v[k];
      lu.at(r + 2, c + 7) = -1.0;
      const double a[4] = {2.0, 6 * T, 12 * P2, 20 * P3};
      for (int k = 0; k < 4; ++k) lu.at(r + 3, c + 2 + k) = a[k];
      lu.at(r + 3, c + 8) = -2.0;
      const double j[3] = {6.0, 24.0 * T, 60.0 * P2};
      for (int k = 0; k < 3; ++k) lu.at(r + 4, c + 3 + k) = j[k];
      lu.at(r + 4, c + 9) = -6.0;
      lu.at(r + 5, c + 4) = 24.0;
      lu.at(r + 5, c + 5) = 120.0 * T;
      lu.at(r + 5, c + 10) = -24.0;
      for (int d = 0; d < D; ++d) b[(size_t)r * D + d] = x[(size_t)d * (M - 1) + i];  // int_wpts (D, M-1) row-major
    }
    {
      const double T = ts[M - 1], P2 = std::pow(T, 2), P3 = std::pow(T, 3), P4 = std::pow(T, 4), P5 = std::pow(T, 5);
      const double r0[6] = {1.0, T, P2, P3, P4, P5};
      const double r1[5] = {1.0, 2 * T, 3 * P2, 4 * P3, 5 * P4};
      const double r2[4] = {2.0, 6 * T, 12 * P2, 20 * P3};
      for (int k = 0; k < 6; ++k) lu.at(N - 3, N - 6 + k) = r0[k];
      for (int k = 0; k < 5; ++k) lu.at(N - 2, N - 5 + k) = r1[k];
      for (int k = 0; k < 4; ++k) lu.at(N - 1, N - 4 + k) = r2[k];
    }
    lu.factor();
    lu.solve(b.data(), D);
    if (p.coeff_eps != 0.0) {
      for (size_t i = 0; i < b.size(); ++i) b[i] *= 1.0 + p.coeff_eps * hash_unit(evals * 1000003ull + i);
    }
  }

  static void jerk_gram(double T, double Q[6][6]) {  // :353-358
    memset(Q, 0, 36 * sizeof(double));
    const double T2 = std::pow(T, 2), T3 = std::pow(T, 3), T4 = std::pow(T, 4), T5 = std::pow(T, 5);
    Q[3][3] = 36 * T;    Q[3][4] = 72 * T2;   Q[3][5] = 120 * T3;
    Q[4][3] = 72 * T2;   Q[4][4] = 192 * T3;  Q[4][5] = 360 * T4;
    Q[5][3] = 120 * T3;  Q[5][4] = 360 * T4;  Q[5][5] = 720 * T5;
  }

  void add_energy_time(bool grad) {
    for (int i = 0; i < M; ++i) {
      const double *c = &coeffs[(size_t)6 * i * D];
      double Q[6][6];
      jerk_gram(ts[i], Q);
      const double T = ts[i];
      for (int d = 0; d < D; ++d) {
        double e = 0.0, jerk = 0.0;
        const double jr[6] = {0, 0, 0, 6, 24 * T, 60 * std::pow(T, 2)};
        for (int a = 3; a < 6; ++a) {
          double qa = 0.0;
          for (int b2 = 3; b2 < 6; ++b2) qa += Q[a][b2] * c[(size_t)b2 * D + d];
          e += c[(size_t)a * D + d] * qa;
          if (grad) gradC[(size_t)(6 * i + a) * D + d] += p.w[0] * 2 * qa;
          jerk += c[(size_t)a * D + d] * jr[a];
        }
        if (!grad) costs[0] += e;
        if (grad) gradT[i] += p.w[0] * jerk * jerk;
      }
      if (!grad) costs[1] += T;
      if (grad) gradT[i] += p.w[1];
    }
  }

  // ---- add_sampled_cost + add_sampled_grad_CT (:392-466), per-sample loop; Real = arithmetic of the samples
  template <typename Real>
  void add_sampled(bool grad) {
    const Real dt = (Real)p.delta_t, vmax2 = (Real)(p.v_max * p.v_max), safe = (Real)p.safe_dis;
    for (int i = 0; i < M; ++i) {
      Real c[6][3];
      for (int k = 0; k < 6; ++k)
        for (int d = 0; d < D; ++d) c[k][d] = (Real)coeffs[(size_t)(6 * i + k) * D + d];
      const int n_i = (int)(ts[i] / p.delta_t);
      for (int j = 0; j < n_i && j < beta_rows; ++j) {
        const double *b = &beta[(size_t)j * 24];
        Real pos[3] = {0, 0, 0}, vel[3] = {0, 0, 0};
        for (int d = 0; d < D; ++d) {
          Real sp = 0, sv = 0;
          for (int k = 0; k < 6; ++k) {
            sp += c[k][d] * (Real)b[k];
            sv += c[k][d] * (Real)b[6 + k];
          }
          pos[d] = sp;
          vel[d] = sv;
        }
        const Real omg = (j == 0 || j == n_i - 1) ? Real(0.5) : Real(1);
        Real v2 = 0;
        for (int d = 0; d < D; ++d) v2 += vel[d] * vel[d];
        const Real vv = v2 - vmax2;
        if (vv > Real(0)) {
          if (!grad) {
            costs[2] += (double)(omg * dt * vv * vv * vv);
          } else {
            Real av = 0;
            for (int d = 0; d < D; ++d) {
              Real acc = 0;
              for (int k = 0; k < 6; ++k) acc += c[k][d] * (Real)b[12 + k];
              av += acc * vel[d];
            }
            const Real dK = Real(3) * dt * omg * vv * vv;
            for (int k = 0; k < 6; ++k)
              for (int d = 0; d < D; ++d)
                gradC[(size_t)(6 * i + k) * D + d] += (double)((Real)p.w[2] * dK * Real(2) * (Real)b[6 + k] * vel[d]);
            gradT[i] += (double)((Real)p.w[2] * (omg * vv * vv * vv / (Real)n_i + dK * Real(2) * av * (Real)j / (Real)n_i));
          }
        }
        Real g[3];
        const Real dis = lookup<Real>(pos, grad ? g : nullptr);
        const Real vd = safe - dis;
        if (vd > Real(0)) {
          if (!grad) {
            costs[3] += (double)(omg * dt * vd * vd * vd);
          } else {
            const Real dK = Real(3) * dt * omg * vd * vd;
            Real gv = 0;
            for (int d = 0; d < D; ++d) gv += g[d] * vel[d];
            for (int k = 0; k < 6; ++k)
              for (int d = 0; d < D; ++d)
                gradC[(size_t)(6 * i + k) * D + d] += (double)((Real)p.w[3] * dK * (-(Real)b[k] * g[d]));
            gradT[i] += (double)((Real)p.w[3] * (omg * vd * vd * vd / (Real)n_i + dK * (-gv) * (Real)j / (Real)n_i));
          }
        }
      }
    }
  }

  // ---- get_cost (:539-558)
  int cost(const double *x, double *f) {
    if (!unpack(x)) return 4;
    get_coeffs(x);
    for (double &c : costs) c = 0.0;
    add_energy_time(false);
    if (p.sample_f32)
      add_sampled<float>(false);
    else
      add_sampled<double>(false);
    *f = costs[0] * p.w[0] + costs[1] * p.w[1] + costs[2] * p.w[2] + costs[3] * p.w[3];
    return 0;
  }

  static void dE_joint(double T, double E[6][6]) {  // rows: waypoint, pos, vel, acc, jerk, snap continuity
    const double T2 = std::pow(T, 2), T3 = std::pow(T, 3), T4 = std::pow(T, 4);
    const double r[6][6] = {{0, 1, 2 * T, 3 * T2, 4 * T3, 5 * T4}, {0, 1, 2 * T, 3 * T2, 4 * T3, 5 * T4},
                            {0, 0, 2, 6 * T, 12 * T2, 20 * T3},    {0, 0, 0, 6, 24 * T, 60 * T2},
                            {0, 0, 0, 0, 24, 120 * T},             {0, 0, 0, 0, 0, 120}};
    memcpy(E, r, sizeof r);
  }

  // ---- get_grad (:560-585)
  int grad(const double *x, double *g) {
    if (!unpack(x)) return 4;
    get_coeffs(x);
    std::fill(gradC.begin(), gradC.end(), 0.0);
    std::fill(gradT.begin(), gradT.end(), 0.0);
    add_energy_time(true);
    if (p.sample_f32)
      add_sampled<float>(true);
    else
      add_sampled<double>(true);
    // propagate_grad_q_tau (:494-537)
    G = gradC;
    lu.solve_transposed(G.data(), D);
    const int nq = D * (M - 1);
    for (int i = 0; i < M - 1; ++i)
      for (int d = 0; d < D; ++d) g[(size_t)d * (M - 1) + i] = G[(size_t)(6 * i + 3) * D + d];
    std::vector<double> gT(M, 0.0);
    double T = 0.0;
    bool have_T = false;
    for (int i = 0; i < M - 1; ++i) {
      T = ts[i];
      have_T = true;
      double E[6][6];
      dE_joint(T, E);
      double tr = 0.0;
      for (int r = 0; r < 6; ++r)
        for (int d = 0; d < D; ++d) {
          double ec = 0.0;
          for (int k = 0; k < 6; ++k) ec += E[r][k] * coeffs[(size_t)(6 * i + k) * D + d];
          tr += G[(size_t)(6 * i + 3 + r) * D + d] * ec;
        }
      gT[i] = gradT[i] - tr;
    }
    if (!p.stale_T || !have_T) T = ts[M - 1];
    {
      double E[6][6];
      dE_joint(T, E);
      double tr = 0.0;
      for (int r = 0; r < 3; ++r)  // rows 1..3 of dE_joint: vel, acc, jerk rows of the tail block
        for (int d = 0; d < D; ++d) {
          double ec = 0.0;
          for (int k = 0; k < 6; ++k) ec += E[r + 1][k] * coeffs[(size_t)(6 * (M - 1) + k) * D + d];
          tr += G[(size_t)(6 * M - 3 + r) * D + d] * ec;
        }
      gT[M - 1] = gradT[M - 1] - tr;
    }
    for (int i = 0; i < M; ++i) {  // get_grad_T2tau (:485-492)
      const double e = std::exp(-tau[i]);
      if ((1.0 + e) > 1.3407807929942596e154) return 4;  // (1 + math.exp(-tau))**2 raises OverflowError
      g[nq + i] = gT[i] * (p.T_max - p.T_min) * e / ((1.0 + e) * (1.0 + e));
    }
    return 0;
  }
};

// ---------------------------------------------------------------- L-BFGS-B backend on plain arrays
struct HostBackend {
  using Vec = std::vector<double>;
  Planner &pl;
  int n, m;
  std::vector<double> S, Y, scal;
  neo::LineSearch lsearch;
  double cost12[12];
  neo::LineSearch &ls() { return lsearch; }
  double *cost_store() { return cost12; }
  HostBackend(Planner &p, int n_, int m_) : pl(p), n(n_), m(m_), S((size_t)n_ * m_), Y((size_t)n_ * m_), scal(2 * m_) {}
  void fit(Vec &v) const {
    if ((int)v.size() != n) v.assign(n, 0.0);
  }
  double dot(const Vec &a, const Vec &b) const {
    double s = 0;
    for (int i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
  }
  double amax(const Vec &a) const {
    double s = 0;
    for (int i = 0; i < n; ++i) s = std::fmax(s, std::fabs(a[i]));
    return s;
  }
  void copy(Vec &d, const Vec &s) const { d = s; }
  void neg(Vec &d, const Vec &s) const {
    fit(d);
    for (int i = 0; i < n; ++i) d[i] = -s[i];
  }
  void axpy(double a, const Vec &x, Vec &y) const {
    for (int i = 0; i < n; ++i) y[i] += a * x[i];
  }
  void lincomb(Vec &out, const Vec &a, double s, const Vec &b) const {
    fit(out);
    for (int i = 0; i < n; ++i) out[i] = a[i] + s * b[i];
  }
  void scale(Vec &v, double s) const {
    for (int i = 0; i < n; ++i) v[i] *= s;
  }
  void hist_put(int slot, const Vec &s, const Vec &y) {
    memcpy(&S[(size_t)slot * n], s.data(), n * sizeof(double));
    memcpy(&Y[(size_t)slot * n], y.data(), n * sizeof(double));
  }
  void hist_get_s(int slot, Vec &v) const { v.assign(&S[(size_t)slot * n], &S[(size_t)slot * n] + n); }
  void hist_get_y(int slot, Vec &v) const { v.assign(&Y[(size_t)slot * n], &Y[(size_t)slot * n] + n); }
  double uni(double v) const { return v; }
  double sdiff(int i, double b) const { return sget(i) - b; }
  double rho_dot(int slot, const Vec &a, const Vec &b) const { return sget(slot) * dot(a, b); }
  void hist_get_sy(int slot, Vec &s, Vec &y) const {
    hist_get_s(slot, s);
    hist_get_y(slot, y);
  }

  void sput(int i, double v) { scal[i] = v; }
  double sget(int i) const { return scal[i]; }
  double *trace = nullptr;  // optional [trace_cap][4]: (f, step, samples, iteration) per counted evaluation
  int trace_cap = 0;
  void note_eval(int nfev, int iter, double stp, double f, const Vec &, const Vec &) {
    if (trace && nfev <= trace_cap) {
      double *r = trace + (size_t)(nfev - 1) * 4;
      int ns = 0;
      for (int i = 0; i < pl.M; ++i) ns += (int)(pl.ts[i] / pl.p.delta_t);
      r[0] = f;
      r[1] = stp;
      r[2] = (double)ns;
      r[3] = (double)iter;
    }
  }
  // SciPy calls fun(x) then jac(x): the reference solves the system twice per trial point
  int eval(const Vec &x, double &f, Vec &g, double *costs4) {
    fit(g);
    pl.evals++;
    int st = pl.cost(x.data(), &f);
    if (st) return st;
    for (int k = 0; k < 4; ++k) costs4[k] = pl.costs[k];
    st = pl.grad(x.data(), g.data());
    if (pl.p.grad_eps != 0.0)
      for (size_t i = 0; i < g.size(); ++i) g[i] *= 1.0 + pl.p.grad_eps * hash_unit(pl.evals * 7000003ull + i);
    return st;
  }
};

int optimize_one(const mc_params &p, const mc_map &m, int M, int D, double *x, const double *head, const double *tail,
                 double *costs4, double *costs4_last, int *nit, int *nfev, int *status, double *trace, int trace_cap) {
  const int n = D * (M - 1) + M;
  Planner pl(p, m, M, D, head, tail);
  HostBackend be(pl, n, 10);
  be.trace = trace;
  be.trace_cap = trace_cap;
  HostBackend::Vec xv(x, x + n);
  neo::LbfgsOpts o{p.ftol, p.gtol, p.maxls, p.maxiter, p.maxfun, 10};
  neo::LbfgsResult res;
  neo::lbfgs_minimize(be, xv, o, res);
  memcpy(x, xv.data(), n * sizeof(double));
  int st = res.status;
  if (res.costs_last[3] * p.w[3] > p.coll_tol) st |= 0x100;  // :233-237
  for (int k = 0; k < 4; ++k) {
    if (costs4) costs4[k] = res.costs[k];
    if (costs4_last) costs4_last[k] = res.costs_last[k];
  }
  *nit = res.nit;
  *nfev = res.nfev;
  *status = st;
  return 0;
}

}  // namespace

extern "C" {

// opaque planner for SciPy-driven runs (oracle/cpu_native/__init__.py)
void *mc_create(const mc_params *p, const mc_map *m, int M, int D, const double *head, const double *tail) {
  if (!p || !m || M < 1 || M > 64 || D < 2 || D > 3) return nullptr;
  return new Planner(*p, *m, M, D, head, tail);
}
void mc_destroy(void *h) { delete static_cast<Planner *>(h); }
int mc_cost(void *h, const double *x, double *f, double *costs4) {
  Planner *pl = static_cast<Planner *>(h);
  pl->evals++;
  const int st = pl->cost(x, f);
  if (costs4)
    for (int k = 0; k < 4; ++k) costs4[k] = pl->costs[k];
  return st;
}
// grad[n]; optional copies of the state the reference leaves on self: coeffs[6M][D], grad_C[6M][D] and grad_T[M]
// (pre-propagation partials, :364-466), ts[M]
int mc_grad(void *h, const double *x, double *grad, double *coeffs, double *grad_C, double *grad_T, double *ts) {
  Planner *pl = static_cast<Planner *>(h);
  const int st = pl->grad(x, grad);
  if (coeffs) memcpy(coeffs, pl->coeffs.data(), pl->coeffs.size() * sizeof(double));
  if (grad_C) memcpy(grad_C, pl->gradC.data(), pl->gradC.size() * sizeof(double));
  if (grad_T) memcpy(grad_T, pl->gradT.data(), pl->gradT.size() * sizeof(double));
  if (ts) memcpy(ts, pl->ts.data(), pl->ts.size() * sizeof(double));
  return st;
}

// cost, cost terms and gradient at E given points x[E][n] of ONE trajectory (head, tail): what the replay tests lay beside
// every point a GPU run evaluated.  st[E] = 0 or 4 (the reference's OverflowError).
int mc_eval_points(const mc_params *p, const mc_map *m, int M, int D, const double *head, const double *tail, int E,
                   const double *x, double *f, double *costs4, double *grad, int32_t *st) {
  if (!p || !m || M < 1 || M > 64 || D < 2 || D > 3 || E < 0) return -1;
  const int n = D * (M - 1) + M;
  Planner pl(*p, *m, M, D, head, tail);
  for (int e = 0; e < E; ++e) {
    pl.evals++;
    int s_ = pl.cost(x + (size_t)e * n, f + e);
    for (int k = 0; k < 4; ++k) costs4[(size_t)e * 4 + k] = pl.costs[k];
    if (!s_) s_ = pl.grad(x + (size_t)e * n, grad + (size_t)e * n);
    st[e] = s_;
  }
  return 0;
}

// plan_once for B trajectories on `threads` host threads, optimiser = csrc/neo_lbfgs.hpp on plain arrays.
// x[B][n] in/out, head/tail [B][3][D]; outputs as neo_optimize_batch.  limit_s > 0: stop handing out new
// trajectories after that many seconds; done[B] = 1 for the finished ones.  trace: optional [B][trace_cap][4]
// records (f, step, samples, iteration) per counted evaluation.  Returns the number finished.
int mc_optimize_batch(const mc_params *p, const mc_map *m, int B, int M, int D, double *x, const double *head,
                      const double *tail, double *costs4, double *costs4_last, int32_t *nit, int32_t *nfev,
                      int32_t *status, int threads, double limit_s, uint8_t *done, double *trace, int trace_cap) {
  if (!p || !m || B < 0 || M < 1 || M > 64 || D < 2 || D > 3) return -1;
  const int n = D * (M - 1) + M;
  std::atomic<int> next{0}, fin{0};
  const auto t0 = std::chrono::steady_clock::now();
  auto work = [&]() {
    for (;;) {
      const int b = next.fetch_add(1);
      if (b >= B) break;
      if (limit_s > 0.0 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s)
        break;
      int it = 0, fe = 0, st = 0;
      optimize_one(*p, *m, M, D, x + (size_t)b * n, head + (size_t)b * 3 * D, tail + (size_t)b * 3 * D,
                   costs4 ? costs4 + (size_t)b * 4 : nullptr, costs4_last ? costs4_last + (size_t)b * 4 : nullptr, &it, &fe,
                   &st, trace ? trace + (size_t)b * trace_cap * 4 : nullptr, trace_cap);
      nit[b] = it;
      nfev[b] = fe;
      status[b] = st;
      if (done) done[b] = 1;
      fin.fetch_add(1);
    }
  };
  if (threads <= 1) {
    work();
  } else {
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t) pool.emplace_back(work);
    for (auto &t : pool) t.join();
  }
  return fin.load();
}

}  // extern "C"
