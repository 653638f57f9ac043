// neo_group_kernel.hpp -- several trajectories per wavefront for small problems (instantiated by neo_disp_group.hip).
//
// A lane group of W = 8 or 16 lanes owns one trajectory (n <= 16 variables, M <= W pieces: the reference's M = 3),
// eight or four groups share a wavefront.  Every round all groups evaluate cost and gradient together (the expensive,
// identical instruction stream), then each group advances its own L-BFGS-B run -- the resumable form of
// neo_lbfgs_sm.hpp -- to its next trial point; what the groups do in between may differ, the hardware masks it.
// A group whose run has ended takes the next trajectory off the launch's ticket counter, so no group idles while
// work is left.  Opt-in (NEO_FLAG_LANE_GROUPS): the sample loop strides a piece's samples over W / M lanes
// instead of 64 / M, so sums are associated differently than in optimize_kernel and results agree with it to
// fp32 rounding, not bit for bit.
#pragma once
#include "neo_device.hpp"
#include "neo_lbfgs_sm.hpp"

namespace neo {

template <int D, int W, int NS, typename Real, class MapT, class LookupT, int SU, typename Num = double>
struct GroupBackend {
  using Hist = Num;  // the stored pairs follow the arithmetic (fp32 in the all-fp32 mode)
  struct Vec {
    Num v[NS];  // FLAT layout inside the group: element e <-> (lane e % W, slot e / W), n <= W * NS
  };
  Traj<D, D, Num> t;
  const DevParams &prm;
  const MapT &map;
  double *xs;       // LDS [W * NS] of this group: FLAT <-> PIECE staging
  double *sc;       // LDS [2m]
  LineSearch *lsp;  // LDS
  double *cst;      // LDS [12]
  Hist *hist;       // LDS [2][m][n]
  int m;

  __device__ GroupBackend(const DevParams &p, const MapT &mp) : prm(p), map(mp) {}

  __device__ __forceinline__ double dot(const Vec &a, const Vec &b) const {
    Num s = Num(0);
#pragma unroll
    for (int k = 0; k < NS; ++k) s += a.v[k] * b.v[k];
    return GroupLanes<W>::sum(s);
  }
  __device__ __forceinline__ double amax(const Vec &a) const {
    Num s = Num(0);
#pragma unroll
    for (int k = 0; k < NS; ++k) s = fmax(s, fabs(a.v[k]));
    return GroupLanes<W>::max_nonneg(s);
  }
  __device__ __forceinline__ void copy(Vec &d, const Vec &s) const {
#pragma unroll
    for (int k = 0; k < NS; ++k) d.v[k] = s.v[k];
  }
  __device__ __forceinline__ void neg(Vec &d, const Vec &s) const {
#pragma unroll
    for (int k = 0; k < NS; ++k) d.v[k] = -s.v[k];
  }
  __device__ __forceinline__ void axpy(double a, const Vec &x, Vec &y) const {
#pragma unroll
    for (int k = 0; k < NS; ++k) y.v[k] += (Num)a * x.v[k];
  }
  __device__ __forceinline__ void lincomb(Vec &o, const Vec &a, double s, const Vec &b) const {
#pragma unroll
    for (int k = 0; k < NS; ++k) o.v[k] = a.v[k] + (Num)s * b.v[k];
  }
  __device__ __forceinline__ void scale(Vec &v, double s) const {
#pragma unroll
    for (int k = 0; k < NS; ++k) v.v[k] *= (Num)s;
  }
  __device__ __forceinline__ void hist_put(int slot, const Vec &s, const Vec &y) {
    const int gl = GroupLanes<W>::lane();
#pragma unroll
    for (int k = 0; k < NS; ++k)
      if (k * W + gl < t.n) {
        hist[slot * t.n + k * W + gl] = s.v[k];
        hist[(m + slot) * t.n + k * W + gl] = y.v[k];
      }
    lds_wave_sync();
  }
  __device__ __forceinline__ void hist_get(int row, Vec &v) const {
    const int gl = GroupLanes<W>::lane();
#pragma unroll
    for (int k = 0; k < NS; ++k) v.v[k] = (k * W + gl < t.n) ? hist[row * t.n + k * W + gl] : Num(0);
  }
  __device__ __forceinline__ void hist_get_s(int slot, Vec &v) const { hist_get(slot, v); }
  __device__ __forceinline__ void hist_get_y(int slot, Vec &v) const { hist_get(m + slot, v); }
  __device__ __forceinline__ void hist_get_sy(int slot, Vec &s, Vec &y) const {
    hist_get(slot, s);
    hist_get(m + slot, y);
  }
  __device__ __forceinline__ void sput(int i, double v) {
    sc[i] = v;
    lds_wave_sync();
  }
  __device__ __forceinline__ double sget(int i) const { return sc[i]; }
  __device__ __forceinline__ double sdiff(int i, double b) const { return sget(i) - b; }
  __device__ __forceinline__ double rho_dot(int slot, const Vec &a, const Vec &b) const { return sget(slot) * dot(a, b); }
  __device__ __forceinline__ double uni(double v) const { return v; }  // (uniform per group only)
  __device__ __forceinline__ LineSearch &ls() { return *lsp; }
  __device__ __forceinline__ double *cost_store() { return cst; }
  __device__ __forceinline__ void note_eval(int, int, double, double, const Vec &, const Vec &) {}  // (no trace in the lane-group kernel)
  __device__ __forceinline__ void sm_stamp(int) {}

  // one evaluation (get_cost + get_grad, :539-585); costs into registers, nsamp = samples visited
  __device__ __forceinline__ int eval(const Vec &x, double &f, Vec &g, double (&costs)[4], int &nsamp) {
    const int lane = GroupLanes<W>::lane();
    const int M = t.M;
    Num *xn = reinterpret_cast<Num *>(xs);  // (the staging holds Num values)
    lds_wave_sync();
#pragma unroll
    for (int k = 0; k < NS; ++k)
      if (k * W + lane < t.n) xn[k * W + lane] = x.v[k];
    lds_wave_sync();
    const bool act = lane < M;
    t.tau = act ? xn[t.nq + lane] : Num(0);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      t.P0[d] = (lane == 0 || !act) ? bstate<D, GroupLanes<W>, false>(t, 0, d) : xn[d * (M - 1) + (lane > 0 ? lane - 1 : 0)];
      t.P1[d] = (lane >= M - 1) ? bstate<D, GroupLanes<W>, true>(t, 0, d) : xn[d * (M - 1) + lane];
    }
    double energy = 0.0, tsum = 0.0;
    const int st = minco_forward<D, GroupLanes<W>, Num>(t, prm, energy, tsum);
    // (a group whose forward pass fails -- exp overflow -- still walks through the rest on the state its last
    //  good evaluation left in `t`: the wavefront-wide loop bounds need finite values, and its own results are
    //  zeroed below exactly as DevBackend::eval returns them)
    nsamp = GroupLanes<W>::sum(act ? t.ns : 0);
    Num gC[6][D], gT = Num(0);
    double cf, ck;
    {
      Real cr[6][D], gCr[6][D], gTr;
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int d = 0; d < D; ++d) cr[k][d] = (Real)t.c[k][d];
      LookupT lk(map);
      const SampleLanes sl = group_sample_lanes<GroupLanes<W>>(M, t.ns);
      minco_sample<Real, D, LookupT, SU, false, GroupLanes<W>>(M, sl, t.ns, cr, prm, lk, gCr, gTr, cf, ck);
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int d = 0; d < D; ++d) gC[k][d] = (Num)gCr[k][d];
      gT = (Num)gTr;
    }
    costs[0] = energy;
    costs[1] = tsum;
    costs[2] = cf;
    costs[3] = ck;
    f = costs[0] * prm.w[0] + costs[1] * prm.w[1] + costs[2] * prm.w[2] + costs[3] * prm.w[3];
    Num gq[D], gtau;
    const int bst = minco_backward<D, GroupLanes<W>, Num>(t, prm, gC, gT, gq, gtau);
    lds_wave_sync();
    if (lane >= 1 && lane < M) {
#pragma unroll
      for (int d = 0; d < D; ++d) xn[d * (M - 1) + lane - 1] = gq[d];
    }
    if (lane < M) xn[t.nq + lane] = gtau;
    lds_wave_sync();
#pragma unroll
    for (int k = 0; k < NS; ++k) g.v[k] = (k * W + lane < t.n) ? xn[k * W + lane] : Num(0);
    if (st != 0) {
      f = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) costs[k] = 0.0;
      return st;
    }
    return bst;
  }
};

// grid = any number of wavefronts (the host sizes it to the chip); groups pull trajectories off *ticket
#ifndef NEO_GRP_OCC
#define NEO_GRP_OCC 1  // (measured at cfg3: 12.2 M traj/s with the whole register file and no spills, 10.8 M at two per SIMD with ~110 spilled)
#endif
#ifndef NEO_GRP_OCC_F32
#define NEO_GRP_OCC_F32 2  // all-fp32 mode: the state fits half the register file
#endif
#ifndef NEO_GRP_U
#define NEO_GRP_U 2
#endif
template <int D, typename Real, class MapT, class LookupT, int W, int NS, typename Num = double>
__global__ __launch_bounds__(kWave, (sizeof(Num) == 4 ? NEO_GRP_OCC_F32 : NEO_GRP_OCC)) void optimize_group_kernel(int B, int M, DevParams prm, const MapT *maps,
                                                                   const double *x0, double *x,
                                                                   const double *__restrict__ head,
                                                                   const double *__restrict__ tail,
                                                                   double *__restrict__ costs4,
                                                                   double *__restrict__ costs4_last,
                                                                   int *__restrict__ nit, int *__restrict__ nfev,
                                                                   int *__restrict__ status,
                                                                   long long *__restrict__ nsamples,
                                                                   const int *__restrict__ order,
                                                                   int *__restrict__ ticket) {
  constexpr int G = kWave / W;
  __shared__ double xs[G][W * NS];
  __shared__ double sc[G][2 * NEO_LBFGS_M];
  __shared__ LineSearch lsm[G];
  __shared__ double cst[G][12];
  __shared__ Num bnd[G][6 * D];  // boundary states of each group's trajectory, copied once when it is taken
  extern __shared__ double dyn_lds[];  // G * 2 * maxcor * n doubles
  using BE = GroupBackend<D, W, NS, Real, MapT, LookupT, NEO_GRP_U, Num>;
  const MapT map = maps[0];
  BE be(prm, map);
  be.t = Traj<D, D, Num>{};
  const int gl = GroupLanes<W>::lane();
  const int g = lane_id() / W;
  const int nq = D * (M - 1), n = nq + M;
  be.xs = xs[g];
  be.sc = sc[g];
  be.lsp = &lsm[g];
  be.cst = cst[g];
  be.m = NEO_LBFGS_M;
  be.hist = reinterpret_cast<typename BE::Hist *>(dyn_lds) + (size_t)g * 2 * NEO_LBFGS_M * n;
  be.t.M = M;
  be.t.nq = nq;
  be.t.n = n;
  be.t.L = GroupLanes<W>::lanes_per_piece(M);
  be.t.bnd = bnd[g];

  LbfgsOpts o{prm.ftol, prm.gtol, prm.maxls, prm.maxiter, prm.maxfun, NEO_LBFGS_M};
  LbfgsMachine<BE> mach(be, o);
  int b = 0;              // the group's trajectory
  bool busy = false;      // it has one
  long long samples = 0;

  // take the next trajectory: lane 0 of the group draws a ticket, the group loads the start point
  auto take = [&]() {
    int tk = 0;
    if (gl == 0) tk = atomicAdd(ticket, 1);
    tk = __shfl(tk, GroupLanes<W>::base(), kWave);
    busy = tk < B;
    b = busy ? (order ? order[tk] : tk) : 0;
    be.t.head = head + (size_t)b * 3 * D;
    be.t.tail = tail + (size_t)b * 3 * D;
    // (read through these pointers -- one per group, so vector loads -- the boundary states cost ~25 global loads an
    //  evaluation, each queued behind the field gathers)
    lds_wave_sync();
    for (int e = gl; e < 6 * D; e += W)
      bnd[g][e] = (Num)(e < 3 * D ? be.t.head[e] : be.t.tail[e - 3 * D]);
    lds_wave_sync();
#pragma unroll
    for (int k = 0; k < NS; ++k) mach.x.v[k] = (k * W + gl < n) ? x0[(size_t)b * n + k * W + gl] : 0.0;
    mach.begin();
    samples = 0;
  };
  auto put = [&]() {
#pragma unroll
    for (int k = 0; k < NS; ++k)
      if (k * W + gl < n) x[(size_t)b * n + k * W + gl] = mach.x.v[k];
    if (gl == 0) {
      LbfgsResult res;
      mach.result(res);
      int st = res.status;
      if (res.costs_last[3] * prm.w[3] > prm.coll_tol) st |= NEO_TRAJ_FLAG_COLLISION;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        costs4[(size_t)b * 4 + k] = res.costs[k];
        if (costs4_last) costs4_last[(size_t)b * 4 + k] = res.costs_last[k];
      }
      nit[b] = res.nit;
      nfev[b] = res.nfev;
      status[b] = st;
      if (nsamples) nsamples[b] = samples;
    }
  };

  take();
  while (__any(busy)) {
    double fnew, c4[4];
    typename BE::Vec gnew;
    int ns;
    const int est = be.eval(mach.x, fnew, gnew, c4, ns);  // every lane of the wavefront, busy group or not
    if (busy) {
      mach.f = fnew;
      mach.g = gnew;
#pragma unroll
      for (int k = 0; k < 4; ++k) be.cst[k] = c4[k];
      samples += ns;
      mach.advance(est);
      if (!mach.need_eval()) {
        put();
        take();
      }
    }
  }
}

}  // namespace neo
