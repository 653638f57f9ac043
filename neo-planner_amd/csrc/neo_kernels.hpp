// neo_kernels.hpp -- the fused kernels of libneo_planner_hip.so as templates (gfx950 only); every translation
// unit of the library instantiates the family it dispatches to (neo_disp_*.hip), so that the families compile in parallel.
//
//   eval_kernel      get_cost + get_grad for a batch            (expert_planner.py:539-585)
//   optimize_kernel  plan_once: the whole L-BFGS-B run on-chip  (expert_planner.py:205-237)
//   sample_kernel    add_sampled_cost + add_sampled_grad_CT     (expert_planner.py:392-466)
//
// x0 / x: the start points are read from x0 and the results written to x; the two may be the same buffer (every lane reads
// its own elements before it writes them), which is the in-place form of neo_optimize_batch_dev.
// One 64-lane workgroup (= one wavefront) per trajectory: the optimiser never leaves the chip,
// finished trajectories free their slot for the next ones, no host round trips.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "../../include/neo_planner.h"
#ifndef NEO_FUSED_U
#define NEO_FUSED_U (sizeof(Real) == 4 ? 4 : 2)
#endif
#ifndef NEO_W2_U
#define NEO_W2_U 1  // (one sample per lane in flight: the two-waves variant then has no VGPR spills; 2 -> 7 % slower)
#endif
#ifndef NEO_W2_OCC
#define NEO_W2_OCC 2  // wavefronts per SIMD the throughput variant is allocated for (experiments: 3)
#endif
#ifndef NEO_X_OCC
// all-fp32 kernels (Num = float) with up to two FLAT slots (n <= 128; four slots -- cfg5 -- spill 167 registers at that
// budget and stay at two per SIMD): allocated for THREE wavefronts per SIMD (168 registers; the cfg2 instantiation
// spills 25 of them, built with -amdgpu-use-amdgpu-trackers: neo_planner_amd/build.py) -- measured 933 k traj/s against
// 889 k at two per SIMD without spills; LDS: fp32 pairs + staging = 12.6 KB of the 13.3 KB a twelfth of a CU has
#define NEO_X_OCC 3
#endif
#ifndef NEO_SM_MAX_SLOTS
#define NEO_SM_MAX_SLOTS 2  // FLAT slots up to which the optimiser runs as the resumable state machine (neo_lbfgs_sm.hpp)
#endif
#ifndef NEO_W2_MAX_SLOTS
#define NEO_W2_MAX_SLOTS 4  // two waves per SIMD for every n <= 256 (four FLAT slots: with fp32 pairs, pairs_in_f32)
#endif
#include "neo_device.hpp"
#include "neo_lbfgs.hpp"
#ifdef NEO_STAMPS
#define NEO_SM_STAMP(i) be.sm_stamp(i)
#endif
#include "neo_lbfgs_sm.hpp"

namespace neo {

// the type of the optimiser's scalars and of the line-search state: fp32 where everything else is (Num = float) and the
// run is the state machine (the straight-line form of the four-slot kernels keeps fp64 scalars)
template <typename Num, int NS>
#ifdef NEO_F64_SCALARS  // comparison builds: the fp64 scalars of rounds 1 - 4 everywhere
using opt_scalar_t = double;
#else
using opt_scalar_t = std::conditional_t<sizeof(Num) == 4 && (NS <= NEO_SM_MAX_SLOTS), float, double>;
#endif

// doubles of LDS staging a wavefront needs: NS * 64 for the FLAT <-> PIECE exchange, 64 rows of [D][8] Reals for the
// per-piece fold of the sampled partials (the two uses never overlap)
template <int D, int NS, typename Real>
__host__ __device__ constexpr int stage_doubles() {
  constexpr int fold = sizeof(Real) == 4 ? kWave * 8 * D * (int)sizeof(Real) : 0;  // (fp32 sampling only)
  return (NS * kWave * 8 > fold ? NS * kWave * 8 : fold) / 8;
}

constexpr int kSlCacheInts = kWave / 4 + kWave + 4;  // DevBackend::sl_cache: a key byte and an assignment word a lane, three uniform ints

// ------------------------------------------------------------------ device backend of the optimiser
// SU: samples per lane in flight in the sample loop (minco_sample)
// LG: lane layout of the PIECE-layout phases -- WaveLanes (lane = piece) or WaveLanesPD<D> (lane = (piece, dimension))
// Storage of the L-BFGS pairs.  fp64, except in the two-waves fp32-sampling kernels with four FLAT slots (n > 128,
// cfg5): their 2 * 10 * n doubles (25.8 KB at n = 161) leave room for six wavefronts per CU, in fp32 for eight -- measured
// at cfg5: 119 k traj/s with one wavefront per SIMD, 105 k with two and fp64 pairs, 155 k with two and fp32 pairs,
// mean nfev 336.7 against 337.5 and the same status histogram (the rounding of the pairs is well below what the fp32
// gradient already carries).
template <typename Real, int NS, int WAVES>
__host__ __device__ constexpr bool pairs_in_f32() {
  return sizeof(Real) == 4 && WAVES == 2 && NS > 2;
}
// PAIRS32: the L-BFGS pairs are stored in fp32
// Num: arithmetic of the coefficient solve, the adjoint and the optimiser vectors (double; float = the all-fp32 mode)
template <int D, int NS, typename Real, class MapT, class LookupT, int SU = NEO_FUSED_U, class LG = WaveLanes,
          bool PAIRS32 = false, typename Num = double>
struct DevBackend {
  static constexpr int DL = LG::dl(D);
  static constexpr int kSlotsN = NS;  // FLAT slots of a Vec
  // joint systems by parallel cyclic reduction in the all-fp32 mode (neo_device.hpp pcr_solve).  Block Thomas wherever the
  // solve is fp64: the parity mode's recorded runs are pinned to its rounding, and in the mixed mode the reduction was
  // measured SLOWER (fp64: 744 k against 778 k traj/s at cfg2, 142 k against 156 k at cfg5 -- five levels of 63 fp64
  // instructions with ~60 live doubles spill where block Thomas does not)
  // (measured again after the spills had gone: in the lane = (piece, dimension) kernel the fp64 reduction fits the
  // registers -- 251 of 256, no spills -- and changes nothing: 774 k traj/s either way)
  static constexpr bool kPcr = sizeof(Num) == 4 && LG::W == kWave;
  // FLAT layout with NS slots: n <= 64 * NS
  struct Vec {
    Num v[NS];
  };
  Traj<D, DL, Num> t;
  const DevParams &prm;
  const MapT &map;
  double *xs;    // LDS [kStage]: FLAT <-> PIECE staging; between scatter_x and the gradient gather the same memory
                 // holds the lane assignment's segment table and the rows of the per-piece fold (minco_sample)
  static constexpr int kStage = stage_doubles<D, NS, Real>();
  // LDS [kSlCacheInts] or nullptr: the lane assignment of the last evaluation (a key byte and a word a lane + three wave-uniform ints).
  // The assignment depends on the pieces' sample counts int(T / delta_t) alone, and along a run 58 % of consecutive
  // iterates have the same counts in every piece (line-search trial points more often still): a hit costs a compare, a
  // ballot and two LDS reads instead of the ~140 vector instructions of balanced_sample_lanes.  Same assignment, same bits.
  int *sl_cache = nullptr;
  bool fold_rows = true;  // xs has all kStage doubles (false: only NS * 64, the fold stays in registers)
  bool fold_acc = false;  // xs holds [M][fold_acc_stride(D)] per-piece accumulators instead of the rows (minco_sample; the kernels that keep
                          // the cyclic reduction's multipliers in LDS: launch_opt)
  // the optimiser's scalars (f, step, g . d, theta) and the line-search state: fp32 in the all-fp32 kernels that run the
  // state machine (round 5: dcsrch / dcstep were ~150 wave-uniform fp64 instructions, eight divisions and a square root
  // among them, on the dependent chain of every evaluation), fp64 otherwise
  using Scalar = opt_scalar_t<Num, NS>;
  LineSearchT<Scalar> *lsp;  // LDS: line-search state (wave-uniform)
  double *cst;      // LDS [12]: cost terms of the last evaluation / current x / previous x
  // LDS [2][m][n]: the stored (s, y) pairs of this trajectory
  using Hist = std::conditional_t<PAIRS32, float, double>;
  Hist *hist;
  int npad, m;
  double *coeff_out;  // optional [6M][D] (eval kernel)
  long long samples = 0;  // quadrature samples visited so far (lane-uniform), for the bench's byte count
  int last_ns = 0;        // samples of the last evaluation
  double *trace = nullptr;  // optional [trace_cap][4] of this trajectory: (f, step, samples, iteration) per evaluation
  double *trace_xg = nullptr;  // optional [trace_cap][2][n]: the evaluated point and its gradient
  int trace_cap = 0;
#ifdef NEO_STAMPS  // timing experiments (tools/gpu_straggler.py): 100 MHz wall-clock ticks per phase
  long long tk[4] = {0, 0, 0, 0};  // forward, sample, backward, evaluations
  long long tl = 0, tl0 = 0;       // two-loop recursion (direction) time
  __device__ __forceinline__ void sm_stamp(int i) {
    if (i == 0) tl0 = wall_clock64(); else tl += wall_clock64() - tl0;
  }
#endif

  __device__ DevBackend(const DevParams &p, const MapT &mp) : prm(p), map(mp) {}

  __device__ __forceinline__ double dot(const Vec &a, const Vec &b) const {
    Num s = Num(0);
#pragma unroll
    for (int k = 0; k < NS; ++k) s += a.v[k] * b.v[k];
    return uni((double)wave_sum(s));
  }
  // a wave-uniform scalar back into scalar registers (two v_readfirstlane): the optimiser's scalar state then costs
  // no vector registers
  __device__ __forceinline__ double uni(double v) const { return uniform(v); }
  __device__ __forceinline__ float uni(float v) const { return uniform(v); }
  __device__ __forceinline__ double amax(const Vec &a) const {
    Num s = Num(0);
#pragma unroll
    for (int k = 0; k < NS; ++k) s = fmax(s, fabs(a.v[k]));
    return wave_max_nonneg(s);
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
  // element k * 64 + lane of a FLAT vector exists.  The kernel with NS slots is launched for kFull * 64 < n <= NS * 64
  // (slots_for: kFull = 0, 1, 2, 2 for NS = 1, 2, 3, 4), so the first kFull slots are full in every lane: no exec masking
  // around their LDS accesses.
  static constexpr int kFull = NS == 3 ? 2 : NS / 2;
  __device__ __forceinline__ bool in_range(int k, int lane) const { return k < kFull || k * kWave + lane < t.n; }
  __device__ __forceinline__ void hist_put(int slot, const Vec &s, const Vec &y) {
    const int lane = lane_id();
#pragma unroll
    for (int k = 0; k < NS; ++k)
      if (in_range(k, lane)) {
        hist[slot * t.n + k * kWave + lane] = (Hist)s.v[k];
        hist[(m + slot) * t.n + k * kWave + lane] = (Hist)y.v[k];
      }
    lds_wave_sync();
  }
  __device__ __forceinline__ void hist_get(int row, Vec &v) const {
    const int lane = lane_id();
#pragma unroll
    for (int k = 0; k < NS; ++k) v.v[k] = in_range(k, lane) ? (Num)hist[row * t.n + k * kWave + lane] : Num(0);
  }
  __device__ __forceinline__ void hist_get_s(int slot, Vec &v) const { hist_get(slot, v); }
  __device__ __forceinline__ void hist_get_y(int slot, Vec &v) const { hist_get(m + slot, v); }
  // s and y of one pair together: the partly filled last slot of both under ONE exec mask
  __device__ __forceinline__ void hist_get_sy(int slot, Vec &s, Vec &y) const {
    const int lane = lane_id();
    const Hist *ps = hist + slot * t.n + lane, *py = hist + (m + slot) * t.n + lane;
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      s.v[k] = Num(0);
      y.v[k] = Num(0);
      if (in_range(k, lane)) {
        s.v[k] = (Num)ps[k * kWave];
        y.v[k] = (Num)py[k * kWave];
      }
    }
  }
  // the 2m wave-uniform scalars of the two-loop recursion (rho, alpha): entry i lives in lane i of one register
  // pair, written with a select and read back with v_readlane -- no LDS round trip on the recursion's dependent chain
  Num sreg = Num(0);
  __device__ __forceinline__ void sput(int i, double v) { sreg = (lane_id() == i) ? (Num)v : sreg; }
  __device__ __forceinline__ double sget(int i) const { return (double)rdlane(sreg, i); }
  // scalar i minus b, in the backend's arithmetic (alpha_k - beta of the two-loop recursion's second loop)
  __device__ __forceinline__ double sdiff(int i, double b) const { return (double)(rdlane(sreg, i) - (Num)b); }
  // rho_slot * (a . b) in the backend's arithmetic (the two-loop recursion's coefficient)
  __device__ __forceinline__ double rho_dot(int slot, const Vec &a, const Vec &b) const {
    Num s = Num(0);
#pragma unroll
    for (int k = 0; k < NS; ++k) s += a.v[k] * b.v[k];
    return (double)(rdlane(sreg, slot) * wave_sum(s));
  }
  // ---- the two-loop recursion two pairs at a time (all-fp32 mode; round 5).  The textbook recursion is 2 * col DEPENDENT
  // steps "read (s, y), dot, wave-wide sum, axpy": a_k = rho_k s_k . q_k needs q_k = q_{k+1} - a_{k+1} y_{k+1}.  Written out for
  // two steps, a_{k-1} = rho_{k-1} (s_{k-1} . q_k - a_k s_{k-1} . y_k): the three dots s_k . q, s_{k-1} . q, s_{k-1} . y_k do not
  // depend on each other, go through ONE batched reduction (wave_sum4) and give both coefficients; likewise the second
  // loop with y_k . r, y_{k+1} . r, y_{k+1} . s_k.  col dependent reductions instead of 2 * col, ~35 vector instructions a
  // pair of steps instead of ~90.  Same mathematics, another rounding (the fp64 modes keep the textbook form: their
  // iterates are pinned to SciPy's, tests/test_lbfgs_host.py).
  static constexpr bool kOwnDirection = sizeof(Num) == 4;
  __device__ __forceinline__ void direction(const Vec &g, Vec &d, int col, int head, int mm, Num theta) {
    if (col == 0) {
      neg(d, g);
      return;
    }
    auto slot_of = [&](int k) { return head + k < mm ? head + k : head + k - mm; };
    auto dot3 = [&](const Vec &a0, const Vec &b0, const Vec &a1, const Vec &b1, const Vec &a2, const Vec &b2, Num &t0, Num &t1,
                    Num &t2) {
      Num p0 = Num(0), p1 = Num(0), p2 = Num(0);
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        p0 += a0.v[k] * b0.v[k];
        p1 += a1.v[k] * b1.v[k];
        p2 += a2.v[k] * b2.v[k];
      }
      Num unused;
      wave_sum4((float)p0, (float)p1, (float)p2, 0.0f, t0, t1, t2, unused);
    };
    copy(d, g);  // d plays q of the recursion
    Vec sA, yA, sB, yB;
    int k = col - 1;
    for (; k >= 1; k -= 2) {
      const int A = slot_of(k), B = slot_of(k - 1);
      hist_get_sy(A, sA, yA);
      hist_get_sy(B, sB, yB);
      Num t0, t1, t2;
      dot3(sA, d, sB, d, sB, yA, t0, t1, t2);
      const Num aA = rdlane(sreg, A) * t0;
      const Num aB = rdlane(sreg, B) * (t1 - aA * t2);
      sput(mm + A, (double)aA);
      sput(mm + B, (double)aB);
#pragma unroll
      for (int q = 0; q < NS; ++q) d.v[q] = (d.v[q] - aA * yA.v[q]) - aB * yB.v[q];
    }
    if (k == 0) {
      const int A = slot_of(0);
      hist_get_sy(A, sA, yA);
      const Num aA = (Num)rho_dot(A, sA, d);
      sput(mm + A, (double)aA);
      axpy(-(double)aA, yA, d);
    }
    scale(d, (double)(Num(1) / theta));
    k = 0;
    for (; k + 1 < col; k += 2) {
      const int A = slot_of(k), B = slot_of(k + 1);
      hist_get_sy(A, sA, yA);
      hist_get_sy(B, sB, yB);
      Num t0, t1, t2;
      dot3(yA, d, yB, d, yB, sA, t0, t1, t2);
      const Num cA = rdlane(sreg, mm + A) - rdlane(sreg, A) * t0;              // alpha_k - beta_k
      const Num cB = rdlane(sreg, mm + B) - rdlane(sreg, B) * (t1 + cA * t2);
#pragma unroll
      for (int q = 0; q < NS; ++q) d.v[q] = (d.v[q] + cA * sA.v[q]) + cB * sB.v[q];
    }
    if (k < col) {
      const int A = slot_of(k);
      hist_get_sy(A, sA, yA);
      const double b = rho_dot(A, yA, d);
      axpy(sdiff(mm + A, b), sA, d);
    }
    scale(d, -1.0);
  }

  // (line-search state and cost terms in LDS: in registers they spill, measured 14.0 against 15.5 ms at cfg2)
  __device__ __forceinline__ LineSearchT<Scalar> &ls() { return *lsp; }
  __device__ __forceinline__ double *cost_store() { return cst; }

  // diagnostics (neo_optimize_trace / neo_optimize_trace_xg): one record per counted evaluation -- (f, step, samples,
  // iteration) and, when asked for, the evaluated point and its gradient (replay tests: every evaluation of a run is laid
  // beside the CPU oracle's at the same point)
  __device__ __forceinline__ void note_eval(int nfev, int iter, double stp, double f, const Vec &x, const Vec &g) {
    // (trace_cap is 0 when neither buffer was given: ONE scalar test in the product path -- the two pointers tested
    //  here were four spilled scalar registers re-read at every evaluation)
    if (nfev > trace_cap) return;
    if (trace != nullptr && lane_id() == 0) {
      double *r = trace + (size_t)(nfev - 1) * 4;
      r[0] = f;
      r[1] = stp;
      r[2] = (double)last_ns;
      r[3] = (double)iter;
    }
    if (trace_xg != nullptr) {
      const int lane = lane_id();
      double *r = trace_xg + (size_t)(nfev - 1) * 2 * t.n;
#pragma unroll
      for (int k = 0; k < NS; ++k)
        if (k * kWave + lane < t.n) {
          r[k * kWave + lane] = (double)x.v[k];
          r[t.n + k * kWave + lane] = (double)g.v[k];
        }
    }
  }

  // FLAT x -> PIECE inputs
  __device__ __forceinline__ void scatter_x(const Vec &x) {
    // (opaque: the LDS addresses below are loop invariant -- hoisted out of the optimiser loop they become nine registers
    // that live across everything and are spilled; formed here they cost one instruction each)
    const int lane = opaque(lane_id());
    lds_wave_sync();
    Num *xn = reinterpret_cast<Num *>(xs);  // (the staging holds Num values)
#pragma unroll
    for (int k = 0; k < NS; ++k)
      if (in_range(k, lane)) xn[k * kWave + lane] = x.v[k];
    lds_wave_sync();
    const int M = t.M;
    const int p = opaque(LG::piece());
    const bool act = p < M;
    t.tau = act ? xn[t.nq + p] : Num(0);
#pragma unroll
    for (int d = 0; d < DL; ++d) {
      const int dg = LG::dim0() + d;  // the dimension: compile-time when the lane holds all of them
      t.P0[d] = (p == 0 || !act) ? bstate<D, LG, false>(t, 0, d) : xn[dg * (M - 1) + (p > 0 ? p - 1 : 0)];
      t.P1[d] = (p >= M - 1) ? bstate<D, LG, true>(t, 0, d) : xn[dg * (M - 1) + p];
    }
  }

  // one evaluation of cost and gradient (get_cost + get_grad, :539-585)
  __device__ __forceinline__ int eval(const Vec &x, double &f, Vec &g, double *costs) {
    const int lane = lane_id();
    const int p = LG::piece();
#ifdef NEO_STAMPS
    const long long s0 = wall_clock64();
#endif
    NEO_MARK("eval_begin");
    // Lane masks such as `piece < M` are loop invariant: hoisted out of the optimiser loop they are scalar register PAIRS
    // that live across everything, the allocator spills them to lanes of a vector register, and every use inside the
    // loop pays two v_readlane on the vector pipe (round 4: ~120 of them an evaluation).  The piece count is re-read
    // through an opaque copy at the head of every phase, so the masks are compared where they are used (one v_cmp).
    const int M_all = t.M;
    t.M = opaque_uniform(M_all);
    scatter_x(x);
    NEO_MARK("scatter_done");
    Num *xn = reinterpret_cast<Num *>(xs);
    double energy, tsum;
    const int st = minco_forward<D, LG, Num, kPcr>(t, prm, energy, tsum);
    NEO_MARK("forward_done");
#ifdef NEO_STAMPS
    const long long s1 = wall_clock64();
#endif
    if (st != 0) {
      f = 0.0;
#pragma unroll
      for (int k = 0; k < 4; ++k) costs[k] = 0.0;
      return st;
    }
    if (coeff_out != nullptr && p < t.M) {
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int d = 0; d < DL; ++d) coeff_out[(size_t)(6 * p + k) * D + LG::dim0() + d] = (double)t.c[k][d];
    }
    Num gC[6][DL], gT = Num(0);
    double cf, ck;
    {
      Real cr[6][DL], gCr[6][DL], gTr;
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int d = 0; d < DL; ++d) cr[k][d] = (Real)t.c[k][d];
#ifdef NEO_STAMPS
      // timing experiment (flags bit 1024): a buffer descriptor with zero records -- the range check drops every gather
      // (it returns 0 without touching memory) while the instruction stream stays: what the sample loop costs without
      // its memory latency
      MapT map_t = map;
      if constexpr (sizeof(MapT) == sizeof(Map3D))
        if (prm.dbg & 1024) map_t.bytes = 0;
      LookupT lk(map_t);
#else
      LookupT lk(map);
#endif
      // lanes in proportion to the pieces' sample counts (xs is free between scatter_x and the gradient gather);
      // the assignment wants the sample count of piece l in lane l
      int ns_by_piece = t.ns;
      if constexpr (LG::S > 1) ns_by_piece = __shfl(t.ns, min(LG::S * lane, kWave - 1), kWave);
      NEO_MARK("assign_begin");
      t.M = opaque_uniform(M_all);
      SampleLanes sl;
      bool cached = false;
      const int ns_key = lane < t.M ? ns_by_piece : 0;  // (what balanced_sample_lanes sees of this lane)
      if (sl_cache != nullptr) cached = __ballot(ns_key == (int)reinterpret_cast<const unsigned char *>(sl_cache)[lane]) == ~0ull;
      if (cached) {
        const unsigned pk = (unsigned)sl_cache[kWave / 4 + lane];
        sl.piece = pk & 63;
        sl.r = (pk >> 6) & 63;
        sl.L = (pk >> 12) & 127;
        sl.act = ((pk >> 19) & 1) != 0;
        sl.first = (pk >> 20) & 63;
        sl.Lp = pk >> 26;  // (<= 32 here: Lp = 64 means a single piece, which has no entry -- see the store)
        sl.rounds = __builtin_amdgcn_readfirstlane(sl_cache[kWave / 4 + kWave]);
        sl.lmax = __builtin_amdgcn_readfirstlane(sl_cache[kWave / 4 + kWave + 1]);
        sl.total = __builtin_amdgcn_readfirstlane(sl_cache[kWave / 4 + kWave + 2]);
      } else {
        sl = balanced_sample_lanes(t.M, ns_by_piece, reinterpret_cast<int *>(xs));
        if (sl_cache != nullptr) {
          // one byte a lane for the key (sample counts stay below T_max / delta_t; 255 = never equal), one word for the
          // assignment; a piece with more than 63 lanes (M = 1) does not fit the word's six bits: not cached
          const bool fits = sl.lmax < 64 && ns_key < 255;
          reinterpret_cast<unsigned char *>(sl_cache)[lane] = (unsigned char)(__ballot(!fits) == 0 ? ns_key : 255);
          sl_cache[kWave / 4 + lane] = (int)((unsigned)sl.piece | ((unsigned)sl.r << 6) | ((unsigned)sl.L << 12) |
                                             ((sl.act ? 1u : 0u) << 19) | ((unsigned)sl.first << 20) | ((unsigned)sl.Lp << 26));
          if (lane == 0) {
            sl_cache[kWave / 4 + kWave] = sl.rounds;
            sl_cache[kWave / 4 + kWave + 1] = sl.lmax;
            sl_cache[kWave / 4 + kWave + 2] = sl.total;
          }
          lds_wave_sync();
        }
      }
      last_ns = sl.total;  // (the lane assignment has summed the sample counts already)
      samples += (long long)last_ns;
      NEO_MARK("assign_done");
      minco_sample<Real, D, LookupT, SU, false, LG>(t.M, sl, t.ns, cr, prm, lk, gCr, gTr, cf, ck,
                                                    fold_rows ? reinterpret_cast<Real *>(xs) : nullptr, fold_acc);
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int d = 0; d < DL; ++d) gC[k][d] = (Num)gCr[k][d];
      gT = (Num)gTr;
    }
#ifdef NEO_STAMPS
    const long long s2 = wall_clock64();
#endif
    costs[0] = uniform(energy);
    costs[1] = uniform(tsum);
    costs[2] = uniform(cf);
    costs[3] = uniform(ck);
    f = uniform(costs[0] * prm.w[0] + costs[1] * prm.w[1] + costs[2] * prm.w[2] + costs[3] * prm.w[3]);
    Num gq[DL], gtau;
    NEO_MARK("sample_done");
    t.M = opaque_uniform(M_all);
    const int bst = minco_backward<D, LG, Num, kPcr>(t, prm, gC, gT, gq, gtau);
    NEO_MARK("backward_done");
    if (bst != 0) return bst;
    // PIECE -> FLAT (addresses formed here, as in scatter_x)
    lds_wave_sync();
    const int pg = opaque(p), lg = opaque(lane);
    if (pg >= 1 && pg < t.M) {
#pragma unroll
      for (int d = 0; d < DL; ++d) xn[(LG::dim0() + d) * (t.M - 1) + pg - 1] = gq[d];
    }
    if (pg < t.M && LG::dim0() == opaque_uniform(0)) xn[t.nq + pg] = gtau;
    lds_wave_sync();
#pragma unroll
    for (int k = 0; k < NS; ++k) g.v[k] = in_range(k, lg) ? xn[k * kWave + lg] : Num(0);
    NEO_MARK("gather_done");
#ifdef NEO_STAMPS
    const long long s3 = wall_clock64();
    tk[0] += s1 - s0;
    tk[1] += s2 - s1;
    tk[2] += s3 - s2;
    tk[3] += 1;
#endif
    return 0;
  }
};

// ------------------------------------------------------------------ suspended runs (neo_optimize_batch_budget_dev)
// A launch with an evaluation budget stops a run that needs more: everything the optimiser carries between two
// evaluations -- the machine's vectors and scalars, the line search, the stored pairs and their rho, the cost terms --
// goes to `st` (doubles: every fp32 value converts exactly both ways) and a later launch picks the run up at the very
// evaluation it was about to make.  Layout: [64 scalars | x g t r d (n each) | rho / alpha lanes (64) | pairs (2 m n)].
__host__ __device__ inline size_t opt_state_doubles(int n, int m) { return 64 + 5 * (size_t)n + kWave + 2 * (size_t)m * (size_t)n; }

template <class BE, class Mach>
__device__ __forceinline__ void save_run(BE &be, const Mach &mc, double *st) {
  const int lane = lane_id(), n = be.t.n;
  lds_wave_sync();
  if (lane == 0) {
    const int iv[9] = {mc.phase, mc.status, mc.nfev, mc.nit, mc.iter, mc.col, mc.head, mc.task, mc.ifun};
#pragma unroll
    for (int k = 0; k < 9; ++k) st[k] = (double)iv[k];
    const double dv[7] = {(double)mc.f, (double)mc.fold, (double)mc.stp, (double)mc.gd, (double)mc.gdold, (double)mc.theta,
                          (double)mc.stp_evaluated};
#pragma unroll
    for (int k = 0; k < 7; ++k) st[9 + k] = dv[k];
    st[16] = (double)be.samples;
    st[17] = (double)be.last_ns;
    const auto &L = *be.lsp;
    const double lv[20] = {(double)L.ftol, (double)L.gtol, (double)L.xtol, (double)L.stpmin, (double)L.stpmax, (double)L.brackt,
                           (double)L.stage, (double)L.ginit, (double)L.gtest, (double)L.gx, (double)L.gy, (double)L.finit,
                           (double)L.fx, (double)L.fy, (double)L.stx, (double)L.sty, (double)L.stmin, (double)L.stmax,
                           (double)L.width, (double)L.width1};
#pragma unroll
    for (int k = 0; k < 20; ++k) st[20 + k] = lv[k];
#pragma unroll
    for (int k = 0; k < 12; ++k) st[40 + k] = be.cst[k];
  }
  double *v = st + 64;
#pragma unroll
  for (int k = 0; k < BE::kSlotsN; ++k)
    if (k * kWave + lane < n) {
      const int e = k * kWave + lane;
      v[e] = (double)mc.x.v[k];
      v[n + e] = (double)mc.g.v[k];
      v[2 * n + e] = (double)mc.t.v[k];
      v[3 * n + e] = (double)mc.r.v[k];
      v[4 * n + e] = (double)mc.d.v[k];
    }
  v[5 * n + lane] = (double)be.sreg;
  double *h = v + 5 * n + kWave;
  for (int i = lane; i < 2 * be.m * n; i += kWave) h[i] = (double)be.hist[i];
}

template <class BE, class Mach>
__device__ __forceinline__ void load_run(BE &be, Mach &mc, const double *st) {
  using Num = decltype(be.sreg);
  const int lane = lane_id(), n = be.t.n;
  auto iu = [&](int k) { return (int)uniform(st[k]); };
  mc.phase = iu(0); mc.status = iu(1); mc.nfev = iu(2); mc.nit = iu(3); mc.iter = iu(4);
  mc.col = iu(5); mc.head = iu(6); mc.task = iu(7); mc.ifun = iu(8);
  using S = typename Mach::S;  // (every fp32 scalar went to the state as a double: exact both ways)
  mc.f = (S)uniform(st[9]); mc.fold = (S)uniform(st[10]); mc.stp = (S)uniform(st[11]); mc.gd = (S)uniform(st[12]);
  mc.gdold = (S)uniform(st[13]); mc.theta = (S)uniform(st[14]); mc.stp_evaluated = (S)uniform(st[15]);
  be.samples = (long long)uniform(st[16]);
  be.last_ns = iu(17);
  if (lane == 0) {
    auto &L = *be.lsp;
    L.ftol = (S)st[20]; L.gtol = (S)st[21]; L.xtol = (S)st[22]; L.stpmin = (S)st[23]; L.stpmax = (S)st[24];
    L.brackt = (int)st[25]; L.stage = (int)st[26];
    L.ginit = (S)st[27]; L.gtest = (S)st[28]; L.gx = (S)st[29]; L.gy = (S)st[30]; L.finit = (S)st[31]; L.fx = (S)st[32];
    L.fy = (S)st[33]; L.stx = (S)st[34]; L.sty = (S)st[35]; L.stmin = (S)st[36]; L.stmax = (S)st[37]; L.width = (S)st[38];
    L.width1 = (S)st[39];
#pragma unroll
    for (int k = 0; k < 12; ++k) be.cst[k] = st[40 + k];
  }
  const double *v = st + 64;
#pragma unroll
  for (int k = 0; k < BE::kSlotsN; ++k) {
    const int e = k * kWave + lane;
    const bool in = e < n;
    mc.x.v[k] = in ? (Num)v[e] : Num(0);
    mc.g.v[k] = in ? (Num)v[n + e] : Num(0);
    mc.t.v[k] = in ? (Num)v[2 * n + e] : Num(0);
    mc.r.v[k] = in ? (Num)v[3 * n + e] : Num(0);
    mc.d.v[k] = in ? (Num)v[4 * n + e] : Num(0);
  }
  be.sreg = (Num)v[5 * n + lane];
  const double *h = v + 5 * n + kWave;
  for (int i = lane; i < 2 * be.m * n; i += kWave) be.hist[i] = (typename BE::Hist)h[i];
  lds_wave_sync();
}

// bnd_lds: 6 * D elements of LDS for the layouts with one dimension per lane (stage_boundary), unused otherwise
template <class LG = WaveLanes, int D, int DL, typename Num>
__device__ __forceinline__ void load_boundary(Traj<D, DL, Num> &t, const double *head, const double *tail, int M,
                                              Num *bnd_lds = nullptr) {
  t.M = M;
  t.nq = D * (M - 1);
  t.n = t.nq + M;
  t.L = sample_lanes_per_piece(M);
  t.head = head;
  t.tail = tail;
  stage_boundary<D, LG>(t, bnd_lds);
}

struct MapTable {
  const void *maps;  // array of MapT indexed by scene slot
};

// ------------------------------------------------------------------ kernels
template <int D, int NS, typename Real, class MapT, class LookupT, class LG = WaveLanes, typename Num = double>
__global__ __launch_bounds__(kWave) void eval_kernel(int B, int M, DevParams prm, MapT map,
                                                      const double *__restrict__ x,
                                                      const double *__restrict__ head,
                                                      const double *__restrict__ tail,
                                                      double *__restrict__ cost, double *__restrict__ costs4,
                                                      double *__restrict__ grad, double *__restrict__ coeffs,
                                                      int *__restrict__ status) {
  __shared__ __attribute__((aligned(16))) double xs[stage_doubles<D, NS, Real>()];
  __shared__ LineSearchT<opt_scalar_t<Num, NS>> lsm;
  __shared__ double cst[12];
  __shared__ Num bnd[6 * D];
  // all-fp32 mode, lane = (piece, dimension): the cyclic reduction's multipliers stay in LDS for the adjoint pass
  constexpr bool kKeep = sizeof(Num) == 4 && LG::S > 1;
  __shared__ __attribute__((aligned(16))) Num mult_s[kKeep ? (5 * 8 + 4) * (kWave / LG::S + 1) : 4];
  const int b = blockIdx.x;
  if (b >= B) return;
  using BE = DevBackend<D, NS, Real, MapT, LookupT, NEO_FUSED_U, LG, false, Num>;
  BE be(prm, map);
  if constexpr (kKeep) be.t.pcr_mult = mult_s;
  be.xs = xs;
  if constexpr (sizeof(Num) == 4 && LG::W == kWave)   // (the staging doubles as the cyclic reduction's exchange table)
    if (pcr_xch_elems(M, LG::dl(D)) * (int)sizeof(Num) <= stage_doubles<D, NS, Real>() * 8) be.t.pcr_xch = reinterpret_cast<Num *>(xs);
  be.lsp = &lsm;
  be.cst = cst;
  be.hist = nullptr;
  be.m = NEO_LBFGS_M;
  load_boundary<LG>(be.t, head + (size_t)b * 3 * D, tail + (size_t)b * 3 * D, M, bnd);
  const int n = be.t.n;
  be.npad = NS * kWave;
  be.coeff_out = coeffs ? coeffs + (size_t)b * 6 * M * D : nullptr;
  const int lane = lane_id();
  typename BE::Vec xv, gv;
#pragma unroll
  for (int k = 0; k < NS; ++k) xv.v[k] = (k * kWave + lane < n) ? x[(size_t)b * n + k * kWave + lane] : 0.0;
  double f;
  double *costs = cst;
  const int st = be.eval(xv, f, gv, costs);
#pragma unroll
  for (int k = 0; k < NS; ++k)
    if (k * kWave + lane < n) grad[(size_t)b * n + k * kWave + lane] = gv.v[k];
  if (lane == 0) {
    cost[b] = f;
#pragma unroll
    for (int k = 0; k < 4; ++k) costs4[(size_t)b * 4 + k] = costs[k];
    if (status) status[b] = st;
  }
}

// WAVES = wavefronts per SIMD the register allocation aims at.  1: the whole file (256 VGPRs + AGPRs) for one
// trajectory -- the shortest evaluation, for batches that leave SIMDs to spare.  2: half the file (the cfg2
// instantiation fits it without spills) -- each evaluation is slower, but two trajectories share a SIMD's issue
// slots, which wins once the batch queues for the 1024 SIMDs anyway.  Same source, same arithmetic, bit-identical
// results.
// Dynamic LDS (launch parameter): `stage` doubles of staging (DevBackend::xs) followed by the 2 * maxcor * n doubles of
// the L-BFGS pairs.  The launcher gives the staging its full size (room for the rows of the per-piece fold) unless that
// would cost a wavefront of occupancy -- then NS * 64 doubles, and the fold runs in registers.
// BUDGET (neo_optimize_batch_budget_dev; instantiated in neo_disp_opt3d_b.hip only): a run that has not terminated after
// `budget` evaluations in this launch is suspended -- its state to `state`, status NEO_TRAJ_SUSPENDED -- and a launch
// with `resume` set continues the suspended runs of its trajectories (the others it leaves untouched).
template <int D, int NS, typename Real, class MapT, class LookupT, int WAVES, class LG = WaveLanes, typename Num = double,
          bool BUDGET = false>
__global__ __launch_bounds__(kWave, (WAVES == 2 ? (sizeof(Num) == 4 && NS <= 2 ? NEO_X_OCC : NEO_W2_OCC) : 1)) void optimize_kernel(int B, int M, DevParams prm, const MapT *maps,
                                                          const int *__restrict__ scene_slot, int nmaps,
                                                          const double *x0, double *x,
                                                          const double *__restrict__ head,
                                                          const double *__restrict__ tail,
                                                          double *__restrict__ costs4,
                                                          double *__restrict__ costs4_last,
                                                          int *__restrict__ nit, int *__restrict__ nfev,
                                                          int *__restrict__ status,
                                                          long long *__restrict__ nsamples,
                                                          const int *__restrict__ order, double *__restrict__ trace,
                                                          double *__restrict__ trace_xg, int trace_cap, int stage, int pcr_off,
                                                          double *state, int state_doubles, int budget, int resume,
                                                          int traj_total) {
  extern __shared__ __attribute__((aligned(16))) double dyn_lds[];
  __shared__ LineSearchT<opt_scalar_t<Num, NS>> lsm;
  __shared__ double cst[12];
  __shared__ Num bnd[6 * D];
  // the all-fp32 kernels keep the last lane assignment (DevBackend::sl_cache; launch_opt counts its 336 bytes)
  constexpr bool kSlCache = sizeof(Num) == 4;
  __shared__ __attribute__((aligned(8))) int slc[kSlCache ? kSlCacheInts : 2];
  if ((int)blockIdx.x >= B) return;
  // workgroups are dispatched in index order: `order` lets the caller start the runs it expects to
  // be long first (list scheduling: a long run that starts last sets the duration of the launch)
  const int b = order ? order[blockIdx.x] : (int)blockIdx.x;
  if constexpr (BUDGET)  // (a subset launch names its trajectories: an entry outside the arrays is skipped)
    if (b < 0 || b >= traj_total) return;
  // the two-waves variant has half the registers: two samples per lane in flight instead of four (with the lean
  // Horner form, cfg2 with three batches in flight: 414 k -> 541 k traj/s; scratch 640 -> 336 B per lane)
  using BE = DevBackend<D, NS, Real, MapT, LookupT, (WAVES == 2 ? NEO_W2_U : NEO_FUSED_U), LG,
                        pairs_in_f32<Real, NS, WAVES>() || sizeof(Num) == 4, Num>;
  // a slot outside the table (a stale or foreign slot array): the trajectory is left untouched and flagged
  const int slot = scene_slot ? scene_slot[b] : 0;
  if (slot < 0 || slot >= nmaps) {
    if (lane_id() == 0) {
      status[b] = NEO_TRAJ_BAD_SCENE;
      nit[b] = 0;
      nfev[b] = 0;
      if constexpr (!BUDGET)  // (the progress counter counts every trajectory of the launch: below)
        if (state != nullptr) __hip_atomic_fetch_add(reinterpret_cast<int *>(state), 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    return;
  }
  const MapT map = maps[slot];
  BE be(prm, map);
  be.xs = dyn_lds;
  be.fold_rows = stage >= stage_doubles<D, NS, Real>();
  // pcr_off != 0 (all-fp32 mode, lane = (piece, dimension), launch_opt): the multipliers of the forward cyclic reduction
  // live at dyn_lds + pcr_off (doubles) for the adjoint pass, and the staging holds per-piece accumulators for the fold
  if constexpr (sizeof(Num) == 4 && LG::S > 1) {
    if (pcr_off != 0) {
      be.t.pcr_mult = reinterpret_cast<Num *>(dyn_lds + pcr_off);
      be.fold_rows = true;
      be.fold_acc = true;
    }
  }
  if constexpr (sizeof(Num) == 4 && LG::W == kWave)   // (the staging doubles as the cyclic reduction's exchange table)
    if (be.fold_rows && pcr_xch_elems(M, LG::dl(D)) * (int)sizeof(Num) <= stage * 8) be.t.pcr_xch = reinterpret_cast<Num *>(dyn_lds);
  be.lsp = &lsm;
  be.cst = cst;
  be.m = NEO_LBFGS_M;
  be.coeff_out = nullptr;
  if constexpr (kSlCache) {
    if (lane_id() < kWave / 4) slc[lane_id()] = -1;  // (key bytes 255: the first evaluation misses)
    be.sl_cache = slc;
  }
  be.trace = trace ? trace + (size_t)b * trace_cap * 4 : nullptr;
  be.trace_cap = (trace != nullptr || trace_xg != nullptr) ? trace_cap : 0;
  be.trace_xg = trace_xg ? trace_xg + (size_t)b * trace_cap * 2 * (D * (M - 1) + M) : nullptr;
  load_boundary<LG>(be.t, head + (size_t)b * 3 * D, tail + (size_t)b * 3 * D, M, bnd);
  const int n = be.t.n;
  be.npad = NS * kWave;
  be.hist = reinterpret_cast<typename BE::Hist *>(dyn_lds + stage);
  const int lane = lane_id();
  typename BE::Vec xv;
#pragma unroll
  for (int k = 0; k < NS; ++k) xv.v[k] = (k * kWave + lane < n) ? x0[(size_t)b * n + k * kWave + lane] : 0.0;
  LbfgsOpts o{prm.ftol, prm.gtol, prm.maxls, prm.maxiter, prm.maxfun, NEO_LBFGS_M};
  LbfgsResult res;
#ifdef NEO_STAMPS
  const long long k0 = wall_clock64();
#endif
  // n <= 128: the run as "evaluate, then advance" (neo_lbfgs_sm.hpp: the same arithmetic and decisions as
  // lbfgs_minimize) -- ONE inlined copy of the evaluation instead of two: 36 % less code and no spills in the cfg2
  // two-waves kernel.  Four FLAT slots (n > 128, cfg5): the compiler keeps the machine's vectors in private memory
  // (1 KB of scratch, 3x slower), so those instantiations run the straight-line form.
  bool suspended = false;
  // (round 5, measured and dropped: wavefronts raising their own issue priority -- s_setprio -- as their evaluations pass
  //  128 / 256 / 384: 1.444 - 1.451 M traj/s against 1.439 - 1.440 M, a single batch unchanged: HISTORY.md)
  if constexpr (NS <= NEO_SM_MAX_SLOTS) {
    LbfgsMachine<BE> mach(be, o);
    if constexpr (BUDGET) {
      double *st = state + (size_t)b * (size_t)state_doubles;
      if (resume) {
        if (status[b] != NEO_TRAJ_SUSPENDED) return;  // (finished in an earlier launch: its results stay as they are)
        load_run(be, mach, st);
      } else {
        mach.x = xv;
        mach.begin();
      }
      int evals = 0;
      while (mach.need_eval()) {
        if (evals >= budget) {
          save_run(be, mach, st);
          suspended = true;
          break;
        }
        double fe;
        const int est = be.eval(mach.x, fe, mach.g, mach.costs());
        mach.f = (typename BE::Scalar)fe;
        mach.advance(est);
        ++evals;
      }
    } else {
      mach.x = xv;
      mach.begin();
      while (mach.need_eval()) {
        double fe;
        const int est = be.eval(mach.x, fe, mach.g, mach.costs());
        mach.f = (typename BE::Scalar)fe;
        mach.advance(est);
      }
    }
    mach.result(res);
    xv = mach.x;
  } else {
    lbfgs_minimize<BE>(be, xv, o, res);
  }
#ifdef NEO_STAMPS
  if (lane == 0 && nsamples) {  // the counter buffer is [B][8] in this build
    long long *o8 = nsamples + (size_t)b * 8;
    o8[1] = be.tk[3]; o8[2] = be.tk[0]; o8[3] = be.tk[1]; o8[4] = be.tk[2];
    o8[5] = wall_clock64() - k0; o8[6] = k0; o8[7] = be.tl;  // (7: two-loop time; was the dispatch slot)
    o8[0] = be.samples;
  }
  nsamples = nullptr;
#endif
#pragma unroll
  for (int k = 0; k < NS; ++k)
    if (k * kWave + lane < n) x[(size_t)b * n + k * kWave + lane] = xv.v[k];
  if (lane == 0) {
    int st = res.status;
    // weighted collision cost of the last evaluated x against the tolerance (:233-237)
    if (res.costs_last[3] * prm.w[3] > prm.coll_tol) st |= NEO_TRAJ_FLAG_COLLISION;
    if (suspended) st = NEO_TRAJ_SUSPENDED;  // (x = the point the run evaluates next, costs4 = the terms at its last iterate)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      costs4[(size_t)b * 4 + k] = res.costs[k];
      if (costs4_last) costs4_last[(size_t)b * 4 + k] = res.costs_last[k];
    }
    nit[b] = res.nit;
    nfev[b] = res.nfev;
    status[b] = st;
    if (nsamples) nsamples[b] = be.samples;
  }
  // Progress counter (neo_optimize_progress_counter, round 6; plain launches only -- their `state` argument carries it): one
  // system-scope release add per finished trajectory, after every lane's result stores.  A host that polls the counter
  // knows how many trajectories are complete while the launch's long runs are still going, and may read x / status of those
  // (status[b] != NEO_TRAJ_RUNNING, preset by the caller) through a copy on another stream.
  if constexpr (!BUDGET) {
    if (state != nullptr && lane == 0)
      __hip_atomic_fetch_add(reinterpret_cast<int *>(state), 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// add_sampled_cost + add_sampled_grad_CT (expert_planner.py:392-466) as a kernel of its own: the ESDF
// lookup kernel.  Input: polynomial coefficients and durations; output: the two sampled cost terms and
// their partials w.r.t. coefficients and durations.  One wavefront per trajectory; every lane reads the
// coefficients of its piece straight into the SAMPLE layout and the first lane of each piece writes the
// piece's partials.  fp32: <= 128 VGPRs, so four waves per SIMD -- the whole cfg2 batch is resident at once and
// the gathers of different trajectories overlap (one sample per lane in flight is enough then).
#ifndef NEO_SAMPLE_U
#define NEO_SAMPLE_U 1
#endif
#ifndef NEO_SAMPLE_OCC
// waves per SIMD the fp32 instantiations are allocated for.  Round 6: five (94 registers suffice): the launch over a whole
// step's 163 840 requests 921 -> 731 us (more gathers in flight per CU); the 4096 launch, whose wavefronts are all
// resident at four per SIMD anyway, is unchanged (23.9 us either way, same box)
#define NEO_SAMPLE_OCC 5
#endif
// the body of sample_kernel (one wavefront per trajectory, lanes per piece) as a function: also the fallback path of
// sample_chunk_kernel (neo_sample_chunk.hpp)
// (round 5, measured and not adopted -- HISTORY.md: the coefficients loaded in the PIECE layout next to the durations and
//  handed to the sample lanes through an LDS table, to take the coefficient load off the chain "durations -> lane
//  assignment -> coefficients": 28.3 us against 27.9 us per 4096 launch; shared tail lanes, 10 rounds instead of 13 at a
//  fresh guess: 31.1 us, tools/probe/neo_sample_shared.hpp.  Round 6, same verdict -- HISTORY.md round 6 (1): the samples as ONE
//  sequence dealt round-robin to the lanes with a candidate list, tools/probe/seq/; floor / clamp / brick address as single
//  instructions through inline assembly: 8 % FEWER vector instructions and 10 % slower, 26.3 against 23.8 us)
// IO: element type of the coefficient / partials buffers (double: neo_sampled_terms_batch[_dev]; float: the _f32 entry
// points of the fp32 sampling path, round 6: half the operand bytes and no conversions)
template <int D, typename Real, class MapT, class LookupT, typename IO = double>
__device__ __forceinline__ void sample_wave_per_piece(int b, int M, const DevParams &prm, const MapT &map,
                                                      const IO *__restrict__ coeffs, const double *__restrict__ ts,
                                                      double *__restrict__ costs2, IO *__restrict__ grad_C,
                                                      IO *__restrict__ grad_T, int *seg, Real *rows) {
  typedef IO IOPair __attribute__((ext_vector_type(2)));
  const int lane = lane_id();
  // lanes in proportion to the pieces' sample counts, as in the fused kernels
  const double Tp = lane < M ? ts[(size_t)b * M + lane] : 1.0;
  const int ns_p = lane < M ? (int)(Tp / prm.delta_t) : 0;
  const SampleLanes sl = balanced_sample_lanes(M, ns_p, seg);
  const int piece = sl.piece, r = sl.r;
  const bool act = sl.act;
  const double T = act ? ts[(size_t)b * M + piece] : 1.0;
  const int ns = act ? (int)(T / prm.delta_t) : 0;
  Real c[6][D], gC[6][D], gT;
  {
    // the 6*D elements of a piece are contiguous and pair-aligned (6*D is even)
    const IOPair *src = reinterpret_cast<const IOPair *>(coeffs + ((size_t)b * 6 * M + 6 * (act ? piece : 0)) * D);
#pragma unroll
    for (int q = 0; q < 3 * D; ++q) {
      const IOPair v = src[q];
      const int e0 = 2 * q, e1 = 2 * q + 1;
      c[e0 / D][e0 % D] = act ? (Real)v.x : Real(0);
      c[e1 / D][e1 % D] = act ? (Real)v.y : Real(0);
    }
  }
  double cf, ck;
  LookupT lk(map);
  minco_sample<Real, D, LookupT, NEO_SAMPLE_U, true>(M, sl, ns, c, prm, lk, gC, gT, cf, ck, sizeof(Real) == 4 ? rows : nullptr);
  if (act && r == 0) {
    IOPair *dst = reinterpret_cast<IOPair *>(grad_C + ((size_t)b * 6 * M + 6 * piece) * D);
#pragma unroll
    for (int q = 0; q < 3 * D; ++q) {
      const int e0 = 2 * q, e1 = 2 * q + 1;
      dst[q] = IOPair{(IO)gC[e0 / D][e0 % D], (IO)gC[e1 / D][e1 % D]};
    }
    grad_T[(size_t)b * M + piece] = (IO)gT;
  }
  if (lane < M && ns_p == 0) {
    // a piece shorter than delta_t has no sample and no sample lane: its partials are zeros (the planner's durations
    // never are -- T > T_min = 5 delta_t --, a caller of neo_sampled_terms_batch may pass any; found by
    // tests/test_gpu_parity.py::test_sampled_terms_on_ragged_durations: the rows were left as the caller's buffer had them)
    IOPair *dst = reinterpret_cast<IOPair *>(grad_C + ((size_t)b * 6 * M + 6 * lane) * D);
#pragma unroll
    for (int q = 0; q < 3 * D; ++q) dst[q] = IOPair{IO(0), IO(0)};
    grad_T[(size_t)b * M + lane] = IO(0);
  }
  if (lane == 0) {
    costs2[(size_t)b * 2 + 0] = cf;
    costs2[(size_t)b * 2 + 1] = ck;
  }
}

template <int D, typename Real, class MapT, class LookupT, typename IO = double>
__global__ __launch_bounds__(kWave, sizeof(Real) == 4 ? NEO_SAMPLE_OCC : 2) void sample_kernel(int B, int M, DevParams prm, MapT map,
                                                                                  const IO *__restrict__ coeffs,
                                                                                  const double *__restrict__ ts,
                                                                                  double *__restrict__ costs2,
                                                                                  IO *__restrict__ grad_C,
                                                                                  IO *__restrict__ grad_T,
                                                                                  const int *__restrict__ order) {
  if ((int)blockIdx.x >= B) return;
  // workgroup i runs on XCD i mod 8 (round-robin dispatch) and each XCD has its own 4 MB L2: `order` lets the caller
  // hand every XCD requests that fly through the same part of the field (BatchPlanner.spatial_order); results stay in
  // the caller's order
  const int b = order ? order[blockIdx.x] : (int)blockIdx.x;
  __shared__ int seg[kWave];
  // rows of the per-piece fold (fp32 sampling, minco_sample)
  __shared__ __attribute__((aligned(16))) Real rows[sizeof(Real) == 4 ? kWave * 8 * D : 4];
  sample_wave_per_piece<D, Real, MapT, LookupT, IO>(b, M, prm, map, coeffs, ts, costs2, grad_C, grad_T, seg, rows);
}

}  // namespace neo
