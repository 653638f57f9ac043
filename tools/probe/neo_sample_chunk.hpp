// neo_sample_chunk.hpp -- the ESDF-lookup kernel (add_sampled_cost + add_sampled_grad_CT, expert_planner.py:392-466)
// with the quadrature samples of a trajectory dealt to the 64 lanes in CONTIGUOUS CHUNKS (gfx950 only).
//
// Why.  sample_kernel gives every piece its own lanes (balanced_sample_lanes: the smallest round count R with
// sum_p ceil(ns_p / R) <= 64).  Lanes are whole numbers: at the cfg2 initial guess (pieces of 37, 25 x 19, 37 samples,
// 549 in all) nine rounds would need 5 + 3 x 19 + 5 = 67 lanes, so the kernel runs THIRTEEN rounds with 44 lanes busy --
// 66 % of the lane-rounds it issues.  The kernel is bound by instruction issue (measured: putting more gathers in
// flight, by LDS-DMA or with more wavefronts per trajectory, made it slower), so idle lane-rounds
// are what there is to win.  Here lane l walks samples [l R, (l + 1) R) of the trajectory's sample sequence with
// R = ceil(S / 64): nine rounds, 61 lanes busy, 95 %.  A chunk may run over a piece boundary; it can do so at most once
// as long as R - 1 <= min_p ns_p (checked per trajectory; otherwise, and for pieces without samples, the trajectory takes
// the per-piece path of sample_kernel inside this kernel).  At the boundary the lane writes the partial sums of the piece
// it leaves to its LDS row, clears them and reads the next piece's coefficients from a table in LDS.  Per-piece totals:
// lane p adds, in lane order, the row of the lane that entered piece p mid-chunk and the rows of the lanes that follow
// up to the piece's last sample.  No atomics; fixed summation order; the same arithmetic per sample as every other
// sampling kernel (piece_pos_vel, Lookup::prepare / load / finish, sample_accumulate).  Consecutive samples of a lane
// are neighbours in space, so its gathers stay on one or two cache lines for several rounds.
// fp32 sampling; fp64 sampling (parity mode) keeps sample_kernel and the summation order its fixtures were made with.
//
// MEASURED AND NOT ADOPTED (round 3, MI355X, cfg2 workload; tools/gpu_sample_variants.py variant 1,
// profiles/r03_sample_variants.log): correct (1e-7 from sample_kernel: another order of the per-piece sums, bit-reproducible)
// and SLOWER -- 44.4 us per 4096 trajectories against 32.8, 576 us per 65 536 against 432 -- although it issues nine
// rounds instead of thirteen.  In sample_kernel the lanes of a piece take ADJACENT samples in the same round, so the 64
// addresses of one gather instruction fall on ~25 cache lines; here the lanes of a round are nine samples apart and
// every address is its own line (61 per instruction): the texture-address / L1 path does 2.4 times the work per
// instruction and the kernel turns from issue-bound to bound by that path.  Kept outside the library as the
// experiment it was (-DNEO_SAMPLE_EXPERIMENTS, NEO_SAMPLE_VARIANT=1).
#pragma once
#include "../../neo-planner_amd/csrc/neo_kernels.hpp"

namespace neo {

// floats per LDS row of the chunked kernel: the 6 D partials (or coefficients) of a piece, dT, padded to whole quads
template <int D>
__host__ __device__ constexpr int chunk_row_floats() { return (6 * D + 1 + 3) / 4 * 4; }
// dynamic LDS of sample_chunk_kernel in bytes: coefficient table [M], lane rows [64], boundary rows [M], 4 int tables
// of 64 -- and never less than the per-piece path needs (64 ints + 64 rows of [D][8])
template <int D>
__host__ __device__ constexpr size_t chunk_lds_bytes(int M) {
  const size_t mine = (size_t)(2 * M + kWave) * chunk_row_floats<D>() * sizeof(float) + 4 * kWave * sizeof(int);
  const size_t per_piece = (size_t)kWave * sizeof(int) + (size_t)kWave * 8 * D * sizeof(float);
  return mine > per_piece ? mine : per_piece;
}

template <int D, class MapT, class LookupT>
__global__ __launch_bounds__(kWave, NEO_SAMPLE_OCC) void sample_chunk_kernel(int B, int M, DevParams prm, MapT map,
                                                                             const double *__restrict__ coeffs,
                                                                             const double *__restrict__ ts,
                                                                             double *__restrict__ costs2,
                                                                             double *__restrict__ grad_C,
                                                                             double *__restrict__ grad_T) {
#pragma clang fp contract(on)  // fuse a*b+c only as written: the same arithmetic whatever the unrolling around it
  typedef float Real;
  typedef Real Quad __attribute__((ext_vector_type(4)));
  constexpr int NC = 6 * D, RW = chunk_row_floats<D>(), NQ = RW / 4;
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  const int b = blockIdx.x;
  if (b >= B) return;
  const int lane = lane_id();
  // ---- sample counts, their prefix sums, the round count
  const double Tp = lane < M ? ts[(size_t)b * M + lane] : 1.0;
  const int ns_p = lane < M ? (int)(Tp / prm.delta_t) : 0;  // int(T / delta_t) (:401)
  const int incl = wave_scan_add(ns_p);
  const int off_p = incl - ns_p;
  const int S = __builtin_amdgcn_readlane(incl, kWave - 1);
  const int R = (S + kWave - 1) / kWave;
  const int min_ns = (1 << 20) - wave_max_nonneg(lane < M ? (1 << 20) - ns_p : 0);
  if (S == 0 || R - 1 > min_ns || min_ns == 0) {
    // a chunk could run over two piece boundaries (or a piece has no samples): lanes per piece, as sample_kernel
    int *seg = reinterpret_cast<int *>(dyn);
    sample_wave_per_piece<D, Real, MapT, LookupT>(b, M, prm, map, coeffs, ts, costs2, grad_C, grad_T, seg,
                                                   reinterpret_cast<Real *>(seg + kWave));
    return;
  }
  Real *cft = dyn;                        // [M][RW]  coefficients of the pieces
  Real *rows0 = cft + (size_t)M * RW;     // [64][RW] partial sums of the piece a lane starts in (or its only piece)
  Real *rows1 = rows0 + (size_t)kWave * RW;  // [M][RW] ... of the piece a lane enters mid-chunk, indexed by rank
  int *t_off = reinterpret_cast<int *>(rows1 + (size_t)M * RW), *t_ns = t_off + kWave, *t_rank = t_ns + kWave;
  int *seg = t_rank + kWave;
  const float rcpR = __frcp_rn((float)R);
  // floor(x / R) for 0 <= x < 2^20: the half keeps the quotient off the integers (balanced_sample_lanes)
  auto div_R = [&](int x) { return (int)(((float)x + 0.5f) * rcpR); };
  // ---- tables: coefficients (lane p converts piece p's 6 D doubles), offsets, counts, rank among the pieces that start
  // inside a chunk
  const int l_first = div_R(off_p);
  const bool mid = lane < M && off_p - l_first * R != 0;
  const int rank_p = wave_scan_add(mid ? 1 : 0) - (mid ? 1 : 0);
  seg[lane] = 0;
  if (lane < M) {
    const double2 *src = reinterpret_cast<const double2 *>(coeffs + ((size_t)b * 6 * M + 6 * lane) * D);
    Real cc[RW];
#pragma unroll
    for (int q = 0; q < RW; ++q) cc[q] = Real(0);
#pragma unroll
    for (int q = 0; q < 3 * D; ++q) {
      const double2 v = src[q];
      cc[2 * q] = (Real)v.x;
      cc[2 * q + 1] = (Real)v.y;
    }
    Quad *dst = reinterpret_cast<Quad *>(cft + (size_t)lane * RW);
#pragma unroll
    for (int q = 0; q < NQ; ++q) dst[q] = Quad{cc[4 * q], cc[4 * q + 1], cc[4 * q + 2], cc[4 * q + 3]};
    t_off[lane] = off_p;
    t_ns[lane] = ns_p;
    t_rank[lane] = rank_p;
  }
  lds_wave_sync();
  // the piece lane l starts in = the last piece whose first sample is at or before l R: every piece marks the first
  // lane that starts at or after its first sample (several may mark one lane: the largest wins), prefix maximum
  if (lane < M) {
    const int tl = div_R(off_p + R - 1);
    if (tl < kWave) atomicMax(&seg[tl], lane + 1);
  }
  lds_wave_sync();
  int p = wave_scan_max_nonneg(seg[lane]) - 1;
  const int q0 = lane * R;
  const bool busy = q0 < S;
  if (!busy) p = 0;
  int off_cur = t_off[p], ns_cur = t_ns[p];
  Real c[6][D];
  {
    const Quad *src = reinterpret_cast<const Quad *>(cft + (size_t)p * RW);
    Real cc[RW];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const Quad v = src[q];
      cc[4 * q] = v.x; cc[4 * q + 1] = v.y; cc[4 * q + 2] = v.z; cc[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int e = 0; e < NC; ++e) c[e / D][e % D] = cc[e];
  }
  const LookupT lk(map);
  const Real dt = (Real)prm.delta_t, vmax2 = (Real)(prm.v_max * prm.v_max), safe = (Real)prm.safe_dis;
  const Real w2 = (Real)prm.w[2], w3 = (Real)prm.w[3];
  Real inv_ns = Real(1) / (Real)ns_cur;
  Real aC[6][D];
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int d = 0; d < D; ++d) aC[k][d] = Real(0);
  Real aT = Real(0), aF = Real(0), aK = Real(0);
  bool crossed = false;
  auto put_row = [&](Real *row) {
    Real f[RW];
#pragma unroll
    for (int q = 0; q < RW; ++q) f[q] = Real(0);
#pragma unroll
    for (int e = 0; e < NC; ++e) f[e] = aC[e / D][e % D];
    f[NC] = aT;
    Quad *dst = reinterpret_cast<Quad *>(row);
#pragma unroll
    for (int q = 0; q < NQ; ++q) dst[q] = Quad{f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]};
  };

  for (int k = 0; k < R; ++k) {
    const int q = q0 + k;
    const bool on = q < S;
    if (on && q >= off_cur + ns_cur) {
      // into the next piece: the partial sums of the piece left behind go to this lane's row
      put_row(rows0 + (size_t)lane * RW);
#pragma unroll
      for (int kk = 0; kk < 6; ++kk)
#pragma unroll
        for (int d = 0; d < D; ++d) aC[kk][d] = Real(0);
      aT = Real(0);
      crossed = true;
      p += 1;
      off_cur += ns_cur;
      ns_cur = t_ns[p];
      inv_ns = Real(1) / (Real)ns_cur;
      const Quad *src = reinterpret_cast<const Quad *>(cft + (size_t)p * RW);
      Real cc[RW];
#pragma unroll
      for (int qq = 0; qq < NQ; ++qq) {
        const Quad v = src[qq];
        cc[4 * qq] = v.x; cc[4 * qq + 1] = v.y; cc[4 * qq + 2] = v.z; cc[4 * qq + 3] = v.w;
      }
#pragma unroll
      for (int e = 0; e < NC; ++e) c[e / D][e % D] = cc[e];
    }
    const int j = q - off_cur;
    const Real s = (Real)((double)j * prm.delta_t);  // beta_full row j: t = j * delta_t (:251)
    // only the position is needed to issue the gathers; the velocity is evaluated while they fly
    Real pos[D], vel[D];
#pragma unroll
    for (int d = 0; d < D; ++d)
      pos[d] = fmaf(fmaf(fmaf(fmaf(fmaf(c[5][d], s, c[4][d]), s, c[3][d]), s, c[2][d]), s, c[1][d]), s, c[0][d]);
    const typename LookupT::Addr ad = lk.template prepare<D>(pos, on);
    const typename LookupT::Raw rw = lk.load(ad);
    {
      Real pos2[D];
      piece_pos_vel<Real, D>(c, s, pos2, vel);
    }
    Real v2 = Real(0);
#pragma unroll
    for (int d = 0; d < D; ++d) v2 += vel[d] * vel[d];
    const Real vv = v2 - vmax2;
    Real gdrop[D];
    const Real vd = safe - lk.template finish<D>(ad, rw, gdrop);
    if (on && (vv > Real(0) || vd > Real(0)))
      sample_accumulate<Real, D, LookupT>(c, j, ns_cur, s, inv_ns, vel, vv, vd, lk, ad, rw, dt, w2, w3, aC, aT, aF, aK);
  }
  if (busy) put_row(crossed ? rows1 + (size_t)t_rank[p] * RW : rows0 + (size_t)lane * RW);
  const double cost_f = (double)wave_sum(busy ? aF : Real(0)), cost_k = (double)wave_sum(busy ? aK : Real(0));
  lds_wave_sync();
  // ---- per-piece totals: lane p adds the rows that hold piece p, in lane order
  if (lane < M) {
    const int l_last = div_R(off_p + ns_p - 1);
    Quad acc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc[q] = Quad{Real(0), Real(0), Real(0), Real(0)};
    if (mid) {
      const Quad *src = reinterpret_cast<const Quad *>(rows1 + (size_t)rank_p * RW);
#pragma unroll
      for (int q = 0; q < NQ; ++q) acc[q] += src[q];
    }
    for (int l = l_first + (mid ? 1 : 0); l <= l_last; ++l) {
      const Quad *src = reinterpret_cast<const Quad *>(rows0 + (size_t)l * RW);
#pragma unroll
      for (int q = 0; q < NQ; ++q) acc[q] += src[q];
    }
    Real f[RW];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      f[4 * q] = acc[q].x; f[4 * q + 1] = acc[q].y; f[4 * q + 2] = acc[q].z; f[4 * q + 3] = acc[q].w;
    }
    double2 *dst = reinterpret_cast<double2 *>(grad_C + ((size_t)b * 6 * M + 6 * lane) * D);
#pragma unroll
    for (int q = 0; q < 3 * D; ++q) dst[q] = make_double2((double)f[2 * q], (double)f[2 * q + 1]);
    grad_T[(size_t)b * M + lane] = (double)f[NC];
  }
  if (lane == 0) {
    costs2[(size_t)b * 2 + 0] = cost_f;
    costs2[(size_t)b * 2 + 1] = cost_k;
  }
}

}  // namespace neo
