// neo_sample_wg.hpp -- the ESDF-lookup kernel (add_sampled_cost + add_sampled_grad_CT, expert_planner.py:392-466) with a
// WORKGROUP of WPT wavefronts per trajectory and the field gathers staged through LDS (gfx950 only).
//
// Why (measured on the MI355X).  sample_kernel gives a trajectory one wavefront: its ~550 samples
// are 13 dependent rounds of "position -> address -> gather -> interpolate -> penalty", one round's gathers in flight per
// lane, and a 4096-trajectory launch is bound by the latency of those rounds (VALU under 30 % busy, 2.3 TB/s of 8).
// Here
//   * 64 * WPT lanes share the samples of one trajectory: 13 rounds become 5 (WPT = 2) or 3 (WPT = 4);
//   * every lane puts the gathers of up to PF rounds in flight at once as LDS-DMA (`buffer_load_dwordx4 ... offen lds`,
//     Lookup3D::load_async): the 32 bytes of a lookup land in LDS without a destination register, so the number in
//     flight is set by the landing room (2 KB per wavefront and round), not by spare VGPRs -- the register-staged form
//     spilled from two rounds on ("more samples in flight");
//   * the per-piece sums go through LDS rows as in minco_sample, across the wavefronts of the workgroup (one barrier).
// Same arithmetic per sample as every other sampling kernel (sample_accumulate, piece_pos_vel, Lookup3D::prepare /
// finish); the lanes of a piece are summed in lane order, so results are bit-reproducible run to run -- they differ from
// sample_kernel's in the last bits only through the order of those sums (fewer, longer partial sums per piece).
// fp32 sampling on yz-quad fields (fp32 or fp16 voxels).
//
// MEASURED AND NOT ADOPTED (round 3, MI355X, cfg2 workload; tools/gpu_sample_variants.py, profiles/r03_sample_variants.log):
// every variant is correct (bit-identical to sample_kernel with one wavefront per trajectory, 2e-7 with more) and every
// one is SLOWER than sample_kernel's 33.0 us per 4096 trajectories / 432 us per 65 536: two rounds in flight 35.7 / 447,
// four 38.9 / 482, six 39.6 / 460, two wavefronts per trajectory 35.7-36.6 / 467-484, four 43.8 / 635.  The kernel is
// bound by instruction issue, not by the latency of its gathers; the product's answer is fewer idle lane-rounds
// (csrc/neo_sample_chunk.hpp).  Kept here, outside the library, as the experiment it was: a library built with
// -DNEO_SAMPLE_EXPERIMENTS dispatches to it when NEO_SAMPLE_VARIANT is set.
#pragma once
#include "../../neo-planner_amd/csrc/neo_kernels.hpp"

namespace neo {

// s_waitcnt vmcnt(N) alone (gfx9 encoding: vmcnt = simm16[15:14]:[3:0], expcnt [6:4] and lgkmcnt [11:8] left at "no wait")
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70);
}

template <int WPT>
struct WgLanes {  // which of the workgroup's 64 * WPT sample lanes walk which piece's samples
  int piece, r, L;
  bool act;
  int rounds;  // workgroup-uniform
  int lmax;    // workgroup-uniform: most lanes any piece has
};

// Lanes in proportion to the pieces' sample counts over the NL = 64 * WPT lanes of the workgroup (balanced_sample_lanes
// for one wavefront): the smallest round count R with sum_p ceil(ns_p / R) <= NL, piece p gets ceil(ns_p / R) adjacent
// lanes.  Every wavefront computes the same table for itself (`seg`: NL ints of LDS per wavefront, no barrier).
// ns_piece: lane p < M of EVERY wavefront holds ns_p.  Lp_out: lanes of piece `lane` (for the zero-sample case).
template <int WPT>
__device__ __forceinline__ WgLanes<WPT> wg_sample_lanes(int M, int ns_piece, int *seg, int wave, int &Lp_out) {
  constexpr int NL = kWave * WPT;
  const int lane = lane_id();
  const int mine = lane < M ? ns_piece : 0;
  const int total = wave_sum(mine);
  int R = max(1, (total + NL - 1) / NL);
  int Lp = 0;
  for (;;) {
    const float fr = (float)R;
    Lp = mine > 0 ? (int)(((float)mine + fr - 0.5f) * __frcp_rn(fr)) : 0;  // ceil(ns / R), see balanced_sample_lanes
    if (wave_sum(Lp) <= NL) break;
    ++R;
  }
  Lp_out = Lp;
  const int incl = wave_scan_add(Lp);
  const int start = incl - Lp;
#pragma unroll
  for (int k = 0; k < WPT; ++k) seg[k * kWave + lane] = 0;
  lds_wave_sync();
  if (Lp > 0) seg[start] = ((lane + 1) << 16) | (Lp << 8) | start;  // (start < 256, Lp <= 255)
  lds_wave_sync();
  // the segment global lane g = wave * 64 + lane falls in: the last start at or before g
  int before = 0;
  for (int k = 0; k < wave; ++k) before = max(before, wave_max_nonneg(seg[k * kWave + lane]));
  const int key = max(before, wave_scan_max_nonneg(seg[wave * kWave + lane]));
  WgLanes<WPT> sl;
  sl.piece = max((key >> 16) - 1, 0);
  sl.L = max((key >> 8) & 0xff, 1);
  sl.r = wave * kWave + lane - (key & 0xff);
  sl.act = key != 0 && sl.r < sl.L;
  sl.rounds = R;
  sl.lmax = wave_max_nonneg(Lp);
  return sl;
}

// OCC: wavefronts per SIMD the kernel is allocated for (4: 128 VGPRs, 3: 168)
template <class LookupT, int WPT, int PF, int OCC>
__global__ __launch_bounds__(kWave * WPT, OCC) void sample_wg_kernel(
    int B, int M, DevParams prm, Map3D map, const double *__restrict__ coeffs, const double *__restrict__ ts,
    double *__restrict__ costs2, double *__restrict__ grad_C, double *__restrict__ grad_T) {
#pragma clang fp contract(on)  // fuse a*b+c only as written: the same arithmetic whatever the unrolling around it
  constexpr int D = 3, NL = kWave * WPT;
  typedef float Real;
  typedef Real Quad __attribute__((ext_vector_type(4)));
  constexpr int kRow = 8 * D;                                       // floats per fold row: [d][8] = (aC[0..5][d], aT, 0)
  constexpr int kLand = PF * LookupT::kAsyncBytes;                  // landing bytes per wavefront
  constexpr int kRowBytes = kWave * kRow * (int)sizeof(Real);       // fold rows of one wavefront's lanes
  constexpr int kWaveBytes = kLand > kRowBytes ? kLand : kRowBytes;  // a wavefront's rows reuse its own landing area
  __shared__ __attribute__((aligned(16))) char stage[WPT * kWaveBytes];
  __shared__ int seg[WPT][NL];
  __shared__ double part[WPT][2];
  const int b = blockIdx.x;
  if (b >= B) return;
  const int lane = lane_id();
  const int wave = (int)threadIdx.x / kWave;  // (wave-uniform: the compiler keeps it in a scalar register)
  const double Tp = lane < M ? ts[(size_t)b * M + lane] : 1.0;
  int Lp_piece;
  const WgLanes<WPT> sl = wg_sample_lanes<WPT>(M, lane < M ? (int)(Tp / prm.delta_t) : 0, seg[wave], wave, Lp_piece);
  const int piece = sl.piece, r = sl.r, L = sl.L;
  const bool act = sl.act;
  const double T = act ? ts[(size_t)b * M + piece] : 1.0;
  const int ns = act ? (int)(T / prm.delta_t) : 0;
  Real c[6][D];
  {
    const double2 *src = reinterpret_cast<const double2 *>(coeffs + ((size_t)b * 6 * M + 6 * (act ? piece : 0)) * D);
#pragma unroll
    for (int q = 0; q < 3 * D; ++q) {
      const double2 v = src[q];
      const int e0 = 2 * q, e1 = 2 * q + 1;
      c[e0 / D][e0 % D] = act ? (Real)v.x : Real(0);
      c[e1 / D][e1 % D] = act ? (Real)v.y : Real(0);
    }
  }
  const LookupT lk(map);
  const Real dt = (Real)prm.delta_t, vmax2 = (Real)(prm.v_max * prm.v_max), safe = (Real)prm.safe_dis;
  const Real w2 = (Real)prm.w[2], w3 = (Real)prm.w[3];
  const Real inv_ns = ns > 0 ? Real(1) / (Real)ns : Real(0);
  Real aC[6][D];
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int d = 0; d < D; ++d) aC[k][d] = Real(0);
  Real aT = Real(0), aF = Real(0), aK = Real(0);
  char *land = stage + wave * kWaveBytes;

  for (int it0 = 0; it0 < sl.rounds; it0 += PF) {
    // ---- issue: positions and addresses of up to PF rounds, their gathers on the way to LDS back to back
    typename LookupT::Addr ad[PF];
    Real sv[PF];
    bool on[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      const int j = r + (it0 + u) * L;
      on[u] = act && j < ns;  // (rounds past the last one: j >= ns for every lane)
      const Real s = (Real)((double)j * prm.delta_t);  // beta_full row j: t = j * delta_t (:251)
      sv[u] = s;
      Real pos[D];
#pragma unroll
      for (int d = 0; d < D; ++d)
        pos[d] = fmaf(fmaf(fmaf(fmaf(fmaf(c[5][d], s, c[4][d]), s, c[3][d]), s, c[2][d]), s, c[1][d]), s, c[0][d]);
      ad[u] = lk.template prepare<D>(pos, on[u]);
      lk.load_async(ad[u], land + u * LookupT::kAsyncBytes);
    }
    // ---- consume round by round as the landings complete (loads return in issue order)
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (u == 0) wait_vmcnt<(PF - 1) * LookupT::kAsyncPieces>();
      if constexpr (PF > 1) if (u == 1) wait_vmcnt<(PF > 1 ? PF - 2 : 0) * LookupT::kAsyncPieces>();
      if constexpr (PF > 2) if (u == 2) wait_vmcnt<(PF > 2 ? PF - 3 : 0) * LookupT::kAsyncPieces>();
      if constexpr (PF > 3) if (u == 3) wait_vmcnt<(PF > 3 ? PF - 4 : 0) * LookupT::kAsyncPieces>();
      if constexpr (PF > 4) if (u == 4) wait_vmcnt<(PF > 4 ? PF - 5 : 0) * LookupT::kAsyncPieces>();
      if constexpr (PF > 5) if (u >= 5) wait_vmcnt<0>();
      lds_wave_sync();
      // (keep the rounds apart: interleaved by the scheduler their temporaries add up and the kernel spills)
      __builtin_amdgcn_sched_barrier(0);
      const typename LookupT::Raw rw = lk.read_staged(land + u * LookupT::kAsyncBytes);
      Real pos[D], vel[D];
      // (the position is evaluated a second time here, with the velocity: hidden from the compiler, which would otherwise
      //  keep the 15 partial Horner sums of every round in flight alive to reuse them -- 20 registers a round)
      Real s_again = sv[u];
      asm volatile("" : "+v"(s_again));
      piece_pos_vel<Real, D>(c, s_again, pos, vel);
      Real v2 = Real(0);
#pragma unroll
      for (int d = 0; d < D; ++d) v2 += vel[d] * vel[d];
      const Real vv = v2 - vmax2;
      Real gdrop[D];
      const Real vd = safe - lk.template finish<D>(ad[u], rw, gdrop);
      if (on[u] && (vv > Real(0) || vd > Real(0)))
        sample_accumulate<Real, D, LookupT>(c, r + (it0 + u) * L, ns, s_again, inv_ns, vel, vv, vd, lk, ad[u], rw, dt, w2, w3, aC, aT,
                                            aF, aK);
    }
    lds_wave_sync();  // (the landing area is rewritten by the next chunk's loads / the rows below)
  }

  // ---- per-piece sums through LDS rows: sample lane g writes row g (in its own wavefront's area), the first lane of
  // every piece adds the piece's rows in lane order
  Real *rows_w = reinterpret_cast<Real *>(land);
  {
    Quad *row = reinterpret_cast<Quad *>(rows_w + (size_t)lane * kRow);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      row[2 * d] = Quad{aC[0][d], aC[1][d], aC[2][d], aC[3][d]};
      row[2 * d + 1] = Quad{aC[4][d], aC[5][d], aT, Real(0)};
    }
  }
  const double pf = (double)wave_sum(act ? aF : Real(0)), pk = (double)wave_sum(act ? aK : Real(0));
  if (lane == 0) {
    part[wave][0] = pf;
    part[wave][1] = pk;
  }
  __syncthreads();
  if (act && r == 0) {
    Quad lo[D], hi[D];
#pragma unroll
    for (int d = 0; d < D; ++d) lo[d] = hi[d] = Quad{Real(0), Real(0), Real(0), Real(0)};
    const int g0 = wave * kWave + lane;
    for (int i = 0; i < L; ++i) {
      const int g = g0 + i;  // global sample lane: wavefront g / 64, lane g % 64
      const Quad *src = reinterpret_cast<const Quad *>(stage + (g / kWave) * kWaveBytes) + (size_t)(g % kWave) * (kRow / 4);
#pragma unroll
      for (int d = 0; d < D; ++d) {
        lo[d] += src[2 * d];
        hi[d] += src[2 * d + 1];
      }
    }
    Real gC[6][D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      gC[0][d] = lo[d].x; gC[1][d] = lo[d].y; gC[2][d] = lo[d].z; gC[3][d] = lo[d].w;
      gC[4][d] = hi[d].x; gC[5][d] = hi[d].y;
    }
    double2 *dst = reinterpret_cast<double2 *>(grad_C + ((size_t)b * 6 * M + 6 * piece) * D);
#pragma unroll
    for (int q = 0; q < 3 * D; ++q) {
      const int e0 = 2 * q, e1 = 2 * q + 1;
      dst[q] = make_double2((double)gC[e0 / D][e0 % D], (double)gC[e1 / D][e1 % D]);
    }
    grad_T[(size_t)b * M + piece] = (double)hi[0].z;
  }
  if (wave == 0) {
    // a piece without samples (T < delta_t) has no lane: its partials are zero
    if (lane < M && Lp_piece == 0) {
      double2 *dst = reinterpret_cast<double2 *>(grad_C + ((size_t)b * 6 * M + 6 * lane) * D);
#pragma unroll
      for (int q = 0; q < 3 * D; ++q) dst[q] = make_double2(0.0, 0.0);
      grad_T[(size_t)b * M + lane] = 0.0;
    }
    if (lane == 0) {
      double cf = 0.0, ck = 0.0;
#pragma unroll
      for (int k = 0; k < WPT; ++k) {
        cf += part[k][0];
        ck += part[k][1];
      }
      costs2[(size_t)b * 2 + 0] = cf;
      costs2[(size_t)b * 2 + 1] = ck;
    }
  }
}

}  // namespace neo
