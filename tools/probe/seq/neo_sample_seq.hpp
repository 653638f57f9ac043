// neo_sample_seq.hpp -- the ESDF-lookup kernel for fp32 sampling on 3-D fields, round 6 body:
// add_sampled_cost + add_sampled_grad_CT (expert_planner.py:392-466) with the trilinear lookups of the 3-D mode.
//
// One wavefront per trajectory, as before; what changed is WHICH lane visits which sample and WHEN the penalties are
// worked out.
//
//  * The samples of a trajectory form ONE sequence g = 0 .. S-1 (piece by piece, j = 0 .. ns_p - 1 inside a piece) and
//    round k of the loop hands sample 64 k + l to lane l.  ceil(S / 64) rounds with every lane busy (a fresh cfg2
//    guess: 9 rounds instead of the 13 of whole lanes per piece with 44 of 64 lanes busy), and the 64 lookups of
//    one gather instruction are 64 CONSECUTIVE points of the path: ~5 neighbouring lanes share a 128-byte brick, a wave
//    instruction touches ~13 lines instead of ~25.  (Round 5 measured the opposite assignment -- contiguous blocks of
//    samples per lane, neighbouring lanes two bricks apart -- 45 % slower; HISTORY.md round 5 (2).)
//    A lane's piece changes from round to round, so the pieces' coefficients live in an LDS table (80 bytes a piece:
//    18 floats, first sample index, sample count) and a lane reads its piece's record each round; the piece of sample g
//    comes from a bit mask of the piece boundaries (one 64-bit word a round: two v_mbcnt).
//  * The loop only DETECTS: position, velocity, the eight corners.  A sample can contribute only where
//    |v|^2 > v_max^2 (tested exactly) or d(p) < safe_dis; d is a convex combination of the corners, so
//    min(corners) - rounding bound >= safe_dis rules the collision term out without interpolating.  Candidates are
//    appended (piece, j) to a list in LDS in (round, lane) order.
//  * The penalties and their partials -- sample_accumulate, the very arithmetic of the fused kernels, sample by sample --
//    are worked out for the LISTED samples only, one lane a sample, 64 at a time: the ~66 instructions of the penalty
//    block used to run for ~4 busy lanes in half of the rounds.  Partials go to the pieces' accumulator rows in LDS with
//    ds_add_f32.  Only this wavefront touches its rows: the adds of one instruction are applied in the hardware's fixed
//    lane order, instructions in program order -- results are bit-reproducible and independent of batch and dispatch
//    order (tests/test_gpu_parity.py, test_gpu_api_edges.py), though no longer summed in the order of the fused kernels.
//
// Trajectories the table does not fit -- a piece without samples or with more than 64, more than 2048 samples in all --
// take the round-4 body (sample_wave_per_piece): decided per wavefront, same results as before.
#pragma once
#include "neo_kernels.hpp"

namespace neo {

constexpr int kSeqRec = 20;         // floats of a piece record: c[6][3], first sample (int), samples (int); also an accumulator row
constexpr int kSeqMaxRounds = 32;   // boundary-mask words
constexpr int kSeqMaxNs = 64;       // entries of the sample-time table
constexpr int kSeqList = 128;       // candidate list: a flush leaves < 64, a round adds <= 64

__host__ __device__ constexpr int seq_lds_bytes(int M) {
  const int seq = 2 * M * kSeqRec * 4 + kSeqMaxRounds * 8 + kSeqMaxNs * 4 + kSeqList * 4;
  const int old = kWave * 4 + kWave * 8 * 3 * 4;  // sample_wave_per_piece: seg + fold rows
  return seq > old ? seq : old;
}

// ---- the detection loop's view of a lookup: the eight corners of the sample's cell and whether it lies inside the field --
// no fractions, no interpolation.  Generic form: the lookup's own prepare / load.  Corner-brick fp32 fields (the bench
// default): the same cell and the same four 8-byte loads with the index arithmetic written for the instruction count
// (v_cvt_flr_i32_f32 and v_med3_i32 for floor / clamp, v_mad_u32_u24 for the brick address; idle and outside lanes read
// their clamped cell instead of cell 0).
__device__ __forceinline__ float seq_min3(float a, float b, float c) {
  float r;
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float seq_absmax3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// min over the corners and max over their magnitudes (NaN corners are skipped, as v_min3 / v_max3 do: such a cell is a
// candidate whenever the others are, and the exact arithmetic of the expansion decides)
__device__ __forceinline__ void seq_corner_range(const float (&c)[2][2][2], float &lo, float &hi) {
  const float m0 = seq_min3(c[0][0][0], c[0][0][1], c[0][1][0]), m1 = seq_min3(c[0][1][1], c[1][0][0], c[1][0][1]);
  lo = seq_min3(c[1][1][0], c[1][1][1], m0);
  asm("v_min_f32 %0, %1, %2" : "=v"(lo) : "v"(lo), "v"(m1));
  const float a0 = seq_absmax3(c[0][0][0], c[0][0][1], c[0][1][0]), a1 = seq_absmax3(c[0][1][1], c[1][0][0], c[1][0][1]);
  hi = seq_absmax3(c[1][1][0], c[1][1][1], a0);
  asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(hi), "v"(a1));
}

template <class LookupT>
struct SeqCorners {
  const LookupT &lk;
  __device__ __forceinline__ explicit SeqCorners(const LookupT &l) : lk(l) {}
  __device__ __forceinline__ bool fetch(const float (&pos)[3], bool on, float (&c)[2][2][2]) const {
    const typename LookupT::Addr ad = lk.template prepare<3>(pos, on);
    const typename LookupT::Raw rw = lk.load(ad);
#pragma unroll
    for (int z = 0; z < 2; ++z)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int x = 0; x < 2; ++x) c[z][y][x] = rw.c[z][y][x];
    return ad.inside;
  }
};
template <>
struct SeqCorners<Lookup3D<float, float, 3>> {
  const Lookup3D<float, float, 3> &lk;
  int hi_i[3];        // n - 2 per axis
  unsigned nbx, nby;  // bricks along x, y
  bool first_alone = false;
  __device__ __forceinline__ explicit SeqCorners(const Lookup3D<float, float, 3> &l) : lk(l) {
    hi_i[0] = l.m.nx - 2;
    hi_i[1] = l.m.ny - 2;
    hi_i[2] = l.m.nz - 2;
    nbx = (unsigned)l.m.nbx;
    nby = (unsigned)l.m.nby;
  }
  __device__ __forceinline__ bool fetch(const float (&pos)[3], bool on, float (&c)[2][2][2]) const {
#pragma clang fp contract(on)
    bool inside = on;
    int i[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float um = fmaf(pos[k], lk.inv, lk.off[k]);  // (the arithmetic of Lookup3D::prepare)
      if (!(um >= -0.5f && um < lk.hi[k])) inside = false;
      int fl;
      asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(fl) : "v"(um));
      asm("v_med3_i32 %0, %1, 0, %2" : "=v"(i[k]) : "v"(fl), "s"(hi_i[k]));
    }
    const unsigned bx = (unsigned)i[0] >> 1, by = (unsigned)i[1] >> 1, bz = (unsigned)i[2] >> 1;
    const unsigned lx = (unsigned)i[0] & 1u, ly = (unsigned)i[1] & 1u, lz = (unsigned)i[2] & 1u;
    unsigned blk, inl;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(blk) : "v"(bz), "s"(nby), "v"(by));
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(blk) : "v"(blk), "s"(nbx), "v"(bx));
    asm("v_mad_u32_u24 %0, %1, 3, %2" : "=v"(inl) : "v"(lz), "v"(ly));
    asm("v_mad_u32_u24 %0, %1, 3, %2" : "=v"(inl) : "v"(inl), "v"(lx));
    const unsigned off = (blk << 7) + (inl << 2);
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(lk.rsrc, (int)(off + (unsigned)((dz * 3 + dy) * 3) * 4u), 0, 0);
        c[dz][dy][0] = __uint_as_float(v[0]);
        c[dz][dy][1] = __uint_as_float(v[1]);
#ifdef NEO_EXPERIMENTS
        if (first_alone && dz == 0 && dy == 0) {  // (experiment: the line's first request alone, the other three once it is back)
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(c[0][0][0]), "+v"(c[0][0][1]) : : "memory");
        }
#endif
      }
    return inside;
  }
};

#ifndef NEO_SAMPLE_SEQ_OCC
#define NEO_SAMPLE_SEQ_OCC 4
#endif

// IO: element type of coeffs / grad_C / grad_T (double: neo_sampled_terms_batch_dev; float: neo_sampled_terms_batch_f32_dev)
template <class LookupT, typename IO>
__global__ __launch_bounds__(kWave, NEO_SAMPLE_SEQ_OCC) void sample_seq_kernel(int B, int M, DevParams prm, Map3D map,
                                                                               const IO *__restrict__ coeffs,
                                                                               const double *__restrict__ ts,
                                                                               double *__restrict__ costs2,
                                                                               IO *__restrict__ grad_C, IO *__restrict__ grad_T,
                                                                               const int *__restrict__ order) {
#pragma clang fp contract(on)
  constexpr int D = 3;
  typedef float Real;
  typedef Real Quad __attribute__((ext_vector_type(4)));
  typedef Real Pair __attribute__((ext_vector_type(2)));
  typedef IO IOPair __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char seq_lds[];
  if ((int)blockIdx.x >= B) return;
  const int b = order ? order[blockIdx.x] : (int)blockIdx.x;
  const int lane = lane_id();

  // sample counts int(T / delta_t) (:401) and the first sample of every piece
  const double Tp = lane < M ? ts[(size_t)b * M + lane] : 1.0;
  const int ns_p = lane < M ? (int)(Tp / prm.delta_t) : 0;
  const int incl = wave_scan_add(ns_p);
  const int start = incl - ns_p;
  const int S = __builtin_amdgcn_readlane(incl, kWave - 1);
  if (__builtin_amdgcn_ballot_w64(lane < M && (ns_p <= 0 || ns_p > kSeqMaxNs)) != 0ull || S > kWave * kSeqMaxRounds) {
    sample_wave_per_piece<D, Real, Map3D, LookupT, IO>(b, M, prm, map, coeffs, ts, costs2, grad_C, grad_T,
                                                       reinterpret_cast<int *>(seq_lds),
                                                       reinterpret_cast<Real *>(seq_lds + kWave * 4));
    return;
  }

  Real *rec = reinterpret_cast<Real *>(seq_lds);
  Real *acc = rec + M * kSeqRec;
  unsigned long long *masks = reinterpret_cast<unsigned long long *>(acc + M * kSeqRec);
  Real *s_tab = reinterpret_cast<Real *>(masks + kSeqMaxRounds);
  int *list = reinterpret_cast<int *>(s_tab + kSeqMaxNs);

  // ---- tables
  if (lane < kSeqMaxRounds) masks[lane] = 0ull;
  s_tab[lane] = (Real)((double)lane * prm.delta_t);  // beta_full row j: t = j * delta_t (:251), rounded as in the fused kernels
  for (int i = lane; i < 5 * M; i += kWave) reinterpret_cast<Quad *>(acc)[i] = Quad{0.0f, 0.0f, 0.0f, 0.0f};
  {
    // the 18 M coefficients of the trajectory are contiguous: pairs, one a lane, into the records
    const IOPair *src = reinterpret_cast<const IOPair *>(coeffs + (size_t)b * 6 * M * D);
    for (int i = lane; i < 9 * M; i += kWave) {
      const int p = (i * 7282) >> 16;  // i / 9 for i < 9 * 64
      const IOPair v = src[i];
      reinterpret_cast<Pair *>(rec)[p * (kSeqRec / 2) + (i - 9 * p)] = Pair{(Real)v.x, (Real)v.y};
    }
    if (lane < M) {
      typedef int IPair __attribute__((ext_vector_type(2)));
      reinterpret_cast<IPair *>(rec)[lane * (kSeqRec / 2) + 9] = IPair{start, ns_p};
    }
  }
  lds_wave_sync();
  // bit (start_q - 1) for the pieces q >= 1: the piece of sample g is the number of bits below position g
  if (lane >= 1 && lane < M) {
    const int at = start - 1;
    __hip_atomic_fetch_or(&masks[at >> 6], 1ull << (at & 63), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  lds_wave_sync();

  const Real vmax2 = par_vmax2<Real>(prm), safe = par_safe<Real>(prm), dt = par_dt<Real>(prm);
  const Real w2 = par_w2<Real>(prm), w3 = par_w3<Real>(prm);
  const LookupT lk(map);
  SeqCorners<LookupT> corners(lk);
#ifdef NEO_EXPERIMENTS
  if constexpr (std::is_same<LookupT, Lookup3D<float, float, 3>>::value) corners.first_alone = NEO_DBG(prm, 1 << 19);
#endif
  Real aF = 0.0f, aK = 0.0f;

  // a piece's record: coefficients [k][d], first sample, samples
  auto read_rec = [&](int p, Real (&c)[6][D], int &first, int &ns) {
    const Quad *r = reinterpret_cast<const Quad *>(seq_lds + __umul24((unsigned)p, (unsigned)(kSeqRec * 4)));
    const Quad q0 = r[0], q1 = r[1], q2 = r[2], q3 = r[3], q4 = r[4];
    c[0][0] = q0.x; c[0][1] = q0.y; c[0][2] = q0.z; c[1][0] = q0.w;
    c[1][1] = q1.x; c[1][2] = q1.y; c[2][0] = q1.z; c[2][1] = q1.w;
    c[2][2] = q2.x; c[3][0] = q2.y; c[3][1] = q2.z; c[3][2] = q2.w;
    c[4][0] = q3.x; c[4][1] = q3.y; c[4][2] = q3.z; c[5][0] = q3.w;
    c[5][1] = q4.x; c[5][2] = q4.y;
    first = __float_as_int(q4.z);
    ns = __float_as_int(q4.w);
  };

  // ---- the listed samples, one a lane: penalties and partials (:404-466), sample_accumulate as in the fused kernels
  auto expand = [&](int entry, bool act) {
    const int p = entry >> 8, j = entry & 255;
    Real c[6][D];
    int first, ns;
    read_rec(p, c, first, ns);
    (void)first;
    const Real s = s_tab[j];
    Real pos[D], vel[D];
    piece_pos_vel<Real, D>(c, s, pos, vel);
    const typename LookupT::Addr ad = lk.template prepare<D>(pos, act);
    const typename LookupT::Raw rw = lk.load(ad);
    Real v2 = 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) v2 += vel[d] * vel[d];
    const Real vv = v2 - vmax2;
    Real gdrop[D];
    const Real vd = safe - lk.template finish<D>(ad, rw, gdrop);
    if (act && (vv > 0.0f || vd > 0.0f)) {
      const Real inv_ns = __builtin_amdgcn_rcpf((Real)ns);
      Real dC[6][D], dT = 0.0f;
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int d = 0; d < D; ++d) dC[k][d] = 0.0f;
      sample_accumulate<Real, D, LookupT>(c, j, ns, s, inv_ns, vel, vv, vd, lk, ad, rw, dt, w2, w3, dC, dT, aF, aK);
      Real *row = acc + p * kSeqRec;
      if (NEO_DBG(prm, 1 << 17)) row = acc + (lane % M) * kSeqRec;   // (timing experiment: no two lanes on one address ... mostly)
      if (!NEO_DBG(prm, 1 << 16)) {
#pragma unroll
      for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int d = 0; d < D; ++d) __hip_atomic_fetch_add(row + k * D + d, dC[k][d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      __hip_atomic_fetch_add(row + 6 * D, dT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      } else { aF += dC[0][0] + dC[5][2] + dT; }
    }
  };

  // ---- the detection loop
  const int rounds = (S + kWave - 1) >> 6;
  int base = 0;  // pieces begun before this round's 64 samples (bits of the earlier mask words)
  int cnt = 0;   // listed samples
#ifdef NEO_EXPERIMENTS
  int total_cand = 0;
#endif
  for (int k = 0; k < rounds; ++k) {
    const int g = k * kWave + lane;
    const bool on = g < S;
    const unsigned long long bk = masks[k];
    const int p = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bk, (unsigned)base));
    base += __builtin_popcountll(__builtin_amdgcn_readfirstlane((unsigned)bk) |
                                 ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(bk >> 32)) << 32));
    Real c[6][D];
    int first, ns;
    read_rec(p, c, first, ns);
    (void)ns;
    const int j = on ? g - first : 0;
    const Real s = s_tab[j];
    Real pos[D], vel[D];
    piece_pos_vel<Real, D>(c, s, pos, vel);
    Real cr[2][2][2];
    const bool inside = corners.fetch(pos, on, cr);
    Real v2 = 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d) v2 += vel[d] * vel[d];
    // the distance is a convex combination of the eight corners: it cannot fall below their minimum by more than the
    // rounding of three nested interpolations (< 2^-21 of the largest corner, 2^-18 taken)
    Real lo, hi;
    seq_corner_range(cr, lo, hi);
    const bool near = inside && !(fmaf(hi, -3.814697265625e-06f, lo) >= safe);
    const bool cand = on && (v2 - vmax2 > 0.0f || near);
    const unsigned long long cm = __builtin_amdgcn_ballot_w64(cand);
    if (cm != 0ull) {
      const int at = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, 0u));
      if (cand) list[at] = (p << 8) | j;
      cnt += __builtin_popcountll(cm);
#ifdef NEO_EXPERIMENTS
      total_cand += __builtin_popcountll(cm);
#endif
      lds_wave_sync();
      if (cnt >= kWave) {
        expand(list[lane], true);
        const int rest = list[kWave + lane];
        lds_wave_sync();
        cnt -= kWave;
        if (lane < cnt) list[lane] = rest;
        lds_wave_sync();
      }
    }
  }
  if (cnt > 0) expand(lane < cnt ? list[lane] : 0, lane < cnt);
  lds_wave_sync();
#ifdef NEO_EXPERIMENTS
  if (NEO_DBG(prm, 1 << 18)) aK = lane == 0 ? (Real)total_cand : 0.0f;
#endif

  // ---- results: the accumulator rows are the pieces' partials, [6][3] then the duration's
  {
    const Pair *rows = reinterpret_cast<const Pair *>(acc);
    IOPair *dst = reinterpret_cast<IOPair *>(grad_C + (size_t)b * 6 * M * D);
    for (int i = lane; i < 9 * M; i += kWave) {
      const int p = (i * 7282) >> 16;
      const Pair v = rows[p * (kSeqRec / 2) + (i - 9 * p)];
      dst[i] = IOPair{(IO)v.x, (IO)v.y};
    }
    if (lane < M) grad_T[(size_t)b * M + lane] = (IO)acc[lane * kSeqRec + 6 * D];
  }
  const Real cf = wave_sum(aF), ck = wave_sum(aK);
  if (lane == 0) {
    costs2[(size_t)b * 2 + 0] = (double)cf;
    costs2[(size_t)b * 2 + 1] = (double)ck;
  }
}

}  // namespace neo
