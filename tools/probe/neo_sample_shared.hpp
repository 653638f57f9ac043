// neo_sample_shared.hpp -- round-5 experiment, measured and NOT adopted (HISTORY.md round 5): the ESDF-lookup kernel's
// body with whole lanes per piece plus SHARED TAIL LANES (the remainders of two neighbouring pieces on one lane), so that
// a fresh guess takes 10 rounds of samples instead of 13.  Fewer vector instructions (2 756 against 3 035 a wavefront) and
// 24 % more L1 -> L2 requests: 29.9 us against 27.9 us per 4096 launch.  Built into sample_kernel only with
// -DNEO_SAMPLE_SHARED_TAILS (tools/gpu_sample_only.py compares the libraries).  Included by neo_kernels.hpp inside
// namespace neo.
#pragma once

// floats per row of sample_wave_shared's LDS tables: [d][6] coefficients / partials, then (samples, lane split) as ints /
// the duration partial
__host__ __device__ constexpr int srow_floats(int D) { return (6 * D + 2 + 3) / 4 * 4; }
// LDS of the fp32 ESDF-lookup kernel: the coefficient table [M] rows, one row of partial sums per lane and one per pair of
// pieces whose remainders share a lane
__host__ __device__ constexpr int sample_lds_bytes(int M, int D) { return (M + kWave + (M + 1) / 2) * srow_floats(D) * (int)sizeof(float); }

// The ESDF-lookup kernel's body for fp32 sampling (round 5).  Whole lanes per piece as in balanced_sample_lanes -- the L
// lanes of a piece walk its samples j = r, r + L, ... , neighbours in space on neighbouring lanes -- but a piece no longer
// rounds its lane count UP: it gets floor(ns / R) full lanes of R samples each, and its remainder (the last ns mod R
// samples) goes to a tail lane which the NEXT piece's remainder shares when the two fit into R rounds.  At a fresh guess
// (19 pieces of 25 samples, 2 of 37) whole lanes need R = 13 rounds with 44 lanes busy (R = 10 would take 65 lanes);
// with shared tails R = 10 fits in 56.  Only the shared lanes change piece inside the loop, once each.
// Also: durations AND coefficients are loaded up front in the PIECE layout (two independent loads in flight instead of
// the coefficient load waiting for the lane assignment, which waits for the durations) and reach the sample lanes
// through an LDS table.
// Sums: every lane leaves its partial sums in its own LDS row, a shared lane its second piece's in the pair's extra row;
// lane p then adds up the rows of piece p in lane order (the extra row last): a fixed order.
template <int D, class MapT, class LookupT>
__device__ __forceinline__ void sample_wave_shared(int b, int M, const DevParams &prm, const MapT &map,
                                                   const double *__restrict__ coeffs, const double *__restrict__ ts,
                                                   double *__restrict__ costs2, double *__restrict__ grad_C,
                                                   double *__restrict__ grad_T, int *seg, float *tab) {
#pragma clang fp contract(on)  // fuse a*b+c only as written
  typedef float Real;
  typedef float Quad __attribute__((ext_vector_type(4)));
  constexpr int RS = srow_floats(D), NQ = RS / 4, kNs = 6 * D, kMeta = 6 * D + 1;
  float *ctab = tab, *rows = tab + (size_t)M * RS, *extra = rows + (size_t)kWave * RS;
  const int lane = lane_id();
  // ---- PIECE layout: durations and coefficients (both loads in flight), sample counts
  const bool have = lane < M;
  const double Tp = have ? ts[(size_t)b * M + lane] : 0.0;
  float f[RS];
  {
    // the 6*D doubles of a piece are contiguous and 16-byte aligned (6*D is even)
    const double2 *src = reinterpret_cast<const double2 *>(coeffs + ((size_t)b * 6 * M + 6 * (have ? lane : 0)) * D);
#pragma unroll
    for (int q = 0; q < 3 * D; ++q) {
      const double2 v = src[q];
      const int e0 = 2 * q, e1 = 2 * q + 1;  // element k * D + d of the piece -> [d][k]
      f[(e0 % D) * 6 + e0 / D] = (float)v.x;
      f[(e1 % D) * 6 + e1 / D] = (float)v.y;
    }
#pragma unroll
    for (int q = kNs; q < RS; ++q) f[q] = 0.0f;
  }
  const int ns_p = have ? (int)(Tp / prm.delta_t) : 0;
  const int total = wave_sum(ns_p);
  // ---- lanes: the smallest R for which the full lanes, the tails and the shared tails fit the wavefront
  int R = max(1, (total + kWave - 1) / kWave);
  int full = 0, rem = 0, Lp = 0;
  bool first_of_pair = false, second_of_pair = false;
  for (;;) {
    const float rR = __builtin_amdgcn_rcpf((float)R);
    full = (int)(((float)ns_p + 0.5f) * rR);  // floor(ns / R): the half keeps the quotient off the integers
    rem = ns_p - full * R;
    const int rem_next = dpp_i<0x130>(rem);   // wave_shl:1 -- lane p <- lane p + 1 (0 past the end)
    first_of_pair = (lane & 1) == 0 && rem > 0 && rem_next > 0 && rem + rem_next <= R;
    second_of_pair = dpp_i<0x138>(first_of_pair ? 1 : 0) != 0;  // wave_shr:1 -- lane p <- lane p - 1
    Lp = full + ((rem > 0 && !second_of_pair) ? 1 : 0);
    if (wave_sum(Lp) <= kWave) break;
    ++R;
  }
  const int start = wave_scan_add(Lp) - Lp;
  if (have) {
    f[kNs] = __int_as_float(ns_p);
    f[kMeta] = __int_as_float(full | (first_of_pair ? 0x10000 : 0) | ((rem > 0 && !second_of_pair) ? 0x20000 : 0));
    Quad *row = reinterpret_cast<Quad *>(ctab + (size_t)lane * RS);
#pragma unroll
    for (int q = 0; q < NQ; ++q) row[q] = Quad{f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]};
  }
  seg[lane] = 0;
  lds_wave_sync();
  if (Lp > 0) seg[start] = ((lane + 1) << 16) | start;
  lds_wave_sync();
  const int key = wave_scan_max_nonneg(seg[lane]);  // the piece this sample lane falls in: the last start at or before it
  int piece = max((key >> 16) - 1, 0);
  // ---- SAMPLE layout
  Real c[6][D];
  int ns = 0;
  auto read_piece = [&](int pc) {
    const Quad *row = reinterpret_cast<const Quad *>(ctab + (size_t)pc * RS);
    float g[RS];
#pragma unroll
    for (int w = 0; w < NQ; ++w) {
      const Quad v = row[w];
      g[4 * w] = v.x; g[4 * w + 1] = v.y; g[4 * w + 2] = v.z; g[4 * w + 3] = v.w;
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int k = 0; k < 6; ++k) c[k][d] = g[d * 6 + k];
    ns = __float_as_int(g[kNs]);
    return __float_as_int(g[kMeta]);
  };
  int j, stride, left;
  bool shared;
  {
    const int meta = read_piece(piece);
    const int fl = meta & 0xffff, idx = lane - (key & 0xffff);
    // (bit 17: the piece has a tail lane of its own -- a remainder that is not hosted by the piece before it)
    const bool act = key != 0 && (idx < fl || (idx == fl && (meta & 0x20000) != 0));
    const bool tail = idx == fl;
    j = tail ? fl * R : idx;
    stride = tail ? 1 : fl;
    left = act ? (tail ? ns - fl * R : R) : 0;
    shared = act && tail && (meta & 0x10000) != 0;
  }
  Real inv_ns = ns > 0 ? __builtin_amdgcn_rcpf((float)ns) : 0.0f;
  const Real dt = par_dt<Real>(prm), vmax2 = par_vmax2<Real>(prm), safe = par_safe<Real>(prm);
  const Real w2 = par_w2<Real>(prm), w3 = par_w3<Real>(prm);
  Real aC[6][D];
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int d = 0; d < D; ++d) aC[k][d] = 0.0f;
  Real aT = 0.0f, aF = 0.0f, aK = 0.0f;
  auto store_row = [&](float *dst) {
    Quad *row = reinterpret_cast<Quad *>(dst);
    float g[RS];
#pragma unroll
    for (int w = 0; w < RS; ++w) g[w] = 0.0f;
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
      for (int k = 0; k < 6; ++k) g[d * 6 + k] = aC[k][d];
    g[6 * D] = aT;
#pragma unroll
    for (int w = 0; w < NQ; ++w) row[w] = Quad{g[4 * w], g[4 * w + 1], g[4 * w + 2], g[4 * w + 3]};
  };
  const LookupT lk(map);
  for (int k = 0; k < R; ++k) {
    if (left > 0) {  // (tail lanes have fewer than R samples: they sit the other rounds out, exec-masked)
      const Real s = (Real)((double)j * prm.delta_t);  // beta_full row j: t = j * delta_t (:251)
      Real pos[D], vel[D];
#pragma unroll
      for (int d = 0; d < D; ++d)
        pos[d] = fmaf(fmaf(fmaf(fmaf(fmaf(c[5][d], s, c[4][d]), s, c[3][d]), s, c[2][d]), s, c[1][d]), s, c[0][d]);
      const typename LookupT::Addr ad = lk.template prepare<D>(pos, true);
      const typename LookupT::Raw rw = lk.load(ad);
      piece_pos_vel<Real, D>(c, s, pos, vel);  // (the velocity while the gathers fly)
      Real v2 = 0.0f;
#pragma unroll
      for (int d = 0; d < D; ++d) v2 += vel[d] * vel[d];
      const Real vv = v2 - vmax2;
      Real gdrop[D];
      const Real vd = safe - lk.template finish<D>(ad, rw, gdrop);
      if (vv > 0.0f || vd > 0.0f)
        sample_accumulate<Real, D, LookupT>(c, j, ns, s, inv_ns, vel, vv, vd, lk, ad, rw, dt, w2, w3, aC, aT, aF, aK);
      j += stride;
      --left;
      if (left == 0 && shared) {
        // a shared tail lane is through with its own piece: those sums into its row, on with the next piece's remainder
        store_row(rows + (size_t)lane * RS);
#pragma unroll
        for (int kk = 0; kk < 6; ++kk)
#pragma unroll
          for (int d = 0; d < D; ++d) aC[kk][d] = 0.0f;
        aT = 0.0f;
        shared = false;
        ++piece;
        const int fl = read_piece(piece) & 0xffff;
        j = fl * R;
        left = ns - fl * R;
        inv_ns = __builtin_amdgcn_rcpf((float)ns);
      }
    }
  }
  // ---- every lane's sums into its row (a lane that went on to a second piece: into the pair's extra row)
  const bool moved_on = piece != max((key >> 16) - 1, 0);  // (always the odd piece of its pair)
  store_row(moved_on ? extra + (size_t)(piece >> 1) * RS : rows + (size_t)lane * RS);
  lds_wave_sync();
  // ---- PIECE layout again: lane p adds up the rows of piece p, first lane first, the shared remainder last
  if (have) {
    Quad sum[NQ];
#pragma unroll
    for (int w = 0; w < NQ; ++w) sum[w] = Quad{0.0f, 0.0f, 0.0f, 0.0f};
    const Quad *src = reinterpret_cast<const Quad *>(rows + (size_t)start * RS);
    for (int i = 0; i < Lp; ++i) {
#pragma unroll
      for (int w = 0; w < NQ; ++w) sum[w] += src[i * NQ + w];
    }
    if (second_of_pair) {
      const Quad *ex = reinterpret_cast<const Quad *>(extra + (size_t)(lane >> 1) * RS);
#pragma unroll
      for (int w = 0; w < NQ; ++w) sum[w] += ex[w];
    }
    float g[RS];
#pragma unroll
    for (int w = 0; w < NQ; ++w) {
      g[4 * w] = sum[w].x; g[4 * w + 1] = sum[w].y; g[4 * w + 2] = sum[w].z; g[4 * w + 3] = sum[w].w;
    }
    double2 *dst = reinterpret_cast<double2 *>(grad_C + ((size_t)b * 6 * M + 6 * lane) * D);
#pragma unroll
    for (int w = 0; w < 3 * D; ++w) {
      const int e0 = 2 * w, e1 = 2 * w + 1;
      dst[w] = make_double2((double)g[(e0 % D) * 6 + e0 / D], (double)g[(e1 % D) * 6 + e1 / D]);
    }
    grad_T[(size_t)b * M + lane] = (double)g[6 * D];
  }
  const double cf = (double)wave_sum(aF), ck = (double)wave_sum(aK);  // (lanes without samples hold zeros)
  if (lane == 0) {
    costs2[(size_t)b * 2 + 0] = cf;
    costs2[(size_t)b * 2 + 1] = ck;
  }
}

