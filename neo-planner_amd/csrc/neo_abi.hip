// neo_abi.hip -- C ABI of libneo_planner_hip.so (include/neo_planner.h) and the small kernels around the hot path:
//
//   query_kernel     ESDF point lookups                         (map_server/esdf.py:53-82)
//   edt / gradient   ESDF.occupancy_map_cb                      (map_server/esdf.py:11-33)
//   traj_state_kernel get_full_state_cmd                        (traj_utils.py:85-195)
//
// The fused kernels (eval / optimize / sample) are templates in neo_kernels.hpp, instantiated per family in the
// neo_disp_*.hip translation units.
#include <cstring>
#include "neo_host.hpp"
#include <rocprim/device/device_radix_sort.hpp>
#include "neo_kernels.hpp"

namespace neo {

template <typename Real, class MapT, class LookupT, int DM>
__global__ void query_kernel(int n, MapT map, const double *__restrict__ pts, double *__restrict__ dist,
                             double *__restrict__ grad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Real pos[DM], g[DM];
#pragma unroll
  for (int d = 0; d < DM; ++d) pos[d] = (Real)pts[(size_t)i * DM + d];
  bool inside;
  LookupT lk(map);
  const Real v = lk.template fetch<DM>(pos, g, inside);
  dist[i] = (double)v;
  if (grad)
#pragma unroll
    for (int d = 0; d < DM; ++d) grad[(size_t)i * DM + d] = (double)g[d];
}

// ---- ESDF construction (esdf.py:23-33) ------------------------------------
// exact Euclidean distance transform of the free cells to the nearest occupied cell, two
// separable passes over integer squared distances (Felzenszwalb & Huttenlocher lower envelope),
// then sqrt * resolution and numpy.gradient with unit spacing.
constexpr int kEdtInf = 1 << 28;

__global__ void edt_columns_kernel(const int8_t *__restrict__ occ, int W, int H, int *__restrict__ g) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= W) return;
  int d = kEdtInf;
  for (int y = 0; y < H; ++y) {
    d = (occ[(size_t)y * W + x] == 100) ? 0 : (d >= kEdtInf ? kEdtInf : d + 1);
    g[(size_t)y * W + x] = d;
  }
  d = kEdtInf;
  for (int y = H - 1; y >= 0; --y) {
    d = (occ[(size_t)y * W + x] == 100) ? 0 : (d >= kEdtInf ? kEdtInf : d + 1);
    const int cur = g[(size_t)y * W + x];
    g[(size_t)y * W + x] = cur < d ? cur : d;
  }
}

// one thread per row; v/z scratch rows live in global memory (W ints / W+1 doubles per row)
__global__ void edt_rows_kernel(const int *__restrict__ g, int W, int H, double res, int *__restrict__ vbuf,
                                double *__restrict__ zbuf, double *__restrict__ dist) {
  const int y = blockIdx.x * blockDim.x + threadIdx.x;
  if (y >= H) return;
  const int *f = g + (size_t)y * W;
  int *v = vbuf + (size_t)y * W;
  double *z = zbuf + (size_t)y * (W + 1);
  int k = -1;
  for (int q = 0; q < W; ++q) {
    if (f[q] >= kEdtInf) continue;
    const double fq = (double)f[q] * (double)f[q] + (double)q * q;
    double s = 0.0;
    while (k >= 0) {
      const int p = v[k];
      const double fp = (double)f[p] * (double)f[p] + (double)p * p;
      s = (fq - fp) / (2.0 * q - 2.0 * p);
      if (s <= z[k]) {
        --k;
      } else {
        break;
      }
    }
    ++k;
    v[k] = q;
    z[k] = (k == 0) ? -1.0e300 : s;
    z[k + 1] = 1.0e300;
  }
  double *out = dist + (size_t)y * W;
  if (k < 0) {
    // no occupied cell in the whole map: scipy.ndimage.distance_transform_edt then measures to a
    // virtual background cell at (row -1, column 0); the reference inherits that (esdf.py:29)
    for (int q = 0; q < W; ++q) {
      const long long sq = (long long)(y + 1) * (y + 1) + (long long)q * q;
      out[q] = sqrt((double)sq) * res;
    }
    return;
  }
  int j = 0;
  for (int q = 0; q < W; ++q) {
    while (z[j + 1] < (double)q) ++j;
    const long long p = v[j];
    const long long dq = q - p;
    const long long sq = dq * dq + (long long)f[p] * f[p];
    out[q] = sqrt((double)sq) * res;
  }
}

// Small maps (the reference's 300 x 300): the same two passes by exhaustive minimisation, one thread per cell --
// W*H*(W+H) integer operations (54 M at 300 x 300) instead of a sequential sweep per line.  Exact integer
// squared distances, hence the same doubles as the sweeps above (and as SciPy).
__global__ void edt2_columns_bf_kernel(const int8_t *__restrict__ occ, int W, int H, int *__restrict__ g) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  int d = kEdtInf;
  for (int q = 0; q < H; ++q) {
    const int dq = q > y ? q - y : y - q;
    if (occ[(size_t)q * W + x] == 100 && dq < d) d = dq;
  }
  g[(size_t)y * W + x] = d;
}
__global__ void edt2_rows_bf_kernel(const int *__restrict__ g, int W, int H, double res, double *__restrict__ dist) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const int *f = g + (size_t)y * W;
  long long best = -1;
  for (int p = 0; p < W; ++p) {
    const int fp = f[p];
    if (fp >= kEdtInf) continue;
    const long long dq = x - p;
    const long long sq = dq * dq + (long long)fp * fp;
    if (best < 0 || sq < best) best = sq;
  }
  // no occupied cell in the whole map: SciPy's virtual background cell at (row -1, column 0), see edt_rows_kernel
  if (best < 0) best = (long long)(y + 1) * (y + 1) + (long long)x * x;
  dist[(size_t)y * W + x] = sqrt((double)best) * res;
}

// ---- 3-D exact EDT (north-star scenes): three separable passes over integers, so the result equals
// scipy.ndimage.distance_transform_edt exactly; the final sqrt * resolution is rounded to fp32.
//   pass X  distance in cells to the nearest occupied voxel of the same x-row: one wavefront per row, two wave scans
//           ("last occupied index at or before me" from the left, the mirror image from the right) -> uint16
//   pass Y  squared distance in the (x, y) plane, D(p) = min_q gx(q)^2 + (p - q)^2 along y; pass Z the same along z on
//           the plane distances, then sqrt * res -> fp32.  The (leftmost) minimiser q*(p) never moves left when p
//           moves right (the cost is totally monotone for ANY f), so the line is solved by monotone minima: the two end
//           points by a full scan, then the midpoint of every gap -- its minimiser lies between its neighbours' -- at
//           spacings 2^k .. 1.  Each level visits at most n + (#points) candidates: O(n log n) comparisons for a line,
//           about a dozen per voxel, whatever the distances are (an outward search from q = p, stopped at d^2 >= best,
//           visits as many candidates as the voxel's distance in cells: 4.2 ms for the y pass of a 300^3 forest scene,
//           most of whose volume is far from everything).  A block holds TX x-columns by the whole line in LDS
//           (squared values + minimisers, 6 bytes a voxel); work items (point, column) are dealt to its 256 threads.
// (Round 1-3 form: one thread per line running the lower-envelope sweep with its stacks in global memory -- 0.75 +
//  1.24 + 2.73 ms for 300^3 and 540 MB of scratch; these kernels need none.)  Dimensions up to 4096 per axis.
constexpr int kXInf = 0x7fff;

__global__ __launch_bounds__(256) void edt3_x_kernel(const uint8_t *__restrict__ occ, int nx, size_t rows,
                                                     uint16_t *__restrict__ gx) {
  extern __shared__ uint16_t x_left[];  // [4][nx]
  const int lane = lane_id(), wave = threadIdx.x / kWave;
  const size_t row = (size_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const uint8_t *o = occ + row * nx;
  uint16_t *left = x_left + (size_t)wave * nx;
  int carry = 0;  // (index + 1) of the last occupied voxel so far, 0 = none
  for (int c0 = 0; c0 < nx; c0 += kWave) {
    const int idx = c0 + lane;
    const int v = (idx < nx && o[idx]) ? idx + 1 : 0;
    const int s = max(wave_scan_max_nonneg(v), carry);
    if (idx < nx) left[idx] = (uint16_t)(s ? min(idx + 1 - s, kXInf) : kXInf);
    carry = __builtin_amdgcn_readlane(s, kWave - 1);
  }
  lds_wave_sync();
  carry = 0;  // nx - index of the nearest occupied voxel to the right so far (>= 1), 0 = none
  const int nchunk = (nx + kWave - 1) / kWave;
  for (int c = nchunk - 1; c >= 0; --c) {
    const int idx = c * kWave + (kWave - 1 - lane);  // lanes walk the chunk from its right end
    const int v = (idx < nx && o[idx]) ? nx - idx : 0;
    const int s = max(wave_scan_max_nonneg(v), carry);
    if (idx < nx) {
      const int right = s ? (nx - s) - idx : kXInf;
      gx[row * nx + idx] = (uint16_t)min(min((int)left[idx], right), kXInf);
    }
    carry = __builtin_amdgcn_readlane(s, kWave - 1);
  }
}

// The same pass with V = 8 or 16 consecutive voxels per lane (rows of up to 64 V voxels, nx a multiple of four): ONE
// pair of 4-byte loads per lane instead of a byte per lane and chunk, one prefix and one suffix scan per row, the
// distances of the lane's voxels by two sweeps in registers, 8-byte stores (round 4: 69 -> see DESIGN.md at 300^3).
template <int V>
__global__ __launch_bounds__(256) void edt3_xv_kernel(const uint8_t *__restrict__ occ, int nx, size_t rows,
                                                      uint16_t *__restrict__ gx) {
  const int lane = lane_id(), wave = threadIdx.x / kWave;
  const size_t row = (size_t)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const uint32_t *o = reinterpret_cast<const uint32_t *>(occ + row * nx);
  const int idx0 = lane * V;
  uint32_t w[V / 4];
#pragma unroll
  for (int k = 0; k < V / 4; ++k) w[k] = idx0 + 4 * k < nx ? o[(idx0 >> 2) + k] : 0u;
  // bit k of m: voxel idx0 + k is occupied
  uint32_t m = 0;
#pragma unroll
  for (int k = 0; k < V / 4; ++k) {
    const uint32_t nz = (((w[k] & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w[k]) & 0x80808080u;  // bit 7 of every non-zero byte
    m |= (((nz >> 7) & 1u) | ((nz >> 14) & 2u) | ((nz >> 21) & 4u) | ((nz >> 28) & 8u)) << (4 * k);
  }
  // nearest occupied voxel before the lane's group (index + 1, 0 = none) and after it (nx - index, 0 = none)
  const int last_in = m ? idx0 + (31 - __clz((int)m)) + 1 : 0;
  const int first_in = m ? nx - (idx0 + __ffs((int)m) - 1) : 0;
  const int pre = wave_scan_max_nonneg(last_in);
  int before = __shfl_up(pre, 1, kWave);
  before = lane == 0 ? 0 : before;
  const int rev = __shfl(first_in, kWave - 1 - lane, kWave);
  const int suf = wave_scan_max_nonneg(rev);  // (in reversed lane order)
  int after = __shfl(suf, kWave - 2 - lane >= 0 ? kWave - 2 - lane : 0, kWave);
  after = lane == kWave - 1 ? 0 : after;
  int left[V];
  int last = before;
#pragma unroll
  for (int k = 0; k < V; ++k) {
    if ((m >> k) & 1u) last = idx0 + k + 1;
    left[k] = last ? idx0 + k + 1 - last : kXInf;
  }
  int nxt = after;
  uint16_t out[V];
#pragma unroll
  for (int k = V - 1; k >= 0; --k) {
    if ((m >> k) & 1u) nxt = nx - (idx0 + k);
    const int right = nxt ? (nx - nxt) - (idx0 + k) : kXInf;
    out[k] = (uint16_t)min(min(left[k], right), kXInf);
  }
  uint2 *dst = reinterpret_cast<uint2 *>(gx + row * nx + idx0);
#pragma unroll
  for (int k = 0; k < V / 4; ++k)
    if (idx0 + 4 * k < nx)
      dst[k] = make_uint2((uint32_t)out[4 * k] | ((uint32_t)out[4 * k + 1] << 16), (uint32_t)out[4 * k + 2] | ((uint32_t)out[4 * k + 3] << 16));
}

constexpr int kSqInf = 1 << 28;

// SrcT = uint16_t: plane distances from row distances (squared while the tile is loaded); uint32_t: volume distances
// from plane distances.  `stride_line` = elements between consecutive voxels of a line, `stride_slab` = elements
// between the slabs a block row works on (grid.y), nline = voxels per line.  Dynamic LDS: nline * TX * 4 (uint16 source) or 6 bytes.
#ifndef NEO_EDT_THREADS
#define NEO_EDT_THREADS 256  // threads of a line-pass block (experiments: 512, 1024)
#endif
constexpr int kEdtThreads = NEO_EDT_THREADS;
template <typename SrcT, int TX, bool FINAL>
__global__ __launch_bounds__(kEdtThreads) void edt3_line_kernel(const SrcT *__restrict__ src, int nx, int nline, size_t stride_line,
                                                        size_t stride_slab, double res, uint32_t *__restrict__ out_sq,
                                                        float *__restrict__ out_dist) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
  // [nline][TX] values: row distances as they are (uint16, squared when read: half the LDS, twice the blocks per CU) or
  // plane distances (already squared); then [nline][TX] minimisers
  using FT = std::conditional_t<sizeof(SrcT) == 2, uint16_t, int>;
  FT *f = reinterpret_cast<FT *>(tile_raw);
  uint16_t *am = reinterpret_cast<uint16_t *>(tile_raw + (size_t)nline * TX * sizeof(FT));
  auto val = [&](int i) -> int {
    const int v = (int)f[i];
    if constexpr (sizeof(SrcT) == 2) return v >= kXInf ? kSqInf : v * v;
    return v;
  };
  const int x0 = blockIdx.x * TX;
  const size_t base = (size_t)blockIdx.y * stride_slab;
  for (int i = threadIdx.x; i < nline * TX; i += kEdtThreads) {
    const int q = i / TX, xl = i - q * TX;
    FT v = sizeof(SrcT) == 2 ? (FT)kXInf : (FT)kSqInf;
    if (x0 + xl < nx) v = (FT)src[base + (size_t)q * stride_line + x0 + xl];
    f[i] = v;
  }
  __syncthreads();
  // One level: `npts` points p = p0 + k * dp, each with the minimiser range of its column taken from the neighbours
  // (or the whole line).  While a level has fewer (point, column) items than the block has threads, G = 2 .. 8 threads
  // share an item -- contiguous parts of its range, combined through `part` (leftmost minimum wins: parts in order,
  // strict comparison) -- so that the coarse levels, few points with long ranges, do not run on a handful of threads.
  // leftmost minimiser of f[q] + (p - q)^2 over a <= q <= b: four candidates' LDS reads in flight at a time, compared in
  // order with a strict "<" (kSqInf + 4095^2 < 2^31)
  auto scan = [&](int p, int xl, int a, int b, int &best, int &arg) {
    int q = a;
    for (; q + 3 <= b; q += 4) {
      const int f0 = val(q * TX + xl), f1 = val((q + 1) * TX + xl), f2 = val((q + 2) * TX + xl), f3 = val((q + 3) * TX + xl);
      const int d0 = p - q, d1 = d0 - 1, d2 = d0 - 2, d3 = d0 - 3;
      const int c0 = f0 + d0 * d0, c1 = f1 + d1 * d1, c2 = f2 + d2 * d2, c3 = f3 + d3 * d3;
      if (c0 < best) { best = c0; arg = q; }
      if (c1 < best) { best = c1; arg = q + 1; }
      if (c2 < best) { best = c2; arg = q + 2; }
      if (c3 < best) { best = c3; arg = q + 3; }
    }
    for (; q <= b; ++q) {
      const int dq = p - q, c = val(q * TX + xl) + dq * dq;
      if (c < best) { best = c; arg = q; }
    }
  };
  __shared__ int part_best[kEdtThreads];
  __shared__ int part_arg[kEdtThreads];
  auto level = [&](int npts, int p0, int dp, int S) {
    int G = 1;
    while (G < 8 && npts * TX * G * 2 <= kEdtThreads) G *= 2;
    const int items = npts * TX * G;
    int p = 0, xl = 0, g = 0;
    const bool mine = (int)threadIdx.x < items || G == 1;
    if (G > 1) {
      if (mine) {
        xl = threadIdx.x % TX;
        g = (threadIdx.x / TX) % G;
        p = min(p0 + (int)(threadIdx.x / (TX * G)) * dp, nline - 1);
        const int lo = S ? am[(p - S) * TX + xl] : 0, hi = S ? am[min(p + S, nline - 1) * TX + xl] : nline - 1;
        const int chunk = (hi - lo + G) / G, a = lo + g * chunk, b = min(hi, a + chunk - 1);
        int best = 0x7fffffff, arg = lo;
        scan(p, xl, a, b, best, arg);
        part_best[threadIdx.x] = best;
        part_arg[threadIdx.x] = arg;
      }
      __syncthreads();
      if (mine && g == 0) {
        int best = part_best[threadIdx.x], arg = part_arg[threadIdx.x];
        for (int j = 1; j < G; ++j) {
          const int c = part_best[threadIdx.x + j * TX];
          if (c < best) { best = c; arg = part_arg[threadIdx.x + j * TX]; }
        }
        am[p * TX + xl] = (uint16_t)arg;
      }
    } else {
      for (int it = threadIdx.x; it < items; it += kEdtThreads) {
        const int k = it / TX;
        xl = it - k * TX;
        p = min(p0 + k * dp, nline - 1);
        const int lo = S ? am[(p - S) * TX + xl] : 0, hi = S ? am[min(p + S, nline - 1) * TX + xl] : nline - 1;
        int best = 0x7fffffff, arg = lo;
        scan(p, xl, lo, hi, best, arg);
        am[p * TX + xl] = (uint16_t)arg;
      }
    }
    __syncthreads();
  };
  level(2, 0, nline - 1, 0);  // the two end points: full scans
  int top = 1;
  while (top < nline - 1) top <<= 1;
  for (int S = top >> 1; S >= 1; S >>= 1)
    level((nline - 1 - S + 2 * S - 1) / (2 * S), S, 2 * S, S);  // points p = S + 2 S k < nline - 1
  for (int i = threadIdx.x; i < nline * TX; i += kEdtThreads) {
    const int p = i / TX, xl = i - p * TX;
    if (x0 + xl >= nx) continue;
    const int q = am[i], dq = p - q;
    const int best = min(val(q * TX + xl) + dq * dq, kSqInf);
    const size_t o = base + (size_t)p * stride_line + x0 + xl;
    if constexpr (FINAL) {
      // no occupied voxel at all: keep a large finite distance (scipy's convention there is an artefact of its
      // virtual background voxel; 3-D scenes always contain the ground slab)
      out_dist[o] = (float)(best >= kSqInf ? 1.0e4 : sqrt((double)best) * res);
    } else {
      out_sq[o] = (uint32_t)best;
    }
  }
}

// The same monotone-minima line pass on PACKED KEYS (round 4), for volumes whose squared diagonal leaves room in 31 bits
// (every scene of BASELINE.json: 300^3 needs 28 bits, 600^3 31).  With h(q) = f(q) + q^2 the cost of candidate q at point
// p is f(q) + (p - q)^2 = h(q) - 2 p q + p^2; the tile holds h(q) << qb (the low qb bits are the slot of the minimiser
// found for POINT q, masked off when q is read as a candidate), and
//     key(p, q) = (h(q) << qb) + q * (1 - (p << (qb + 1)))        [= ((cost - p^2) << qb) | q]
// orders the candidates of one point by (cost, q): the leftmost minimiser is ONE v_mad_i32_i24 and half a v_min3_i32 per
// candidate instead of square / add / compare / two selects (the line passes are bound by vector issue: 4.0 k vector
// instructions a wavefront, 72 % of the SIMDs' cycles, `tools/probe/pmc_edt.sh`).  Unreachable voxels enter as
// `big` = nx^2 + ny^2 + nz^2 + 1, above every real squared distance, so a line's minimum is a real candidate whenever it
// has one.  Same levels, same work distribution, bit-equal results (integers).
template <typename SrcT, int TX, bool FINAL>
__global__ __launch_bounds__(kEdtThreads) void edt3_line_keys_kernel(const SrcT *__restrict__ src, int nx, int nline,
                                                                     size_t stride_line, size_t stride_slab, int nslab, double res,
                                                                     int big, int qb, uint32_t *__restrict__ out_sq,
                                                                     float *__restrict__ out_dist) {
  extern __shared__ __attribute__((aligned(16))) unsigned char tile_raw[];
  static_assert((TX & (TX - 1)) == 0, "TX is a power of two");
  constexpr int LX = TX == 32 ? 5 : (TX == 16 ? 4 : (TX == 8 ? 3 : (TX == 4 ? 2 : 1)));
  int *hk = reinterpret_cast<int *>(tile_raw);  // [nline][TX] keys (and, in their low bits, the minimisers found)
  // the minimiser found for point p lives in the low qb bits of hk[p] (they hold p itself until then, and a candidate's
  // position is added back from its index): 4 bytes a voxel -- 8 workgroups a CU at 300 voxels a line, 4 at 600 (with a
  // separate 2-byte array: 5 and 2; 600^3 5.1 -> 4.0 ms the build).  A candidate's key is masked when it is read.
  const int himask = ~((1 << qb) - 1);
  auto am_get = [&](int i) { return hk[i] & ~himask; };
  auto am_put = [&](int i, int arg) { hk[i] = (hk[i] & himask) | arg; };
#define NEO_EDT_K(x) ((x) & himask)
#define NEO_EDT_NEGP(p) (1 - ((p) << (qb + 1)))  // per unit of q: -(p << (qb + 1)) for the cost, + 1 for q in the low bits
  // 1-D grid, XCD-aware: workgroup b runs on XCD b mod 8; the tiles of one slab share their rows' 128-byte lines (a row
  // of a 16-column tile is 32 or 64 bytes), so a slab's tiles go to ONE XCD, one after the other, and meet in its L2
  const int ntx = (nx + TX - 1) / TX;
  const int bj = blockIdx.x >> 3, slab = (blockIdx.x & 7) + 8 * (bj / ntx);
  if (slab >= nslab) return;
  const int x0 = (bj % ntx) * TX;
  const size_t base = (size_t)slab * stride_slab;
  const int qmask = (1 << qb) - 1;
  // the tile: several loads in flight per thread (one load per thread and trip left the pass waiting on memory latency:
  // 84 of the y pass's 149 us at 300^3 were this loop and the store loop with the levels switched off) -- four
  // neighbouring columns per load where the rows are aligned for it, eight rows per thread in flight
  auto put = [&](int q, int xl, int v, bool in) {
    int c = big;
    if (in) {
      if constexpr (sizeof(SrcT) == 2)
        c = v >= kXInf ? big : v * v;
      else
        c = v >= kSqInf ? big : v;
    }
    hk[q * TX + xl] = ((c + q * q) << qb) | q;
  };
  if (TX >= 4 && ((nx | stride_line | stride_slab) & 3) == 0) {
    struct alignas(4 * sizeof(SrcT)) Vec4 { SrcT v[4]; };
    constexpr int CPR = TX / 4 > 0 ? TX / 4 : 1, RPP = kEdtThreads / CPR;  // threads a row, rows a pass of the block
    const int xl = (threadIdx.x % CPR) * 4, r0 = threadIdx.x / CPR;
    const bool in = x0 + xl < nx;  // (nx is a multiple of four: the four columns are in or out together)
    constexpr int U = 8;
    for (int qa = r0; qa < nline; qa += U * RPP) {
      Vec4 w[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int q = qa + u * RPP;
        if (q < nline && in) w[u] = *reinterpret_cast<const Vec4 *>(src + base + (size_t)q * stride_line + x0 + xl);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int q = qa + u * RPP;
        if (q < nline) {
#pragma unroll
          for (int e = 0; e < 4; ++e) put(q, xl + e, in ? (int)w[u].v[e] : 0, in);
        }
      }
    }
  } else {
    constexpr int U = 8;
    for (int ia = threadIdx.x; ia < nline * TX; ia += U * kEdtThreads) {
      int v[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = ia + u * kEdtThreads, q = i >> LX, xl = i & (TX - 1);
        v[u] = (i < nline * TX && x0 + xl < nx) ? (int)src[base + (size_t)q * stride_line + x0 + xl] : 0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = ia + u * kEdtThreads, q = i >> LX, xl = i & (TX - 1);
        if (i < nline * TX) put(q, xl, v[u], x0 + xl < nx);
      }
    }
  }
  __syncthreads();
  // smallest key of the candidates a <= q <= b of column xl for the point with negP = -(p << (qb + 1)).  (No unrolling
  // beyond the four written out: most ranges of the fine levels hold one to three candidates, and the kernel is bound
  // by the instructions around the loop.)
  auto scan = [&](int negP, int xl, int a, int b, int &best) {
    int q = a;
    const int *h = hk + a * TX + xl;
    int t = __mul24(q, negP);
#pragma clang loop unroll(disable)
    for (; q + 3 <= b; q += 4, h += 4 * TX, t += 4 * negP) {
      const int k0 = NEO_EDT_K(h[0]), k1 = NEO_EDT_K(h[TX]), k2 = NEO_EDT_K(h[2 * TX]), k3 = NEO_EDT_K(h[3 * TX]);
      best = min(min(best, k0 + t), k1 + t + negP);
      best = min(min(best, k2 + t + 2 * negP), k3 + t + 3 * negP);
    }
#pragma clang loop unroll(disable)
    for (; q <= b; ++q, h += TX, t += negP) best = min(best, NEO_EDT_K(h[0]) + t);
  };
  __shared__ int part_key[kEdtThreads];
  const int my_xl = threadIdx.x & (TX - 1), my_k = threadIdx.x >> LX;
  auto level = [&](int npts, int p0, int dp, int S) {
    int G = 1, lg = 0;
    while (G < 8 && npts * TX * G * 2 <= kEdtThreads) G *= 2, ++lg;
    if (G > 1) {
      const int items = npts * TX * G;
      const bool mine = (int)threadIdx.x < items;
      int p = 0, g = 0;
      const int xl = my_xl;
      if (mine) {
        g = my_k & (G - 1);
        p = min(p0 + (int)(threadIdx.x >> (LX + lg)) * dp, nline - 1);
        const int lo = S ? am_get((p - S) * TX + xl) : 0, hi = S ? am_get(min(p + S, nline - 1) * TX + xl) : nline - 1;
        const int chunk = (hi - lo + G) >> lg, a = lo + g * chunk, b = min(hi, a + chunk - 1);
        int best = 0x7fffffff;
        scan(NEO_EDT_NEGP(p), xl, a, b, best);
        part_key[threadIdx.x] = best;
      }
      __syncthreads();
      if (mine && g == 0) {
        int best = part_key[threadIdx.x];
        for (int j = 1; j < G; ++j) best = min(best, part_key[threadIdx.x + j * TX]);
        am_put(p * TX + xl, best & qmask);
      }
    } else {
      // (every point of these levels has both neighbours: p = S + 2 S k < nline - 1)
      const int xl = my_xl;
      for (int k = my_k; k < npts; k += kEdtThreads / TX) {
        const int p = p0 + k * dp;
        const int lo = am_get((p - S) * TX + xl), hi = am_get(min(p + S, nline - 1) * TX + xl);
        int arg = lo;
        if (lo != hi) {  // (neighbours with the same minimiser: it is this point's too)
          int best = 0x7fffffff;
          scan(NEO_EDT_NEGP(p), xl, lo, hi, best);
          arg = best & qmask;
        }
        am_put(p * TX + xl, arg);
      }
    }
    __syncthreads();
  };
  level(2, 0, nline - 1, 0);  // the two end points: full scans
  int top = 1;
  while (top < nline - 1) top <<= 1;
  for (int S = top >> 1; S >= 1; S >>= 1) level((nline - 1 - S + 2 * S - 1) / (2 * S), S, 2 * S, S);
  auto result = [&](int p, int xl) {  // squared distance of voxel p of column xl (>= big: nothing occupied in reach)
    const int q = am_get(p * TX + xl);
    const int key = NEO_EDT_K(hk[q * TX + xl]) + q * NEO_EDT_NEGP(p);
    return (key >> qb) + p * p;  // (arithmetic shift: the key is ((cost - p^2) << qb) | q)
  };
  auto emit = [&](int c) {
    // (the correctly rounded fp64 root is 7 of the z pass's 157 us at 300^3 -- measured with an fp32 root in its place)
    if constexpr (FINAL) return (float)(c >= big ? 1.0e4 : sqrt((double)c) * res);
    else return c >= big ? (uint32_t)kSqInf : (uint32_t)c;
  };
  using OutT = std::conditional_t<FINAL, float, uint32_t>;
  OutT *out = nullptr;
  if constexpr (FINAL) out = out_dist; else out = out_sq;
  if (TX >= 4 && ((nx | stride_line | stride_slab) & 3) == 0) {
    struct alignas(16) Out4 { OutT v[4]; };
    constexpr int CPR = TX / 4 > 0 ? TX / 4 : 1, RPP = kEdtThreads / CPR;
    const int xl = (threadIdx.x % CPR) * 4;
    if (x0 + xl < nx) {
      for (int p = threadIdx.x / CPR; p < nline; p += RPP) {
        Out4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o.v[e] = emit(result(p, xl + e));
        *reinterpret_cast<Out4 *>(out + base + (size_t)p * stride_line + x0 + xl) = o;
      }
    }
  } else {
    for (int i = threadIdx.x; i < nline * TX; i += kEdtThreads) {
      const int p = i >> LX, xl = i & (TX - 1);
      if (x0 + xl >= nx) continue;
      out[base + (size_t)p * stride_line + x0 + xl] = emit(result(p, xl));
    }
  }
}

#undef NEO_EDT_K
#undef NEO_EDT_NEGP

// numpy.gradient, unit spacing: central differences inside, one-sided at the borders
__global__ void gradient_pack_kernel(const double *__restrict__ dist, int W, int H, double4 *__restrict__ rec,
                                     double *__restrict__ gx_out, double *__restrict__ gy_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= W * H) return;
  const int y = i / W, x = i - y * W;
  double gx, gy;
  if (W == 1) gx = 0.0;
  else if (x == 0) gx = dist[i + 1] - dist[i];
  else if (x == W - 1) gx = dist[i] - dist[i - 1];
  else gx = (dist[i + 1] - dist[i - 1]) / 2.0;
  if (H == 1) gy = 0.0;
  else if (y == 0) gy = dist[i + W] - dist[i];
  else if (y == H - 1) gy = dist[i] - dist[i - W];
  else gy = (dist[i + W] - dist[i - W]) / 2.0;
  rec[i] = make_double4(dist[i], gx, gy, 0.0);
  if (gx_out) gx_out[i] = gx;
  if (gy_out) gy_out[i] = gy;
}

__global__ void pack2d_kernel(const double *__restrict__ dist, const double *__restrict__ gx,
                              const double *__restrict__ gy, int n, double4 *__restrict__ rec) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rec[i] = make_double4(dist[i], gx[i], gy[i], 0.0);
}

// 3-D field: convert element type (linear layout)
template <typename SrcT, typename DstT>
__global__ void pack3d_kernel(const SrcT *__restrict__ src, int nx, int ny, int nz, DstT *__restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)nx * ny * nz;
  if (i >= total) return;
  const float v = (float)src[i];
  if constexpr (sizeof(DstT) == 2)
    dst[i] = __float2half(v);
  else
    dst[i] = (DstT)v;
}

// yz-quad layout: dst[voxel][w] = src at (ix, iy + (w & 1), iz + (w >> 1)), clamped at the upper faces
template <typename SrcT, typename DstT>
__global__ void pack3d_yz4_kernel(const SrcT *__restrict__ src, int nx, int ny, int nz, DstT *__restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)nx * ny * nz;
  if (i >= total) return;
  const int ix = (int)(i % nx), iy = (int)((i / nx) % ny), iz = (int)(i / ((size_t)nx * ny));
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int y = min(iy + (w & 1), ny - 1), z = min(iz + (w >> 1), nz - 1);
    const float v = (float)src[((size_t)z * ny + y) * nx + ix];
    if constexpr (sizeof(DstT) == 2)
      dst[i * 4 + w] = __float2half(v);
    else
      dst[i * 4 + w] = (DstT)v;
  }
}

// cell-packed layout: dst[cell][dz][dy][dx] = src at (ix+dx, iy+dy, iz+dz), clamped at the upper faces
template <typename SrcT, typename DstT>
__global__ void pack3d_cell8_kernel(const SrcT *__restrict__ src, int nx, int ny, int nz, DstT *__restrict__ dst) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)nx * ny * nz;
  if (i >= total) return;
  const int ix = (int)(i % nx), iy = (int)((i / nx) % ny), iz = (int)(i / ((size_t)nx * ny));
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    const int x = min(ix + (w & 1), nx - 1), y = min(iy + ((w >> 1) & 1), ny - 1), z = min(iz + (w >> 2), nz - 1);
    const float v = (float)src[((size_t)z * ny + y) * nx + x];
    if constexpr (sizeof(DstT) == 2)
      dst[i * 8 + w] = __float2half(v);
    else
      dst[i * 8 + w] = (DstT)v;
  }
}

// corner-brick layout: one 128-byte line per block of 2 x 2 x 2 cells (fp32: 32 elements a line, 27 used) or 4 x 2 x 2
// cells (fp16: 64 elements, 45 used); element ((cz * 3 + cy) * CX + cx) of block (bx, by, bz) = src at the block's corner
// (cx, cy, cz), clamped at the upper faces; the rest of the line is zero.
// A workgroup packs kBrickXB bricks of one (by, bz) row: their nine source rows (3 y x 3 z) come in with coalesced loads
// through LDS, then every thread assembles 16-byte (fp32) / 8-byte (fp16) pieces of the lines.  (Round 4, first form: one
// thread per four stored elements reading its four corners straight from memory -- 153 us at 300^3, bound by the
// address processing of the scattered 4-byte loads.)
constexpr int kBrickXB = 64;  // (32: 145 us at 300^3 against 126)
template <typename SrcT, typename DstT>
__global__ __launch_bounds__(256) void pack3d_brick_kernel(const SrcT *__restrict__ src, int nx, int ny, int nz, int nbx, int nby,
                                                           DstT *__restrict__ dst) {
  constexpr int SHX = sizeof(DstT) == 4 ? 1 : 2, CX = (1 << SHX) + 1, PER = 128 / (int)sizeof(DstT);
  constexpr int NCOL = (kBrickXB << SHX) + 1;  // corners along x the workgroup's bricks touch
  __shared__ float tile[9 * NCOL];
  const int bx0 = blockIdx.x * kBrickXB, by = blockIdx.y, bz = blockIdx.z;
  {
    constexpr int NL = (9 * NCOL + 255) / 256;  // loads a thread, all in flight before the first is stored
    SrcT w[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int i = min((int)threadIdx.x + u * 256, 9 * NCOL - 1), r = i / NCOL, cxl = i - r * NCOL;
      const int x = min((bx0 << SHX) + cxl, nx - 1), y = min(2 * by + r % 3, ny - 1), z = min(2 * bz + r / 3, nz - 1);
      w[u] = src[((size_t)z * ny + y) * nx + x];
    }
#pragma unroll
    for (int u = 0; u < NL; ++u) {
      const int i = threadIdx.x + u * 256;
      if (i < 9 * NCOL) tile[i] = (float)w[u];
    }
  }
  __syncthreads();
  constexpr int Q = PER / 4;  // four-element pieces a line
  DstT *line0 = dst + ((size_t)((size_t)bz * nby + by) * nbx + bx0) * PER;
  for (int i = threadIdx.x; i < kBrickXB * Q; i += 256) {
    const int bl = i / Q, e0 = (i - bl * Q) * 4;
    if (bx0 + bl >= nbx) break;  // (bl grows with i)
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = e0 + k, r = e / CX, cx = e - r * CX;  // r = cz * 3 + cy
      v[k] = e < 9 * CX ? tile[r * NCOL + (bl << SHX) + cx] : 0.0f;
    }
    DstT *o = line0 + (size_t)bl * PER + e0;
    if constexpr (sizeof(DstT) == 2) {
      const __half2 lo = __floats2half2_rn(v[0], v[1]), hi = __floats2half2_rn(v[2], v[3]);
      uint2 u;
      u.x = *reinterpret_cast<const unsigned int *>(&lo);
      u.y = *reinterpret_cast<const unsigned int *>(&hi);
      *reinterpret_cast<uint2 *>(o) = u;
    } else {
      *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// get_full_state_cmd (traj_utils.py:85-195): one wavefront per trajectory solves the
// coefficients, then its lanes walk the sample times.
template <int D>
__global__ __launch_bounds__(kWave) void traj_state_kernel(int B, int M, DevParams prm,
                                                            const double *__restrict__ x,
                                                            const double *__restrict__ head,
                                                            const double *__restrict__ tail, double hz, int K,
                                                            double *__restrict__ state, int *__restrict__ count) {
  __shared__ double xs[kSlots * kWave];
  __shared__ double cs[kWave * 6 * D];
  __shared__ double tcum[kWave + 1];
  const int b = blockIdx.x;
  if (b >= B) return;
  struct NoMap {};
  struct NoLookup {
    __device__ explicit NoLookup(const NoMap &) {}
  };
  DevParams p = prm;
  NoMap nm;
  DevBackend<D, kSlots, double, NoMap, NoLookup> be(p, nm);
  be.xs = xs;
  be.hist = nullptr;
  be.m = NEO_LBFGS_M;
  be.coeff_out = nullptr;
  load_boundary(be.t, head + (size_t)b * 3 * D, tail + (size_t)b * 3 * D, M);
  const int n = be.t.n;
  const int lane = lane_id();
  typename DevBackend<D, kSlots, double, NoMap, NoLookup>::Vec xv;
#pragma unroll
  for (int k = 0; k < kSlots; ++k) xv.v[k] = (k * kWave + lane < n) ? x[(size_t)b * n + k * kWave + lane] : 0.0;
  be.scatter_x(xv);
  double e, ts;
  const int st = minco_forward<D>(be.t, p, e, ts);
  if (st != 0) {
    if (lane == 0) count[b] = -1;
    return;
  }
  if (lane < M) {
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int d = 0; d < D; ++d) cs[(lane * 6 + k) * D + d] = be.t.c[k][d];
  }
  // sequential prefix sums like Python's sum(ts[:k]) (traj_utils.py:98-101)
  if (lane == 0) tcum[0] = 0.0;
  for (int pce = 0; pce < M; ++pce) {
    const double Tp = rdlane(be.t.T, pce);
    if (lane == 0) tcum[pce + 1] = tcum[pce] + Tp;
  }
  __syncthreads();
  const double total = tcum[M];
  const double step = 1.0 / hz;
  const int cnt = (int)ceil(total / step);  // len(np.arange(0, total, 1/hz))
  if (lane == 0) count[b] = cnt;
  for (int k = lane; k < K; k += kWave) {
    double *out = state + ((size_t)b * K + k) * 3 * D;
    if (k >= cnt) {
#pragma unroll
      for (int q = 0; q < 3 * D; ++q) out[q] = 0.0;
      continue;
    }
    double tt = (double)k * step;
    if (tt > total) tt = total;
    int pc = 0;
    while (pc < M - 1 && tcum[pc + 1] < tt) ++pc;
    const double T = tt - tcum[pc];
    const double T2 = T * T, T3 = T2 * T, T4 = T2 * T2, T5 = T4 * T;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const double c0 = cs[(pc * 6 + 0) * D + d], c1 = cs[(pc * 6 + 1) * D + d], c2 = cs[(pc * 6 + 2) * D + d];
      const double c3 = cs[(pc * 6 + 3) * D + d], c4 = cs[(pc * 6 + 4) * D + d], c5 = cs[(pc * 6 + 5) * D + d];
      out[0 * D + d] = c0 + c1 * T + c2 * T2 + c3 * T3 + c4 * T4 + c5 * T5;
      out[1 * D + d] = c1 + 2.0 * c2 * T + 3.0 * c3 * T2 + 4.0 * c4 * T3 + 5.0 * c5 * T4;
      out[2 * D + d] = 2.0 * c2 + 6.0 * c3 * T + 12.0 * c4 * T2 + 20.0 * c5 * T3;
    }
  }
}

}  // namespace neo

// =================================================================== host side
using namespace neo;

namespace {

void fill_dev_params(neo_ctx *c) {
  const neo_params &p = c->params;
  DevParams &d = c->dev;
  d.v_max = p.v_max;
  d.T_min = p.T_min;
  d.T_max = p.T_max;
  d.safe_dis = p.safe_dis;
  d.delta_t = p.delta_t;
  for (int k = 0; k < 4; ++k) d.w[k] = p.weights[k];
  d.coll_tol = p.collision_cost_tol;
  d.ftol = p.ftol;
  d.gtol = p.gtol;
  d.maxls = p.maxls;
  d.maxiter = p.maxiter;
  d.maxfun = p.maxfun;
  d.stale_T = p.bugcompat_stale_T;
  d.dbg = p.flags;
  d.derive();
}

int ensure_scratch(neo_ctx *c, size_t bytes) {
  if (bytes <= c->scratch_bytes) return NEO_OK;
  if (c->scratch) hipFree(c->scratch);
  c->scratch = nullptr;
  c->scratch_bytes = 0;
  HIPCHK(c, hipMalloc(&c->scratch, bytes));
  c->scratch_bytes = bytes;
  return NEO_OK;
}

int ensure_pinned(neo_ctx *c, size_t bytes) {
  if (bytes <= c->pinned_bytes) return NEO_OK;
  if (c->pinned) hipHostFree(c->pinned);
  c->pinned = nullptr;
  c->pinned_bytes = 0;
  HIPCHK(c, hipHostMalloc(&c->pinned, bytes, hipHostMallocDefault));
  c->pinned_bytes = bytes;
  return NEO_OK;
}

// bump allocator over the scratch buffer
struct Carver {
  char *base;
  size_t off = 0;
  explicit Carver(void *b) : base(static_cast<char *>(b)) {}
  template <typename T>
  T *take(size_t count) {
    off = (off + 255) & ~size_t(255);
    T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
};

int rebuild_tables(neo_ctx *c) {
  if (!c->table_dirty) return NEO_OK;
  std::vector<Map2D> t2;
  std::vector<Map3D> t3;
  for (auto &kv : c->maps) {
    MapEntry &e = kv.second;
    if (e.kind == 0) {
      e.slot = (int)t2.size();
      t2.push_back(e.m2);
    } else if (e.kind == 1) {
      e.slot = (int)t3.size();
      t3.push_back(e.m3);
    }
  }
  if (c->table2d) hipFree(c->table2d);
  if (c->table3d) hipFree(c->table3d);
  c->table2d = c->table3d = nullptr;
  if (!t2.empty()) {
    HIPCHK(c, hipMalloc(&c->table2d, t2.size() * sizeof(Map2D)));
    HIPCHK(c, hipMemcpyAsync(c->table2d, t2.data(), t2.size() * sizeof(Map2D), hipMemcpyHostToDevice, c->stream));
  }
  if (!t3.empty()) {
    HIPCHK(c, hipMalloc(&c->table3d, t3.size() * sizeof(Map3D)));
    HIPCHK(c, hipMemcpyAsync(c->table3d, t3.data(), t3.size() * sizeof(Map3D), hipMemcpyHostToDevice, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->n2d = (int)t2.size();
  c->n3d = (int)t3.size();
  c->table_dirty = false;
  return NEO_OK;
}

// (callers do not hold the context lock yet: take it for the error string)
int check_shape(neo_ctx *c, int B, int M, int D) {
  if (!c) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  if (B < 0 || M < 1 || M > NEO_MAX_PIECES || D < 2 || D > NEO_MAX_DIM)
    return fail(c, NEO_ERR_INVALID, "shape out of range (1 <= M <= 64, D in {2,3})");
  if (D * (M - 1) + M > kSlots * kWave) return fail(c, NEO_ERR_INVALID, "n = D(M-1)+M exceeds 256");
  return NEO_OK;
}


// From this batch size on the kernels take the two-wavefronts-per-SIMD register allocation (three in the all-fp32
// mode, always).  Measured in round 3, one launch at a time, two waves against one: fp64 mode +4 % at 1024, +9 % at 2048,
// +23 % at 3072 trajectories; mixed mode +2 %, +5 %, +23 % (round 2 had found one wave faster up to 2048: 9.1 against
// 10.1 ms -- before the per-evaluation loads and the spills of the two-waves kernels were gone).
constexpr int kTwoWavesFromBatch = 1024;

int fail_locked(neo_ctx *c, int code, const char *msg) {
  std::lock_guard<std::recursive_mutex> g(c->mu);
  return fail(c, code, msg);
}

int dispatch_opt(neo_ctx *c, int kind, int elem, int layout, int D, const OptArgs &a) {
  const bool f32 = c->params.sample_dtype == NEO_F32;
  if (kind == 0) {
    // small planar problems on the reference's own map (M = 3 -> n = 7): eight replans per wavefront, opt-in
    if ((c->params.flags & NEO_FLAG_LANE_GROUPS) && D == 2 && !a.slots && a.M <= 16 && D * (a.M - 1) + a.M <= 32)
      return launch_opt_groups_2d(c, f32, a);
    const int fl2 = c->params.flags;
    // all-fp32 mode on the reference's own map: the timed arithmetic, pinned to the reference's fixtures there
    if ((fl2 & NEO_FLAG_F32_SOLVE) && f32) return launch_opt_2d_x(c, D, a);
    if (D == 2 && slots_for(a.M, D) <= 2 &&
        ((a.B >= kTwoWavesFromBatch && !(fl2 & NEO_FLAG_ONE_WAVE_PER_SIMD)) ||
         (fl2 & NEO_FLAG_TWO_WAVES_PER_SIMD)))
      return launch_opt_2d_w2(c, f32, a);
    return launch_opt_2d(c, D, f32, a);
  }
  if (D != 3) return fail(c, NEO_ERR_INVALID, "a 3-D map needs D = 3");
  const int fl = c->params.flags;
  if ((fl & NEO_FLAG_LANE_GROUPS) && f32 && (layout == NEO_LAYOUT_LINEAR || layout == NEO_LAYOUT_YZ4 || layout == NEO_LAYOUT_BRICK) && !a.slots &&
      a.M <= 16 && D * (a.M - 1) + a.M <= 32)
    return launch_opt_groups(c, elem, layout, a);
  if ((fl & NEO_FLAG_F32_SOLVE) && f32) return launch_opt_3d_x(c, elem, layout, a);  // all-fp32 mode
  // two trajectories per SIMD for calls that queue for the SIMDs anyway (3-D fields, fp32 sampling; beyond n = 128 with
  // the pairs stored in fp32 so that eight wavefronts still fit a CU's LDS, neo_kernels.hpp pairs_in_f32)
  const bool two = f32 && slots_for(a.M, D) <= NEO_W2_MAX_SLOTS &&
                   ((a.B >= kTwoWavesFromBatch && !(fl & NEO_FLAG_ONE_WAVE_PER_SIMD)) || (fl & NEO_FLAG_TWO_WAVES_PER_SIMD));
  if (two) return launch_opt_3d_w2(c, elem, layout, a);
  // fp64 sampling (the parity mode): two wavefronts per SIMD for n <= 128 when the batch queues for the SIMDs (the
  // unit is compiled so that these kernels spill 0 - 13 registers, build.py; cfg2 360 k -> 637 k traj/s, M = 25 239 k
  // -> 400 k, M = 32 162 k -> 241 k; four FLAT slots would spill 112)
  if (!f32 && slots_for(a.M, D) <= 2 &&
      ((a.B >= kTwoWavesFromBatch && !(fl & NEO_FLAG_ONE_WAVE_PER_SIMD)) || (fl & NEO_FLAG_TWO_WAVES_PER_SIMD)))
    return launch_opt_3d_f64_w2(c, elem, layout, a);
  return f32 ? launch_opt_3d_f32(c, elem, layout, a) : launch_opt_3d_f64(c, elem, layout, a);
}

void drain_profile(neo_ctx *c) {
  for (int k = 0; k < NEO_KERNEL_COUNT; ++k) {
    for (auto &pr : c->prof[k].pending) {
      float ms = 0.f;
      hipEventSynchronize(pr.second);
      hipEventElapsedTime(&ms, pr.first, pr.second);
      c->prof[k].ms += ms;
      c->prof[k].launches += 1;
      hipEventDestroy(pr.first);
      hipEventDestroy(pr.second);
    }
    c->prof[k].pending.clear();
  }
}

}  // namespace

// =================================================================== C ABI
extern "C" {

int neo_abi_version(void) { return NEO_ABI_VERSION; }

int neo_params_default(neo_params *p) {
  if (!p) return NEO_ERR_INVALID;
  memset(p, 0, sizeof(*p));
  // launch/config/planner_config.yaml:2-13
  p->v_max = 1.0;
  p->T_min = 0.5;
  p->T_max = 5.0;
  p->safe_dis = 0.7;
  p->delta_t = 0.1;
  p->weights[0] = 1.0;
  p->weights[1] = 1.0;
  p->weights[2] = 1.0;
  p->weights[3] = 10000.0;
  p->collision_cost_tol = 5.0;
  // expert_planner.py:213-225
  p->ftol = 1e-4;
  p->gtol = 1e-4;
  p->maxls = 20;
  p->maxiter = 15000;
  p->maxfun = 15000;
  p->bugcompat_stale_T = 1;
  p->sample_dtype = NEO_F64;
  return NEO_OK;
}

int neo_ctx_create(int device_id, void *stream, neo_ctx **out) {
  if (!out) return NEO_ERR_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return NEO_ERR_HIP;
  if (device_id < 0 || device_id >= count) return NEO_ERR_INVALID;
  if (hipSetDevice(device_id) != hipSuccess) return NEO_ERR_HIP;
  neo_ctx *c = new neo_ctx();
  c->device = device_id;
  if (stream) {
    c->stream = static_cast<hipStream_t>(stream);
  } else {
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
      delete c;
      return NEO_ERR_HIP;
    }
    c->own_stream = true;
  }
  c->home_stream = c->stream;
  neo_params_default(&c->params);
  fill_dev_params(c);
  *out = c;
  return NEO_OK;
}

int neo_ctx_set_stream(neo_ctx *c, void *stream) {
  if (!c) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  c->stream = stream ? static_cast<hipStream_t>(stream) : c->home_stream;
  return NEO_OK;
}

int neo_ctx_destroy(neo_ctx *c) {
  if (!c) return NEO_ERR_INVALID;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  if (c->stream != c->home_stream) hipStreamSynchronize(c->home_stream);
  drain_profile(c);
  for (auto &kv : c->maps)
    if (kv.second.data) hipFree(kv.second.data);
  if (c->order_buf) hipFree(c->order_buf);
  if (c->sample_order) hipFree(c->sample_order);
  if (c->tickets) hipFree(c->tickets);
  if (c->table2d) hipFree(c->table2d);
  if (c->table3d) hipFree(c->table3d);
  if (c->scratch) hipFree(c->scratch);
  if (c->pinned) hipHostFree(c->pinned);
  if (c->own_stream) hipStreamDestroy(c->home_stream);
  delete c;
  return NEO_OK;
}

const char *neo_last_error(neo_ctx *c) { return c ? c->err.c_str() : "null context"; }

int neo_params_set(neo_ctx *c, const neo_params *p) {
  if (!c || !p) return NEO_ERR_INVALID;
  if (!(p->T_max > p->T_min) || !(p->delta_t > 0.0) || p->maxls < 1)
    return fail(c, NEO_ERR_INVALID, "bad parameters");
  if (p->sample_dtype != NEO_F64 && p->sample_dtype != NEO_F32)
    return fail(c, NEO_ERR_INVALID, "sample_dtype must be NEO_F64 or NEO_F32");
  std::lock_guard<std::recursive_mutex> g(c->mu);
  c->params = *p;
  fill_dev_params(c);
  return NEO_OK;
}

int neo_ctx_synchronize(neo_ctx *c) {
  if (!c) return NEO_ERR_INVALID;
  hipStream_t st;
  {
    std::lock_guard<std::recursive_mutex> g(c->mu);
    st = c->stream;
  }
  const hipError_t e = hipStreamSynchronize(st);  // (not under the lock: other threads may keep launching)
  if (e != hipSuccess) return fail_locked(c, NEO_ERR_HIP, hipGetErrorString(e));
  return NEO_OK;
}

// Map updates wait for ALL device work first (hipDeviceSynchronize), not only for the context's current stream:
// `_dev` launches of neo_ctx_set_stream callers may still be reading the scene's buffer on other streams.
static int drop_locked(neo_ctx *c, int scene_id) {
  auto it = c->maps.find(scene_id);
  if (it != c->maps.end()) {
    hipDeviceSynchronize();
    if (it->second.data) hipFree(it->second.data);
    c->maps.erase(it);
    c->table_dirty = true;
  }
  return NEO_OK;
}

int neo_esdf_drop(neo_ctx *c, int scene_id) {
  if (!c) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  hipSetDevice(c->device);
  return drop_locked(c, scene_id);
}

int neo_esdf_upload_2d(neo_ctx *c, int scene_id, const double *dist, const double *gx, const double *gy, int W,
                       int H, double res, double ox, double oy) {
  if (!c || !dist || !gx || !gy || W < 1 || H < 1 || !(res > 0.0)) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  hipSetDevice(c->device);
  drop_locked(c, scene_id);
  const size_t ncell = (size_t)W * H;
  int rc = ensure_scratch(c, 3 * ncell * sizeof(double) + 1024);
  if (rc) return rc;
  Carver cv(c->scratch);
  double *d0 = cv.take<double>(ncell), *d1 = cv.take<double>(ncell), *d2 = cv.take<double>(ncell);
  MapEntry e;
  e.kind = 0;
  e.elem = NEO_F64;
  DevBuf rec;
  HIPCHK(c, rec.alloc(ncell * sizeof(double4)));
  HIPCHK(c, hipMemcpyAsync(d0, dist, ncell * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d1, gx, ncell * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d2, gy, ncell * sizeof(double), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(pack2d_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, c->stream, d0, d1, d2,
                     (int)ncell, rec.as<double4>());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  e.data = rec.release();
  e.m2 = Map2D{static_cast<const double4 *>(e.data), W, H, res, ox, oy};
  c->maps[scene_id] = e;
  c->table_dirty = true;
  return NEO_OK;
}

// the device work of neo_esdf_build_2d into `rec`; scratch carved by the caller
static int build_2d_into(neo_ctx *c, double4 *rec, const int8_t *occ, int W, int H, double res, double *out_dist,
                         double *out_gx, double *out_gy) {
  const size_t ncell = (size_t)W * H;
  Carver cv(c->scratch);
  int8_t *d_occ = cv.take<int8_t>(ncell);
  int *d_g = cv.take<int>(ncell);
  int *d_v = cv.take<int>(ncell);
  double *d_z = cv.take<double>((size_t)H * (W + 1));
  double *d_dist = cv.take<double>(ncell);
  double *d_gx = cv.take<double>(ncell);
  double *d_gy = cv.take<double>(ncell);
  HIPCHK(c, hipMemcpyAsync(d_occ, occ, ncell, hipMemcpyHostToDevice, c->stream));
  {
    ProfScope ps(c, NEO_KERNEL_ESDF_BUILD);
    if (W <= 512 && H <= 512) {
      const dim3 grid((W + 63) / 64, H);
      hipLaunchKernelGGL(edt2_columns_bf_kernel, grid, dim3(64), 0, c->stream, d_occ, W, H, d_g);
      hipLaunchKernelGGL(edt2_rows_bf_kernel, grid, dim3(64), 0, c->stream, d_g, W, H, res, d_dist);
    } else {
      hipLaunchKernelGGL(edt_columns_kernel, dim3((W + 63) / 64), dim3(64), 0, c->stream, d_occ, W, H, d_g);
      hipLaunchKernelGGL(edt_rows_kernel, dim3((H + 63) / 64), dim3(64), 0, c->stream, d_g, W, H, res, d_v, d_z,
                         d_dist);
    }
    hipLaunchKernelGGL(gradient_pack_kernel, dim3((unsigned)((ncell + 255) / 256)), dim3(256), 0, c->stream,
                       d_dist, W, H, rec, d_gx, d_gy);
  }
  HIPCHK(c, hipGetLastError());
  if (out_dist) HIPCHK(c, hipMemcpyAsync(out_dist, d_dist, ncell * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (out_gx) HIPCHK(c, hipMemcpyAsync(out_gx, d_gx, ncell * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (out_gy) HIPCHK(c, hipMemcpyAsync(out_gy, d_gy, ncell * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return NEO_OK;
}

int neo_esdf_build_2d(neo_ctx *c, int scene_id, const int8_t *occ, int W, int H, double res, double ox, double oy,
                      double *out_dist, double *out_gx, double *out_gy) {
  if (!c || !occ || W < 1 || H < 1 || !(res > 0.0)) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  hipSetDevice(c->device);
  const size_t ncell = (size_t)W * H;
  const size_t need = ncell * (sizeof(int8_t) + 2 * sizeof(int) + 3 * sizeof(double)) +
                      (size_t)H * (W + 1) * sizeof(double) + 4096;
  int rc = ensure_scratch(c, need);
  if (rc) return rc;
  // a map update of the same size rewrites the scene's record buffer in place (no hipFree / hipMalloc per
  // update), once nothing on the device can still be reading it
  auto it = c->maps.find(scene_id);
  const bool reuse = it != c->maps.end() && it->second.kind == 0 && it->second.data &&
                     (size_t)it->second.m2.W * it->second.m2.H == ncell;
  if (reuse) {
    HIPCHK(c, hipDeviceSynchronize());
    rc = build_2d_into(c, static_cast<double4 *>(it->second.data), occ, W, H, res, out_dist, out_gx, out_gy);
    if (rc) {
      drop_locked(c, scene_id);  // the buffer may be half rewritten: the scene has no map any more
      return rc;
    }
    const Map2D m{static_cast<const double4 *>(it->second.data), W, H, res, ox, oy};
    if (memcmp(&m, &it->second.m2, sizeof(m)) != 0) c->table_dirty = true;  // same buffer: usually the same descriptor
    it->second.m2 = m;
    return NEO_OK;
  }
  drop_locked(c, scene_id);
  DevBuf rec;
  HIPCHK(c, rec.alloc(ncell * sizeof(double4)));
  rc = build_2d_into(c, rec.as<double4>(), occ, W, H, res, out_dist, out_gx, out_gy);
  if (rc) return rc;
  MapEntry e;
  e.kind = 0;
  e.elem = NEO_F64;
  e.data = rec.release();
  e.m2 = Map2D{static_cast<const double4 *>(e.data), W, H, res, ox, oy};
  c->maps[scene_id] = e;
  c->table_dirty = true;
  return NEO_OK;
}

int neo_esdf_upload_3d(neo_ctx *c, int scene_id, const void *dist, int src_dtype, int src_is_device, int nx, int ny,
                       int nz, double res, const double origin[3], int store_dtype, int layout) {
  if (!c || !dist || !origin || nx < 2 || ny < 2 || nz < 2 || !(res > 0.0)) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  if (src_dtype != NEO_F64 && src_dtype != NEO_F32) return fail(c, NEO_ERR_INVALID, "src_dtype must be f64 or f32");
  if (store_dtype != NEO_F32 && store_dtype != NEO_F16)
    return fail(c, NEO_ERR_INVALID, "store_dtype must be f32 or f16");
  if (layout != NEO_LAYOUT_LINEAR && layout != NEO_LAYOUT_YZ4 && layout != NEO_LAYOUT_CELL8 && layout != NEO_LAYOUT_BRICK)
    return fail(c, NEO_ERR_INVALID, "bad layout");
  // the lookups form voxel indices with 24-bit multiplies: (iz * ny + iy) * nx + ix
  if ((size_t)ny * nz >= ((size_t)1 << 24) || (size_t)nx >= ((size_t)1 << 24))
    return fail(c, NEO_ERR_INVALID, "field too large: ny * nz and nx must be below 2^24");
  hipSetDevice(c->device);
  drop_locked(c, scene_id);
  const size_t nvox = (size_t)nx * ny * nz;
  const size_t ssz = src_dtype == NEO_F64 ? 8 : 4, dsz = store_dtype == NEO_F32 ? 4 : 2;
  // corner bricks: blocks of 2 x 2 x 2 cells (fp32) / 4 x 2 x 2 cells (fp16), one 128-byte line each
  const int bshx = store_dtype == NEO_F32 ? 1 : 2;
  const int nbx = ((nx - 1) + (1 << bshx) - 1) >> bshx, nby = (ny - 1 + 1) / 2, nbz = (nz - 1 + 1) / 2;
  const size_t nstore = layout == NEO_LAYOUT_YZ4 ? nvox * 4 : (layout == NEO_LAYOUT_CELL8 ? nvox * 8 :
                        (layout == NEO_LAYOUT_BRICK ? (size_t)nbx * nby * nbz * (128 / dsz) : nvox));
  if ((nstore + 64) * dsz >= (size_t)4 << 30) return fail(c, NEO_ERR_INVALID, "field too large for 32-bit buffer offsets in this layout");
  if (layout == NEO_LAYOUT_BRICK && (nby > 65535 || nbz > 65535))
    return fail(c, NEO_ERR_INVALID, "field too large for the brick layout: at most 131070 voxels along y and z");
  const void *src = dist;
  DevBuf staged, field;
  if (!src_is_device) {
    HIPCHK(c, staged.alloc(nvox * ssz));
    HIPCHK(c, hipMemcpyAsync(staged.p, dist, nvox * ssz, hipMemcpyHostToDevice, c->stream));
    src = staged.p;
  }
  MapEntry e;
  e.kind = 1;
  e.elem = store_dtype;
  // +64 elements of slack: the x-pair load of the last voxel row touches one element past the end
  HIPCHK(c, field.alloc((nstore + 64) * dsz));
  // (the pack kernels write every stored element; only the 64 elements of padding behind them need zeroing)
  HIPCHK(c, hipMemsetAsync(static_cast<char *>(field.p) + nstore * dsz, 0, 64 * dsz, c->stream));
  const dim3 grid((unsigned)((nvox + 255) / 256)), blk(256);
#define NEO_PACK(KERNEL)                                                                                               \
  do {                                                                                                             \
    if (src_dtype == NEO_F64 && store_dtype == NEO_F32)                                                            \
      hipLaunchKernelGGL((KERNEL<double, float>), grid, blk, 0, c->stream, (const double *)src, nx, ny, nz, (float *)field.p);   \
    else if (src_dtype == NEO_F64)                                                                                 \
      hipLaunchKernelGGL((KERNEL<double, __half>), grid, blk, 0, c->stream, (const double *)src, nx, ny, nz, (__half *)field.p); \
    else if (store_dtype == NEO_F32)                                                                               \
      hipLaunchKernelGGL((KERNEL<float, float>), grid, blk, 0, c->stream, (const float *)src, nx, ny, nz, (float *)field.p);     \
    else                                                                                                           \
      hipLaunchKernelGGL((KERNEL<float, __half>), grid, blk, 0, c->stream, (const float *)src, nx, ny, nz, (__half *)field.p);   \
  } while (0)
  if (layout == NEO_LAYOUT_BRICK) {
    const dim3 gb((unsigned)((nbx + kBrickXB - 1) / kBrickXB), (unsigned)nby, (unsigned)nbz);
    if (src_dtype == NEO_F64 && store_dtype == NEO_F32)
      hipLaunchKernelGGL((pack3d_brick_kernel<double, float>), gb, blk, 0, c->stream, (const double *)src, nx, ny, nz, nbx, nby, (float *)field.p);
    else if (src_dtype == NEO_F64)
      hipLaunchKernelGGL((pack3d_brick_kernel<double, __half>), gb, blk, 0, c->stream, (const double *)src, nx, ny, nz, nbx, nby, (__half *)field.p);
    else if (store_dtype == NEO_F32)
      hipLaunchKernelGGL((pack3d_brick_kernel<float, float>), gb, blk, 0, c->stream, (const float *)src, nx, ny, nz, nbx, nby, (float *)field.p);
    else
      hipLaunchKernelGGL((pack3d_brick_kernel<float, __half>), gb, blk, 0, c->stream, (const float *)src, nx, ny, nz, nbx, nby, (__half *)field.p);
  } else if (layout == NEO_LAYOUT_CELL8)
    NEO_PACK(pack3d_cell8_kernel);
  else if (layout == NEO_LAYOUT_YZ4)
    NEO_PACK(pack3d_yz4_kernel);
  else
    NEO_PACK(pack3d_kernel);
#undef NEO_PACK
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipStreamSynchronize(c->stream));
  e.data = field.release();
  e.m3 = Map3D{e.data, nx, ny, nz, layout, res, origin[0], origin[1], origin[2], (unsigned int)((nstore + 64) * dsz)};
  e.m3.derive();
  e.m3.nbx = nbx;
  e.m3.nby = nby;
  c->maps[scene_id] = e;
  c->table_dirty = true;
  return NEO_OK;
}

int neo_esdf_build_3d(neo_ctx *c, int scene_id, const uint8_t *occ, int occ_is_device, int nx, int ny, int nz,
                      double res, const double origin[3], int store_dtype, int layout, float *out_dist) {
  if (!c || !occ || !origin || nx < 2 || ny < 2 || nz < 2 || !(res > 0.0)) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  if (store_dtype != NEO_F32 && store_dtype != NEO_F16) return fail(c, NEO_ERR_INVALID, "store_dtype must be f32 or f16");
  if (nx > 4096 || ny > 4096 || nz > 4096) return fail(c, NEO_ERR_INVALID, "3-D ESDF build: at most 4096 voxels per axis");
  const size_t nvox = (size_t)nx * ny * nz;
  // the passes' intermediates and the fp32 distances live in the context's scratch buffer (kept between calls: a map
  // update at sensor rate then allocates nothing but the field itself)
  float *d_dist_p = nullptr;
  {
    hipSetDevice(c->device);
    DevBuf d_occ;
    const uint8_t *src = occ;
    if (!occ_is_device) {
      HIPCHK(c, d_occ.alloc(nvox));
      HIPCHK(c, hipMemcpyAsync(d_occ.p, occ, nvox, hipMemcpyHostToDevice, c->stream));
      src = d_occ.as<uint8_t>();
    }
    int rc = ensure_scratch(c, nvox * (sizeof(uint16_t) + sizeof(uint32_t) + sizeof(float)) + 4 * 256);
    if (rc) return rc;
    Carver cv(c->scratch);
    uint16_t *d_gx = cv.take<uint16_t>(nvox);
    uint32_t *d_sq = cv.take<uint32_t>(nvox);
    d_dist_p = cv.take<float>(nvox);
    {
      ProfScope ps(c, NEO_KERNEL_ESDF_BUILD);
      const size_t rows = (size_t)ny * nz;
      // (the occupancy's rows must be 4-byte aligned for the wide form: nx a multiple of four and an aligned base)
      const bool wide = nx % 4 == 0 && (reinterpret_cast<uintptr_t>(src) & 3) == 0;
      if (wide && nx <= 8 * kWave)
        hipLaunchKernelGGL(edt3_xv_kernel<8>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, c->stream, src, nx, rows, d_gx);
      else if (wide && nx <= 16 * kWave)
        hipLaunchKernelGGL(edt3_xv_kernel<16>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, c->stream, src, nx, rows, d_gx);
      else
        hipLaunchKernelGGL(edt3_x_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 4 * (size_t)nx * sizeof(uint16_t),
                           c->stream, src, nx, rows, d_gx);
      // tiles of at most 64 KB of LDS INCLUDING the kernel's 2 KB of static part_best / part_arg (4 bytes a voxel in the y
      // pass, 6 in the z pass): TX x-columns by the whole line
      constexpr size_t kEdtTile = 65536 - 2 * kEdtThreads * sizeof(int);
      const size_t plane = (size_t)nx * ny;
#define NEO_EDT_LINE(SRC, TXV, FINAL, srcp, nline, sline, sslab, nslab, outsq, outd)                                 \
  hipLaunchKernelGGL((edt3_line_kernel<SRC, TXV, FINAL>), dim3((unsigned)((nx + TXV - 1) / TXV), (unsigned)(nslab)), \
                     dim3(kEdtThreads), (size_t)(nline) * TXV * (sizeof(SRC) == 2 ? 4 : 6), c->stream, srcp, nx, nline, sline, \
                     sslab, res, outsq, outd)
      // 16 columns a tile while that keeps four or more blocks on a CU (measured at 300^3: y pass 248 -> 204 us, z pass
      // 384 -> 252 us against 32 columns; 8 columns in the z pass: 278), else the widest tile that fits 64 KB
#define NEO_EDT_PASS(SRC, FINAL, srcp, nline, sline, sslab, nslab, outsq, outd)                             \
  if ((size_t)(nline) * 16 * (sizeof(SRC) == 2 ? 4 : 6) <= 40960)                                           \
    NEO_EDT_LINE(SRC, 16, FINAL, srcp, nline, sline, sslab, nslab, outsq, outd);                             \
  else if ((size_t)(nline) * 32 * 6 <= kEdtTile) NEO_EDT_LINE(SRC, 32, FINAL, srcp, nline, sline, sslab, nslab, outsq, outd); \
  else if ((size_t)(nline) * 16 * 6 <= kEdtTile) NEO_EDT_LINE(SRC, 16, FINAL, srcp, nline, sline, sslab, nslab, outsq, outd); \
  else if ((size_t)(nline) * 8 * 6 <= kEdtTile) NEO_EDT_LINE(SRC, 8, FINAL, srcp, nline, sline, sslab, nslab, outsq, outd);   \
  else NEO_EDT_LINE(SRC, 2, FINAL, srcp, nline, sline, sslab, nslab, outsq, outd)
      // packed-key form of the same passes where (squared diagonal + 2 nline^2) << bits(nline) fits 31 bits
#define NEO_EDT_KEYS(SRC, TXV, FINAL, srcp, nline, sline, sslab, nslab, outsq, outd, qb)                                  \
  hipLaunchKernelGGL((edt3_line_keys_kernel<SRC, TXV, FINAL>),                                                          \
                     dim3((unsigned)(((nx + TXV - 1) / TXV) * (((nslab) + 7) / 8) * 8)), dim3(kEdtThreads),                \
                     (size_t)(nline) * TXV * NEO_EDT_KEY_BYTES, c->stream, srcp, nx, nline, sline, sslab, (int)(nslab), res, big, qb, outsq, outd)
#define NEO_EDT_KEY_BYTES 4
#define NEO_EDT_KEYS_PASS(SRC, FINAL, srcp, nline, sline, sslab, nslab, outsq, outd, qb)                                    \
  if ((size_t)(nline) * 16 * NEO_EDT_KEY_BYTES <= 40960) NEO_EDT_KEYS(SRC, 16, FINAL, srcp, nline, sline, sslab, nslab, outsq, outd, qb);     \
  else if ((size_t)(nline) * 32 * NEO_EDT_KEY_BYTES <= kEdtTile) NEO_EDT_KEYS(SRC, 32, FINAL, srcp, nline, sline, sslab, nslab, outsq, outd, qb); \
  else if ((size_t)(nline) * 16 * NEO_EDT_KEY_BYTES <= kEdtTile) NEO_EDT_KEYS(SRC, 16, FINAL, srcp, nline, sline, sslab, nslab, outsq, outd, qb); \
  else if ((size_t)(nline) * 8 * NEO_EDT_KEY_BYTES <= kEdtTile) NEO_EDT_KEYS(SRC, 8, FINAL, srcp, nline, sline, sslab, nslab, outsq, outd, qb);   \
  else NEO_EDT_KEYS(SRC, 2, FINAL, srcp, nline, sline, sslab, nslab, outsq, outd, qb)
      const long long diag2 = (long long)nx * nx + (long long)ny * ny + (long long)nz * nz + 1;
      const int big = (int)std::min<long long>(diag2, 1 << 30);
      auto key_bits = [&](int nline) {  // bits of the minimiser field, or 0 when the keys do not fit
        int qb = 1;
        while ((1 << qb) < nline) ++qb;
        const long long span = (diag2 + 2LL * nline * nline) << qb;
        return (span < (1LL << 31) && ((long long)nline << (qb + 1)) < (1LL << 23)) ? qb : 0;
      };
      // (neo_esdf_build_config(ctx, NEO_EDT_GENERIC_LINES): the general form for every volume -- the tests run both)
      const bool generic = (c->edt_flags & NEO_EDT_GENERIC_LINES) != 0;
      const int qby = generic ? 0 : key_bits(ny), qbz = generic ? 0 : key_bits(nz);
      // pass Y: lines along y (stride nx) in every z slab; pass Z: lines along z (stride nx * ny) for every y row
      if (qby) {
        NEO_EDT_KEYS_PASS(uint16_t, false, d_gx, ny, (size_t)nx, plane, nz, d_sq, (float *)nullptr, qby);
      } else {
        NEO_EDT_PASS(uint16_t, false, d_gx, ny, (size_t)nx, plane, nz, d_sq, (float *)nullptr);
      }
      if (qbz) {
        NEO_EDT_KEYS_PASS(uint32_t, true, d_sq, nz, plane, (size_t)nx, ny, (uint32_t *)nullptr, d_dist_p, qbz);
      } else {
        NEO_EDT_PASS(uint32_t, true, d_sq, nz, plane, (size_t)nx, ny, (uint32_t *)nullptr, d_dist_p);
      }
#undef NEO_EDT_KEYS_PASS
#undef NEO_EDT_KEYS
#undef NEO_EDT_PASS
#undef NEO_EDT_LINE
    }
    HIPCHK(c, hipGetLastError());
    if (out_dist) HIPCHK(c, hipMemcpyAsync(out_dist, d_dist_p, nvox * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  const int rc_up = neo_esdf_upload_3d(c, scene_id, d_dist_p, NEO_F32, 1, nx, ny, nz, res, origin, store_dtype, layout);
  // the intermediates stay in the context's scratch for the next update of a scene of this size -- up to 512 MB (a 300^3
  // scene: 270 MB); beyond that (600^3: 2.2 GB, 1000^3: 10 GB) they are released, a map build does not pin gigabytes of
  // HBM for the lifetime of the context
  if (c->scratch_bytes > ((size_t)512 << 20)) {
    hipStreamSynchronize(c->stream);
    hipFree(c->scratch);
    c->scratch = nullptr;
    c->scratch_bytes = 0;
  }
  return rc_up;
}

int neo_esdf_query(neo_ctx *c, int scene_id, int n, const double *pts, double *dist, double *grad) {
  if (!c || n < 0 || !pts || !dist) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  hipSetDevice(c->device);
  auto it = c->maps.find(scene_id);
  if (it == c->maps.end()) return fail(c, NEO_ERR_NO_MAP, "no ESDF for this scene");
  if (n == 0) return NEO_OK;
  const MapEntry &e = it->second;
  const int dm = e.kind == 0 ? 2 : 3;
  int rc = ensure_scratch(c, (size_t)n * (2 * dm + 1) * sizeof(double) + 1024);
  if (rc) return rc;
  Carver cv(c->scratch);
  double *d_p = cv.take<double>((size_t)n * dm), *d_d = cv.take<double>(n), *d_g = cv.take<double>((size_t)n * dm);
  HIPCHK(c, hipMemcpyAsync(d_p, pts, (size_t)n * dm * sizeof(double), hipMemcpyHostToDevice, c->stream));
  const dim3 grid((n + 127) / 128), blk(128);
  if (e.kind == 0)
    hipLaunchKernelGGL((query_kernel<double, Map2D, Lookup2D<double>, 2>), grid, blk, 0, c->stream, n, e.m2, d_p, d_d,
                       d_g);
  else if (e.elem == NEO_F32)
    hipLaunchKernelGGL((query_kernel<double, Map3D, Lookup3D<double, float, 9>, 3>), grid, blk, 0, c->stream, n, e.m3,
                       d_p, d_d, d_g);
  else
    hipLaunchKernelGGL((query_kernel<double, Map3D, Lookup3D<double, __half, 9>, 3>), grid, blk, 0, c->stream, n, e.m3,
                       d_p, d_d, d_g);
  HIPCHK(c, hipMemcpyAsync(dist, d_d, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (grad) HIPCHK(c, hipMemcpyAsync(grad, d_g, (size_t)n * dm * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return NEO_OK;
}

int neo_cost_grad_batch_dev(neo_ctx *c, int scene_id, int B, int M, int D, const double *x, const double *head,
                            const double *tail, double *cost, double *costs4, double *grad, double *coeffs,
                            int32_t *status) {
  int rc = check_shape(c, B, M, D);
  if (rc) return rc;
  if (!x || !head || !tail || !cost || !costs4 || !grad) return fail_locked(c, NEO_ERR_INVALID, "null buffer");
  std::lock_guard<std::recursive_mutex> g(c->mu);
  hipSetDevice(c->device);
  auto it = c->maps.find(scene_id);
  if (it == c->maps.end()) return fail(c, NEO_ERR_NO_MAP, "no ESDF for this scene");
  if (B == 0) return NEO_OK;
  ProfScope ps(c, NEO_KERNEL_EVAL);
  const EvalArgs ea{B, M, x, head, tail, cost, costs4, grad, coeffs, status};
  rc = dispatch_eval(c, it->second, D, ea);
  if (rc) return rc;
  HIPCHK(c, hipGetLastError());
  return NEO_OK;
}

int neo_cost_grad_batch(neo_ctx *c, int scene_id, int B, int M, int D, const double *x, const double *head,
                        const double *tail, double *cost, double *costs4, double *grad, double *coeffs,
                        int32_t *status) {
  int rc = check_shape(c, B, M, D);
  if (rc) return rc;
  if (!x || !head || !tail || !cost || !costs4 || !grad) return fail_locked(c, NEO_ERR_INVALID, "null buffer");
  if (B == 0) return NEO_OK;
  std::lock_guard<std::recursive_mutex> whole_call(c->mu);  // scratch buffers stay ours until the copies back are done
  const size_t n = (size_t)D * (M - 1) + M, bs = (size_t)B;
  // scratch layout [x | head | tail | cost | costs4 | grad | coeffs | status]; small calls (get_cost / get_grad of one
  // trajectory) through the pinned mirror: one copy each way, as in neo_optimize_batch
  size_t o_x, o_h, o_t, o_c, o_c4, o_g, o_co, o_st, o_end;
  {
    Carver lay(nullptr);
    auto at = [&](size_t bytes) { lay.take<char>(0); const size_t o = lay.off; lay.off += bytes; return o; };
    o_x = at(bs * n * sizeof(double));
    o_h = at(bs * 3 * D * sizeof(double));
    o_t = at(bs * 3 * D * sizeof(double));
    o_c = at(bs * sizeof(double));
    o_c4 = at(bs * 4 * sizeof(double));
    o_g = at(bs * n * sizeof(double));
    o_co = at(bs * 6 * M * D * sizeof(double));
    o_st = at(bs * sizeof(int));
    o_end = lay.off;
  }
  const bool staged = o_end <= (size_t)256 * 1024;
  double *dx, *dh, *dt, *dc, *dc4, *dg, *dco;
  int *dst;
  {
    std::lock_guard<std::recursive_mutex> g(c->mu);
    hipSetDevice(c->device);
    rc = ensure_scratch(c, o_end + 256);
    if (rc) return rc;
    if (staged) {
      rc = ensure_pinned(c, o_end + 256);
      if (rc) return rc;
    }
    char *dbase = static_cast<char *>(c->scratch), *hbase = static_cast<char *>(c->pinned);
    dx = reinterpret_cast<double *>(dbase + o_x);
    dh = reinterpret_cast<double *>(dbase + o_h);
    dt = reinterpret_cast<double *>(dbase + o_t);
    dc = reinterpret_cast<double *>(dbase + o_c);
    dc4 = reinterpret_cast<double *>(dbase + o_c4);
    dg = reinterpret_cast<double *>(dbase + o_g);
    dco = reinterpret_cast<double *>(dbase + o_co);
    dst = reinterpret_cast<int *>(dbase + o_st);
    if (staged) {
      std::memcpy(hbase + o_x, x, bs * n * sizeof(double));
      std::memcpy(hbase + o_h, head, bs * 3 * D * sizeof(double));
      std::memcpy(hbase + o_t, tail, bs * 3 * D * sizeof(double));
      HIPCHK(c, hipMemcpyAsync(dbase + o_x, hbase + o_x, o_c - o_x, hipMemcpyHostToDevice, c->stream));
    } else {
      HIPCHK(c, hipMemcpyAsync(dx, x, bs * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipMemcpyAsync(dh, head, bs * 3 * D * sizeof(double), hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipMemcpyAsync(dt, tail, bs * 3 * D * sizeof(double), hipMemcpyHostToDevice, c->stream));
    }
  }
  rc = neo_cost_grad_batch_dev(c, scene_id, B, M, D, dx, dh, dt, dc, dc4, dg, coeffs ? dco : nullptr, dst);
  if (rc) return rc;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  if (staged) {
    char *dbase = static_cast<char *>(c->scratch), *hbase = static_cast<char *>(c->pinned);
    HIPCHK(c, hipMemcpyAsync(hbase + o_c, dbase + o_c, o_end - o_c, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::memcpy(cost, hbase + o_c, bs * sizeof(double));
    std::memcpy(costs4, hbase + o_c4, bs * 4 * sizeof(double));
    std::memcpy(grad, hbase + o_g, bs * n * sizeof(double));
    if (coeffs) std::memcpy(coeffs, hbase + o_co, bs * 6 * M * D * sizeof(double));
    if (status) std::memcpy(status, hbase + o_st, bs * sizeof(int));
    return NEO_OK;
  }
  HIPCHK(c, hipMemcpyAsync(cost, dc, bs * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(costs4, dc4, bs * 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(grad, dg, bs * n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (coeffs) HIPCHK(c, hipMemcpyAsync(coeffs, dco, bs * 6 * M * D * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  if (status) HIPCHK(c, hipMemcpyAsync(status, dst, bs * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return NEO_OK;
}

// IO32: coeffs / grad_C / grad_T are floats (the _f32 entry points; fp32 sampling only)
static int sampled_terms_dev(neo_ctx *c, int scene_id, int B, int M, int D, const void *coeffs, const double *ts, double *costs2,
                             void *grad_C, void *grad_T, bool io32) {
  int rc = check_shape(c, B, M, D);
  if (rc) return rc;
  if (!coeffs || !ts || !costs2 || !grad_C || !grad_T) return fail_locked(c, NEO_ERR_INVALID, "null buffer");
  if (((uintptr_t)coeffs | (uintptr_t)grad_C) & (io32 ? 7 : 15))
    return fail_locked(c, NEO_ERR_INVALID, io32 ? "coeffs and grad_C must be 8-byte aligned" : "coeffs and grad_C must be 16-byte aligned");
  std::lock_guard<std::recursive_mutex> g(c->mu);
  hipSetDevice(c->device);
  auto it = c->maps.find(scene_id);
  if (it == c->maps.end()) return fail(c, NEO_ERR_NO_MAP, "no ESDF for this scene");
  if (B == 0) return NEO_OK;
  ProfScope ps(c, NEO_KERNEL_ESDF_SAMPLE);
  SampleArgs sa{B, M, coeffs, ts, costs2, grad_C, grad_T};
  sa.io32 = io32;
  rc = dispatch_sample(c, it->second, D, sa);
  if (rc) return rc;
  HIPCHK(c, hipGetLastError());
  return NEO_OK;
}

int neo_sampled_terms_batch_dev(neo_ctx *c, int scene_id, int B, int M, int D, const double *coeffs,
                                const double *ts, double *costs2, double *grad_C, double *grad_T) {
  return sampled_terms_dev(c, scene_id, B, M, D, coeffs, ts, costs2, grad_C, grad_T, false);
}

int neo_sampled_terms_batch_f32_dev(neo_ctx *c, int scene_id, int B, int M, int D, const float *coeffs,
                                    const double *ts, double *costs2, float *grad_C, float *grad_T) {
  return sampled_terms_dev(c, scene_id, B, M, D, coeffs, ts, costs2, grad_C, grad_T, true);
}

static int sampled_terms_host(neo_ctx *c, int scene_id, int B, int M, int D, const void *coeffs, const double *ts, double *costs2,
                              void *grad_C, void *grad_T, bool io32) {
  int rc = check_shape(c, B, M, D);
  if (rc) return rc;
  if (!coeffs || !ts || !costs2 || !grad_C || !grad_T) return fail_locked(c, NEO_ERR_INVALID, "null buffer");
  if (B == 0) return NEO_OK;
  std::lock_guard<std::recursive_mutex> whole_call(c->mu);  // scratch buffers stay ours until the copies back are done
  const size_t bs = (size_t)B, nc = (size_t)6 * M * D, es = io32 ? sizeof(float) : sizeof(double);
  void *dco, *dgc, *dgt;
  double *dts, *dc2;
  {
    std::lock_guard<std::recursive_mutex> g(c->mu);
    hipSetDevice(c->device);
    rc = ensure_scratch(c, bs * (2 * nc + 2 * M + 2) * sizeof(double) + 6 * 256);
    if (rc) return rc;
    Carver cv(c->scratch);
    dco = cv.take<double>(bs * nc);  // (sized for doubles either way)
    dts = cv.take<double>(bs * M);
    dc2 = cv.take<double>(bs * 2);
    dgc = cv.take<double>(bs * nc);
    dgt = cv.take<double>(bs * M);
    HIPCHK(c, hipMemcpyAsync(dco, coeffs, bs * nc * es, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dts, ts, bs * M * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  rc = sampled_terms_dev(c, scene_id, B, M, D, dco, dts, dc2, dgc, dgt, io32);
  if (rc) return rc;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  HIPCHK(c, hipMemcpyAsync(costs2, dc2, bs * 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(grad_C, dgc, bs * nc * es, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(grad_T, dgt, bs * M * es, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return NEO_OK;
}

int neo_sampled_terms_batch(neo_ctx *c, int scene_id, int B, int M, int D, const double *coeffs, const double *ts,
                            double *costs2, double *grad_C, double *grad_T) {
  return sampled_terms_host(c, scene_id, B, M, D, coeffs, ts, costs2, grad_C, grad_T, false);
}

int neo_sampled_terms_batch_f32(neo_ctx *c, int scene_id, int B, int M, int D, const float *coeffs, const double *ts,
                                double *costs2, float *grad_C, float *grad_T) {
  return sampled_terms_host(c, scene_id, B, M, D, coeffs, ts, costs2, grad_C, grad_T, true);
}

// the L-BFGS pairs (2 * maxcor * n doubles per trajectory) live in LDS: no HBM workspace
size_t neo_optimize_workspace_bytes(int, int, int) { return 0; }

int neo_optimize_batch_dev(neo_ctx *c, int scene_id, const int32_t *scene_ids, int B, int M, int D, double *x,
                           const double *head, const double *tail, double *costs4, double *costs4_last, int32_t *nit,
                           int32_t *nfev, int32_t *status) {
  return neo_optimize_batch_from_dev(c, scene_id, scene_ids, B, M, D, x, x, head, tail, costs4, costs4_last, nit, nfev,
                                     status);
}

int neo_optimize_batch_from_dev(neo_ctx *c, int scene_id, const int32_t *scene_ids, int B, int M, int D, const double *x0,
                                double *x, const double *head, const double *tail, double *costs4, double *costs4_last,
                                int32_t *nit, int32_t *nfev, int32_t *status) {
  int rc = check_shape(c, B, M, D);
  if (rc) return rc;
  if (!x0 || !x || !head || !tail || !costs4 || !nit || !nfev || !status)
    return fail_locked(c, NEO_ERR_INVALID, "null buffer");
  std::lock_guard<std::recursive_mutex> g(c->mu);
  hipSetDevice(c->device);
  if (B == 0) return NEO_OK;
  rc = rebuild_tables(c);
  if (rc) return rc;
  // scene_ids (device array) holds map-table SLOTS when given; a single scene_id is looked up here
  int kind, elem, layout = 0;
  const void *table;
  const int *slots = scene_ids;
  std::vector<int> one;
  int nmaps = 1;
  if (scene_ids) {
    // one kernel instantiation serves the whole call: every map a slot can name (all maps of the reference
    // scene's kind in this context) must share its element type and layout
    auto it = c->maps.find(scene_id);
    if (it == c->maps.end()) return fail(c, NEO_ERR_NO_MAP, "no ESDF for the reference scene");
    kind = it->second.kind;
    elem = it->second.elem;
    layout = it->second.m3.layout;
    for (const auto &kv : c->maps)
      if (kv.second.kind == kind && (kv.second.elem != elem || (kind == 1 && kv.second.m3.layout != layout)))
        return fail(c, NEO_ERR_INVALID, "multi-scene call: the context holds maps of this kind with different element "
                                        "types or layouts");
    table = kind == 0 ? c->table2d : c->table3d;
    nmaps = kind == 0 ? c->n2d : c->n3d;
  } else {
    auto it = c->maps.find(scene_id);
    if (it == c->maps.end()) return fail(c, NEO_ERR_NO_MAP, "no ESDF for this scene");
    kind = it->second.kind;
    elem = it->second.elem;
    layout = it->second.m3.layout;
    const char *base = static_cast<const char *>(kind == 0 ? c->table2d : c->table3d);
    table = base + (size_t)it->second.slot * (kind == 0 ? sizeof(Map2D) : sizeof(Map3D));
  }
  ProfScope ps(c, NEO_KERNEL_OPTIMIZE);
  const OptArgs oa{B, M, table, slots, nmaps, x0, x, head, tail, costs4, costs4_last, nit, nfev, status};
  rc = dispatch_opt(c, kind, elem, layout, D, oa);
  if (rc) return rc;
  HIPCHK(c, hipGetLastError());
  return NEO_OK;
}

size_t neo_optimize_state_bytes(int M, int D) {
  if (M < 1 || D < 1) return 0;
  return opt_state_doubles(D * (M - 1) + M, NEO_LBFGS_M) * sizeof(double);
}

int neo_optimize_batch_budget_dev(neo_ctx *c, int scene_id, int B, int M, int D, const double *x0, double *x, const double *head,
                                  const double *tail, double *costs4, double *costs4_last, int32_t *nit, int32_t *nfev,
                                  int32_t *status, void *state, int eval_budget, const int32_t *subset, int n_subset,
                                  int resume) {
  int rc = check_shape(c, B, M, D);
  if (rc) return rc;
  if (!x0 || !x || !head || !tail || !costs4 || !nit || !nfev || !status || !state)
    return fail_locked(c, NEO_ERR_INVALID, "null buffer");
  if (eval_budget < 1) return fail_locked(c, NEO_ERR_INVALID, "eval_budget < 1");
  if (subset && (n_subset < 0 || n_subset > B)) return fail_locked(c, NEO_ERR_INVALID, "bad subset size");
  std::lock_guard<std::recursive_mutex> g(c->mu);
  hipSetDevice(c->device);
  if (B == 0 || (subset && n_subset == 0)) return NEO_OK;
  rc = rebuild_tables(c);
  if (rc) return rc;
  auto it = c->maps.find(scene_id);
  if (it == c->maps.end()) return fail(c, NEO_ERR_NO_MAP, "no ESDF for this scene");
  if (it->second.kind == 0 || D != 3) return fail(c, NEO_ERR_INVALID, "budgeted launches: 3-D fields, D = 3");
  if (c->trace || c->trace_xg) return fail(c, NEO_ERR_INVALID, "budgeted launches do not record traces");
  const char *base = static_cast<const char *>(c->table3d);
  const void *table = base + (size_t)it->second.slot * sizeof(Map3D);
  ProfScope ps(c, NEO_KERNEL_OPTIMIZE);
  OptArgs oa{B, M, table, nullptr, 1, x0, x, head, tail, costs4, costs4_last, nit, nfev, status};
  oa.state = static_cast<double *>(state);
  oa.state_doubles = (int)opt_state_doubles(D * (M - 1) + M, NEO_LBFGS_M);
  oa.budget = eval_budget;
  oa.resume = resume ? 1 : 0;
  oa.subset = subset;
  oa.n_subset = subset ? n_subset : 0;
  oa.traj_total = B;
  rc = launch_opt_3d_budget(c, it->second.elem, it->second.m3.layout, oa);
  if (rc) return rc;
  HIPCHK(c, hipGetLastError());
  return NEO_OK;
}

int neo_scene_slot(neo_ctx *c, int scene_id) {
  if (!c) return -1;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  hipSetDevice(c->device);
  if (rebuild_tables(c)) return -1;
  auto it = c->maps.find(scene_id);
  return it == c->maps.end() ? -1 : it->second.slot;
}

int neo_optimize_batch(neo_ctx *c, int scene_id, const int32_t *scene_ids, int B, int M, int D, double *x,
                       const double *head, const double *tail, double *costs4, double *costs4_last, int32_t *nit,
                       int32_t *nfev, int32_t *status) {
  int rc = check_shape(c, B, M, D);
  if (rc) return rc;
  if (!x || !head || !tail || !costs4 || !nit || !nfev || !status) return fail_locked(c, NEO_ERR_INVALID, "null buffer");
  if (B == 0) return NEO_OK;
  std::lock_guard<std::recursive_mutex> whole_call(c->mu);  // scratch buffers stay ours until the copies back are done
  const size_t n = (size_t)D * (M - 1) + M, bs = (size_t)B;
  double *dx, *dh, *dt, *dc4, *dc4l;
  int *dnit, *dnfev, *dst, *dslots = nullptr;
  std::vector<int> slots;
  if (scene_ids) {
    slots.resize(bs);
    int kind0 = -1;
    for (size_t i = 0; i < bs; ++i) {
      const int s = neo_scene_slot(c, scene_ids[i]);
      if (s < 0) return fail(c, NEO_ERR_NO_MAP, "no ESDF for one of scene_ids");
      const int k = c->maps.find(scene_ids[i])->second.kind;
      if (kind0 < 0) kind0 = k;
      if (k != kind0) return fail(c, NEO_ERR_INVALID, "scene_ids mix 2-D and 3-D maps");
      slots[i] = s;
    }
  }
  // Scratch layout [head | tail | slots | x | costs4 | costs4_last | nit | nfev | status]: the inputs are one contiguous
  // range ending with x, the outputs one starting with it.  The host side of both copies is a pinned mirror of the
  // same layout -- one hipMemcpyAsync each way (a single plan() of the reference's shape: 0.36 -> 0.27 ms; ten copies
  // from and to pageable memory cost almost as much as the kernel).
  size_t o_h, o_t, o_s, o_x, o_c4, o_c4l, o_nit, o_nfev, o_st, o_end;
  {
    Carver lay(nullptr);
    auto at = [&](size_t bytes) { lay.take<char>(0); const size_t o = lay.off; lay.off += bytes; return o; };
    o_h = at(bs * 3 * D * sizeof(double));
    o_t = at(bs * 3 * D * sizeof(double));
    o_s = at(bs * sizeof(int));
    o_x = at(bs * n * sizeof(double));
    o_c4 = at(bs * 4 * sizeof(double));
    o_c4l = at(bs * 4 * sizeof(double));
    o_nit = at(bs * sizeof(int));
    o_nfev = at(bs * sizeof(int));
    o_st = at(bs * sizeof(int));
    o_end = lay.off;
  }
  // (small calls only: from a megabyte on, the extra pass over the data costs more than the copies' latencies --
  //  8192 replans of the reference's shape per call: 1.46 M/s direct, 1.28 M/s staged)
  const bool staged = o_end <= (size_t)256 * 1024;
  {
    std::lock_guard<std::recursive_mutex> g(c->mu);
    hipSetDevice(c->device);
    rc = ensure_scratch(c, o_end + 256);
    if (rc) return rc;
    if (staged) {
      rc = ensure_pinned(c, o_end + 256);
      if (rc) return rc;
    }
    char *dbase = static_cast<char *>(c->scratch), *hbase = static_cast<char *>(c->pinned);
    dh = reinterpret_cast<double *>(dbase + o_h);
    dt = reinterpret_cast<double *>(dbase + o_t);
    dslots = reinterpret_cast<int *>(dbase + o_s);
    dx = reinterpret_cast<double *>(dbase + o_x);
    dc4 = reinterpret_cast<double *>(dbase + o_c4);
    dc4l = reinterpret_cast<double *>(dbase + o_c4l);
    dnit = reinterpret_cast<int *>(dbase + o_nit);
    dnfev = reinterpret_cast<int *>(dbase + o_nfev);
    dst = reinterpret_cast<int *>(dbase + o_st);
    if (staged) {
      std::memcpy(hbase + o_h, head, bs * 3 * D * sizeof(double));
      std::memcpy(hbase + o_t, tail, bs * 3 * D * sizeof(double));
      if (scene_ids) std::memcpy(hbase + o_s, slots.data(), bs * sizeof(int));
      std::memcpy(hbase + o_x, x, bs * n * sizeof(double));
      HIPCHK(c, hipMemcpyAsync(dbase + o_h, hbase + o_h, o_c4 - o_h, hipMemcpyHostToDevice, c->stream));
    } else {
      HIPCHK(c, hipMemcpyAsync(dx, x, bs * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipMemcpyAsync(dh, head, bs * 3 * D * sizeof(double), hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipMemcpyAsync(dt, tail, bs * 3 * D * sizeof(double), hipMemcpyHostToDevice, c->stream));
      if (scene_ids)
        HIPCHK(c, hipMemcpyAsync(dslots, slots.data(), bs * sizeof(int), hipMemcpyHostToDevice, c->stream));
    }
  }
  rc = neo_optimize_batch_dev(c, scene_ids ? scene_ids[0] : scene_id, scene_ids ? dslots : nullptr, B, M, D, dx, dh,
                              dt, dc4, dc4l, dnit, dnfev, dst);
  if (rc) return rc;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  if (!staged) {
    HIPCHK(c, hipMemcpyAsync(x, dx, bs * n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(costs4, dc4, bs * 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (costs4_last) HIPCHK(c, hipMemcpyAsync(costs4_last, dc4l, bs * 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(nit, dnit, bs * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(nfev, dnfev, bs * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(status, dst, bs * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return NEO_OK;
  }
  char *dbase = static_cast<char *>(c->scratch), *hbase = static_cast<char *>(c->pinned);
  HIPCHK(c, hipMemcpyAsync(hbase + o_x, dbase + o_x, o_end - o_x, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memcpy(x, hbase + o_x, bs * n * sizeof(double));
  std::memcpy(costs4, hbase + o_c4, bs * 4 * sizeof(double));
  if (costs4_last) std::memcpy(costs4_last, hbase + o_c4l, bs * 4 * sizeof(double));
  std::memcpy(nit, hbase + o_nit, bs * sizeof(int));
  std::memcpy(nfev, hbase + o_nfev, bs * sizeof(int));
  std::memcpy(status, hbase + o_st, bs * sizeof(int));
  return NEO_OK;
}

int neo_eval_traj_batch(neo_ctx *c, int B, int M, int D, const double *x, const double *head, const double *tail,
                        double hz, int K, double *state, int32_t *count) {
  int rc = check_shape(c, B, M, D);
  if (rc) return rc;
  if (!x || !head || !tail || !state || !count || K < 0 || !(hz > 0.0)) return fail_locked(c, NEO_ERR_INVALID, "bad argument");
  if (B == 0 || K == 0) return NEO_OK;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  hipSetDevice(c->device);
  const size_t n = (size_t)D * (M - 1) + M, bs = (size_t)B;
  // scratch layout [x | head | tail | state | count]; small calls through the pinned mirror (one copy each way, as in
  // neo_optimize_batch)
  size_t o_x, o_h, o_t, o_s, o_c, o_end;
  {
    Carver lay(nullptr);
    auto at = [&](size_t bytes) { lay.take<char>(0); const size_t o = lay.off; lay.off += bytes; return o; };
    o_x = at(bs * n * sizeof(double));
    o_h = at(bs * 3 * D * sizeof(double));
    o_t = at(bs * 3 * D * sizeof(double));
    o_s = at(bs * K * 3 * D * sizeof(double));
    o_c = at(bs * sizeof(int));
    o_end = lay.off;
  }
  const bool staged = o_end <= (size_t)256 * 1024;
  rc = ensure_scratch(c, o_end + 256);
  if (rc) return rc;
  if (staged) {
    rc = ensure_pinned(c, o_end + 256);
    if (rc) return rc;
  }
  char *dbase = static_cast<char *>(c->scratch), *hbase = static_cast<char *>(c->pinned);
  double *dx = reinterpret_cast<double *>(dbase + o_x), *dh = reinterpret_cast<double *>(dbase + o_h);
  double *dt = reinterpret_cast<double *>(dbase + o_t), *ds = reinterpret_cast<double *>(dbase + o_s);
  int *dcnt = reinterpret_cast<int *>(dbase + o_c);
  if (staged) {
    std::memcpy(hbase + o_x, x, bs * n * sizeof(double));
    std::memcpy(hbase + o_h, head, bs * 3 * D * sizeof(double));
    std::memcpy(hbase + o_t, tail, bs * 3 * D * sizeof(double));
    HIPCHK(c, hipMemcpyAsync(dbase + o_x, hbase + o_x, o_s - o_x, hipMemcpyHostToDevice, c->stream));
  } else {
    HIPCHK(c, hipMemcpyAsync(dx, x, bs * n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dh, head, bs * 3 * D * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(dt, tail, bs * 3 * D * sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  if (D == 2)
    hipLaunchKernelGGL((traj_state_kernel<2>), dim3(B), dim3(kWave), 0, c->stream, B, M, c->dev, dx, dh, dt, hz, K, ds,
                       dcnt);
  else
    hipLaunchKernelGGL((traj_state_kernel<3>), dim3(B), dim3(kWave), 0, c->stream, B, M, c->dev, dx, dh, dt, hz, K, ds,
                       dcnt);
  HIPCHK(c, hipGetLastError());
  if (staged) {
    HIPCHK(c, hipMemcpyAsync(hbase + o_s, dbase + o_s, o_end - o_s, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::memcpy(state, hbase + o_s, bs * K * 3 * D * sizeof(double));
    std::memcpy(count, hbase + o_c, bs * sizeof(int));
    return NEO_OK;
  }
  HIPCHK(c, hipMemcpyAsync(state, ds, bs * K * 3 * D * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(count, dcnt, bs * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return NEO_OK;
}

int neo_optimize_progress_counter(neo_ctx *c, int32_t *counter) {
  if (!c) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  c->progress = counter;
  return NEO_OK;
}

int neo_optimize_sample_counter(neo_ctx *c, int64_t *dev_counts) {
  if (!c) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  c->sample_counter = reinterpret_cast<long long *>(dev_counts);
  return NEO_OK;
}

int neo_optimize_trace(neo_ctx *c, double *dev_trace, int cap) {
  if (!c || cap < 0) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  if (dev_trace && cap > 0 && c->trace_xg && c->trace_cap != cap)
    return fail(c, NEO_ERR_INVALID, "neo_optimize_trace: cap differs from neo_optimize_trace_xg's");
  c->trace = (dev_trace && cap > 0) ? dev_trace : nullptr;
  if (c->trace) c->trace_cap = cap;
  if (!c->trace_xg && !c->trace) c->trace_cap = 0;
  return NEO_OK;
}

int neo_optimize_trace_xg(neo_ctx *c, double *dev_xg, int cap) {
  if (!c || cap < 0) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  if (dev_xg && cap > 0 && c->trace && c->trace_cap != cap)
    return fail(c, NEO_ERR_INVALID, "neo_optimize_trace_xg: cap differs from neo_optimize_trace's");
  c->trace_xg = (dev_xg && cap > 0) ? dev_xg : nullptr;
  if (c->trace_xg) c->trace_cap = cap;
  if (!c->trace_xg && !c->trace) c->trace_cap = 0;
  return NEO_OK;
}

int neo_optimize_dispatch_order(neo_ctx *c, const int32_t *dev_order, int B) {
  if (!c || B < 0) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  c->dispatch_order = (dev_order && B > 0) ? dev_order : nullptr;
  c->order_B = c->dispatch_order ? B : 0;
  return NEO_OK;
}

int neo_optimize_dispatch_order_host(neo_ctx *c, const int32_t *host_order, int B) {
  if (!c || B < 0) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  hipSetDevice(c->device);
  if (!host_order || B == 0) {
    c->dispatch_order = nullptr;
    c->order_B = 0;
    return NEO_OK;
  }
  if ((size_t)B > c->order_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->order_buf) hipFree(c->order_buf);
    c->order_buf = nullptr;
    c->order_cap = 0;
    HIPCHK(c, hipMalloc((void **)&c->order_buf, (size_t)B * sizeof(int)));
    c->order_cap = (size_t)B;
  }
  HIPCHK(c, hipMemcpyAsync(c->order_buf, host_order, (size_t)B * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));  // the host array may go away after the call
  c->dispatch_order = c->order_buf;
  c->order_B = B;
  return NEO_OK;
}

namespace neo {
// rows [x (n) | total cost | 4 cost terms] in fp32 for the result gather (neo_planner_amd/sharding.py)
__global__ void pack_results_kernel(int B, int n, const double *__restrict__ x, const double *__restrict__ costs4, double w0, double w1,
                                    double w2, double w3, float *__restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t row = i / (size_t)(n + 5);
  if (row >= (size_t)B) return;
  const int col = (int)(i - row * (size_t)(n + 5));
  const double *c4 = costs4 + row * 4;
  double v;
  if (col < n)
    v = x[row * (size_t)n + col];
  else if (col == n)
    v = ((c4[0] * w0 + c4[1] * w1) + c4[2] * w2) + c4[3] * w3;  // (costs * weights).sum(dim=1), left to right
  else
    v = c4[col - n - 1];
  out[i] = (float)v;
}
}  // namespace neo

int neo_pack_results_dev(neo_ctx *c, int B, int n, const double *x, const double *costs4, const double *weights4, float *out) {
  if (!c || B < 0 || n < 1) return NEO_ERR_INVALID;
  if (!x || !costs4 || !weights4 || !out) return fail_locked(c, NEO_ERR_INVALID, "null buffer");
  if (B == 0) return NEO_OK;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  hipSetDevice(c->device);
  const size_t total = (size_t)B * (size_t)(n + 5);
  hipLaunchKernelGGL(neo::pack_results_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, B, n, x, costs4,
                     weights4[0], weights4[1], weights4[2], weights4[3], out);
  HIPCHK(c, hipGetLastError());
  return NEO_OK;
}

namespace neo {
// ---- the expected-effort dispatch order on the device (round 6; host form: BatchPlanner.expected_effort_order)
// key = time slack of the initial guess, sum(T) v_max / |goal - start| (a guess that is far too slow sheds duration over many
// iterations: rank correlation 0.5 with the evaluation count at cfg2); NaN -> 0 so that the keys are totally ordered
__global__ void effort_keys_kernel(int B, int M, int D, double T_min, double T_max, double v_max, const double *__restrict__ x0,
                                   const double *__restrict__ head, const double *__restrict__ tail, double *__restrict__ keys,
                                   int *__restrict__ idx) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  idx[b] = b;
  const int nq = D * (M - 1), n = nq + M;
  double sum = 0.0;
  for (int p = 0; p < M; ++p) sum += (T_max - T_min) / (1.0 + exp(-x0[(size_t)b * n + nq + p])) + T_min;  // map_tau2T (:477-483)
  double d2 = 0.0;
  for (int d = 0; d < D; ++d) {
    const double e = tail[(size_t)b * 3 * D + d] - head[(size_t)b * 3 * D + d];
    d2 += e * e;
  }
  const double k = sum * v_max / fmax(sqrt(d2), 1e-9);
  keys[b] = (k == k) ? k : 0.0;
}
}  // namespace neo

// The order itself: a stable descending sort of (key, index) -- rocPRIM's radix sort on the doubles (LSD radix: equal keys
// keep index order).  Round 6 first ranked by B^2 comparisons (one thread per element: 45 us at 4096 on 16 CUs; tiled over a
// 2-D grid with integer partial counts: three launches, fine at 4096, 0.6 ms at cfg3's 65 536).
static size_t effort_sort_temp_bytes(int B) {
  size_t tb = 0;
  (void)rocprim::radix_sort_pairs_desc(nullptr, tb, (const double *)nullptr, (double *)nullptr, (const int *)nullptr, (int *)nullptr,
                                       (unsigned)B, 0, 64, (hipStream_t)0);
  return (tb + 255) / 256 * 256;
}

static size_t align256(size_t v) { return (v + 255) / 256 * 256; }

size_t neo_effort_order_scratch_bytes(int B) {
  if (B <= 0) return 0;
  // keys in, keys out (doubles), indices in (ints), the sort's temporary storage: every part 256-byte aligned
  return align256((size_t)B * 16) + align256((size_t)B * 4) + effort_sort_temp_bytes(B);
}

int neo_effort_order_dev(neo_ctx *c, int B, int M, int D, const double *x0, const double *head, const double *tail, void *scratch,
                         int32_t *order) {
  int rc = check_shape(c, B, M, D);
  if (rc) return rc;
  if (!x0 || !head || !tail || !scratch || !order) return fail_locked(c, NEO_ERR_INVALID, "null buffer");
  if (((uintptr_t)scratch) & 255) return fail_locked(c, NEO_ERR_INVALID, "scratch must be 256-byte aligned");
  if (B == 0) return NEO_OK;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  hipSetDevice(c->device);
  double *keys = static_cast<double *>(scratch), *keys_out = keys + B;
  int *idx = reinterpret_cast<int *>(reinterpret_cast<char *>(scratch) + align256((size_t)B * 16));
  void *temp = reinterpret_cast<char *>(idx) + align256((size_t)B * 4);
  size_t tb = effort_sort_temp_bytes(B);
  hipLaunchKernelGGL(neo::effort_keys_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, c->stream, B, M, D, c->params.T_min,
                     c->params.T_max, c->params.v_max, x0, head, tail, keys, idx);
  HIPCHK(c, rocprim::radix_sort_pairs_desc(temp, tb, (const double *)keys, keys_out, (const int *)idx, (int *)order, (unsigned)B, 0, 64,
                                           c->stream));
  HIPCHK(c, hipGetLastError());
  return NEO_OK;
}

int neo_sampled_terms_dispatch_order(neo_ctx *c, const int32_t *order, int on_device, int B) {
  if (!c || B < 0) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  hipSetDevice(c->device);
  if (!order || B == 0) {
    c->sample_order_B = 0;
    return NEO_OK;
  }
  if ((size_t)B > c->sample_order_cap) {
    HIPCHK(c, hipStreamSynchronize(c->stream));  // (launches that read the old copy have finished)
    if (c->sample_order) hipFree(c->sample_order);
    c->sample_order = nullptr;
    c->sample_order_cap = 0;
    c->sample_order_B = 0;
    HIPCHK(c, hipMalloc((void **)&c->sample_order, (size_t)B * sizeof(int)));
    c->sample_order_cap = (size_t)B;
  }
  // copied into the context's own buffer before the call returns: the caller's array may go away right after it
  HIPCHK(c, hipMemcpyAsync(c->sample_order, order, (size_t)B * sizeof(int),
                           on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->sample_order_B = B;
  return NEO_OK;
}

int neo_esdf_build_config(neo_ctx *c, int flags) {
  if (!c || (flags & ~NEO_EDT_GENERIC_LINES)) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  c->edt_flags = flags;
  return NEO_OK;
}

int neo_profile_enable(neo_ctx *c, int on) {
  if (!c) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  c->profile = on != 0;
  return NEO_OK;
}

int neo_profile_read(neo_ctx *c, int kernel, int64_t *launches, double *total_ms) {
  if (!c || kernel < 0 || kernel >= NEO_KERNEL_COUNT) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  hipSetDevice(c->device);
  drain_profile(c);
  if (launches) *launches = c->prof[kernel].launches;
  if (total_ms) *total_ms = c->prof[kernel].ms;
  return NEO_OK;
}

int neo_profile_reset(neo_ctx *c) {
  if (!c) return NEO_ERR_INVALID;
  std::lock_guard<std::recursive_mutex> g(c->mu);
  hipSetDevice(c->device);
  drain_profile(c);
  for (int k = 0; k < NEO_KERNEL_COUNT; ++k) {
    c->prof[k].launches = 0;
    c->prof[k].ms = 0.0;
  }
  return NEO_OK;
}

}  // extern "C"
