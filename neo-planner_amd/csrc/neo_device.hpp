// neo_device.hpp -- device-side MINCO cost/gradient for one trajectory per lane group (gfx950): the whole
// wavefront (WaveLanes) or, for small problems, 16 or 8 lanes (GroupLanes<W>).  ONE source for both.
//
// One 64-lane wavefront owns one trajectory.  Two lane layouts are used:
//   PIECE  layout: lane p  <-> polynomial piece p (p < M <= 64) and joint p (its start);
//   SAMPLE layout: lane l  <-> (piece l / L, residue l % L): the piece's quadrature samples
//                  j = r, r+L, r+2L, ... are walked by its L lanes, gradients accumulate in
//                  registers and are folded over the L lanes once at the end;
//   FLAT   layout: element e of x / grad <-> (lane e & 63, slot e >> 6)  (optimiser vectors).
// No LDS atomics, no global atomics: every sum has a fixed order, results are bit-reproducible.
//
// What is computed (reference: src/planner/scripts/traj_planner/expert_planner.py):
//   forward  = map_tau2T (:477-483) + get_coeffs (:261-336) + add_energy_cost/add_time_cost (:345-390)
//   sample   = add_sampled_cost + add_sampled_grad_CT (:392-466) with the map lookups of
//              map_server/esdf.py:53-82 (or the trilinear 3-D mode)
//   backward = add_energy_grad_CT/add_time_grad_CT (:361-390) + propagate_grad_q_tau (:494-537)
//              + get_grad_T2tau (:485-492)
//
// get_coeffs solves a dense 6M x 6M system; here the same coefficients come from the
// Hermite form of each quintic (fixed by position/velocity/acceleration at its two ends) plus a
// block-tridiagonal system with 2x2 blocks in the unknown (v_j, a_j) of the interior joints
// (jerk and snap continuity, rows 6i+7 and 6i+8 of the reference's A); the adjoint solve of
// propagate_grad_q_tau becomes the transposed 2x2 block system.  tools/proto_reduced.py checks
// the algebra against the reference formulation to 1e-13, including the stale-T quirk.
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

namespace neo {

constexpr int kWave = 64;
constexpr int kSlots = 4;  // FLAT layout: n <= 256

struct DevParams {
  double v_max, T_min, T_max, safe_dis, delta_t;
  double w[4];
  double coll_tol;
  double ftol, gtol;
  int maxls, maxiter, maxfun, stale_T;
  int dbg;  // timing experiments only (neo_params.flags): 1 skip sample loop, 2 skip joint sweeps, 4 no history -- read by
            // the kernels of -DNEO_EXPERIMENTS builds alone (NEO_DBG below); the product's kernels have no such paths
  // fp32 forms of what the fp32 arithmetic uses, derived on the host (derive()): kernel arguments arrive in scalar
  // registers, whereas a (float)prm.x inside a kernel is a vector conversion whose (wave-uniform) result the compiler
  // hoists and then keeps in a vector register across the whole optimiser loop -- nine of them, spilled to scratch in the
  // three-waves-per-SIMD kernels
  struct F32 {
    float T_min, T_span;  // (float)T_max - (float)T_min, rounded as the device expression was
    float dt, inv_dt;     // (float)delta_t, (float)(1 / delta_t)
    float vmax2, safe;
    float w[4];
  } f;
  __host__ __device__ void derive() {
    f.T_min = (float)T_min;
    f.T_span = (float)T_max - (float)T_min;
    f.dt = (float)delta_t;
    f.inv_dt = (float)(1.0 / delta_t);
    f.vmax2 = (float)(v_max * v_max);
    f.safe = (float)safe_dis;
    for (int k = 0; k < 4; ++k) f.w[k] = (float)w[k];
  }
};
// phase switches of the timing experiments (tools/gpu_phase_bench.py, tools/probe/phase_counts.py): compiled in only with
// -DNEO_EXPERIMENTS; in the product the tests are the constant 0 and the switched-off paths do not exist
#ifdef NEO_EXPERIMENTS
#define NEO_DBG(prm, bits) (((prm).dbg & (bits)) != 0)
#else
#define NEO_DBG(prm, bits) false
#endif
// a parameter in the arithmetic N: the double itself, or its fp32 form from DevParams::f
#define NEO_PARAM(name, dexpr, fexpr)                                                  \
  template <typename N>                                                                \
  __device__ __forceinline__ N name(const DevParams &p) {                              \
    if constexpr (sizeof(N) == 4) return (fexpr); else return (N)(dexpr);              \
  }
NEO_PARAM(par_T_min, p.T_min, p.f.T_min)
NEO_PARAM(par_T_span, N(p.T_max) - N(p.T_min), p.f.T_span)
NEO_PARAM(par_dt, p.delta_t, p.f.dt)
NEO_PARAM(par_inv_dt, 1.0 / p.delta_t, p.f.inv_dt)
NEO_PARAM(par_vmax2, p.v_max * p.v_max, p.f.vmax2)
NEO_PARAM(par_safe, p.safe_dis, p.f.safe)
NEO_PARAM(par_w0, p.w[0], p.f.w[0])
NEO_PARAM(par_w1, p.w[1], p.f.w[1])
NEO_PARAM(par_w2, p.w[2], p.f.w[2])
NEO_PARAM(par_w3, p.w[3], p.f.w[3])
#undef NEO_PARAM

// 2-D reference map: one 32-byte record per cell {dist, grad_x, grad_y, 0}
struct Map2D {
  const double4 *rec;
  int W, H;
  double res, ox, oy;
};
// 3-D distance field, element type E
struct Map3D {
  const void *data;
  int nx, ny, nz;
  int layout;  // 0 linear [z][y][x], 1 = yz-quads (the 2x2 (y,z) neighbourhood of every voxel contiguous, x-major),
               // 2 = cell-packed (8 corners of every cell contiguous), 3 = corner bricks (one 128-byte line per block of
               // 2 x 2 x 2 cells (fp32: its 27 corners) or 4 x 2 x 2 cells (fp16: its 45 corners))
  double res, ox, oy, oz;
  unsigned int bytes;  // size of the stored field (buffer-descriptor range)
  // fp32 constants of the lookup (Lookup3D), derived on the host: cell coordinate minus one half = pos * f_inv + f_off,
  // inside <=> -0.5 <= that < f_hi.  Computed in the kernel they are fp64 divisions whose results the compiler hoists
  // out of the optimiser loop and keeps in vector registers (spilled at three wavefronts per SIMD); as part of the
  // map record they arrive in scalar registers.
  float f_inv, f_off[3], f_hi[3];
  int nbx, nby;  // corner-brick layout: blocks along x and y (set by the upload for layout 3)
  double d_inv;  // 1 / res for the fp64 lookups (round 5: three fp64 divisions a sample -- ~10 instructions each with
                 // v_rcp_f64 and its refinement -- became multiplications; the trilinear mode is defined by the oracle,
                 // which divides: cell coordinates differ by an ulp at most, distances by ~1e-13 relative)
  __host__ __device__ void derive() {
    d_inv = 1.0 / res;
    f_inv = (float)(1.0 / res);
    f_off[0] = (float)(-ox / res - 0.5);
    f_off[1] = (float)(-oy / res - 0.5);
    f_off[2] = (float)(-oz / res - 0.5);
    f_hi[0] = (float)nx - 0.5f;
    f_hi[1] = (float)ny - 0.5f;
    f_hi[2] = (float)nz - 0.5f;
  }
};

// ------------------------------------------------------------------ wave helpers
__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }

// Ordering of LDS traffic inside ONE wavefront (every workgroup of these kernels is a single wavefront): the LDS
// executes a wavefront's DS instructions in issue order, so a read issued after a write sees it -- whichever lanes
// wrote and read.  All that is needed is that the compiler keeps the order: a wavefront-scope fence and a scheduling
// barrier, no s_barrier and no wait for every outstanding LDS operation as __syncthreads() would add.
__device__ __forceinline__ void lds_wave_sync() {
#ifdef NEO_STRONG_SYNC  // (diagnostic builds: a workgroup barrier with its full waits)
  __syncthreads();
#else
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
#endif
}

__device__ __forceinline__ double rdlane(double v, int src /*wave-uniform*/) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float rdlane(float v, int src /*wave-uniform*/) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
}
// reciprocal to working precision: v_rcp_f64 (4.6e-8 raw on gfx950) with two Newton steps, v_rcp_f32 (1 ulp) as it is
__device__ __forceinline__ double precise_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ float precise_rcp(float d) { return __builtin_amdgcn_rcpf(d); }
__device__ __forceinline__ double uniform(double v) {
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
// position markers in the assembly listing (tools/probe/mark_counts.py prices the phases between them); no code
#ifdef NEO_MARKS
#define NEO_MARK(name) asm volatile("; NEOMARK " name)
#else
#define NEO_MARK(name)
#endif
// the value, hidden from loop-invariant code motion and common-subexpression elimination: what is computed from it is
// computed where it is written (no instruction; used where a hoisted address costs a register across a whole loop)
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
// the same for a wave-uniform value (it stays in a scalar register)
__device__ __forceinline__ int opaque_uniform(int v) {
  asm volatile("" : "+s"(v));
  return v;
}
// lane predicates compared WHERE THEY ARE USED: the bound goes through an opaque scalar copy, so the compare cannot be
// hoisted out of the optimiser loop -- where it would be a scalar register pair that lives across the whole loop, is
// spilled to a lane of a vector register and costs two v_readlane at every use instead of one v_cmp (DevBackend::eval)
__device__ __forceinline__ bool lane_lt(int bound) { return (int)__lane_id() < opaque_uniform(bound); }
__device__ __forceinline__ bool lane_ge(int bound) { return (int)__lane_id() >= opaque_uniform(bound); }
__device__ __forceinline__ bool lane_eq(int which) { return (int)__lane_id() == opaque_uniform(which); }
__device__ __forceinline__ float uniform(float v) {
  return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v)));
}
// ---- DPP cross-lane moves (no LDS round trip).  ctrl: 0x110+n = row_shr:n (lane i <- lane i-n inside
// its row of 16), 0x142 / 0x143 = row_bcast:15 / row_bcast:31, 0x130 / 0x138 = wave_shl:1 / wave_shr:1.
// Lanes without a valid source (or masked off by row_mask) receive 0.  With every row enabled that is the
// instruction's own bound_ctrl zero fill: no register has to be preset to 0 ahead of each move (two v_mov_b32 per
// fp64 step, 8 of the 34 instructions of a wave_sum); with a row mask the masked rows keep `old`, which must be the 0.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(dpp_i<CTRL, ROW_MASK>(__float_as_int(v)));
}
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_d(double v) {
  const int lo = dpp_i<CTRL, ROW_MASK>(__double2loint(v));
  const int hi = dpp_i<CTRL, ROW_MASK>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
// The two row broadcasts of a reduction that is read at lane 63 only: rows masked off by row_mask are left UNDEFINED
// (no preset register, v_mov_dpp with an undefined `old`).  Lane 63 depends only on written rows: row_bcast:15 (rows 1,
// 3) gives lane 31 = S1 + S0 and lane 63 = S3 + S2, row_bcast:31 (rows 2, 3) adds lane 31 to lane 63.  Not for scans.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_any_i(int v) {
  return __builtin_amdgcn_mov_dpp(v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_any_d(double v) {
  const int lo = dpp_any_i<CTRL, ROW_MASK>(__double2loint(v));
  const int hi = dpp_any_i<CTRL, ROW_MASK>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
// wave-wide sum / max with a fixed association order; the result is returned wave-uniform
// (taken from lane 63 through v_readlane).  Inclusive scan inside rows, then the two row broadcasts.
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_d<0x111>(v);
  v += dpp_d<0x112>(v);
  v += dpp_d<0x114>(v);
  v += dpp_d<0x118>(v);
  v += dpp_any_d<0x142, 0xa>(v);
  v += dpp_any_d<0x143, 0xc>(v);
  return rdlane(v, 63);
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0x111>(v);
  v += dpp_f<0x112>(v);
  v += dpp_f<0x114>(v);
  v += dpp_f<0x118>(v);
  v += __int_as_float(dpp_any_i<0x142, 0xa>(__float_as_int(v)));
  v += __int_as_float(dpp_any_i<0x143, 0xc>(__float_as_int(v)));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_sum(int v) {
  v += dpp_i<0x111>(v);
  v += dpp_i<0x112>(v);
  v += dpp_i<0x114>(v);
  v += dpp_i<0x118>(v);
  v += dpp_any_i<0x142, 0xa>(v);
  v += dpp_any_i<0x143, 0xc>(v);
  return __builtin_amdgcn_readlane(v, 63);
}
// Four wave-wide sums for little more than the price of one: the four per-lane values are first folded onto one
// register -- v_permlane32_swap / v_permlane16_swap (gfx950) exchange half-waves and odd/even rows of two registers, so
// two adds leave the 64 partials of value k on the 16 lanes of row k -- then ONE row-wise DPP scan finishes all four
// (lane 15 of row k holds the total of value k).  One dependent chain of 7 additions instead of four of 6, 37
// instructions instead of 80; fixed association order.
__device__ __forceinline__ void swap_half_waves(double &x, double &y) {  // lanes 32..63 of x <-> lanes 0..31 of y
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  x = __hiloint2double((int)hi[0], (int)lo[0]);
  y = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void swap_odd_even_rows(double &x, double &y) {  // odd rows of x <-> even rows of y
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
  x = __hiloint2double((int)hi[0], (int)lo[0]);
  y = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void wave_sum4(double a, double b, double c, double d, double &ta, double &tb, double &tc,
                                          double &td) {
  swap_half_waves(a, c);
  double x = a + c;  // lanes 0..31: a folded to 32 values, lanes 32..63: c
  swap_half_waves(b, d);
  double y = b + d;
  swap_odd_even_rows(x, y);
  double z = x + y;  // row 0: a, row 1: b, row 2: c, row 3: d (16 partials each)
  z += dpp_d<0x111>(z);
  z += dpp_d<0x112>(z);
  z += dpp_d<0x114>(z);
  z += dpp_d<0x118>(z);
  ta = rdlane(z, 15);
  tb = rdlane(z, 31);
  tc = rdlane(z, 47);
  td = rdlane(z, 63);
}

// the same for four fp32 values: 2 + 1 register swaps, 3 adds, one row-wise DPP scan, 4 v_readlane -- the price of about one
// and a half wave_sum(float) for four sums on ONE dependent chain (the paired two-loop recursion batches its dots).
// Fixed association order: half-waves first, then odd / even rows, then the 16 lanes of a row left to right.
__device__ __forceinline__ void wave_sum4(float a, float b, float c, float d, float &ta, float &tb, float &tc, float &td) {
  auto swap32 = [](float &x, float &y) {  // lanes 32..63 of x <-> lanes 0..31 of y
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r[0]);
    y = __uint_as_float(r[1]);
  };
  auto swap16 = [](float &x, float &y) {  // odd rows of x <-> even rows of y
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    x = __uint_as_float(r[0]);
    y = __uint_as_float(r[1]);
  };
  swap32(a, c);
  float x = a + c;  // lanes 0..31: a folded to 32 values, lanes 32..63: c
  swap32(b, d);
  float y = b + d;
  swap16(x, y);
  float z = x + y;  // row 0: a, row 1: b, row 2: c, row 3: d (16 partials each)
  z += dpp_f<0x111>(z);
  z += dpp_f<0x112>(z);
  z += dpp_f<0x114>(z);
  z += dpp_f<0x118>(z);
  ta = rdlane(z, 15);
  tb = rdlane(z, 31);
  tc = rdlane(z, 47);
  td = rdlane(z, 63);
}

// inclusive prefix sum / prefix maximum over the lanes of the wavefront (non-negative ints; the same DPP sequence as
// wave_sum, which is that scan read at lane 63)
__device__ __forceinline__ int wave_scan_add(int v) {
  v += dpp_i<0x111>(v);
  v += dpp_i<0x112>(v);
  v += dpp_i<0x114>(v);
  v += dpp_i<0x118>(v);
  v += dpp_i<0x142, 0xa>(v);
  v += dpp_i<0x143, 0xc>(v);
  return v;
}
__device__ __forceinline__ int wave_scan_max_nonneg(int v) {
  v = max(v, dpp_i<0x111>(v));
  v = max(v, dpp_i<0x112>(v));
  v = max(v, dpp_i<0x114>(v));
  v = max(v, dpp_i<0x118>(v));
  v = max(v, dpp_i<0x142, 0xa>(v));
  v = max(v, dpp_i<0x143, 0xc>(v));
  return v;
}
// maxima of NON-NEGATIVE values (the 0 fill of the DPP moves is then neutral)
__device__ __forceinline__ double wave_max_nonneg(double v) {
  v = fmax(v, dpp_d<0x111>(v));
  v = fmax(v, dpp_d<0x112>(v));
  v = fmax(v, dpp_d<0x114>(v));
  v = fmax(v, dpp_d<0x118>(v));
  v = fmax(v, dpp_any_d<0x142, 0xa>(v));
  v = fmax(v, dpp_any_d<0x143, 0xc>(v));
  return rdlane(v, 63);
}
__device__ __forceinline__ float wave_max_nonneg(float v) {
  v = fmaxf(v, dpp_f<0x111>(v));
  v = fmaxf(v, dpp_f<0x112>(v));
  v = fmaxf(v, dpp_f<0x114>(v));
  v = fmaxf(v, dpp_f<0x118>(v));
  v = fmaxf(v, __int_as_float(dpp_any_i<0x142, 0xa>(__float_as_int(v))));
  v = fmaxf(v, __int_as_float(dpp_any_i<0x143, 0xc>(__float_as_int(v))));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_max_nonneg(int v) {
  v = max(v, dpp_i<0x111>(v));
  v = max(v, dpp_i<0x112>(v));
  v = max(v, dpp_i<0x114>(v));
  v = max(v, dpp_i<0x118>(v));
  v = max(v, dpp_any_i<0x142, 0xa>(v));
  v = max(v, dpp_any_i<0x143, 0xc>(v));
  return __builtin_amdgcn_readlane(v, 63);
}
// value of lane (l-1) / (l+1); lanes without such a neighbour get `fill`
__device__ __forceinline__ double from_prev(double v, double fill) {
  const double o = dpp_d<0x138>(v);  // wave_shr:1
  return lane_eq(0) ? fill : o;
}
__device__ __forceinline__ double from_next(double v, double fill) {
  const double o = dpp_d<0x130>(v);  // wave_shl:1
  return lane_eq(kWave - 1) ? fill : o;
}
__device__ __forceinline__ float from_prev(float v, float fill) {
  const float o = dpp_f<0x138>(v);
  return lane_eq(0) ? fill : o;
}
__device__ __forceinline__ float from_next(float v, float fill) {
  const float o = dpp_f<0x130>(v);
  return lane_eq(kWave - 1) ? fill : o;
}

// ------------------------------------------------------------------ map lookups
// esdf.py:53-82: nearest cell with int() truncation; out of range -> 10000 / zero gradient.
// Returns the distance; the gradient (metres per cell, as np.gradient leaves it) is only
// fetched when the caller needs it -- it lives in the same 32-byte record.
// Lookups are split into prepare (index arithmetic) / load (the gathers) / finish (interpolation) so
// that the sample loop can put the loads of several samples in flight before it consumes any.
template <typename Real>
struct Lookup2D {
  const Map2D &m;
  __device__ __forceinline__ explicit Lookup2D(const Map2D &m_) : m(m_) {}
  struct Addr {
    int idx;
    bool inside;
  };
  struct Raw {
    double4 r;
  };
  // `on` = false (an idle sample slot): treated like a point outside the map, so that all idle lanes of a
  // wavefront read one and the same record instead of 64 scattered ones
  template <int D>
  __device__ __forceinline__ Addr prepare(const Real (&pos)[D], bool on = true) const {
    // index arithmetic in fp64 with a true division, exactly like int((y - origin.y) / res)
    const double fy = ((double)pos[1] - m.oy) / m.res;
    const double fx = ((double)pos[0] - m.ox) / m.res;
    Addr a;
    a.idx = 0;
    a.inside = false;
    if (!on || !(fabs(fy) < 1.0e9) || !(fabs(fx) < 1.0e9)) return a;
    const int row = (int)fy, col = (int)fx;  // C casts truncate toward zero, like int()
    if (row < 0 || row >= m.H || col < 0 || col >= m.W) return a;
    a.inside = true;
    a.idx = row * m.W + col;
    return a;
  }
  __device__ __forceinline__ Raw load(const Addr &a) const {
    Raw q;
    q.r = m.rec[a.idx];  // idx = 0 when outside: a valid, ignored record
    return q;
  }
  template <int D>
  __device__ __forceinline__ Real finish(const Addr &a, const Raw &q, Real (&g)[D]) const {
#pragma unroll
    for (int d = 0; d < D; ++d) g[d] = Real(0);
    if (!a.inside) return Real(10000);
    g[0] = (Real)q.r.y;
    g[1] = (Real)q.r.z;
    return (Real)q.r.x;
  }
  template <int D>
  __device__ __forceinline__ Real fetch(const Real (&pos)[D], Real (&g)[D], bool &inside) const {
    const Addr a = prepare<D>(pos);
    inside = a.inside;
    return finish<D>(a, load(a), g);
  }
};

template <typename E>
__device__ __forceinline__ float elem_to_float(E v);
template <>
__device__ __forceinline__ float elem_to_float<float>(float v) { return v; }
template <>
__device__ __forceinline__ float elem_to_float<__half>(__half v) { return __half2float(v); }

// two x-adjacent voxels in ONE load that is only element-aligned: an integer of twice the element
// size with reduced alignment (gfx950 global loads accept dword-aligned dwordx2), split afterwards.
// (Loading a two-member struct instead gets scalarised into two loads before the backend sees it.)
typedef unsigned long long u64_align4 __attribute__((aligned(4)));
typedef unsigned int u32_align2 __attribute__((aligned(2)));
template <typename E>
__device__ __forceinline__ void load_pair(const E *p, float &a, float &b);
template <>
__device__ __forceinline__ void load_pair<float>(const float *p, float &a, float &b) {
  const unsigned long long u = *reinterpret_cast<const u64_align4 *>(p);
  a = __uint_as_float((unsigned int)(u & 0xffffffffull));
  b = __uint_as_float((unsigned int)(u >> 32));
}
// The compiler splits a dword-aligned 8-byte *global* load into two dword loads; a raw buffer load
// of 8 bytes at a dword-aligned offset is one instruction and returns the right data on gfx950
// (tools/probe/unaligned_pair.hip).  `off` = byte offset into the field.
// `soff` is a wave-uniform byte offset (an SGPR operand of the instruction: the four x-pairs of a lookup share
// one address register).
__device__ __forceinline__ void buffer_load_pair_f32(__amdgpu_buffer_rsrc_t rsrc, unsigned int off, unsigned int soff,
                                                     float &a, float &b) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off, (int)soff, 0);
  a = __uint_as_float(v[0]);
  b = __uint_as_float(v[1]);
}
template <>
__device__ __forceinline__ void load_pair<__half>(const __half *p, float &a, float &b) {
  const unsigned int u = *reinterpret_cast<const u32_align2 *>(p);
  a = __half2float(__ushort_as_half((unsigned short)(u & 0xffffu)));
  b = __half2float(__ushort_as_half((unsigned short)(u >> 16)));
}

// trilinear distance + analytic gradient (oracle/minco_np.py:Grid3DESDF defines the semantics)
// LAYOUT is a template parameter (0 linear, 1 yz-quads, 2 cell-packed, 9 = read Map3D::layout at run time,
// for the point-query kernel only): a run-time branch on the layout inside the sample
// loop makes the compiler join the two load paths and wait for each sample's loads right there,
// which defeats keeping several samples' gathers in flight.
template <typename Real, typename E, int LAYOUT>
struct Lookup3D {
  const Map3D &m;
  __amdgpu_buffer_rsrc_t rsrc;
  // fp32 arithmetic: cell coordinate minus one half in ONE fma, um = pos * inv + off (all wave-uniform operands)
  float inv, off[3], hi[3];
  __device__ __forceinline__ explicit Lookup3D(const Map3D &m_)
      : m(m_), rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(m_.data), 0, (int)m_.bytes, 0x00020000)) {
    inv = m_.f_inv;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      off[k] = m_.f_off[k];
      hi[k] = m_.f_hi[k];
    }
  }
  struct Addr {
    int i0[3];
    Real fr[3];
    bool inside;
  };
  struct Raw {
    float c[2][2][2];
  };

  // `on` = false (an idle sample slot): treated like a point outside the field -- corner (0,0,0) for every
  // idle lane, i.e. one cache line per wavefront instead of 64 scattered gathers
  template <int D>
  __device__ __forceinline__ Addr prepare(const Real (&pos)[D], bool on = true) const {
#pragma clang fp contract(on)  // fuse a*b+c only as written: the same arithmetic whatever the unrolling around it
    static_assert(D == 3, "the 3-D map needs D = 3");
    const int n[3] = {m.nx, m.ny, m.nz};
    const double org[3] = {m.ox, m.oy, m.oz};
    Addr a;
    a.inside = on;
    if constexpr (sizeof(Real) == 4) {
      // (measured and rejected: clamping the cell coordinate with v_med3_f32 before truncating -- 7 instructions an axis
      // instead of 10, but nine more wave-uniform float constants, of which an instruction can name only one as a
      // scalar operand; the integer clamps below take theirs from scalar registers: 1.22 M against 1.25 M traj/s at cfg2)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float um = fmaf(pos[k], inv, off[k]);
        if (!(um >= -0.5f && um < hi[k])) a.inside = false;
        const int i = min(max((int)floorf(um), 0), n[k] - 2);
        a.i0[k] = i;
        a.fr[k] = __builtin_amdgcn_fmed3f(um - (float)i, 0.0f, 1.0f);
      }
      if (!a.inside) a.i0[0] = a.i0[1] = a.i0[2] = 0;
      return a;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      Real u = (Real)(((double)pos[k] - org[k]) * m.d_inv);  // oracle Grid3DESDF._cell divides by res: an ulp apart at most
      if (!(u >= Real(0) && u < (Real)n[k])) a.inside = false;
      u -= Real(0.5);
      const int i = min(max((int)floor(u), 0), n[k] - 2);
      a.i0[k] = i;
      a.fr[k] = fmin(fmax(u - (Real)i, Real(0)), Real(1));
    }
    if (!a.inside) a.i0[0] = a.i0[1] = a.i0[2] = 0;
    return a;
  }
  __device__ __forceinline__ Raw load(const Addr &a) const {
    const E *vox = static_cast<const E *>(m.data);
    Raw q;
    if (LAYOUT == 2 || (LAYOUT == 9 && m.layout == 2)) {
      // cell-packed: the 8 corners [dz][dy][dx] of cell (ix, iy, iz) are contiguous and 16-byte aligned:
      // one lookup = one 32-byte (fp32) or 16-byte (fp16) read instead of four gathers
      const unsigned int cell = __umul24(__umul24((unsigned)a.i0[2], (unsigned)m.ny) + (unsigned)a.i0[1], (unsigned)m.nx) +
                                (unsigned)a.i0[0];
      if constexpr (sizeof(E) == 4) {
        const auto lo = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cell * 32u), 0, 0);
        const auto hi = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cell * 32u + 16u), 0, 0);
        q.c[0][0][0] = __uint_as_float(lo[0]); q.c[0][0][1] = __uint_as_float(lo[1]);
        q.c[0][1][0] = __uint_as_float(lo[2]); q.c[0][1][1] = __uint_as_float(lo[3]);
        q.c[1][0][0] = __uint_as_float(hi[0]); q.c[1][0][1] = __uint_as_float(hi[1]);
        q.c[1][1][0] = __uint_as_float(hi[2]); q.c[1][1][1] = __uint_as_float(hi[3]);
      } else {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cell * 16u), 0, 0);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          q.c[w >> 1][w & 1][0] = __half2float(__ushort_as_half((unsigned short)(v[w] & 0xffffu)));
          q.c[w >> 1][w & 1][1] = __half2float(__ushort_as_half((unsigned short)(v[w] >> 16)));
        }
      }
    } else if (LAYOUT == 3 || (LAYOUT == 9 && m.layout == 3)) {
      // corner bricks: the cells are grouped in blocks of 2 x 2 x 2 (fp32) or 4 x 2 x 2 (fp16: x is the direction most
      // requests fly in), and all (2+1)^3 = 27 (or 5 x 3 x 3 = 45) corners of a block sit in ONE 128-byte line, [z][y][x]
      // inside the line.  A lookup is four x-pairs of that line -- one address register, four immediate offsets -- and
      // a path stays on the line for two cells in EVERY direction, where the yz-quad line is eight cells along x and one
      // along y and z: fewer lines per path (tools/sim_esdf_locality.py: -27 % lines fetched at cfg2), the same memory
      // (16 bytes a cell in fp32).
      constexpr int SHX = sizeof(E) == 4 ? 1 : 2, CX = (1 << SHX) + 1;  // cells (log2) and corners of a block along x
      const unsigned int bx = (unsigned)a.i0[0] >> SHX, by = (unsigned)a.i0[1] >> 1, bz = (unsigned)a.i0[2] >> 1;
      const unsigned int lx = (unsigned)a.i0[0] & ((1u << SHX) - 1u), ly = (unsigned)a.i0[1] & 1u, lz = (unsigned)a.i0[2] & 1u;
      const unsigned int blk = __umul24(__umul24(bz, (unsigned)m.nby) + by, (unsigned)m.nbx) + bx;
      const unsigned int in_line = (lz * 3u + ly) * (unsigned)CX + lx;  // element of corner (0,0,0) inside the line
#pragma unroll
      for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const unsigned int so = (unsigned)((dz * 3 + dy) * CX);  // compile-time: the instruction's immediate offset
          if constexpr (sizeof(E) == 4) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(blk * 128u + in_line * 4u + so * 4u), 0, 0);
            q.c[dz][dy][0] = __uint_as_float(v[0]);
            q.c[dz][dy][1] = __uint_as_float(v[1]);
          } else {
            load_pair<E>(vox + blk * 64u + in_line + so, q.c[dz][dy][0], q.c[dz][dy][1]);
          }
        }
    } else if (LAYOUT == 0 || (LAYOUT == 9 && m.layout == 0)) {
      // 32-bit element index of corner (0,0,0); the other three x-pairs sit at +nx, +nx*ny, +nx*ny+nx
      const unsigned int base = __umul24(__umul24((unsigned)a.i0[2], (unsigned)m.ny) + (unsigned)a.i0[1], (unsigned)m.nx) +
                                (unsigned)a.i0[0];
      const unsigned int sy = (unsigned)m.nx, sz = (unsigned)m.nx * (unsigned)m.ny;
#pragma unroll
      for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const unsigned int so = (dy ? sy : 0u) + (dz ? sz : 0u);
          if constexpr (sizeof(E) == 4)
            buffer_load_pair_f32(rsrc, base * 4u, so * 4u, q.c[dz][dy][0], q.c[dz][dy][1]);
          else  // (a 4-byte buffer load at a 2-byte-aligned offset works too, tools/probe/unaligned_half_pair.hip,
                //  but measured 2 % slower at cfg5 than the global load)
            load_pair<E>(vox + base + so, q.c[dz][dy][0], q.c[dz][dy][1]);
        }
    } else {
      // yz-quads: voxel (ix, iy, iz) stores {d(iy,iz), d(iy+1,iz), d(iy,iz+1), d(iy+1,iz+1)} at x = ix, records in
      // [z][y][x] order: the 8 corners of a cell are the records ix and ix + 1 = 32 contiguous bytes (fp32; 16 for fp16),
      // and x-adjacent cells share half of them -- 4x the memory of the linear layout instead of 8x, one line per
      // lookup instead of four, and a sample triple walking along x stays on its 128-byte line for 8 cells
      const unsigned int cell = __umul24(__umul24((unsigned)a.i0[2], (unsigned)m.ny) + (unsigned)a.i0[1], (unsigned)m.nx) +
                                (unsigned)a.i0[0];
      if constexpr (sizeof(E) == 4) {
        const auto lo = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cell * 16u), 0, 0);
        const auto hi = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cell * 16u + 16u), 0, 0);
        q.c[0][0][0] = __uint_as_float(lo[0]); q.c[0][1][0] = __uint_as_float(lo[1]);
        q.c[1][0][0] = __uint_as_float(lo[2]); q.c[1][1][0] = __uint_as_float(lo[3]);
        q.c[0][0][1] = __uint_as_float(hi[0]); q.c[0][1][1] = __uint_as_float(hi[1]);
        q.c[1][0][1] = __uint_as_float(hi[2]); q.c[1][1][1] = __uint_as_float(hi[3]);
      } else {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(cell * 8u), 0, 0);
#pragma unroll
        for (int dx = 0; dx < 2; ++dx)
#pragma unroll
          for (int dz = 0; dz < 2; ++dz) {
            const unsigned int u = v[2 * dx + dz];
            q.c[dz][0][dx] = __half2float(__ushort_as_half((unsigned short)(u & 0xffffu)));
            q.c[dz][1][dx] = __half2float(__ushort_as_half((unsigned short)(u >> 16)));
          }
      }
    }
    return q;
  }
  // ---- the gathers of load() as LDS-DMA (yz-quad layout only): the 32 (fp32) / 16 (fp16) bytes of a lookup go from
  // HBM / L2 straight into LDS -- `buffer_load_dwordx4 ... offen lds`, no destination registers -- so a lane can have as
  // many lookups in flight as the wavefront has landing room, instead of as many as it has spare VGPRs.  One
  // wave-instruction lands lane-linear: piece k of the wavefront's 64 lookups at land + k * 1024 + lane * 16.
  static constexpr int kAsyncPieces = sizeof(E) == 4 ? 2 : 1;   // 16-byte pieces per lookup
  static constexpr int kAsyncBytes = kAsyncPieces * 1024;       // landing bytes per wavefront and lookup round
  typedef __attribute__((address_space(3))) void *LdsPtr;
  __device__ __forceinline__ void load_async(const Addr &a, char *land /*wave-uniform LDS address*/) const {
    static_assert(LAYOUT == 1, "LDS-DMA gathers are written for the yz-quad layout");
    const unsigned int cell = __umul24(__umul24((unsigned)a.i0[2], (unsigned)m.ny) + (unsigned)a.i0[1], (unsigned)m.nx) +
                              (unsigned)a.i0[0];
    if constexpr (sizeof(E) == 4) {
      // (the second half at its own register offset, hidden from the compiler: folded into the instruction's immediate
      //  offset -- which it does with a constant in either offset operand -- the 16 would be added to the LDS address as
      //  well as to the memory address and shift the landing by one lane)
      const unsigned int off0 = cell * 16u;
      unsigned int off1 = off0 + 16u;
      asm volatile("" : "+v"(off1));
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsPtr)land, 16, (int)off0, 0, 0, 0);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsPtr)(land + 1024), 16, (int)off1, 0, 0, 0);
    } else {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LdsPtr)land, 16, (int)(cell * 8u), 0, 0, 0);
    }
  }
  // the lane's own lookup back out of the landing area (after the wavefront has waited for its LDS-DMA loads)
  __device__ __forceinline__ Raw read_staged(const char *land) const {
    typedef unsigned int U4 __attribute__((ext_vector_type(4)));
    Raw q;
    const U4 lo = *reinterpret_cast<const U4 *>(land + lane_id() * 16);
    if constexpr (sizeof(E) == 4) {
      const U4 hi = *reinterpret_cast<const U4 *>(land + 1024 + lane_id() * 16);
      q.c[0][0][0] = __uint_as_float(lo[0]); q.c[0][1][0] = __uint_as_float(lo[1]);
      q.c[1][0][0] = __uint_as_float(lo[2]); q.c[1][1][0] = __uint_as_float(lo[3]);
      q.c[0][0][1] = __uint_as_float(hi[0]); q.c[0][1][1] = __uint_as_float(hi[1]);
      q.c[1][0][1] = __uint_as_float(hi[2]); q.c[1][1][1] = __uint_as_float(hi[3]);
    } else {
#pragma unroll
      for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int dz = 0; dz < 2; ++dz) {
          const unsigned int u = lo[2 * dx + dz];
          q.c[dz][0][dx] = __half2float(__ushort_as_half((unsigned short)(u & 0xffffu)));
          q.c[dz][1][dx] = __half2float(__ushort_as_half((unsigned short)(u >> 16)));
        }
    }
    return q;
  }
  template <int D>
  __device__ __forceinline__ Real finish(const Addr &a, const Raw &q, Real (&g)[D]) const {
#pragma clang fp contract(on)  // fuse a*b+c only as written: the same arithmetic whatever the unrolling around it
#pragma unroll
    for (int d = 0; d < D; ++d) g[d] = Real(0);
    if (!a.inside) return Real(10000);
    const Real fx = a.fr[0], fy = a.fr[1], fz = a.fr[2];
    const Real inv_res = sizeof(Real) == 4 ? (Real)m.f_inv : (Real)m.d_inv;
    const Real c000 = (Real)q.c[0][0][0], c100 = (Real)q.c[0][0][1], c010 = (Real)q.c[0][1][0], c110 = (Real)q.c[0][1][1];
    const Real c001 = (Real)q.c[1][0][0], c101 = (Real)q.c[1][0][1], c011 = (Real)q.c[1][1][0], c111 = (Real)q.c[1][1][1];
    const Real dx00 = c100 - c000, dx10 = c110 - c010, dx01 = c101 - c001, dx11 = c111 - c011;
    const Real c00 = c000 + fx * dx00, c10 = c010 + fx * dx10;
    const Real c01 = c001 + fx * dx01, c11 = c011 + fx * dx11;
    const Real c0 = c00 + fy * (c10 - c00), c1 = c01 + fy * (c11 - c01);
    const Real dx0 = dx00 + fy * (dx10 - dx00), dx1 = dx01 + fy * (dx11 - dx01);
    const Real dy0 = c10 - c00, dy1 = c11 - c01;
    g[0] = (dx0 + fz * (dx1 - dx0)) * inv_res;
    g[1] = (dy0 + fz * (dy1 - dy0)) * inv_res;
    g[2] = (c1 - c0) * inv_res;
    return c0 + fz * (c1 - c0);
  }
  template <int D>
  __device__ __forceinline__ Real fetch(const Real (&pos)[D], Real (&g)[D], bool &inside) const {
    const Addr a = prepare<D>(pos);
    inside = a.inside;
    return finish<D>(a, load(a), g);
  }
};

__host__ __device__ __forceinline__ int sample_lanes_per_piece_fwd(int M);

// ------------------------------------------------------------------ lane groups
// The per-trajectory functions below are written once for a GROUP of lanes that owns one trajectory:
//   WaveLanes      the whole 64-lane wavefront (optimize / eval / sample kernels);
//   GroupLanes<W>  W = 16 or 8 lanes: four or eight small trajectories share a wavefront (optimize_group_kernel) and
//                  run the same instruction stream in lock step.
// `lane()` is the lane inside the group; cross-lane traffic (sums, neighbour moves, broadcasts) stays inside the
// group, which for W <= 16 lies inside one 16-lane DPP row.  All lanes of a wavefront call the functions together.
//   WaveLanesPD<D> the whole wavefront with lane = (piece, dimension): D lanes per piece, ONE dimension's state per lane
//                  (D * M <= 64: cfg2's M = 21 uses 63 lanes instead of 21 in the PIECE-layout phases).  Everything that
//                  is per dimension -- boundary states, velocities, coefficients, every right-hand side of the joint
//                  solves, partials -- shrinks to a third per lane: a third of the registers and of the instructions;
//                  what depends on the durations only (the factorisation) is computed alike by the D lanes of a piece.
// `piece()` is the piece (= joint) of the lane, `S` the lanes per piece, `dl(D)` the dimensions held per lane and
// `dim0()` the first of them; `prev` / `next` / `read` address pieces, not lanes.
struct WaveLanes {
  static constexpr int W = kWave;
  static constexpr int S = 1;
  static constexpr int dl(int D) { return D; }
  static __device__ __forceinline__ int lane() { return lane_id(); }
  static __device__ __forceinline__ int piece() { return lane_id(); }
  static __device__ __forceinline__ int dim0() { return 0; }
  static __device__ __forceinline__ double sum_dims(double v) { return v; }
  static __device__ __forceinline__ int base() { return 0; }
  static __device__ __forceinline__ double read(double v, int src /*wave-uniform*/) { return rdlane(v, src); }
  static __device__ __forceinline__ double prev(double v, double fill) { return from_prev(v, fill); }
  static __device__ __forceinline__ double next(double v, double fill) { return from_next(v, fill); }
  static __device__ __forceinline__ double sum(double v) { return wave_sum(v); }
  static __device__ __forceinline__ int sum(int v) { return wave_sum(v); }
  static __device__ __forceinline__ int any(int pred) { return __any(pred); }
  static __device__ __forceinline__ float sum_dims(float v) { return v; }
  static __device__ __forceinline__ float read(float v, int src) { return rdlane(v, src); }
  static __device__ __forceinline__ float prev(float v, float fill) { return from_prev(v, fill); }
  static __device__ __forceinline__ float next(float v, float fill) { return from_next(v, fill); }
  static __device__ __forceinline__ float sum(float v) { return wave_sum(v); }
  static __host__ __device__ __forceinline__ int lanes_per_piece(int M);
};

template <int W_>
struct GroupLanes {
  static_assert(W_ == 16 || W_ == 8, "a group is a DPP row or half of one");
  static constexpr int W = W_;
  static constexpr int S = 1;
  static constexpr int dl(int D) { return D; }
  static __device__ __forceinline__ int piece() { return lane_id() & (W - 1); }
  static __device__ __forceinline__ int dim0() { return 0; }
  static __device__ __forceinline__ double sum_dims(double v) { return v; }
  static __device__ __forceinline__ int lane() { return lane_id() & (W - 1); }
  static __device__ __forceinline__ int base() { return lane_id() & ~(W - 1); }
  // sums / maxima over the group, result in every lane.  W = 16: the in-row part of wave_sum (row_shr scan, then the
  // last lane's value to everybody).  W = 8: xor butterfly (quad_perm swaps, then row_half_mirror), which never reads
  // the other group of the row.  Both associate ((v0+v1)+(v2+v3)) + ((v4+v5)+(v6+v7)) [+ the same of the upper half].
  template <class T, class Op>
  static __device__ __forceinline__ T reduce(T v, Op op) {
    if constexpr (W == 16) {
      if constexpr (sizeof(T) == 8) {
        v = op(v, dpp_d<0x111>(v)); v = op(v, dpp_d<0x112>(v)); v = op(v, dpp_d<0x114>(v)); v = op(v, dpp_d<0x118>(v));
      } else if constexpr (std::is_same<T, float>::value) {
        v = op(v, dpp_f<0x111>(v)); v = op(v, dpp_f<0x112>(v)); v = op(v, dpp_f<0x114>(v)); v = op(v, dpp_f<0x118>(v));
      } else {
        v = op(v, dpp_i<0x111>(v)); v = op(v, dpp_i<0x112>(v)); v = op(v, dpp_i<0x114>(v)); v = op(v, dpp_i<0x118>(v));
      }
      return __shfl(v, base() + W - 1, kWave);
    } else {
      // quad_perm:[1,0,3,2] = 0xB1, quad_perm:[2,3,0,1] = 0x4E, row_half_mirror = 0x141
      if constexpr (sizeof(T) == 8) {
        v = op(v, dpp_d<0xB1>(v)); v = op(v, dpp_d<0x4E>(v)); v = op(v, dpp_d<0x141>(v));
      } else if constexpr (std::is_same<T, float>::value) {
        v = op(v, dpp_f<0xB1>(v)); v = op(v, dpp_f<0x4E>(v)); v = op(v, dpp_f<0x141>(v));
      } else {
        v = op(v, dpp_i<0xB1>(v)); v = op(v, dpp_i<0x4E>(v)); v = op(v, dpp_i<0x141>(v));
      }
      return v;
    }
  }
  static __device__ __forceinline__ double sum(double v) {
    return reduce(v, [](double a, double b) { return a + b; });
  }
  static __device__ __forceinline__ int sum(int v) {
    return reduce(v, [](int a, int b) { return a + b; });
  }
  static __device__ __forceinline__ double max_nonneg(double v) {
    return reduce(v, [](double a, double b) { return fmax(a, b); });
  }
  static __device__ __forceinline__ int any(int pred) {
    return reduce(pred ? 1 : 0, [](int a, int b) { return a | b; });
  }
  static __device__ __forceinline__ double read(double v, int src /* lane inside the group, the same for all groups */) {
    return __shfl(v, base() + src, kWave);
  }
  static __device__ __forceinline__ float sum(float v) {
    return reduce(v, [](float a, float b) { return a + b; });
  }
  static __device__ __forceinline__ float max_nonneg(float v) {
    return reduce(v, [](float a, float b) { return fmaxf(a, b); });
  }
  static __device__ __forceinline__ float sum_dims(float v) { return v; }
  static __device__ __forceinline__ float read(float v, int src) { return __shfl(v, base() + src, kWave); }
  static __device__ __forceinline__ float prev(float v, float fill) {
    const float o = dpp_f<0x138>(v);
    return lane() == 0 ? fill : o;
  }
  static __device__ __forceinline__ float next(float v, float fill) {
    const float o = dpp_f<0x130>(v);
    return lane() == W - 1 ? fill : o;
  }
  static __device__ __forceinline__ double prev(double v, double fill) {
    const double o = dpp_d<0x138>(v);  // wave_shr:1
    return lane() == 0 ? fill : o;
  }
  static __device__ __forceinline__ double next(double v, double fill) {
    const double o = dpp_d<0x130>(v);  // wave_shl:1
    return lane() == W - 1 ? fill : o;
  }
  static __host__ __device__ __forceinline__ int lanes_per_piece(int M) {
    int L = W / M;
    if (L < 1) L = 1;
    if (L >= 8) L = (L >= 16 && W >= 16) ? 16 : 8;
    return L;
  }
};

template <int D_>
struct WaveLanesPD {
  static constexpr int W = kWave;
  static constexpr int S = D_;
  static constexpr int dl(int) { return 1; }
  static __device__ __forceinline__ int lane() { return lane_id(); }
  static __device__ __forceinline__ int base() { return 0; }
  static __device__ __forceinline__ int piece() { return (lane_id() * ((65536 + S - 1) / S)) >> 16; }  // lane / S
  // (round 4: an opaque piece number -- every mask compared at its use -- removes 19 more of the 45 remaining scalar
  //  reloads an evaluation and adds 65 vector instructions: 1.457 against 1.455 M traj/s, not kept)
  static __device__ __forceinline__ int dim0() { return lane_id() - S * piece(); }
  // value held by piece q (any of its lanes: used for quantities that depend on the durations only)
  static __device__ __forceinline__ double read(double v, int q /*wave-uniform*/) { return rdlane(v, S * q); }
  // value of the same dimension in the previous / next piece: S single-lane DPP shifts
  static __device__ __forceinline__ double prev(double v, double fill) {
#pragma unroll
    for (int k = 0; k < S; ++k) v = dpp_d<0x138>(v);  // wave_shr:1
    return lane_lt(S) ? fill : v;
  }
  static __device__ __forceinline__ double next(double v, double fill) {
#pragma unroll
    for (int k = 0; k < S; ++k) v = dpp_d<0x130>(v);  // wave_shl:1
    return lane_ge(kWave - S) ? fill : v;
  }
  static __device__ __forceinline__ double sum(double v) { return wave_sum(v); }
  static __device__ __forceinline__ int sum(int v) { return wave_sum(v); }
  static __device__ __forceinline__ int any(int pred) { return __any(pred); }
  static __device__ __forceinline__ float read(float v, int q) { return rdlane(v, S * q); }
  // (one ds_bpermute instead of the S chained shifts was measured on the MI355X and lost: cfg2 1.39 -> 1.37 M traj/s, a
  //  single batch 692 k -> 660 k -- the LDS round trip sits on the dependent chain, and the LDS pipe is busy too)
  static __device__ __forceinline__ float prev(float v, float fill) {
#pragma unroll
    for (int k = 0; k < S; ++k) v = dpp_f<0x138>(v);
    return lane_lt(S) ? fill : v;
  }
  static __device__ __forceinline__ float next(float v, float fill) {
#pragma unroll
    for (int k = 0; k < S; ++k) v = dpp_f<0x130>(v);
    return lane_ge(kWave - S) ? fill : v;
  }
  static __device__ __forceinline__ float sum(float v) { return wave_sum(v); }
  static __device__ __forceinline__ float sum_dims(float v) {
    float acc = v, t = v;
#pragma unroll
    for (int k = 1; k < S; ++k) {
      t = dpp_f<0x130>(t);
      acc += t;
    }
    return acc;
  }
  // sum over the S lanes (dimensions) of a piece; valid in the piece's first lane
  static __device__ __forceinline__ double sum_dims(double v) {
    double acc = v, t = v;
#pragma unroll
    for (int k = 1; k < S; ++k) {
      t = dpp_d<0x130>(t);
      acc += t;
    }
    return acc;
  }
  static __host__ __device__ __forceinline__ int lanes_per_piece(int M) { return sample_lanes_per_piece_fwd(M); }
};

// ------------------------------------------------------------------ per-trajectory state
// DL = dimensions held per lane: D (PIECE layout: lane = piece) or 1 (lane = (piece, dimension), WaveLanesPD)
template <int D, int DL = D, typename Num = double>
struct Traj {
  // wave-uniform
  int M, n, nq, L;
  // PIECE layout (lane p < M; with DL = 1 every lane of the piece)
  Num T, tau;                  // tau: the decision variable on entry to minco_forward, exp(-tau) after it
  Num i1, i2, i3, i4;          // T^-1 .. T^-4
  Num P0[DL], P1[DL];          // positions at the start / end joint
  Num V0[DL], A0[DL], V1[DL], A1[DL];
  Num c[6][DL];                // polynomial coefficients
  int ns;                         // samples of this piece: int(T / delta_t)
  Num N[2][2];                 // pivot-block inverse of the joint system (lane = joint)
  const double *head, *tail;      // boundary states [3][D] in global memory (wave-uniform scalar loads)
  const Num *bnd;                 // lane = (piece, dimension) and lane groups: head [3][D] then tail [3][D] in LDS
  Num *pcr_xch = nullptr;         // LDS [pcr_xch_elems]: exchange table of the cyclic reduction (the caller's staging buffer; nullptr:
                                  // the neighbours' blocks travel by ds_bpermute)
  Num *pcr_mult = nullptr;        // LDS [levels][M][8] + [M][4]: the multipliers of the forward cyclic reduction, kept for the
                                  // adjoint pass (pcr_solve / pcr_solve_transposed); nullptr: the adjoint reduces K^T itself
};

// element (row k, dimension of this lane's local index dl) of the head (TAIL = false) or tail boundary state: a uniform
// scalar load when the lane holds all dimensions.  When it holds one, an LDS read of the copy stage_boundary() made
// once per trajectory: read through the global pointers at every evaluation, the compiler keeps one 64-bit address per
// lane and row alive across the optimiser loop (12 registers, all spilled in the three-waves-per-SIMD kernels) and the
// loads queue behind the other wavefronts' field gathers -- measured 1.06 M -> 1.20 M traj/s at cfg2 without them.
template <int D, class LG, bool TAIL, int DL, typename Num>
__device__ __forceinline__ Num bstate(const Traj<D, DL, Num> &t, int k, int dl) {
  if constexpr (LG::S == 1 && LG::W == kWave)
    return (Num)(TAIL ? t.tail : t.head)[k * D + dl];
  else  // (lane groups: every group has its own trajectory, so the pointers differ by lane and the loads would be vector loads)
    return t.bnd[((TAIL ? 3 : 0) + k) * D + (LG::S > 1 ? LG::dim0() : dl)];
}
// lds: 6 * D elements of the wavefront's own
template <int D, class LG, int DL, typename Num>
__device__ __forceinline__ void stage_boundary(Traj<D, DL, Num> &t, Num *lds) {
  if constexpr (LG::S > 1) {
    const int lane = lane_id();
    if (lane < 3 * D) lds[lane] = (Num)t.head[lane];
    if (lane < 3 * D) lds[3 * D + lane] = (Num)t.tail[lane];
    lds_wave_sync();
    t.bnd = lds;
  }
}

// Block-tridiagonal systems with 2x2 blocks on the interior joints p = 1..M-1 (blocks on lane p):
//     Lo_p y_{p-1} + Di_p y_p + Up_p y_{p+1} = R_p,      y_0 and y_M given.
// Block Thomas, no pivoting across blocks.  thomas_factor runs the right-hand-side independent part
// once per evaluation: N_p = (Di_p - Lo_p E_{p-1})^-1, E_p = N_p Up_p (sequential over joints, lane p-1
// hands E to lane p through v_readlane).  thomas_solve then needs only N, E and the sub-diagonal.
// The transposed system of the adjoint pass reuses the same pivot inverses: the Schur complements of
// K^T are the transposes of those of K, so its N is N^T and its E is N^T Lo_{p+1}^T -- no second
// factorisation, no second set of divisions.
template <class LG = WaveLanes, typename Num = double>
__device__ __forceinline__ void thomas_factor(int M, const Num (&Lo)[2][2], const Num (&Di)[2][2],
                                              const Num (&Up)[2][2], Num (&N)[2][2], Num (&E)[2][2]) {
  // One step per joint: lane p forms D_p = Di_p - Lo_p E_{p-1}, inverts it and keeps N_p = D_p^-1, E_p = N_p Up_p.
  // The optimiser kernels run two wavefronts per SIMD and are bound by instruction issue more than by the length of
  // this chain, so the step is written for the fewest instructions: the reciprocal of the determinant is v_rcp_f64
  // with two Newton steps (5 instructions; measured on gfx950, tools/probe/rcp_precision.hip: 4.6e-8 raw, 2.2e-15 after
  // one step, 1.1e-16 after two; a correctly rounded division is 12 instructions), everything
  // is computed by lane p under its own exec mask straight into N and E, and only E travels to the next lane (8
  // v_readlane).  35 instructions a joint; carrying E as a fraction to keep the division out of the chain (the
  // round-1 form) cost 52 and two divisions after the loop.
  const int lane = LG::piece();  // (every lane of a piece computes the same factors)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) N[i][j] = E[i][j] = Num(0.0);  // lane 0: E_0 = 0; lanes outside 1..M-1 stay 0
  for (int p = 1; p < M; ++p) {
    const Num e00 = LG::read(E[0][0], p - 1), e01 = LG::read(E[0][1], p - 1);
    const Num e10 = LG::read(E[1][0], p - 1), e11 = LG::read(E[1][1], p - 1);
    if (lane == p) {
      const Num h00 = fma(-Lo[0][1], e10, fma(-Lo[0][0], e00, Di[0][0]));
      const Num h01 = fma(-Lo[0][1], e11, fma(-Lo[0][0], e01, Di[0][1]));
      const Num h10 = fma(-Lo[1][1], e10, fma(-Lo[1][0], e00, Di[1][0]));
      const Num h11 = fma(-Lo[1][1], e11, fma(-Lo[1][0], e01, Di[1][1]));
      const Num det = fma(h00, h11, -(h01 * h10));
      const Num r = precise_rcp(det);
      N[0][0] = h11 * r;
      N[0][1] = -h01 * r;
      N[1][0] = -h10 * r;
      N[1][1] = h00 * r;
      E[0][0] = fma(N[0][0], Up[0][0], N[0][1] * Up[1][0]);
      E[0][1] = fma(N[0][0], Up[0][1], N[0][1] * Up[1][1]);
      E[1][0] = fma(N[1][0], Up[0][0], N[1][1] * Up[1][0]);
      E[1][1] = fma(N[1][0], Up[0][1], N[1][1] * Up[1][1]);
    }
  }
}

// Solve with the factors of thomas_factor.  Both substitution sweeps are linear recurrences,
//     f_p = N_p R_p - (N_p Lo_p) f_{p-1}        (f_0 = y_0),
//     y_p = f_p     -  E_p       y_{p+1}        (y_M given),
// i.e. compositions of affine maps v -> A v + b with 2x2 A.  Instead of walking the joints one by
// one (M-1 dependent steps each, one useful lane per step) they are evaluated as Kogge-Stone
// prefix / suffix scans over the lanes: ceil(log2 M) steps, every lane busy, 10x shorter
// dependent chain.  |A| < 1 for these diagonally dominant systems, so the products decay
// (checked against the sequential sweep to 6e-15 over T in [0.5,5]^M, M <= 64).
template <int DL, class LG = WaveLanes, typename Num = double>
__device__ __forceinline__ void thomas_solve(int M, const Num (&Lo)[2][2], const Num (&N)[2][2],
                                             const Num (&E)[2][2], const Num (&R)[2][DL],
                                             const Num (&y0)[2][DL], const Num (&yM)[2][DL],
                                             Num (&y)[2][DL]) {
  const int lane = LG::piece();
  Num A[2][2], b[2][DL];
  // ---- forward: lane 0 is the constant map v -> y_0
  {
    const bool in = lane >= 1 && lane < M;
    A[0][0] = in ? -(N[0][0] * Lo[0][0] + N[0][1] * Lo[1][0]) : Num(0.0);
    A[0][1] = in ? -(N[0][0] * Lo[0][1] + N[0][1] * Lo[1][1]) : Num(0.0);
    A[1][0] = in ? -(N[1][0] * Lo[0][0] + N[1][1] * Lo[1][0]) : Num(0.0);
    A[1][1] = in ? -(N[1][0] * Lo[0][1] + N[1][1] * Lo[1][1]) : Num(0.0);
#pragma unroll
    for (int d = 0; d < DL; ++d) {
      const Num r0 = N[0][0] * R[0][d] + N[0][1] * R[1][d];
      const Num r1 = N[1][0] * R[0][d] + N[1][1] * R[1][d];
      b[0][d] = lane == 0 ? y0[0][d] : (in ? r0 : Num(0.0));
      b[1][d] = lane == 0 ? y0[1][d] : (in ? r1 : Num(0.0));
    }
  }
  for (int s = 1; s < M; s <<= 1) {
    Num As[2][2], bs[2][DL];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) As[i][j] = __shfl_up(A[i][j], s * LG::S, kWave);
#pragma unroll
      for (int d = 0; d < DL; ++d) bs[i][d] = __shfl_up(b[i][d], s * LG::S, kWave);
    }
    if (lane >= s && lane < M) {
#pragma unroll
      for (int d = 0; d < DL; ++d) {
        const Num n0 = A[0][0] * bs[0][d] + A[0][1] * bs[1][d] + b[0][d];
        const Num n1 = A[1][0] * bs[0][d] + A[1][1] * bs[1][d] + b[1][d];
        b[0][d] = n0;
        b[1][d] = n1;
      }
      const Num a00 = A[0][0] * As[0][0] + A[0][1] * As[1][0], a01 = A[0][0] * As[0][1] + A[0][1] * As[1][1];
      const Num a10 = A[1][0] * As[0][0] + A[1][1] * As[1][0], a11 = A[1][0] * As[0][1] + A[1][1] * As[1][1];
      A[0][0] = a00; A[0][1] = a01; A[1][0] = a10; A[1][1] = a11;
    }
  }
  // b = f_p now.  ---- backward over lanes 1..M-1; lane M-1 absorbs y_M and becomes a constant map
  {
    const bool in = lane >= 1 && lane < M - 1;
#pragma unroll
    for (int d = 0; d < DL; ++d) {
      if (lane == M - 1) {
        b[0][d] -= E[0][0] * yM[0][d] + E[0][1] * yM[1][d];
        b[1][d] -= E[1][0] * yM[0][d] + E[1][1] * yM[1][d];
      }
    }
    A[0][0] = in ? -E[0][0] : Num(0.0);
    A[0][1] = in ? -E[0][1] : Num(0.0);
    A[1][0] = in ? -E[1][0] : Num(0.0);
    A[1][1] = in ? -E[1][1] : Num(0.0);
  }
  for (int s = 1; s < M; s <<= 1) {
    Num As[2][2], bs[2][DL];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) As[i][j] = __shfl_down(A[i][j], s * LG::S, kWave);
#pragma unroll
      for (int d = 0; d < DL; ++d) bs[i][d] = __shfl_down(b[i][d], s * LG::S, kWave);
    }
    if (lane >= 1 && lane + s <= M - 1) {
#pragma unroll
      for (int d = 0; d < DL; ++d) {
        const Num n0 = A[0][0] * bs[0][d] + A[0][1] * bs[1][d] + b[0][d];
        const Num n1 = A[1][0] * bs[0][d] + A[1][1] * bs[1][d] + b[1][d];
        b[0][d] = n0;
        b[1][d] = n1;
      }
      const Num a00 = A[0][0] * As[0][0] + A[0][1] * As[1][0], a01 = A[0][0] * As[0][1] + A[0][1] * As[1][1];
      const Num a10 = A[1][0] * As[0][0] + A[1][1] * As[1][0], a11 = A[1][0] * As[0][1] + A[1][1] * As[1][1];
      A[0][0] = a00; A[0][1] = a01; A[1][0] = a10; A[1][1] = a11;
    }
  }
#pragma unroll
  for (int d = 0; d < DL; ++d) {
    y[0][d] = b[0][d];
    y[1][d] = b[1][d];
  }
}

// The same block-tridiagonal systems by PARALLEL CYCLIC REDUCTION (all-fp32 mode, whole-wavefront lane groups).
// Block Thomas above walks the joints: 20 dependent steps at cfg2 in which ONE lane (three with lane = (piece, dimension))
// does the work of an instruction every lane pays for -- 18 % of the kernel's vector instructions -- followed by two
// Kogge-Stone scans per right-hand side (15 %).  In a kernel bound by the issue of its vector instructions that is the
// wrong trade.  PCR eliminates, in every equation at once, the neighbours at distance s = 1, 2, 4, ...:
//     a = -L_i D_{i-s}^-1,   g = -U_i D_{i+s}^-1,
//     L_i' = a L_{i-s},   U_i' = g U_{i+s},   D_i' = D_i + a U_{i-s} + g L_{i+s},   R_i' = R_i + a R_{i-s} + g R_{i+s};
// after ceil(log2(M-1)) levels the equations are uncoupled, y_i = D_i^-1 R_i.  63 vector instructions a level with
// every lane busy: 330 per solve instead of ~540, no factors to keep between the forward and the adjoint pass (the
// adjoint runs the reduction again on the transposed blocks: cheaper than carrying five levels of multipliers in
// registers, and the four registers of the pivot inverses are free across the sample loop).  Neighbours' blocks come
// through ds_bpermute (28 a level).  The systems are block diagonally dominant (|N Lo| < 1 above), so elimination without
// pivoting is as stable here as it is for block Thomas; in fp32 the coefficients agree with the fp64 solve to the same
// 1e-7 (tests/test_gpu_parity.py).  The fp64 modes keep block Thomas: their runs are pinned to its rounding.
// In: L, Dg, U, R of joint p on the lanes of piece p (1 <= p <= M-1; boundary values already folded into R, L_1 = 0,
// U_{M-1} = 0); other lanes are made identity rows here.  Out: y on those lanes.
// `mult` (LDS, may be nullptr): when given, every level's multipliers a, g of every joint and the final pivot inverses are
// written there -- [level][M + 1][8] (negated) then [M + 1][4] -- for pcr_solve_transposed: K^-1 = Dfin^-1 P_last ... P_1 with P_k = I + (a, g of
// level k), hence K^-T = P_1^T ... P_last^T Dfin^-T, and the adjoint system K^T lambda = r needs no reduction of its own.
template <int DL, class LG, typename Num>
__device__ __forceinline__ void pcr_solve(int M, Num (&L)[2][2], Num (&Dg)[2][2], Num (&U)[2][2], Num (&R)[2][DL],
                                          Num (&y)[2][DL], Num *mult = nullptr, Num *xch = nullptr) {
  static_assert(LG::W == kWave, "written for lane groups that span the wavefront");
  const int p = LG::piece();
  const bool in = p >= 1 && p < M;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      L[i][j] = in ? L[i][j] : Num(0.0);
      U[i][j] = in ? U[i][j] : Num(0.0);
      Dg[i][j] = in ? Dg[i][j] : Num(i == j ? 1.0 : 0.0);
    }
#pragma unroll
    for (int d = 0; d < DL; ++d) R[i][d] = in ? R[i][d] : Num(0.0);
  }
  const int lane = lane_id();
  // fp32: at most five levels.  The coupling left after a level is the square of the one before (measured over random
  // durations in [0.1, 5] s: <= 19, 4.8, 0.14, 8e-5, 6e-11 relative to the diagonal after levels 1 .. 5), so what a
  // sixth level (M > 33: cfg5) would eliminate is five orders below the rounding of the fp32 solve.
  const int s_end = sizeof(Num) == 4 ? min(M - 1, 32) : M - 1;
  typedef Num Quad __attribute__((ext_vector_type(4)));
  const bool writer = mult != nullptr && in && LG::dim0() == opaque_uniform(0);  // one lane per joint writes
  int level = 0;
  if constexpr (sizeof(Num) == 4) {
    // fp32: the 2 x 2 blocks as ROWS in register pairs, their products as packed operations (v_pk_mul_f32 / v_pk_fma_f32
    // with the left factor's element broadcast through op_sel): 4 instructions a 2 x 2 product instead of 8 -- 39 vector
    // instructions a level instead of 63, in a kernel bound by the issue of its vector instructions.  Every product is
    // formed by the same multiply / fused multiply-adds in the same order as the scalar form below: the same bits.
    typedef float f2 __attribute__((ext_vector_type(2)));
    auto fma2 = [](f2 x, f2 y, f2 z) { return __builtin_elementwise_fma(x, y, z); };
    f2 Lr[2] = {{L[0][0], L[0][1]}, {L[1][0], L[1][1]}}, Dr[2] = {{Dg[0][0], Dg[0][1]}, {Dg[1][0], Dg[1][1]}};
    f2 Ur[2] = {{U[0][0], U[0][1]}, {U[1][0], U[1][1]}}, Ir[2];
    auto invert = [&]() {
      const float det = fmaf(Dr[0].x, Dr[1].y, -(Dr[0].y * Dr[1].x));
      const float r = precise_rcp(det);
      Ir[0].x = Dr[1].y * r;
      Ir[0].y = -Dr[0].y * r;
      Ir[1].x = -Dr[1].x * r;
      Ir[1].y = Dr[0].x * r;
    };
    auto fetch = [&](const f2 v, int src) { return f2{__shfl(v.x, src, kWave), __shfl(v.y, src, kWave)}; };
    // Lanes without a neighbour at distance s fetch from whatever lane the index wraps to: their coupling block is an exact
    // zero by then (L_i = 0 for i <= s, U_i = 0 for i + s >= M, by induction over the levels), and zero times a finite
    // value is the zero the clamped fetch gave.  The last level's L' and U' (zero up to rounding, never used) are formed
    // like the others: cheaper than the selects that kept them out.  Every lane stores its multipliers -- the lanes of a
    // joint the same values to the same place, the lanes without a joint into slots nobody reads: piece 0, and slot M, which
    // every level and the table of pivot inverses have for themselves (M + 1 slots each: pcr_mult_elems).
    const bool keep = mult != nullptr;
    const int ps = min(p, M);  // (store slot: every lane beyond the last piece shares the one past it)
    for (int s = 1; s < s_end; s <<= 1, ++level) {
      invert();
      const int sp = lane - s * LG::S, sn = lane + s * LG::S;
      f2 Ip[2], Lp[2], Upv[2], In[2], Ln[2], Un[2];
      Num Rp[2][DL], Rn[2][DL];
      if (xch != nullptr) {
        // The neighbours' blocks through an LDS table instead of 24 + 4 DL ds_bpermute: every piece writes its I, L, U as
        // three 16-byte stores (the lanes of a piece the same values), every lane reads the two neighbouring pieces'
        // with six 16-byte loads; the right-hand sides travel as 8-byte pairs per lane.  Measured on the MI355X
        // (tools/probe/valu_rate.hip): a ds_bpermute_b32 occupies the CU's LDS for as long as three quarters of a
        // 16-byte access that moves four values -- and with four launches in flight the LDS pipe is what this kernel
        // waits for (SQ_LDS_IDX_ACTIVE per evaluation x evaluations in flight ~ the CU's cycles).  Pieces without a
        // neighbour at distance s read piece 0 / M - 1 instead: finite values times their exact-zero coupling block.
        Quad *tab = reinterpret_cast<Quad *>(xch);
        f2 *rx = reinterpret_cast<f2 *>(xch + (size_t)(M + 1) * 12);
        lds_wave_sync();
        tab[ps * 3 + 0] = Quad{Ir[0].x, Ir[0].y, Ir[1].x, Ir[1].y};
        tab[ps * 3 + 1] = Quad{Lr[0].x, Lr[0].y, Lr[1].x, Lr[1].y};
        tab[ps * 3 + 2] = Quad{Ur[0].x, Ur[0].y, Ur[1].x, Ur[1].y};
#pragma unroll
        for (int d = 0; d < DL; ++d) rx[lane * DL + d] = f2{R[0][d], R[1][d]};
        lds_wave_sync();
        // (both clamped into 0 .. M - 1, the slots that are always written -- also for the lanes beyond the last piece,
        //  whose own values must stay finite: wrapped right-hand-side reads of real joints land on them)
        const int pp = min(max(p - s, 0), M - 1), pn = min(p + s, M - 1);
        const Quad qi = tab[pp * 3 + 0], ql = tab[pp * 3 + 1], qu = tab[pp * 3 + 2];
        const Quad ni = tab[pn * 3 + 0], nl = tab[pn * 3 + 1], nu = tab[pn * 3 + 2];
        Ip[0] = qi.xy; Ip[1] = qi.zw; Lp[0] = ql.xy; Lp[1] = ql.zw; Upv[0] = qu.xy; Upv[1] = qu.zw;
        In[0] = ni.xy; In[1] = ni.zw; Ln[0] = nl.xy; Ln[1] = nl.zw; Un[0] = nu.xy; Un[1] = nu.zw;
#pragma unroll
        for (int d = 0; d < DL; ++d) {
          const f2 rp = rx[(sp & (kWave - 1)) * DL + d], rn = rx[(sn & (kWave - 1)) * DL + d];
          Rp[0][d] = rp.x; Rp[1][d] = rp.y;
          Rn[0][d] = rn.x; Rn[1][d] = rn.y;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          Ip[i] = fetch(Ir[i], sp);
          Upv[i] = fetch(Ur[i], sp);
          In[i] = fetch(Ir[i], sn);
          Ln[i] = fetch(Lr[i], sn);
          Lp[i] = fetch(Lr[i], sp);
          Un[i] = fetch(Ur[i], sn);
#pragma unroll
          for (int d = 0; d < DL; ++d) {
            Rp[i][d] = __shfl(R[i][d], sp, kWave);
            Rn[i][d] = __shfl(R[i][d], sn, kWave);
          }
        }
      }
      f2 pa[2], pg[2];  // -a, -g: the products as they come (a = -L Ip, g = -U In)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        pa[i] = fma2(Lr[i].xx, Ip[0], Lr[i].yy * Ip[1]);
        pg[i] = fma2(Ur[i].xx, In[0], Ur[i].yy * In[1]);
      }
      if (keep) {
        Quad *dst = reinterpret_cast<Quad *>(mult + ((size_t)level * (M + 1) + ps) * 8);
        dst[0] = Quad{pa[0].x, pa[0].y, pa[1].x, pa[1].y};  // (stored negated: what sits in the registers)
        dst[1] = Quad{pg[0].x, pg[0].y, pg[1].x, pg[1].y};
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const f2 ai = -pa[i], gi = -pg[i];
        const f2 ln2 = fma2(ai.xx, Lp[0], ai.yy * Lp[1]);
        const f2 un2 = fma2(gi.xx, Un[0], gi.yy * Un[1]);
        Dr[i] = fma2(gi.yy, Ln[1], fma2(gi.xx, Ln[0], fma2(ai.yy, Upv[1], fma2(ai.xx, Upv[0], Dr[i]))));
#pragma unroll
        for (int d = 0; d < DL; ++d)
          R[i][d] = fmaf(gi.y, Rn[1][d], fmaf(gi.x, Rn[0][d], fmaf(ai.y, Rp[1][d], fmaf(ai.x, Rp[0][d], R[i][d]))));
        Lr[i] = ln2;
        Ur[i] = un2;
      }
    }
    invert();
    if (xch != nullptr) lds_wave_sync();  // (the exchange table is the caller's staging buffer again)
    if (keep) *reinterpret_cast<Quad *>(mult + (size_t)level * (M + 1) * 8 + (size_t)ps * 4) = Quad{Ir[0].x, Ir[0].y, Ir[1].x, Ir[1].y};
#pragma unroll
    for (int d = 0; d < DL; ++d) {
      y[0][d] = fmaf(Ir[0].x, R[0][d], Ir[0].y * R[1][d]);
      y[1][d] = fmaf(Ir[1].x, R[0][d], Ir[1].y * R[1][d]);
    }
    return;
  }
  Num I[2][2];
  auto invert = [&]() {
    const Num det = fma(Dg[0][0], Dg[1][1], -(Dg[0][1] * Dg[1][0]));
    const Num r = precise_rcp(det);
    I[0][0] = Dg[1][1] * r;
    I[0][1] = -Dg[0][1] * r;
    I[1][0] = -Dg[1][0] * r;
    I[1][1] = Dg[0][0] * r;
  };
  for (int s = 1; s < s_end; s <<= 1, ++level) {
    invert();
    // neighbours at distance s (lanes without one read themselves: their coupling block is zero by then)
    const int lp = lane - s * LG::S, ln = lane + s * LG::S;
    const int sp = lp >= 0 ? lp : lane, sn = ln < kWave ? ln : lane;
    // (the last level leaves uncoupled equations: its L' and U' -- zero -- are not formed, nor their inputs fetched)
    const bool last = 2 * s >= s_end;
    Num Ip[2][2], Lp[2][2], Upv[2][2], Rp[2][DL], In[2][2], Ln[2][2], Un[2][2], Rn[2][DL];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        Ip[i][j] = __shfl(I[i][j], sp, kWave);
        Upv[i][j] = __shfl(U[i][j], sp, kWave);
        In[i][j] = __shfl(I[i][j], sn, kWave);
        Ln[i][j] = __shfl(L[i][j], sn, kWave);
        Lp[i][j] = Un[i][j] = Num(0.0);
        if (!last) {
          Lp[i][j] = __shfl(L[i][j], sp, kWave);
          Un[i][j] = __shfl(U[i][j], sn, kWave);
        }
      }
#pragma unroll
      for (int d = 0; d < DL; ++d) {
        Rp[i][d] = __shfl(R[i][d], sp, kWave);
        Rn[i][d] = __shfl(R[i][d], sn, kWave);
      }
    }
    Num a[2][2], g[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        a[i][j] = -fma(L[i][0], Ip[0][j], L[i][1] * Ip[1][j]);
        g[i][j] = -fma(U[i][0], In[0][j], U[i][1] * In[1][j]);
      }
    if (writer) {
      Quad *dst = reinterpret_cast<Quad *>(mult + ((size_t)level * (M + 1) + p) * 8);
      // (stored NEGATED: a and g are formed as -(...), and the products themselves are what sits in registers)
      dst[0] = Quad{-a[0][0], -a[0][1], -a[1][0], -a[1][1]};
      dst[1] = Quad{-g[0][0], -g[0][1], -g[1][0], -g[1][1]};
    }
    Num Ln2[2][2], Un2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        Ln2[i][j] = Un2[i][j] = Num(0.0);
        if (!last) {
          Ln2[i][j] = fma(a[i][0], Lp[0][j], a[i][1] * Lp[1][j]);
          Un2[i][j] = fma(g[i][0], Un[0][j], g[i][1] * Un[1][j]);
        }
        Dg[i][j] = fma(g[i][1], Ln[1][j], fma(g[i][0], Ln[0][j], fma(a[i][1], Upv[1][j], fma(a[i][0], Upv[0][j], Dg[i][j]))));
      }
#pragma unroll
      for (int d = 0; d < DL; ++d)
        R[i][d] = fma(g[i][1], Rn[1][d], fma(g[i][0], Rn[0][d], fma(a[i][1], Rp[1][d], fma(a[i][0], Rp[0][d], R[i][d]))));
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        L[i][j] = Ln2[i][j];
        U[i][j] = Un2[i][j];
      }
  }
  invert();
  if (writer) *reinterpret_cast<Quad *>(mult + (size_t)level * (M + 1) * 8 + (size_t)p * 4) = Quad{I[0][0], I[0][1], I[1][0], I[1][1]};
#pragma unroll
  for (int d = 0; d < DL; ++d) {
    y[0][d] = fma(I[0][0], R[0][d], I[0][1] * R[1][d]);
    y[1][d] = fma(I[1][0], R[0][d], I[1][1] * R[1][d]);
  }
}

// levels pcr_solve runs for M pieces (the loop above), and the LDS floats its multipliers take
__host__ __device__ __forceinline__ int pcr_levels(int M, bool f32 = true) {
  const int s_end = f32 ? (M - 1 < 32 ? M - 1 : 32) : M - 1;
  int n = 0;
  for (int s = 1; s < s_end; s <<= 1) ++n;
  return n;
}
// ([level][M + 1][8] multipliers, then [M + 1][4] pivot inverses: slot M of every region takes the stores of the lanes
//  beyond the last piece, so no region's stores land in another one)
// floats of LDS the exchange table of pcr_solve takes: I, L, U of pieces 0 .. M, then one pair per lane and held dimension
__host__ __device__ __forceinline__ int pcr_xch_elems(int M, int DL) { return (M + 1) * 12 + kWave * 2 * DL; }
__host__ __device__ __forceinline__ int pcr_mult_elems(int M) { return (pcr_levels(M) * 8 + 4) * (M + 1); }

// The adjoint system K^T lambda = r from the multipliers pcr_solve left in `mult`:
//     lambda = P_1^T ... P_last^T (Dfin^-T r),     (P_k^T w)_j = w_j + a_{j+s}^T w_{j+s} + g_{j-s}^T w_{j-s},   s = 2^k.
// A level is two 16-byte LDS reads, eight multiply-adds and 4 DL neighbour fetches -- against 63 vector instructions and
// 28 fetches for a level of the reduction itself: the adjoint pass no longer pays for a second reduction of a matrix the
// forward pass has already reduced (the joint matrix depends on the durations only, and those do not change between the
// two passes of an evaluation).
template <int DL, class LG, typename Num>
__device__ __forceinline__ void pcr_solve_transposed(int M, const Num *mult, const Num (&R)[2][DL], Num (&y)[2][DL]) {
  static_assert(LG::W == kWave, "written for lane groups that span the wavefront");
  typedef Num Quad __attribute__((ext_vector_type(4)));
  const int p = LG::piece();
  const bool in = p >= 1 && p < M;
  const int pj = in ? p : 1;  // (a valid address for the lanes that hold no joint; their values are masked)
  const int lane = lane_id();
  const int nlev = pcr_levels(M, sizeof(Num) == 4);
  Num w[2][DL];
  {
    const Quad I = *reinterpret_cast<const Quad *>(mult + (size_t)nlev * (M + 1) * 8 + (size_t)pj * 4);
#pragma unroll
    for (int d = 0; d < DL; ++d) {
      w[0][d] = in ? fma(I.x, R[0][d], I.z * R[1][d]) : Num(0.0);  // Dfin^-T r
      w[1][d] = in ? fma(I.y, R[0][d], I.w * R[1][d]) : Num(0.0);
    }
  }
  for (int k = nlev - 1; k >= 0; --k) {
    const int s = 1 << k;
    const Quad *src = reinterpret_cast<const Quad *>(mult + ((size_t)k * (M + 1) + pj) * 8);
    const Quad a = src[0], g = src[1];
    const int ln = lane + s * LG::S, lp = lane - s * LG::S;
    const bool hn = ln < kWave, hp = lp >= 0;
#pragma unroll
    for (int d = 0; d < DL; ++d) {
      // a^T w and g^T w of this joint (zero on lanes without a joint: their w is zero)
      const Num pa0 = fma(a.x, w[0][d], a.z * w[1][d]), pa1 = fma(a.y, w[0][d], a.w * w[1][d]);
      const Num qg0 = fma(g.x, w[0][d], g.z * w[1][d]), qg1 = fma(g.y, w[0][d], g.w * w[1][d]);
      const Num n0 = __shfl(pa0, hn ? ln : lane, kWave), n1 = __shfl(pa1, hn ? ln : lane, kWave);
      const Num m0 = __shfl(qg0, hp ? lp : lane, kWave), m1 = __shfl(qg1, hp ? lp : lane, kWave);
      if (in) {  // (minus: the multipliers are stored negated)
        w[0][d] -= (hn ? n0 : Num(0.0)) + (hp ? m0 : Num(0.0));
        w[1][d] -= (hn ? n1 : Num(0.0)) + (hp ? m1 : Num(0.0));
      }
    }
  }
#pragma unroll
  for (int d = 0; d < DL; ++d) {
    y[0][d] = w[0][d];
    y[1][d] = w[1][d];
  }
}

// joint-system blocks of lane p (joint p between piece p-1 "a" and piece p "b")
// (a1..a3 = T^-1..T^-3 of piece p-1, fetched from the neighbour lane by the caller)
template <class TrajT, typename Num>
__device__ __forceinline__ void joint_blocks(const TrajT &t, Num a1, Num a2, Num a3,
                                             Num (&Lo)[2][2], Num (&Di)[2][2], Num (&Up)[2][2]) {
  Lo[0][0] = -Num(24.0) * a2;  Lo[0][1] = -Num(3.0) * a1;
  Lo[1][0] = -Num(168.0) * a3; Lo[1][1] = -Num(24.0) * a2;
  Di[0][0] = -Num(36.0) * a2 + Num(36.0) * t.i2;    Di[0][1] = Num(9.0) * a1 + Num(9.0) * t.i1;
  Di[1][0] = -Num(192.0) * a3 - Num(192.0) * t.i3;  Di[1][1] = Num(36.0) * a2 - Num(36.0) * t.i2;
  Up[0][0] = Num(24.0) * t.i2;   Up[0][1] = -Num(3.0) * t.i1;
  Up[1][0] = -Num(168.0) * t.i3; Up[1][1] = Num(24.0) * t.i2;
}

// forward pass.  Inputs (PIECE layout): t.tau, t.P0, t.P1 set by the caller, head/tail uniform.
// Returns 0 or NUMERIC_RANGE (4) when exp(-tau) overflows like math.exp does (:481).
// PCR: solve the joint system by parallel cyclic reduction (pcr_solve) instead of block Thomas -- the all-fp32 mode on a
// wavefront-wide lane group; fp64 solves keep block Thomas (the parity mode's recorded runs are pinned to that rounding,
// and the mixed mode measured slower with the reduction in fp64).  The caller passes the same value to minco_backward.
template <int D, class LG = WaveLanes, typename Num = double, bool PCR = (sizeof(Num) == 4 && LG::W == kWave)>
__device__ __forceinline__ int minco_forward(Traj<D, LG::dl(D), Num> &t, const DevParams &prm, double &energy,
                                             double &time_sum) {
  constexpr int DL = LG::dl(D);
  const int lane = LG::piece();
  const bool act = lane < t.M;
  int bad = 0;
  // map_tau2T (:477-483)
  {
    const Num tau = act ? t.tau : Num(0.0);
    if (-tau > Num(709.782712893384)) bad = 1;
    // (fp32: expf overflows to +inf beyond 88.72 -- T is then T_min exactly, as it is in fp64 to rounding from -tau = 37
    //  on; the statuses stay those of the fp64 modes because the range tests are made on tau, not on exp(-tau))
    Num ex;
    if constexpr (sizeof(Num) == 8 || LG::W != kWave) {
      ex = exp(-tau);
      t.T = par_T_span<Num>(prm) / (Num(1.0) + ex) + par_T_min<Num>(prm);
    } else {
      // fp32: v_exp_f32 and v_rcp_f32 (2 + 1 instructions) instead of expf and a correctly rounded division (~30): a
      // few units in the last place of T, the level the fp32 solve works at anyway
      ex = __expf(-tau);
      t.T = fmaf(par_T_span<Num>(prm), precise_rcp(Num(1.0) + ex), par_T_min<Num>(prm));
    }
    // get_grad_T2tau needs exp(-tau) again (:490): fp64 keeps it instead of tau; fp32 keeps -tau (exp(-tau) may be inf)
    if constexpr (sizeof(Num) == 8) t.tau = ex; else t.tau = -tau;
  }
  if (LG::any(bad)) return 4;
  t.i1 = (sizeof(Num) == 8 || LG::W != kWave) ? Num(1.0) / t.T : precise_rcp(t.T);
  t.i2 = t.i1 * t.i1;
  t.i3 = t.i2 * t.i1;
  t.i4 = t.i2 * t.i2;
  // int(T / delta_t) (:401); fp32: times the reciprocal formed in fp64 (10.0f exactly for delta_t = 0.1) instead of a
  // correctly rounded fp32 division by 0.1f -- which is not 0.1 either
  if constexpr (sizeof(Num) == 8 || LG::W != kWave)
    t.ns = act ? (int)(t.T / par_dt<Num>(prm)) : 0;
  else
    t.ns = act ? (int)(t.T * par_inv_dt<Num>(prm)) : 0;

  if (t.M > 1) {
    Num Lo[2][2], Di[2][2], Up[2][2], E[2][2], R[2][DL], y0[2][DL], yM[2][DL], y[2][DL];
    const Num a1 = LG::prev(t.i1, Num(1.0)), a2 = a1 * a1, a3 = a2 * a1, a4 = a2 * a2;
    joint_blocks(t, a1, a2, a3, Lo, Di, Up);
    constexpr bool kPcr = PCR;  // fp32-sampling modes: parallel cyclic reduction (pcr_solve)
    if constexpr (!kPcr) thomas_factor<LG, Num>(NEO_DBG(prm, 2 | 8) ? 1 : t.M, Lo, Di, Up, t.N, E);
#pragma unroll
    for (int d = 0; d < DL; ++d) {
      // displacement of piece p-1 and of piece p
      const Num dPb = t.P1[d] - t.P0[d];
      const Num dPa = LG::prev(dPb, Num(0.0));
      R[0][d] = -(Num(60.0) * a3 * dPa - Num(60.0) * t.i3 * dPb);
      R[1][d] = -(Num(360.0) * a4 * dPa + Num(360.0) * t.i4 * dPb);
      y0[0][d] = bstate<D, LG, false>(t, 1, d);
      y0[1][d] = bstate<D, LG, false>(t, 2, d);
      yM[0][d] = bstate<D, LG, true>(t, 1, d);
      yM[1][d] = bstate<D, LG, true>(t, 2, d);
    }
    if constexpr (kPcr) {
      // the boundary states move to the right-hand sides of the first and the last joint
#pragma unroll
      for (int d = 0; d < DL; ++d) {
        if (lane == 1) {
          R[0][d] -= Lo[0][0] * y0[0][d] + Lo[0][1] * y0[1][d];
          R[1][d] -= Lo[1][0] * y0[0][d] + Lo[1][1] * y0[1][d];
        }
        if (lane == t.M - 1) {
          R[0][d] -= Up[0][0] * yM[0][d] + Up[0][1] * yM[1][d];
          R[1][d] -= Up[1][0] * yM[0][d] + Up[1][1] * yM[1][d];
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          Lo[i][j] = lane == 1 ? Num(0.0) : Lo[i][j];
          Up[i][j] = lane == t.M - 1 ? Num(0.0) : Up[i][j];
        }
      NEO_MARK("fwd_pcr_begin");
      pcr_solve<DL, LG, Num>(t.M, Lo, Di, Up, R, y, t.pcr_mult, t.pcr_xch);
      NEO_MARK("fwd_pcr_end");
    } else {
      thomas_solve<DL, LG, Num>(NEO_DBG(prm, 2 | 16) ? 1 : t.M, Lo, t.N, E, R, y0, yM, y);
    }
#pragma unroll
    for (int d = 0; d < DL; ++d) {
      t.V0[d] = lane == 0 ? bstate<D, LG, false>(t, 1, d) : y[0][d];
      t.A0[d] = lane == 0 ? bstate<D, LG, false>(t, 2, d) : y[1][d];
    }
  } else {
#pragma unroll
    for (int d = 0; d < DL; ++d) {
      t.V0[d] = bstate<D, LG, false>(t, 1, d);
      t.A0[d] = bstate<D, LG, false>(t, 2, d);
    }
  }
#pragma unroll
  for (int d = 0; d < DL; ++d) {
    const Num v1 = LG::next(t.V0[d], Num(0.0)), a1 = LG::next(t.A0[d], Num(0.0));
    t.V1[d] = (lane == t.M - 1) ? bstate<D, LG, true>(t, 1, d) : v1;
    t.A1[d] = (lane == t.M - 1) ? bstate<D, LG, true>(t, 2, d) : a1;
  }
  // Hermite form of the quintic
  Num e = Num(0.0);
  const Num T = t.T, T2 = T * T, T3 = T2 * T, T4 = T2 * T2, T5 = T4 * T;
#pragma unroll
  for (int d = 0; d < DL; ++d) {
    const Num ep = t.P1[d] - t.P0[d] - T * t.V0[d] - Num(0.5) * T2 * t.A0[d];
    const Num ev = t.V1[d] - t.V0[d] - T * t.A0[d];
    const Num ea = t.A1[d] - t.A0[d];
    t.c[0][d] = t.P0[d];
    t.c[1][d] = t.V0[d];
    t.c[2][d] = Num(0.5) * t.A0[d];
    t.c[3][d] = (Num(10.0) * ep - Num(4.0) * T * ev + Num(0.5) * T2 * ea) * t.i3;
    t.c[4][d] = (-Num(15.0) * ep + Num(7.0) * T * ev - T2 * ea) * t.i4;
    t.c[5][d] = (Num(6.0) * ep - Num(3.0) * T * ev + Num(0.5) * T2 * ea) * t.i4 * t.i1;
    // add_energy_cost (:345-359): c^T Q(T) c with the closed-form jerk Gram matrix
    const Num c3 = t.c[3][d], c4 = t.c[4][d], c5 = t.c[5][d];
    e += Num(36.0) * T * c3 * c3 + Num(144.0) * T2 * c3 * c4 + Num(240.0) * T3 * c3 * c5 + Num(192.0) * T3 * c4 * c4 +
         Num(720.0) * T4 * c4 * c5 + Num(720.0) * T5 * c5 * c5;
  }
  energy = LG::sum(act ? e : Num(0.0));
  time_sum = LG::sum((act && LG::dim0() == opaque_uniform(0)) ? t.T : Num(0.0));  // add_time_cost (:386-387): once per piece
  return 0;
}

// position and velocity of a piece at local time s.  fp64 follows the reference's monomial sums.  fp32 runs
// Horner on p and p' together: 9 dependent fma per axis, but no pre-scaled copies k*c_k of the coefficients in
// registers -- that is what lets the stand-alone kernel fit 128 VGPRs and the two-waves optimiser variant spill a
// third as much (2 % slower than two independent chains where registers are plentiful).  Every fp32 kernel uses
// this one form, so the sampled terms of eval / optimize / sample kernels agree bit for bit.
template <typename Real, int D>
__device__ __forceinline__ void piece_pos_vel(const Real (&c)[6][D], Real s, Real (&pos)[D], Real (&vel)[D]) {
#pragma clang fp contract(on)  // fuse a*b+c only as written: the same arithmetic whatever the unrolling around it
#pragma unroll
  for (int d = 0; d < D; ++d) {
    if constexpr (sizeof(Real) == 4) {
      Real a = c[5][d], b = c[5][d];
      a = fmaf(a, s, c[4][d]); b = fmaf(b, s, a);
      a = fmaf(a, s, c[3][d]); b = fmaf(b, s, a);
      a = fmaf(a, s, c[2][d]); b = fmaf(b, s, a);
      a = fmaf(a, s, c[1][d]); b = fmaf(b, s, a);
      a = fmaf(a, s, c[0][d]);
      pos[d] = a;
      vel[d] = b;
    } else {
      pos[d] = c[0][d] + s * (c[1][d] + s * (c[2][d] + s * (c[3][d] + s * (c[4][d] + s * c[5][d]))));
      vel[d] = c[1][d] + s * (Real(2) * c[2][d] + s * (Real(3) * c[3][d] + s * (Real(4) * c[4][d] + s * (Real(5) * c[5][d]))));
    }
  }
}

// lanes per piece in the SAMPLE layout: floor(64 / M), rounded down to a power of two from 8 up -- the piece's
// lanes then form an aligned group inside a 16-lane DPP row (or whole rows) and fold with DPP shifts instead of
// LDS shuffles (M = 3: 16 lanes per piece instead of 21, still two rounds of samples, 550 fewer instructions)
__host__ __device__ __forceinline__ int sample_lanes_per_piece(int M) {
  int L = kWave / M;
  if (L < 1) L = 1;
  if (L >= 8) L = L >= 64 ? 64 : (L >= 32 ? 32 : (L >= 16 ? 16 : 8));
  return L;
}

__host__ __device__ __forceinline__ int WaveLanes::lanes_per_piece(int M) { return sample_lanes_per_piece(M); }
__host__ __device__ __forceinline__ int sample_lanes_per_piece_fwd(int M) { return sample_lanes_per_piece(M); }

template <typename Real, int CTRL>
__device__ __forceinline__ Real dpp_real(Real v) {
  if constexpr (sizeof(Real) == 4)
    return dpp_f<CTRL>(v);
  else
    return dpp_d<CTRL>(v);
}

// sum over the L lanes (residues r = 0..L-1, adjacent lanes) of each piece; valid in the lane with r = 0.
// L <= 4 (M >= 16): wave_shl:1 DPP moves, summed left to right ((v_0 + v_1) + v_2) + v_3.
// L = 8, 16, 32, 64: halving tree with row_shl DPP moves (lane i <- lane i + s inside its row of 16), rows joined
// through LDS shuffles.  Other L: a shuffle tree.  Fixed order in every case.
// All N values of a lane are folded together so that the case distinction on L (wave-uniform) is made ONCE: as 21
// separate calls it cost 21 branch ladders per evaluation (and 1250 instructions of code in the stand-alone kernel).
template <typename Real, int N>
__device__ __forceinline__ void fold_piece_lanes(Real (&v)[N], int L, int r) {
  if (L <= 4) {
    Real t[N];
#pragma unroll
    for (int q = 0; q < N; ++q) t[q] = v[q];
    for (int i = 1; i < L; ++i) {
#pragma unroll
      for (int q = 0; q < N; ++q) {
        t[q] = dpp_real<Real, 0x130>(t[q]);
        v[q] += t[q];
      }
    }
    return;
  }
  if ((L & (L - 1)) == 0) {
    if (L >= 16) {
#pragma unroll
      for (int q = 0; q < N; ++q) v[q] += dpp_real<Real, 0x108>(v[q]);  // row_shl:8
    }
#pragma unroll
    for (int q = 0; q < N; ++q) {
      v[q] += dpp_real<Real, 0x104>(v[q]);  // row_shl:4
      v[q] += dpp_real<Real, 0x102>(v[q]);
      v[q] += dpp_real<Real, 0x101>(v[q]);
    }
    if (L >= 32) {
#pragma unroll
      for (int q = 0; q < N; ++q) v[q] += __shfl_down(v[q], 16, kWave);
    }
    if (L >= 64) {
#pragma unroll
      for (int q = 0; q < N; ++q) v[q] += __shfl_down(v[q], 32, kWave);
    }
    return;
  }
  for (int sft = 1; sft < L; sft <<= 1) {
#pragma unroll
    for (int q = 0; q < N; ++q) {
      const Real o = __shfl_down(v[q], sft, kWave);
      if (r + sft < L) v[q] += o;
    }
  }
}

// ---- SAMPLE layout: which lanes walk which piece's quadrature samples
struct SampleLanes {
  int piece;   // piece of this sample lane (valid when act)
  int r;       // its residue: it walks samples j = r, r + L, r + 2L, ...
  int L;       // lanes of its piece
  bool act;
  int rounds;  // wave-uniform: rounds of the sample loop
  int lmax;    // wave-uniform: largest L (fold depth); 0 = every piece has the same L (fast DPP folds)
  int first;   // PIECE layout (lane p < M): first sample lane of piece p
  int Lp;      // PIECE layout (lane p < M): number of sample lanes of piece p
  int total;   // wave-uniform: samples of the whole trajectory (balanced_sample_lanes; -1: not formed)
};

// the same L = sample_lanes_per_piece(M) lanes for every piece (lane groups; pieces with equal sample counts)
template <class LG>
__device__ __forceinline__ SampleLanes fixed_sample_lanes(int M, int L, int ns_of_my_piece_if_known = -1) {
  (void)ns_of_my_piece_if_known;
  const int lane = LG::lane();
  SampleLanes sl;
  sl.piece = (lane * ((65536 + L - 1) / L)) >> 16;  // lane / L for lane < 64
  sl.r = lane - sl.piece * L;
  sl.L = L;
  sl.act = sl.piece < M;
  sl.rounds = -1;  // from the sample counts, in minco_sample
  sl.lmax = 0;
  sl.first = lane * L;
  sl.Lp = L;
  sl.total = -1;
  return sl;
}

// Lane groups: the W - M * (W / M) lanes a fixed split leaves idle go to the pieces with the most samples, one each.
// cfg3 (M = 3 in eight lanes): the first and the last piece of a fresh guess last 1.5 times as long as the middle one --
// 25, 17, 25 samples: 13 rounds with two lanes each, 9 with 3 + 2 + 3.  ns_piece: PIECE layout (group lane p < M).
template <class LG>
__device__ __forceinline__ SampleLanes group_sample_lanes(int M, int ns_piece) {
  constexpr int W = LG::W;
  const int L0 = W / M, extra = W - M * L0;  // (M <= W)
  if (extra == 0) return fixed_sample_lanes<LG>(M, L0);
  const int lane = LG::lane();
  const int base = LG::base();
  const int mine = lane < M ? ns_piece : -1;
  int rank = 0;  // pieces with more samples than mine (ties: the lower index first)
  for (int q = 0; q < M; ++q) {
    const int o = __shfl(mine, base + q, kWave);
    rank += (o > mine || (o == mine && q < lane)) ? 1 : 0;
  }
  const int Lp = lane < M ? L0 + (rank < extra ? 1 : 0) : 0;
  SampleLanes sl;
  sl.piece = 0;
  sl.r = 0;
  sl.L = L0;
  sl.first = 0;
  int acc = 0;  // first sample lane of piece q
  for (int q = 0; q < M; ++q) {
    const int lq = __shfl(Lp, base + q, kWave);
    if (q == lane) sl.first = acc;
    if (lane >= acc && lane < acc + lq) {
      sl.piece = q;
      sl.r = lane - acc;
      sl.L = lq;
    }
    acc += lq;
  }
  sl.act = lane < acc;
  sl.rounds = -1;  // from the sample counts, in minco_sample
  sl.lmax = L0 + 1;
  sl.Lp = Lp;
  sl.total = -1;
  return sl;
}

// Lanes in proportion to the pieces' sample counts (whole wavefront, one trajectory): the smallest number of rounds
// R for which sum_p ceil(ns_p / R) lanes fit the wavefront, piece p then gets ceil(ns_p / R) adjacent lanes.  Durations
// are optimisation variables, so the pieces of a trajectory under optimisation differ widely in length: with
// the same three lanes for every piece the longest piece sets the rounds (cfg2: 8.6 on average against 4.9 for a
// perfect split).  ns_piece: PIECE layout (lane p < M).  seg: 64 ints of LDS scratch.
__device__ __forceinline__ SampleLanes balanced_sample_lanes(int M, int ns_piece, int *seg) {
  const int lane = lane_id();
  // `seg` is the caller's staging buffer seen as ints; the caller has just READ it as its own type (scatter_x: the
  // lane's waypoints).  Those loads must stay ahead of the stores below, and type-based alias analysis says an int store
  // and a double load cannot touch the same memory: with -sink-insts-to-avoid-spills the compiler moved the loads of
  // P0 / P1 below `seg[lane] = 0` in one instantiation (linear fp16 field, fp64, three FLAT slots) and the optimiser ran
  // on zeros (round 5; the library is built with -fno-strict-aliasing since, and the hand-over is fenced here).
  lds_wave_sync();
  const int mine = lane < M ? ns_piece : 0;
  const int total = wave_sum(mine);
  int R = max(1, (total + kWave - 1) / kWave);
  int Lp = 0;
  for (;;) {
    // ceil(ns / R) = floor((ns + R - 1/2) / R): the half keeps the quotient off the integers, so that neither the rounding
    // of the product nor the last bit of the reciprocal can tip it (exact for ns + R < 2^20)
    const float fr = (float)R;
    Lp = mine > 0 ? (int)(((float)mine + fr - 0.5f) * __builtin_amdgcn_rcpf(fr)) : 0;  // (v_rcp_f32: 1 ulp is plenty here)
    if (wave_sum(Lp) <= kWave) break;
    ++R;
  }
  const int incl = wave_scan_add(Lp);
  const int start = incl - Lp;
  seg[lane] = 0;
  lds_wave_sync();
  if (Lp > 0) seg[start] = ((lane + 1) << 16) | (Lp << 8) | start;
  lds_wave_sync();
  const int key = wave_scan_max_nonneg(seg[lane]);  // the segment this lane falls in: the last start at or before it
  lds_wave_sync();                                  // (seg is the caller's staging buffer again after this)
  SampleLanes sl;
  sl.piece = max((key >> 16) - 1, 0);
  sl.L = max((key >> 8) & 0xff, 1);
  sl.r = lane - (key & 0xff);
  sl.act = key != 0 && sl.r < sl.L;
  sl.rounds = R;
  sl.lmax = wave_max_nonneg(Lp);
  sl.first = start;
  sl.Lp = Lp;
  sl.total = total;
  return sl;
}

// sum over the lanes of each piece when the pieces have different numbers of lanes: suffix doubling with LDS
// shuffles, lane r of a piece adds the value sft lanes on while that lane still belongs to the piece.  Fixed order.
template <typename Real, int N>
__device__ __forceinline__ void fold_piece_segments(Real (&v)[N], int r, int L, int lmax) {
  for (int sft = 1; sft < lmax; sft <<= 1) {
#pragma unroll
    for (int q = 0; q < N; ++q) {
      const Real o = __shfl_down(v[q], sft, kWave);
      if (r + sft < L) v[q] += o;
    }
  }
}

// One quadrature sample's contribution to the two sampled cost terms and their partials (:404-466): j = sample index in
// its piece (of ns), s = its local time, vel = velocity there, vv = |vel|^2 - v_max^2, vd = safe_dis - d(pos); the map
// gradient is taken from the lookup only when the collision penalty is active.  One source for every sampling kernel
// (fused, stand-alone, workgroup-per-trajectory): the same arithmetic sample by sample.
template <typename Real, int D, class LookupT>
__device__ __forceinline__ void sample_accumulate(const Real (&c)[6][D], int j, int ns, Real s, Real inv_ns, const Real (&vel)[D],
                                                  Real vv, Real vd, const LookupT &lk, const typename LookupT::Addr &ad,
                                                  const typename LookupT::Raw &rw, Real dt, Real w2, Real w3, Real (&aC)[6][D],
                                                  Real &aT, Real &aF, Real &aK) {
#pragma clang fp contract(on)  // fuse a*b+c only as written: the same arithmetic whatever the unrolling around it
  const Real omg = (j == 0 || j == ns - 1) ? Real(0.5) : Real(1);
  const Real s2 = s * s, s3 = s2 * s, s4 = s2 * s2, s5 = s4 * s;
  // dynamic feasibility
  if (vv > Real(0)) {
    const Real vq = vv;
    aF += omg * dt * vq * vq * vq;
    Real av = Real(0);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const Real acc = Real(2) * c[2][d] + s * (Real(6) * c[3][d] + s * (Real(12) * c[4][d] + s * (Real(20) * c[5][d])));
      av += acc * vel[d];
    }
    const Real dK = Real(3) * dt * omg * vq * vq;
    const Real b1[6] = {Real(0), Real(1), Real(2) * s, Real(3) * s2, Real(4) * s3, Real(5) * s4};
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const Real uu = w2 * dK * Real(2) * vel[d];
#pragma unroll
      for (int k = 1; k < 6; ++k) aC[k][d] += b1[k] * uu;
    }
    aT += w2 * (omg * vq * vq * vq * inv_ns + dK * Real(2) * av * (Real)j * inv_ns);
  }
  // collision
  if (vd > Real(0)) {
    Real g[D];
    (void)lk.template finish<D>(ad, rw, g);
    const Real vq = vd;
    aK += omg * dt * vq * vq * vq;
    const Real dK = Real(3) * dt * omg * vq * vq;
    const Real b0[6] = {Real(1), s, s2, s3, s4, s5};
    Real gv = Real(0);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      gv += g[d] * vel[d];
      const Real uu = -(w3 * dK * g[d]);
#pragma unroll
      for (int k = 0; k < 6; ++k) aC[k][d] += b0[k] * uu;
    }
    aT += w3 * (omg * vq * vq * vq * inv_ns + dK * (-gv) * (Real)j * inv_ns);
  }
}

// floats per piece of the accumulator fold (minco_sample, fold_acc)
__host__ __device__ constexpr int fold_acc_stride(int D) { return (6 * D + 1 + 3) / 4 * 4; }

// sampled feasibility + collision terms (:392-466), SAMPLE layout.
// SAMPLE_IO = false (fused kernels): in (PIECE layout) cp = coefficients of the lane's piece, ns_in = its sample
//   count; out (PIECE layout) gC, gT = weighted partials of the two sampled terms.
// SAMPLE_IO = true (stand-alone kernel): cp / ns_in are already those of the SAMPLE-layout lane's piece, and
//   gC / gT are left in the first lane (r = 0) of each piece.
// Costs are returned wave-uniform.  U = samples per lane whose gathers are put in flight together.
template <typename Real, int D, class LookupT, int U, bool SAMPLE_IO = false, class LG = WaveLanes>
__device__ __forceinline__ void minco_sample(int M, const SampleLanes &sl, int ns_in, const Real (&cp)[6][LG::dl(D)],
                                             const DevParams &prm, const LookupT &lk, Real (&gC)[6][LG::dl(D)], Real &gT,
                                             double &cost_feas, double &cost_coll, Real *fold_rows = nullptr,
                                             bool fold_acc = false) {
#pragma clang fp contract(on)  // fuse a*b+c only as written: the same arithmetic whatever the unrolling around it
  const int lane = LG::lane();
  const int piece = sl.piece, r = sl.r, L = sl.L;  // (L: per lane when the pieces have different numbers of lanes)
  const bool act = sl.act;
  Real c[6][D];
  int ns;
  if constexpr (SAMPLE_IO) {
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int d = 0; d < D; ++d) c[k][d] = cp[k][d];
    ns = act ? ns_in : 0;
  } else {
    // hand the piece data to its L sample lanes (every sample lane needs all D dimensions of its piece)
    // (round 4: through a table in the staging buffer -- two stores, five 16-byte loads, three syncs -- instead of the 6 D + 1
    //  ds_bpermute, and the adjoint's four fetches a level the same way: measured slower, cfg2 1.413 -> 1.405 M traj/s and a
    //  single batch 705 -> 643 k: the write -> read round trips sit on the dependent chain)
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int d = 0; d < D; ++d) {
        if constexpr (LG::S == 1)
          c[k][d] = __shfl(cp[k][d], LG::base() + piece, kWave);
        else
          c[k][d] = __shfl(cp[k][0], LG::S * piece + d, kWave);
      }
    const int ns_sh = __shfl(ns_in, LG::base() + LG::S * piece, kWave);
    ns = act ? ns_sh : 0;
  }
  const int iters = NEO_DBG(prm, 1) ? 0 : (sl.rounds >= 0 ? sl.rounds : wave_max_nonneg((ns + L - 1) / L));

  const Real dt = par_dt<Real>(prm), vmax2 = par_vmax2<Real>(prm), safe = par_safe<Real>(prm);
  const Real w2 = par_w2<Real>(prm), w3 = par_w3<Real>(prm);
  // (fp32: v_rcp_f32 instead of the 11-instruction correctly rounded division; the fp64 parity mode divides)
  Real inv_ns;
  if constexpr (sizeof(Real) == 4 && LG::W == kWave)
    inv_ns = ns > 0 ? __builtin_amdgcn_rcpf((float)ns) : Real(0);
  else
    inv_ns = ns > 0 ? Real(1) / (Real)ns : Real(0);
  Real aC[6][D];
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int d = 0; d < D; ++d) aC[k][d] = Real(0);
  Real aT = Real(0), aF = Real(0), aK = Real(0);

  // U samples per lane are prepared together and their gathers issued back to back before any of
  // them is consumed
  NEO_MARK("loop_begin");
  for (int it0 = 0; it0 < iters; it0 += U) {
    Real sv[U], vel[U][D];
    typename LookupT::Addr ad[U];
    typename LookupT::Raw rw[U];
    bool on[U];
    // stand-alone kernel: lanes whose piece has run out of samples sit the round out (exec-masked)
    if constexpr (SAMPLE_IO)
      if (r + it0 * L >= ns) continue;
    // only the position is needed to issue the gathers; the velocity is evaluated while they fly
    // (round 4, measured again on the brick layout: velocity with the position before the gathers 27.8 -> 28.4 us per 4096
    //  launch; the sample time as an fp32 product instead of the rounded fp64 one: no change)
    constexpr bool kVelLate = SAMPLE_IO && sizeof(Real) == 4;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = r + (it0 + u) * L;
      on[u] = j < ns;
      const Real s = (Real)((double)j * prm.delta_t);  // beta_full row j: t = j * delta_t (:251)
      sv[u] = s;
      Real pos[D];
      if constexpr (kVelLate) {
#pragma unroll
        for (int d = 0; d < D; ++d)
          pos[d] = fmaf(fmaf(fmaf(fmaf(fmaf(c[5][d], s, c[4][d]), s, c[3][d]), s, c[2][d]), s, c[1][d]), s, c[0][d]);
      } else {
        piece_pos_vel<Real, D>(c, s, pos, vel[u]);
      }
      ad[u] = lk.template prepare<D>(pos, on[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) rw[u] = lk.load(ad[u]);
    if constexpr (kVelLate) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        Real pos[D];
        piece_pos_vel<Real, D>(c, sv[u], pos, vel[u]);
      }
    }
    // violations are rare: first only the two penalties' arguments for the U samples, one test for the
    // whole group, and the per-sample accumulation code only if some lane of the wave needs it
    Real vv[U], vd[U];
    bool any_violation = false;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      Real v2 = Real(0);
#pragma unroll
      for (int d = 0; d < D; ++d) v2 += vel[u][d] * vel[u][d];
      vv[u] = v2 - vmax2;
      Real gdrop[D];
      vd[u] = safe - lk.template finish<D>(ad[u], rw[u], gdrop);
      if (on[u] && (vv[u] > Real(0) || vd[u] > Real(0))) any_violation = true;
    }
    if (any_violation) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (!on[u]) continue;
        sample_accumulate<Real, D, LookupT>(c, r + (it0 + u) * L, ns, sv[u], inv_ns, vel[u], vv[u], vd[u], lk, ad[u], rw[u], dt, w2,
                                            w3, aC, aT, aF, aK);
      }
    }
  }
  NEO_MARK("loop_end");
  // (fp32 sampling only: the fp64 parity mode keeps the summation order below, the one its runs were pinned to the
  //  reference's recorded iterates with)
  if constexpr (sizeof(Real) == 4 && !SAMPLE_IO && LG::S > 1) {
    if (fold_rows != nullptr && fold_acc && sl.lmax != 0) {
      // The same per-piece sums in 80 bytes a piece instead of 96 a sample lane (the kernels that keep the cyclic
      // reduction's multipliers in LDS across this phase have no room for the rows): `fold_rows` is [M][fold_acc_stride(D)]
      // = per piece [d][6] partials, the duration partial, padding to 16 bytes (D = 3: 20 floats).  The sample lanes of a piece add their values one after the other --
      // residue 0 writes, residue 1 adds to what it reads back, ... -- which is the order the rows below are summed in:
      // the same bits.  lmax * (5 reads + 19 adds + 5 writes) under an exec mask of one lane per piece.
      typedef Real Quad __attribute__((ext_vector_type(4)));
      constexpr int RS = fold_acc_stride(D), NQ = RS / 4;  // floats a piece: [d][6], the duration partial, padding
      Quad v[NQ];
      {
        Real f[RS];
#pragma unroll
        for (int q = 0; q < RS; ++q) f[q] = Real(0);
#pragma unroll
        for (int d = 0; d < D; ++d)
#pragma unroll
          for (int k = 0; k < 6; ++k) f[d * 6 + k] = aC[k][d];
        f[6 * D] = aT;
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[q] = Quad{f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]};
      }
      Quad *mine = reinterpret_cast<Quad *>(fold_rows + (size_t)(act ? piece : 0) * RS);
      lds_wave_sync();
      for (int i = 0; i < sl.lmax; ++i) {
        if (act && r == i) {
          if (i == 0) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) mine[q] = v[q];
          } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) mine[q] = mine[q] + v[q];
          }
        }
        lds_wave_sync();
      }
      {
        const int pc = LG::piece();
        const int Lp = __shfl(sl.Lp, pc < M ? pc : 0, kWave);
        const bool have = pc < M && Lp > 0;
        // (may_alias: the accumulators were written as 16-byte vectors)
        typedef Real Pair __attribute__((ext_vector_type(2), may_alias));
        const Pair *src = reinterpret_cast<const Pair *>(fold_rows + (size_t)(have ? pc : 0) * RS + LG::dim0() * 6);
        const Pair p0 = src[0], p1 = src[1], p2 = src[2];
        const Real tpart = fold_rows[(size_t)(have ? pc : 0) * RS + 6 * D];
        gC[0][0] = have ? p0.x : Real(0); gC[1][0] = have ? p0.y : Real(0);
        gC[2][0] = have ? p1.x : Real(0); gC[3][0] = have ? p1.y : Real(0);
        gC[4][0] = have ? p2.x : Real(0); gC[5][0] = have ? p2.y : Real(0);
        gT = have ? tpart : Real(0);
      }
      lds_wave_sync();  // (the staging buffer is the caller's again)
      cost_feas = (double)LG::sum(act ? aF : Real(0));
      cost_coll = (double)LG::sum(act ? aK : Real(0));
      return;
    }
  }
  if constexpr (sizeof(Real) == 4) {
    if (fold_rows != nullptr && sl.lmax != 0) {
      // Pieces with different numbers of sample lanes: the per-piece sums go through LDS.  Every sample lane writes its
      // row [d][8] = (aC[0..5][d], aT, 0) (two 16-byte stores a dimension); the receiving lane -- the lane of (piece,
      // dimension) in the fused kernels, the first sample lane of the piece in the stand-alone kernel -- then adds up
      // the rows first .. first + Lp - 1 of its piece, one 32-byte block a row and dimension, in that order.
      // lmax * (2 reads + 7 adds) instructions a dimension, against log2(lmax) * 21 * (shuffle + select + add) and 21
      // more shuffles to bring the result home for the in-register fold below; the two costs need no fold at all.
      typedef Real Quad __attribute__((ext_vector_type(4)));
      constexpr int DR = LG::dl(D);  // dimensions the receiving lane keeps (SAMPLE_IO: LG = WaveLanes, all D)
      Quad *row = reinterpret_cast<Quad *>(fold_rows + (size_t)lane * (8 * D));
#pragma unroll
      for (int d = 0; d < D; ++d) {
        row[2 * d] = Quad{aC[0][d], aC[1][d], aC[2][d], aC[3][d]};
        row[2 * d + 1] = Quad{aC[4][d], aC[5][d], aT, Real(0)};
      }
      lds_wave_sync();
      int first, Lp, dim_base = 0;
      if constexpr (SAMPLE_IO) {
        first = lane;
        Lp = (act && r == 0) ? L : 0;
      } else {
        first = sl.first;
        Lp = sl.Lp;
        if constexpr (LG::S > 1) {
          first = __shfl(sl.first, LG::piece(), kWave);
          Lp = __shfl(sl.Lp, LG::piece(), kWave);
        }
        if (LG::piece() >= M) Lp = 0;
        dim_base = LG::dim0();
      }
      Quad lo[DR], hi[DR];
#pragma unroll
      for (int dd = 0; dd < DR; ++dd) lo[dd] = hi[dd] = Quad{Real(0), Real(0), Real(0), Real(0)};
      const Quad *src = reinterpret_cast<const Quad *>(fold_rows + (size_t)first * (8 * D)) + 2 * dim_base;
      for (int i = 0; i < sl.lmax; ++i) {
        if (i < Lp) {
#pragma unroll
          for (int dd = 0; dd < DR; ++dd) {
            lo[dd] += src[i * 2 * D + 2 * dd];
            hi[dd] += src[i * 2 * D + 2 * dd + 1];
          }
        }
      }
      lds_wave_sync();  // (the rows are the caller's staging buffer again)
#pragma unroll
      for (int dd = 0; dd < DR; ++dd) {
        gC[0][dd] = lo[dd].x; gC[1][dd] = lo[dd].y; gC[2][dd] = lo[dd].z; gC[3][dd] = lo[dd].w;
        gC[4][dd] = hi[dd].x; gC[5][dd] = hi[dd].y;
      }
      gT = hi[0].z;
      // (summed in the sampling arithmetic: 10 instructions a reduction in fp32 against 26 in fp64)
      cost_feas = (double)LG::sum(act ? aF : Real(0));
      cost_coll = (double)LG::sum(act ? aK : Real(0));
      return;
    }
  }
  // fold the L lanes of each piece (fixed order); PIECE-layout callers get lane piece*L moved to lane piece
  Real fv[6 * D + 3];
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int d = 0; d < D; ++d) fv[k * D + d] = aC[k][d];
  fv[6 * D] = aT;
  fv[6 * D + 1] = aF;
  fv[6 * D + 2] = aK;
  if (sl.lmax == 0)
    fold_piece_lanes<Real, 6 * D + 3>(fv, L, r);
  else
    fold_piece_segments<Real, 6 * D + 3>(fv, r, L, sl.lmax);
  if constexpr (!SAMPLE_IO) {
    // back to the lanes of the piece (sl.first is held by lane p for piece p)
    int src = LG::base() + sl.first;
    if constexpr (LG::S > 1) src = __shfl(sl.first, LG::piece(), kWave);
#pragma unroll
    for (int q = 0; q < 6 * D + 3; ++q) fv[q] = __shfl(fv[q], src, kWave);
  }
  if constexpr (LG::S == 1) {
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int d = 0; d < D; ++d) gC[k][d] = fv[k * D + d];
  } else {
    // one dimension per lane: keep this lane's column of the partials
    const int dm = LG::dim0();
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      Real v = fv[k * D];
#pragma unroll
      for (int j = 1; j < D; ++j) v = (dm == j) ? fv[k * D + j] : v;
      gC[k][0] = v;
    }
  }
  gT = fv[6 * D];
  const Real pf = fv[6 * D + 1], pk = fv[6 * D + 2];
  const bool mine = SAMPLE_IO ? (act && r == 0) : (LG::piece() < M && LG::dim0() == opaque_uniform(0));  // every piece once
  cost_feas = LG::sum(mine ? (double)pf : 0.0);
  cost_coll = LG::sum(mine ? (double)pk : 0.0);
}

// backward pass (PIECE layout): gC = dW/dc incl. sampled part, gT = direct dW/dT incl. sampled part
// (energy and time parts are added here).  Outputs grad wrt the start-joint position of the lane
// (gq, valid for lanes 1..M-1) and grad wrt tau (gtau, lanes 0..M-1).
// Returns 0, or 4 where the reference would leave through OverflowError: it raises Python floats to
// a power in two places, `(np.dot(c, beta3).item())**2` (:382) and `(1+math.exp(-tau))**2` (:490),
// and Python raises once such a result exceeds the double range instead of returning inf.
template <int D, class LG = WaveLanes, typename Num = double, bool PCR = (sizeof(Num) == 4 && LG::W == kWave)>
__device__ __forceinline__ int minco_backward(const Traj<D, LG::dl(D), Num> &t, const DevParams &prm,
                                              Num (&gC)[6][LG::dl(D)], Num gT, Num (&gq)[LG::dl(D)], Num &gtau) {
  constexpr int DL = LG::dl(D);
  const int lane = LG::piece();
  const int M = t.M;
  int pow_overflow = 0;
  const Num a1 = LG::prev(t.i1, Num(1.0)), a2 = a1 * a1, a3 = a2 * a1, a4 = a2 * a2;  // piece p-1
  const Num T = t.T, T2 = T * T, T3 = T2 * T, T4 = T2 * T2, T5 = T4 * T;
  const Num w0 = par_w0<Num>(prm);
  Num jerk_end[DL], snap_end[DL], crackle[DL];
  // add_energy_grad_CT (:361-384), add_time_grad_CT (:389-390)
  // (gT: the sampled partial and the time weight enter once per piece -- in the lane of its first dimension)
  gT = (LG::dim0() == opaque_uniform(0)) ? gT + par_w1<Num>(prm) : Num(0.0);
#pragma unroll
  for (int d = 0; d < DL; ++d) {
    const Num c3 = t.c[3][d], c4 = t.c[4][d], c5 = t.c[5][d];
    gC[3][d] += Num(2.0) * w0 * (Num(36.0) * T * c3 + Num(72.0) * T2 * c4 + Num(120.0) * T3 * c5);
    gC[4][d] += Num(2.0) * w0 * (Num(72.0) * T2 * c3 + Num(192.0) * T3 * c4 + Num(360.0) * T4 * c5);
    gC[5][d] += Num(2.0) * w0 * (Num(120.0) * T3 * c3 + Num(360.0) * T4 * c4 + Num(720.0) * T5 * c5);
    jerk_end[d] = Num(6.0) * c3 + Num(24.0) * T * c4 + Num(60.0) * T2 * c5;
    snap_end[d] = Num(24.0) * c4 + Num(120.0) * T * c5;
    crackle[d] = Num(120.0) * c5;
    // `(...).item()**2` (:382) raises beyond sqrt(DBL_MAX); fp32 cannot hold such a value: there the flag goes up when
    // the jerk itself has left the fp32 range (inf / NaN), which is where its square would poison the gradient
    if constexpr (sizeof(Num) == 8) {
      if (lane < M && fabs(jerk_end[d]) > Num(1.3407807929942596e154)) pow_overflow = 1;  // sqrt(DBL_MAX)
    } else {
      if (lane < M && !(fabs(jerk_end[d]) <= Num(3.4028234663852886e38))) pow_overflow = 1;
    }
    gT += w0 * jerk_end[d] * jerk_end[d];
  }
  // gz = H(T)^T gC : sensitivity wrt the end states Z = (p0, v0, a0, p1, v1, a1)
  Num gz[6][DL];
#pragma unroll
  for (int d = 0; d < DL; ++d) {
    const Num gep = Num(10.0) * gC[3][d] * t.i3 - Num(15.0) * gC[4][d] * t.i4 + Num(6.0) * gC[5][d] * t.i4 * t.i1;
    const Num gev = -Num(4.0) * gC[3][d] * t.i2 + Num(7.0) * gC[4][d] * t.i3 - Num(3.0) * gC[5][d] * t.i4;
    const Num gea = Num(0.5) * gC[3][d] * t.i1 - gC[4][d] * t.i2 + Num(0.5) * gC[5][d] * t.i3;
    gz[0][d] = gC[0][d] - gep;
    gz[1][d] = gC[1][d] - T * gep - gev;
    gz[2][d] = Num(0.5) * gC[2][d] - Num(0.5) * T2 * gep - T * gev - gea;
    gz[3][d] = gep;
    gz[4][d] = gev;
    gz[5][d] = gea;
  }
  // S_p = dW/d(state of joint p) = gz_{p-1}[3:6] + gz_p[0:3]   (lanes 1..M-1)
  Num S[3][DL];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int d = 0; d < DL; ++d) S[k][d] = LG::prev(gz[3 + k][d], Num(0.0)) + gz[k][d];

  Num lam[2][DL];
#pragma unroll
  for (int d = 0; d < DL; ++d) lam[0][d] = lam[1][d] = Num(0.0);
  constexpr bool kPcr = PCR;  // the transposed system by pcr_solve, too
  if constexpr (kPcr) {
    if (M > 1 && t.pcr_mult != nullptr) {
      Num R[2][DL], y[2][DL];
#pragma unroll
      for (int d = 0; d < DL; ++d) {
        R[0][d] = S[1][d];
        R[1][d] = S[2][d];
      }
      NEO_MARK("bwd_pcr_begin");
      pcr_solve_transposed<DL, LG, Num>(M, t.pcr_mult, R, y);
      NEO_MARK("bwd_pcr_end");
#pragma unroll
      for (int d = 0; d < DL; ++d) {
        lam[0][d] = (lane >= 1 && lane < M) ? y[0][d] : Num(0.0);
        lam[1][d] = (lane >= 1 && lane < M) ? y[1][d] : Num(0.0);
      }
    } else if (M > 1) {
      Num LoT[2][2], DiT[2][2], UpT[2][2], R[2][DL], y[2][DL];
      LoT[0][0] = Num(24.0) * a2;  LoT[0][1] = -Num(168.0) * a3;   // Up_{p-1}^T (piece p-1 = "a")
      LoT[1][0] = -Num(3.0) * a1;  LoT[1][1] = Num(24.0) * a2;
      DiT[0][0] = -Num(36.0) * a2 + Num(36.0) * t.i2;  DiT[0][1] = -Num(192.0) * a3 - Num(192.0) * t.i3;  // Di_p^T
      DiT[1][0] = Num(9.0) * a1 + Num(9.0) * t.i1;     DiT[1][1] = Num(36.0) * a2 - Num(36.0) * t.i2;
      UpT[0][0] = -Num(24.0) * t.i2;  UpT[0][1] = -Num(168.0) * t.i3;  // Lo_{p+1}^T (piece p = "b")
      UpT[1][0] = -Num(3.0) * t.i1;   UpT[1][1] = -Num(24.0) * t.i2;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          LoT[i][j] = lane == 1 ? Num(0.0) : LoT[i][j];        // (zero boundary values: nothing to fold)
          UpT[i][j] = lane == M - 1 ? Num(0.0) : UpT[i][j];
        }
#pragma unroll
      for (int d = 0; d < DL; ++d) {
        R[0][d] = S[1][d];
        R[1][d] = S[2][d];
      }
      NEO_MARK("bwd_pcr_begin");
      pcr_solve<DL, LG, Num>(M, LoT, DiT, UpT, R, y);
      NEO_MARK("bwd_pcr_end");
#pragma unroll
      for (int d = 0; d < DL; ++d) {
        lam[0][d] = (lane >= 1 && lane < M) ? y[0][d] : Num(0.0);
        lam[1][d] = (lane >= 1 && lane < M) ? y[1][d] : Num(0.0);
      }
    }
  } else if (M > 1) {
    // transposed system: row p of K^T has Up_{p-1}^T, Di_p^T, Lo_{p+1}^T; pivot inverses are N^T
    Num LoT[2][2], NT[2][2], ET[2][2], R[2][DL], z0[2][DL], y[2][DL];
    LoT[0][0] = Num(24.0) * a2;  LoT[0][1] = -Num(168.0) * a3;   // Up_{p-1}^T (piece p-1 = "a")
    LoT[1][0] = -Num(3.0) * a1;  LoT[1][1] = Num(24.0) * a2;
    NT[0][0] = t.N[0][0]; NT[0][1] = t.N[1][0]; NT[1][0] = t.N[0][1]; NT[1][1] = t.N[1][1];
    {
      const Num u00 = -Num(24.0) * t.i2, u01 = -Num(168.0) * t.i3;  // Lo_{p+1}^T (piece p = "b")
      const Num u10 = -Num(3.0) * t.i1, u11 = -Num(24.0) * t.i2;
      ET[0][0] = NT[0][0] * u00 + NT[0][1] * u10;
      ET[0][1] = NT[0][0] * u01 + NT[0][1] * u11;
      ET[1][0] = NT[1][0] * u00 + NT[1][1] * u10;
      ET[1][1] = NT[1][0] * u01 + NT[1][1] * u11;
    }
#pragma unroll
    for (int d = 0; d < DL; ++d) {
      R[0][d] = S[1][d];
      R[1][d] = S[2][d];
      z0[0][d] = z0[1][d] = Num(0.0);
    }
    thomas_solve<DL, LG, Num>(NEO_DBG(prm, 2 | 16) ? 1 : M, LoT, NT, ET, R, z0, z0, y);
#pragma unroll
    for (int d = 0; d < DL; ++d) {
      lam[0][d] = (lane >= 1 && lane < M) ? y[0][d] : Num(0.0);
      lam[1][d] = (lane >= 1 && lane < M) ? y[1][d] : Num(0.0);
    }
  }
  // dW/dq: G[6i+3] of the reference (:506-508)
  Num Gt[3][DL];  // sensitivity wrt the tail state (lane M-1), = G[-3:] of the reference
#pragma unroll
  for (int d = 0; d < DL; ++d) {
    const Num l1 = lam[0][d], l2 = lam[1][d];
    const Num dp_prev = -Num(60.0) * a3 * l1 - Num(360.0) * a4 * l2;
    const Num dp_here = (Num(60.0) * a3 + Num(60.0) * t.i3) * l1 + (Num(360.0) * a4 - Num(360.0) * t.i4) * l2;
    const Num dp_next = -Num(60.0) * t.i3 * l1 + Num(360.0) * t.i4 * l2;
    const Num from_left = LG::prev(dp_next, Num(0.0));   // joint p-1 pushes on p_{p}
    const Num from_right = LG::next(dp_prev, Num(0.0));  // joint p+1 pushes on p_{p}
    gq[d] = S[0][d] - dp_here - (lane >= 2 ? from_left : Num(0.0)) - (lane + 1 <= M - 1 ? from_right : Num(0.0));
    // tail sensitivity on lane M-1: S_M = gz[3:6] of the last piece, minus joint M-1's pull
    const Num lt1 = (M > 1) ? l1 : Num(0.0), lt2 = (M > 1) ? l2 : Num(0.0);
    Gt[0][d] = gz[3][d] - (-Num(60.0) * t.i3 * lt1 + Num(360.0) * t.i4 * lt2);
    Gt[1][d] = gz[4][d] - (Num(24.0) * t.i2 * lt1 - Num(168.0) * t.i3 * lt2);
    Gt[2][d] = gz[5][d] - (-Num(3.0) * t.i1 * lt1 + Num(24.0) * t.i2 * lt2);
  }
  // dW/dT (:511-533)
  Num gTt = gT;
#pragma unroll
  for (int d = 0; d < DL; ++d) {
    const Num v1 = t.V1[d], a1 = t.A1[d], je = jerk_end[d];
    gTt -= gz[3][d] * v1 + gz[4][d] * a1 + gz[5][d] * je;
    // joint p+1 (this piece ends there): rows +je.Z, +se.Z
    const Num ln1 = LG::next(lam[0][d], Num(0.0)), ln2 = LG::next(lam[1][d], Num(0.0));
    if (lane + 1 <= M - 1) {
      const Num d_je = snap_end[d] - (Num(60.0) * t.i3 * v1 - Num(36.0) * t.i2 * a1 + Num(9.0) * t.i1 * je);
      const Num d_se = crackle[d] - (Num(360.0) * t.i4 * v1 - Num(192.0) * t.i3 * a1 + Num(36.0) * t.i2 * je);
      gTt -= ln1 * d_je + ln2 * d_se;
    }
    // joint p (this piece starts there): rows -js.Z, -ss.Z
    if (lane >= 1) {
      const Num d_js = -(Num(60.0) * t.i3 * v1 - Num(24.0) * t.i2 * a1 + Num(3.0) * t.i1 * je);
      const Num d_ss = -(-Num(360.0) * t.i4 * v1 + Num(168.0) * t.i3 * a1 - Num(24.0) * t.i2 * je);
      gTt += lam[0][d] * d_js + lam[1][d] * d_ss;
    }
  }
  // the reference evaluates the tail rows' d/dT with the previous piece's duration (:528-533)
  {
    const Num Ts = LG::prev(T, T);
    if (prm.stale_T && M >= 2 && lane == M - 1) {
      const Num S2 = Ts * Ts, S3 = S2 * Ts, S4 = S2 * S2;
#pragma unroll
      for (int d = 0; d < DL; ++d) {
        const Num c1 = t.c[1][d], c2 = t.c[2][d], c3 = t.c[3][d], c4 = t.c[4][d], c5 = t.c[5][d];
        const Num velL = c1 + Num(2.0) * T * c2 + Num(3.0) * T2 * c3 + Num(4.0) * T3 * c4 + Num(5.0) * T4 * c5;
        const Num velS = c1 + Num(2.0) * Ts * c2 + Num(3.0) * S2 * c3 + Num(4.0) * S3 * c4 + Num(5.0) * S4 * c5;
        const Num accL = Num(2.0) * c2 + Num(6.0) * T * c3 + Num(12.0) * T2 * c4 + Num(20.0) * T3 * c5;
        const Num accS = Num(2.0) * c2 + Num(6.0) * Ts * c3 + Num(12.0) * S2 * c4 + Num(20.0) * S3 * c5;
        const Num jrkL = Num(6.0) * c3 + Num(24.0) * T * c4 + Num(60.0) * T2 * c5;
        const Num jrkS = Num(6.0) * c3 + Num(24.0) * Ts * c4 + Num(60.0) * S2 * c5;
        gTt += Gt[0][d] * (velL - velS) + Gt[1][d] * (accL - accS) + Gt[2][d] * (jrkL - jrkS);
      }
    }
  }
  // get_grad_T2tau (:485-492)
  // `(1 + math.exp(-tau))**2` (:490) is a Python-float power too: OverflowError beyond sqrt(DBL_MAX)
  if constexpr (sizeof(Num) == 8) {
    const Num ex = t.tau;  // exp(-tau), left there by minco_forward
    if (lane < M && (Num(1.0) + ex) > Num(1.3407807929942596e154)) pow_overflow = 1;
    gtau = LG::sum_dims(gTt) * (Num(prm.T_max) - Num(prm.T_min)) * ex / ((Num(1.0) + ex) * (Num(1.0) + ex));  // (valid in the piece's first lane)
  } else {
    // fp32: minco_forward left -tau.  The same range test on tau itself (1 + exp(-tau) > sqrt(DBL_MAX) <=> -tau >
    // 354.89), and the factor exp(-tau) / (1 + exp(-tau))^2 as (e s) s with s = 1 / (1 + e), e = exp(-tau) clamped to
    // FLT_MAX: a value below 1e-38 (as in fp64) instead of inf / inf once expf overflows (-tau > 88.72)
    const Num nt = t.tau;
    if (lane < M && nt > Num(354.891356446692)) pow_overflow = 1;
    Num ex, s;
    if constexpr (LG::W == kWave) {
      ex = fmin(__expf(nt), Num(3.4028234663852886e38));
      s = precise_rcp(Num(1.0) + ex);
    } else {
      ex = fmin(exp(nt), Num(3.4028234663852886e38));
      s = Num(1.0) / (Num(1.0) + ex);
    }
    gtau = LG::sum_dims(gTt) * par_T_span<Num>(prm) * ((ex * s) * s);
  }
  return LG::any(pow_overflow) ? 4 : 0;
}

}  // namespace neo
