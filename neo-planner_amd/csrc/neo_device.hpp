// neo_device.hpp -- device-side MINCO cost/gradient for one trajectory per wavefront (gfx950).
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
};

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
  int layout;  // 0 linear [z][y][x], 1 = 4x4x4 bricks
  int bx, by;  // bricks per axis (layout 1)
  double res, ox, oy, oz;
};

// ------------------------------------------------------------------ wave helpers
__device__ __forceinline__ int lane_id() { return (int)__lane_id(); }

__device__ __forceinline__ double rdlane(double v, int src /*wave-uniform*/) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double uniform(double v) {
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, kWave);
  return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) {
    T o = __shfl_xor(v, s, kWave);
    v = o > v ? o : v;
  }
  return v;
}
// value of lane (l-1) / (l+1); lanes without such a neighbour get `fill`
__device__ __forceinline__ double from_prev(double v, double fill) {
  double o = __shfl_up(v, 1, kWave);
  return lane_id() == 0 ? fill : o;
}
__device__ __forceinline__ double from_next(double v, double fill) {
  double o = __shfl_down(v, 1, kWave);
  return lane_id() == kWave - 1 ? fill : o;
}

// ------------------------------------------------------------------ map lookups
// esdf.py:53-82: nearest cell with int() truncation; out of range -> 10000 / zero gradient.
// Returns the distance; the gradient (metres per cell, as np.gradient leaves it) is only
// fetched when the caller needs it -- it lives in the same 32-byte record.
template <typename Real>
struct Lookup2D {
  const Map2D &m;
  __device__ __forceinline__ explicit Lookup2D(const Map2D &m_) : m(m_) {}
  static constexpr int kGradDims = 2;
  template <int D>
  __device__ __forceinline__ Real fetch(const Real (&pos)[D], Real (&g)[D], bool &inside) const {
    // index arithmetic in fp64 with a true division, exactly like int((y - origin.y) / res)
    const double fy = ((double)pos[1] - m.oy) / m.res;
    const double fx = ((double)pos[0] - m.ox) / m.res;
#pragma unroll
    for (int d = 0; d < D; ++d) g[d] = Real(0);
    inside = false;
    if (!(fabs(fy) < 1.0e9) || !(fabs(fx) < 1.0e9)) return Real(10000);
    const int row = (int)fy, col = (int)fx;  // C casts truncate toward zero, like int()
    if (row < 0 || row >= m.H || col < 0 || col >= m.W) return Real(10000);
    inside = true;
    const double4 r = m.rec[(size_t)row * m.W + col];
    g[0] = (Real)r.y;
    g[1] = (Real)r.z;
    return (Real)r.x;
  }
};

template <typename E>
__device__ __forceinline__ float elem_to_float(E v);
template <>
__device__ __forceinline__ float elem_to_float<float>(float v) { return v; }
template <>
__device__ __forceinline__ float elem_to_float<__half>(__half v) { return __half2float(v); }

// two x-adjacent voxels in one (dword-aligned) load
template <typename E>
struct __attribute__((packed, aligned(sizeof(E)))) Pair {
  E a, b;
};

// trilinear distance + analytic gradient (oracle/minco_np.py:Grid3DESDF defines the semantics)
template <typename Real, typename E>
struct Lookup3D {
  const Map3D &m;
  __device__ __forceinline__ explicit Lookup3D(const Map3D &m_) : m(m_) {}
  static constexpr int kGradDims = 3;

  __device__ __forceinline__ size_t addr(int ix, int iy, int iz) const {
    if (m.layout == 0) return ((size_t)iz * m.ny + iy) * m.nx + ix;
    const size_t brick = ((size_t)(iz >> 2) * m.by + (iy >> 2)) * m.bx + (ix >> 2);
    return brick * 64 + ((iz & 3) << 4) + ((iy & 3) << 2) + (ix & 3);
  }

  template <int D>
  __device__ __forceinline__ Real fetch(const Real (&pos)[D], Real (&g)[D], bool &inside) const {
    static_assert(D == 3, "the 3-D map needs D = 3");
    const E *vox = static_cast<const E *>(m.data);
    const int n[3] = {m.nx, m.ny, m.nz};
    const double org[3] = {m.ox, m.oy, m.oz};
    int i0[3];
    Real fr[3];
    inside = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      Real u;
      if constexpr (sizeof(Real) == 8)
        u = (Real)(((double)pos[a] - org[a]) / m.res);  // as oracle Grid3DESDF._cell
      else
        u = (pos[a] - (Real)org[a]) * (Real)(1.0 / m.res);
      if (!(u >= Real(0) && u < (Real)n[a])) inside = false;
      u -= Real(0.5);
      int i = (int)floor(u);
      i = i < 0 ? 0 : (i > n[a] - 2 ? n[a] - 2 : i);
      Real f = u - (Real)i;
      f = f < Real(0) ? Real(0) : (f > Real(1) ? Real(1) : f);
      i0[a] = i;
      fr[a] = f;
    }
#pragma unroll
    for (int d = 0; d < D; ++d) g[d] = Real(0);
    if (!inside) return Real(10000);
    Real c[2][2][2];
    if (m.layout == 0) {
#pragma unroll
      for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
          const Pair<E> p = *reinterpret_cast<const Pair<E> *>(vox + addr(i0[0], i0[1] + dy, i0[2] + dz));
          c[dz][dy][0] = (Real)elem_to_float<E>(p.a);
          c[dz][dy][1] = (Real)elem_to_float<E>(p.b);
        }
    } else {
#pragma unroll
      for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx)
            c[dz][dy][dx] = (Real)elem_to_float<E>(vox[addr(i0[0] + dx, i0[1] + dy, i0[2] + dz)]);
    }
    const Real fx = fr[0], fy = fr[1], fz = fr[2];
    const Real inv_res = (Real)(1.0 / m.res);
    const Real dx00 = c[0][0][1] - c[0][0][0], dx10 = c[0][1][1] - c[0][1][0];
    const Real dx01 = c[1][0][1] - c[1][0][0], dx11 = c[1][1][1] - c[1][1][0];
    const Real c00 = c[0][0][0] + fx * dx00, c10 = c[0][1][0] + fx * dx10;
    const Real c01 = c[1][0][0] + fx * dx01, c11 = c[1][1][0] + fx * dx11;
    const Real c0 = c00 + fy * (c10 - c00), c1 = c01 + fy * (c11 - c01);
    const Real dx0 = dx00 + fy * (dx10 - dx00), dx1 = dx01 + fy * (dx11 - dx01);
    const Real dy0 = c10 - c00, dy1 = c11 - c01;
    g[0] = (dx0 + fz * (dx1 - dx0)) * inv_res;
    g[1] = (dy0 + fz * (dy1 - dy0)) * inv_res;
    g[2] = (c1 - c0) * inv_res;
    return c0 + fz * (c1 - c0);
  }
};

// ------------------------------------------------------------------ per-trajectory state
template <int D>
struct Traj {
  // wave-uniform
  int M, n, nq, L;
  // PIECE layout (lane p < M)
  double T, tau;
  double i1, i2, i3, i4;          // T^-1 .. T^-4
  double a1, a2, a3, a4;          // the same of piece p-1 (lane p >= 1)
  double P0[D], P1[D];            // positions at the start / end joint
  double V0[D], A0[D], V1[D], A1[D];
  double c[6][D];                 // polynomial coefficients
  int ns;                         // samples of this piece: int(T / delta_t)
  double head[3][D], tail[3][D];  // boundary states (uniform)
};

// Solve  Lo_p y_{p-1} + Di_p y_p + Up_p y_{p+1} = R_p  (p = 1..M-1) for 2-vectors y with D
// right-hand sides, y_0 and y_M given.  Blocks live on lane p.  Block Thomas: the sweep is
// sequential over joints; lane p-1 hands (E, f) to lane p through v_readlane.
template <int D>
__device__ __forceinline__ void block_thomas(int M, const double (&Lo)[2][2], const double (&Di)[2][2],
                                             const double (&Up)[2][2], const double (&R)[2][D],
                                             const double (&y0)[2][D], const double (&yM)[2][D],
                                             double (&y)[2][D]) {
  const int lane = lane_id();
  double E[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
  double f[2][D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    f[0][d] = y0[0][d];
    f[1][d] = y0[1][d];
  }
  for (int p = 1; p < M; ++p) {
    double Ep[2][2], fp[2][D];
#pragma unroll
    for (int a = 0; a < 2; ++a) {
#pragma unroll
      for (int b = 0; b < 2; ++b) Ep[a][b] = rdlane(E[a][b], p - 1);
#pragma unroll
      for (int d = 0; d < D; ++d) fp[a][d] = rdlane(f[a][d], p - 1);
    }
    // Dp = Di - Lo E_{p-1};  Rp = R - Lo f_{p-1}
    const double d00 = Di[0][0] - (Lo[0][0] * Ep[0][0] + Lo[0][1] * Ep[1][0]);
    const double d01 = Di[0][1] - (Lo[0][0] * Ep[0][1] + Lo[0][1] * Ep[1][1]);
    const double d10 = Di[1][0] - (Lo[1][0] * Ep[0][0] + Lo[1][1] * Ep[1][0]);
    const double d11 = Di[1][1] - (Lo[1][0] * Ep[0][1] + Lo[1][1] * Ep[1][1]);
    const double idet = 1.0 / (d00 * d11 - d01 * d10);
    const double n00 = d11 * idet, n01 = -d01 * idet, n10 = -d10 * idet, n11 = d00 * idet;
    if (lane == p) {
      E[0][0] = n00 * Up[0][0] + n01 * Up[1][0];
      E[0][1] = n00 * Up[0][1] + n01 * Up[1][1];
      E[1][0] = n10 * Up[0][0] + n11 * Up[1][0];
      E[1][1] = n10 * Up[0][1] + n11 * Up[1][1];
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const double r0 = R[0][d] - (Lo[0][0] * fp[0][d] + Lo[0][1] * fp[1][d]);
        const double r1 = R[1][d] - (Lo[1][0] * fp[0][d] + Lo[1][1] * fp[1][d]);
        f[0][d] = n00 * r0 + n01 * r1;
        f[1][d] = n10 * r0 + n11 * r1;
      }
    }
  }
  // back substitution: y_p = f_p - E_p y_{p+1}, y_M given
#pragma unroll
  for (int d = 0; d < D; ++d) {
    y[0][d] = yM[0][d];
    y[1][d] = yM[1][d];
  }
  // lane M (virtual) holds y_M: keep it in every lane >= M so that readlane(M) is valid for M < 64
  for (int p = M - 1; p >= 1; --p) {
    double yn[2][D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      if (p + 1 < kWave && p + 1 < M) {
        yn[0][d] = rdlane(y[0][d], p + 1);
        yn[1][d] = rdlane(y[1][d], p + 1);
      } else {
        yn[0][d] = yM[0][d];
        yn[1][d] = yM[1][d];
      }
    }
    if (lane == p) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        y[0][d] = f[0][d] - (E[0][0] * yn[0][d] + E[0][1] * yn[1][d]);
        y[1][d] = f[1][d] - (E[1][0] * yn[0][d] + E[1][1] * yn[1][d]);
      }
    }
  }
}

// joint-system blocks of lane p (joint p between piece p-1 "a" and piece p "b")
template <int D>
__device__ __forceinline__ void joint_blocks(const Traj<D> &t, double (&Lo)[2][2], double (&Di)[2][2],
                                             double (&Up)[2][2]) {
  Lo[0][0] = -24.0 * t.a2;  Lo[0][1] = -3.0 * t.a1;
  Lo[1][0] = -168.0 * t.a3; Lo[1][1] = -24.0 * t.a2;
  Di[0][0] = -36.0 * t.a2 + 36.0 * t.i2;    Di[0][1] = 9.0 * t.a1 + 9.0 * t.i1;
  Di[1][0] = -192.0 * t.a3 - 192.0 * t.i3;  Di[1][1] = 36.0 * t.a2 - 36.0 * t.i2;
  Up[0][0] = 24.0 * t.i2;   Up[0][1] = -3.0 * t.i1;
  Up[1][0] = -168.0 * t.i3; Up[1][1] = 24.0 * t.i2;
}

// forward pass.  Inputs (PIECE layout): t.tau, t.P0, t.P1 set by the caller, head/tail uniform.
// Returns 0 or NUMERIC_RANGE (4) when exp(-tau) overflows like math.exp does (:481).
template <int D>
__device__ __forceinline__ int minco_forward(Traj<D> &t, const DevParams &prm, double &energy,
                                             double &time_sum) {
  const int lane = lane_id();
  const bool act = lane < t.M;
  int bad = 0;
  // map_tau2T (:477-483)
  {
    const double tau = act ? t.tau : 0.0;
    if (-tau > 709.782712893384) bad = 1;
    t.T = (prm.T_max - prm.T_min) / (1.0 + exp(-tau)) + prm.T_min;
  }
  if (__any(bad)) return 4;
  t.i1 = 1.0 / t.T;
  t.i2 = t.i1 * t.i1;
  t.i3 = t.i2 * t.i1;
  t.i4 = t.i2 * t.i2;
  t.a1 = from_prev(t.i1, 1.0);
  t.a2 = from_prev(t.i2, 1.0);
  t.a3 = from_prev(t.i3, 1.0);
  t.a4 = from_prev(t.i4, 1.0);
  t.ns = act ? (int)(t.T / prm.delta_t) : 0;  // int(T / delta_t) (:401)

  if (t.M > 1) {
    double Lo[2][2], Di[2][2], Up[2][2], R[2][D], y0[2][D], yM[2][D], y[2][D];
    joint_blocks(t, Lo, Di, Up);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      // displacement of piece p-1 and of piece p
      const double dPb = t.P1[d] - t.P0[d];
      const double dPa = from_prev(dPb, 0.0);
      R[0][d] = -(60.0 * t.a3 * dPa - 60.0 * t.i3 * dPb);
      R[1][d] = -(360.0 * t.a4 * dPa + 360.0 * t.i4 * dPb);
      y0[0][d] = t.head[1][d];
      y0[1][d] = t.head[2][d];
      yM[0][d] = t.tail[1][d];
      yM[1][d] = t.tail[2][d];
    }
    block_thomas<D>(t.M, Lo, Di, Up, R, y0, yM, y);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      t.V0[d] = lane == 0 ? t.head[1][d] : y[0][d];
      t.A0[d] = lane == 0 ? t.head[2][d] : y[1][d];
    }
  } else {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      t.V0[d] = t.head[1][d];
      t.A0[d] = t.head[2][d];
    }
  }
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const double v1 = from_next(t.V0[d], 0.0), a1 = from_next(t.A0[d], 0.0);
    t.V1[d] = (lane == t.M - 1) ? t.tail[1][d] : v1;
    t.A1[d] = (lane == t.M - 1) ? t.tail[2][d] : a1;
  }
  // Hermite form of the quintic
  double e = 0.0;
  const double T = t.T, T2 = T * T, T3 = T2 * T, T4 = T2 * T2, T5 = T4 * T;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const double ep = t.P1[d] - t.P0[d] - T * t.V0[d] - 0.5 * T2 * t.A0[d];
    const double ev = t.V1[d] - t.V0[d] - T * t.A0[d];
    const double ea = t.A1[d] - t.A0[d];
    t.c[0][d] = t.P0[d];
    t.c[1][d] = t.V0[d];
    t.c[2][d] = 0.5 * t.A0[d];
    t.c[3][d] = (10.0 * ep - 4.0 * T * ev + 0.5 * T2 * ea) * t.i3;
    t.c[4][d] = (-15.0 * ep + 7.0 * T * ev - T2 * ea) * t.i4;
    t.c[5][d] = (6.0 * ep - 3.0 * T * ev + 0.5 * T2 * ea) * t.i4 * t.i1;
    // add_energy_cost (:345-359): c^T Q(T) c with the closed-form jerk Gram matrix
    const double c3 = t.c[3][d], c4 = t.c[4][d], c5 = t.c[5][d];
    e += 36.0 * T * c3 * c3 + 144.0 * T2 * c3 * c4 + 240.0 * T3 * c3 * c5 + 192.0 * T3 * c4 * c4 +
         720.0 * T4 * c4 * c5 + 720.0 * T5 * c5 * c5;
  }
  energy = wave_sum(act ? e : 0.0);
  time_sum = wave_sum(act ? t.T : 0.0);  // add_time_cost (:386-387)
  return 0;
}

// sampled feasibility + collision terms (:392-466), SAMPLE layout.
// Outputs in PIECE layout: gC (added to), gT (added to); costs wave-uniform.
template <typename Real, int D, class LookupT>
__device__ __forceinline__ void minco_sample(const Traj<D> &t, const DevParams &prm, const LookupT &lk,
                                             double (&gC)[6][D], double &gT, double &cost_feas,
                                             double &cost_coll) {
  const int lane = lane_id();
  const int L = t.L;
  const int piece = lane / L, r = lane - piece * L;
  const bool act = piece < t.M;
  // hand the piece data to its L sample lanes
  Real c[6][D];
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int d = 0; d < D; ++d) c[k][d] = __shfl((Real)t.c[k][d], piece, kWave);
  const int ns_piece = __shfl(t.ns, piece, kWave);
  const int ns = act ? ns_piece : 0;
  const int iters = wave_max((ns + L - 1) / L);

  const Real dt = (Real)prm.delta_t, vmax2 = (Real)(prm.v_max * prm.v_max), safe = (Real)prm.safe_dis;
  const Real w2 = (Real)prm.w[2], w3 = (Real)prm.w[3];
  const Real inv_ns = ns > 0 ? Real(1) / (Real)ns : Real(0);
  Real aC[6][D];
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int d = 0; d < D; ++d) aC[k][d] = Real(0);
  Real aT = Real(0), aF = Real(0), aK = Real(0);

  for (int it = 0; it < iters; ++it) {
    const int j = r + it * L;
    if (j < ns) {
      const Real s = (Real)((double)j * prm.delta_t);  // beta_full row j: t = j * delta_t (:251)
      Real pos[D], vel[D];
#pragma unroll
      for (int d = 0; d < D; ++d) {
        pos[d] = c[0][d] + s * (c[1][d] + s * (c[2][d] + s * (c[3][d] + s * (c[4][d] + s * c[5][d]))));
        vel[d] = c[1][d] + s * (Real(2) * c[2][d] + s * (Real(3) * c[3][d] + s * (Real(4) * c[4][d] + s * (Real(5) * c[5][d]))));
      }
      const Real omg = (j == 0 || j == ns - 1) ? Real(0.5) : Real(1);
      const Real s2 = s * s, s3 = s2 * s, s4 = s2 * s2, s5 = s4 * s;
      // dynamic feasibility
      Real v2 = Real(0);
#pragma unroll
      for (int d = 0; d < D; ++d) v2 += vel[d] * vel[d];
      const Real vv = v2 - vmax2;
      if (vv > Real(0)) {
        aF += omg * dt * vv * vv * vv;
        Real av = Real(0);
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const Real acc = Real(2) * c[2][d] + s * (Real(6) * c[3][d] + s * (Real(12) * c[4][d] + s * (Real(20) * c[5][d])));
          av += acc * vel[d];
        }
        const Real dK = Real(3) * dt * omg * vv * vv;
        const Real b1[6] = {Real(0), Real(1), Real(2) * s, Real(3) * s2, Real(4) * s3, Real(5) * s4};
#pragma unroll
        for (int d = 0; d < D; ++d) {
          const Real u = w2 * dK * Real(2) * vel[d];
#pragma unroll
          for (int k = 1; k < 6; ++k) aC[k][d] += b1[k] * u;
        }
        aT += w2 * (omg * vv * vv * vv * inv_ns + dK * Real(2) * av * (Real)j * inv_ns);
      }
      // collision
      Real g[D];
      bool inside;
      const Real dist = lk.template fetch<D>(pos, g, inside);
      const Real vd = safe - dist;
      if (vd > Real(0)) {
        aK += omg * dt * vd * vd * vd;
        const Real dK = Real(3) * dt * omg * vd * vd;
        const Real b0[6] = {Real(1), s, s2, s3, s4, s5};
        Real gv = Real(0);
#pragma unroll
        for (int d = 0; d < D; ++d) {
          gv += g[d] * vel[d];
          const Real u = -(w3 * dK * g[d]);
#pragma unroll
          for (int k = 0; k < 6; ++k) aC[k][d] += b0[k] * u;
        }
        aT += w3 * (omg * vd * vd * vd * inv_ns + dK * (-gv) * (Real)j * inv_ns);
      }
    }
  }
  // fold the L lanes of each piece (fixed tree), then move lane piece*L -> lane piece
  auto fold = [&](Real v) -> double {
    for (int sft = 1; sft < L; sft <<= 1) {
      const Real o = __shfl_down(v, sft, kWave);
      if (r + sft < L) v += o;
    }
    return (double)__shfl(v, lane * L, kWave);
  };
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int d = 0; d < D; ++d) gC[k][d] += fold(aC[k][d]);
  gT += fold(aT);
  const double pf = fold(aF), pk = fold(aK);
  cost_feas = wave_sum(lane < t.M ? pf : 0.0);
  cost_coll = wave_sum(lane < t.M ? pk : 0.0);
}

// backward pass (PIECE layout): gC = dW/dc incl. sampled part, gT = direct dW/dT incl. sampled part
// (energy and time parts are added here).  Outputs grad wrt the start-joint position of the lane
// (gq, valid for lanes 1..M-1) and grad wrt tau (gtau, lanes 0..M-1).
// Returns 0, or 4 where the reference would leave through OverflowError: it raises Python floats to
// a power in two places, `(np.dot(c, beta3).item())**2` (:382) and `(1+math.exp(-tau))**2` (:490),
// and Python raises once such a result exceeds the double range instead of returning inf.
template <int D>
__device__ __forceinline__ int minco_backward(const Traj<D> &t, const DevParams &prm, double (&gC)[6][D],
                                              double gT, double (&gq)[D], double &gtau) {
  const int lane = lane_id();
  const int M = t.M;
  int pow_overflow = 0;
  const double T = t.T, T2 = T * T, T3 = T2 * T, T4 = T2 * T2, T5 = T4 * T;
  const double w0 = prm.w[0];
  double jerk_end[D], snap_end[D], crackle[D];
  // add_energy_grad_CT (:361-384), add_time_grad_CT (:389-390)
  gT += prm.w[1];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const double c3 = t.c[3][d], c4 = t.c[4][d], c5 = t.c[5][d];
    gC[3][d] += 2.0 * w0 * (36.0 * T * c3 + 72.0 * T2 * c4 + 120.0 * T3 * c5);
    gC[4][d] += 2.0 * w0 * (72.0 * T2 * c3 + 192.0 * T3 * c4 + 360.0 * T4 * c5);
    gC[5][d] += 2.0 * w0 * (120.0 * T3 * c3 + 360.0 * T4 * c4 + 720.0 * T5 * c5);
    jerk_end[d] = 6.0 * c3 + 24.0 * T * c4 + 60.0 * T2 * c5;
    snap_end[d] = 24.0 * c4 + 120.0 * T * c5;
    crackle[d] = 120.0 * c5;
    if (lane < M && fabs(jerk_end[d]) > 1.3407807929942596e154) pow_overflow = 1;  // sqrt(DBL_MAX)
    gT += w0 * jerk_end[d] * jerk_end[d];
  }
  // gz = H(T)^T gC : sensitivity wrt the end states Z = (p0, v0, a0, p1, v1, a1)
  double gz[6][D];
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const double gep = 10.0 * gC[3][d] * t.i3 - 15.0 * gC[4][d] * t.i4 + 6.0 * gC[5][d] * t.i4 * t.i1;
    const double gev = -4.0 * gC[3][d] * t.i2 + 7.0 * gC[4][d] * t.i3 - 3.0 * gC[5][d] * t.i4;
    const double gea = 0.5 * gC[3][d] * t.i1 - gC[4][d] * t.i2 + 0.5 * gC[5][d] * t.i3;
    gz[0][d] = gC[0][d] - gep;
    gz[1][d] = gC[1][d] - T * gep - gev;
    gz[2][d] = 0.5 * gC[2][d] - 0.5 * T2 * gep - T * gev - gea;
    gz[3][d] = gep;
    gz[4][d] = gev;
    gz[5][d] = gea;
  }
  // S_p = dW/d(state of joint p) = gz_{p-1}[3:6] + gz_p[0:3]   (lanes 1..M-1)
  double S[3][D];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int d = 0; d < D; ++d) S[k][d] = from_prev(gz[3 + k][d], 0.0) + gz[k][d];

  double lam[2][D];
#pragma unroll
  for (int d = 0; d < D; ++d) lam[0][d] = lam[1][d] = 0.0;
  if (M > 1) {
    double Lo[2][2], Di[2][2], Up[2][2];
    joint_blocks(t, Lo, Di, Up);
    // transposed system: row p of K^T has Up_{p-1}^T, Di_p^T, Lo_{p+1}^T
    double LoT[2][2], DiT[2][2], UpT[2][2], R[2][D], z0[2][D], y[2][D];
    LoT[0][0] = 24.0 * t.a2;  LoT[0][1] = -168.0 * t.a3;   // Up_{p-1}^T (piece p-1 = "a")
    LoT[1][0] = -3.0 * t.a1;  LoT[1][1] = 24.0 * t.a2;
    DiT[0][0] = Di[0][0]; DiT[0][1] = Di[1][0]; DiT[1][0] = Di[0][1]; DiT[1][1] = Di[1][1];
    UpT[0][0] = -24.0 * t.i2; UpT[0][1] = -168.0 * t.i3;   // Lo_{p+1}^T (piece p = "b")
    UpT[1][0] = -3.0 * t.i1;  UpT[1][1] = -24.0 * t.i2;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      R[0][d] = S[1][d];
      R[1][d] = S[2][d];
      z0[0][d] = z0[1][d] = 0.0;
    }
    block_thomas<D>(M, LoT, DiT, UpT, R, z0, z0, y);
#pragma unroll
    for (int d = 0; d < D; ++d) {
      lam[0][d] = (lane >= 1 && lane < M) ? y[0][d] : 0.0;
      lam[1][d] = (lane >= 1 && lane < M) ? y[1][d] : 0.0;
    }
  }
  // dW/dq: G[6i+3] of the reference (:506-508)
  double Gt[3][D];  // sensitivity wrt the tail state (lane M-1), = G[-3:] of the reference
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const double l1 = lam[0][d], l2 = lam[1][d];
    const double dp_prev = -60.0 * t.a3 * l1 - 360.0 * t.a4 * l2;
    const double dp_here = (60.0 * t.a3 + 60.0 * t.i3) * l1 + (360.0 * t.a4 - 360.0 * t.i4) * l2;
    const double dp_next = -60.0 * t.i3 * l1 + 360.0 * t.i4 * l2;
    const double from_left = from_prev(dp_next, 0.0);   // joint p-1 pushes on p_{p}
    const double from_right = from_next(dp_prev, 0.0);  // joint p+1 pushes on p_{p}
    gq[d] = S[0][d] - dp_here - (lane >= 2 ? from_left : 0.0) - (lane + 1 <= M - 1 ? from_right : 0.0);
    // tail sensitivity on lane M-1: S_M = gz[3:6] of the last piece, minus joint M-1's pull
    const double lt1 = (M > 1) ? l1 : 0.0, lt2 = (M > 1) ? l2 : 0.0;
    Gt[0][d] = gz[3][d] - (-60.0 * t.i3 * lt1 + 360.0 * t.i4 * lt2);
    Gt[1][d] = gz[4][d] - (24.0 * t.i2 * lt1 - 168.0 * t.i3 * lt2);
    Gt[2][d] = gz[5][d] - (-3.0 * t.i1 * lt1 + 24.0 * t.i2 * lt2);
  }
  // dW/dT (:511-533)
  double gTt = gT;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const double v1 = t.V1[d], a1 = t.A1[d], je = jerk_end[d];
    gTt -= gz[3][d] * v1 + gz[4][d] * a1 + gz[5][d] * je;
    // joint p+1 (this piece ends there): rows +je.Z, +se.Z
    const double ln1 = from_next(lam[0][d], 0.0), ln2 = from_next(lam[1][d], 0.0);
    if (lane + 1 <= M - 1) {
      const double d_je = snap_end[d] - (60.0 * t.i3 * v1 - 36.0 * t.i2 * a1 + 9.0 * t.i1 * je);
      const double d_se = crackle[d] - (360.0 * t.i4 * v1 - 192.0 * t.i3 * a1 + 36.0 * t.i2 * je);
      gTt -= ln1 * d_je + ln2 * d_se;
    }
    // joint p (this piece starts there): rows -js.Z, -ss.Z
    if (lane >= 1) {
      const double d_js = -(60.0 * t.i3 * v1 - 24.0 * t.i2 * a1 + 3.0 * t.i1 * je);
      const double d_ss = -(-360.0 * t.i4 * v1 + 168.0 * t.i3 * a1 - 24.0 * t.i2 * je);
      gTt += lam[0][d] * d_js + lam[1][d] * d_ss;
    }
  }
  // the reference evaluates the tail rows' d/dT with the previous piece's duration (:528-533)
  {
    const double Ts = from_prev(T, T);
    if (prm.stale_T && M >= 2 && lane == M - 1) {
      const double S2 = Ts * Ts, S3 = S2 * Ts, S4 = S2 * S2;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const double c1 = t.c[1][d], c2 = t.c[2][d], c3 = t.c[3][d], c4 = t.c[4][d], c5 = t.c[5][d];
        const double velL = c1 + 2.0 * T * c2 + 3.0 * T2 * c3 + 4.0 * T3 * c4 + 5.0 * T4 * c5;
        const double velS = c1 + 2.0 * Ts * c2 + 3.0 * S2 * c3 + 4.0 * S3 * c4 + 5.0 * S4 * c5;
        const double accL = 2.0 * c2 + 6.0 * T * c3 + 12.0 * T2 * c4 + 20.0 * T3 * c5;
        const double accS = 2.0 * c2 + 6.0 * Ts * c3 + 12.0 * S2 * c4 + 20.0 * S3 * c5;
        const double jrkL = 6.0 * c3 + 24.0 * T * c4 + 60.0 * T2 * c5;
        const double jrkS = 6.0 * c3 + 24.0 * Ts * c4 + 60.0 * S2 * c5;
        gTt += Gt[0][d] * (velL - velS) + Gt[1][d] * (accL - accS) + Gt[2][d] * (jrkL - jrkS);
      }
    }
  }
  // get_grad_T2tau (:485-492)
  const double ex = exp(-t.tau);
  // `(1 + math.exp(-tau))**2` (:490) is a Python-float power too: OverflowError beyond sqrt(DBL_MAX)
  if (lane < M && (1.0 + ex) > 1.3407807929942596e154) pow_overflow = 1;
  gtau = gTt * (prm.T_max - prm.T_min) * ex / ((1.0 + ex) * (1.0 + ex));
  return __any(pow_overflow) ? 4 : 0;
}

}  // namespace neo
