// probe: variants of the stand-alone ESDF sample kernel (D = 3, fp32 arithmetic, fp32 linear field), timed
// with HIP events.  Built by tools/gpu_sample_bench.py into tools/probe/_build/libsample_variants.so.
// Not part of the product; winners move into neo_device.hpp / neo_kernels.hpp.
#include "../../neo-planner_amd/csrc/neo_device.hpp"
#include <cstdio>
#include <type_traits>

using namespace neo;

namespace {

struct Field {
  __amdgpu_buffer_rsrc_t rsrc;
  float inv_res;
  float off[3];   // -origin * inv_res - 0.5
  float hi[3];    // n - 0.5
  int nm2[3];     // n - 2
  unsigned nx, ny, nxny;
};

template <int U>
struct Stage {
  float s[U], vv[U], fr[U][3];
  bool inside[U];
  float raw[U][8];
};

__device__ __forceinline__ float med3f(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }

// position + velocity by Horner on p and p' together: no pre-scaled coefficient copies
__device__ __forceinline__ void horner_pv(const float (&c)[6][3], float s, float (&p)[3], float (&v)[3]) {
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float a = c[5][d], b = c[5][d];
    a = fmaf(a, s, c[4][d]); b = fmaf(b, s, a);
    a = fmaf(a, s, c[3][d]); b = fmaf(b, s, a);
    a = fmaf(a, s, c[2][d]); b = fmaf(b, s, a);
    a = fmaf(a, s, c[1][d]); b = fmaf(b, s, a);
    a = fmaf(a, s, c[0][d]);
    p[d] = a; v[d] = b;
  }
}

template <int U>
__device__ __forceinline__ void stage(const Field &f, const float (&c)[6][3], int r, int L, int it0, double delta_t,
                                      float vmax2, Stage<U> &st) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int j = r + (it0 + u) * L;
    const float s = (float)((double)j * delta_t);
    st.s[u] = s;
    float p[3], v[3];
    horner_pv(c, s, p, v);
    st.vv[u] = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] - vmax2;
    bool in = true;
    int i0[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float um = fmaf(p[k], f.inv_res, f.off[k]);
      if (!(um >= -0.5f && um < f.hi[k])) in = false;
      const int i = min(max((int)floorf(um), 0), f.nm2[k]);
      i0[k] = i;
      st.fr[u][k] = med3f(um - (float)i, 0.0f, 1.0f);
    }
    st.inside[u] = in;
    unsigned base = __umul24(__umul24((unsigned)i0[2], f.ny) + (unsigned)i0[1], f.nx) + (unsigned)i0[0];
    base = in ? base * 4u : 0u;
    const auto q0 = __builtin_amdgcn_raw_buffer_load_b64(f.rsrc, (int)base, 0, 0);
    const auto q1 = __builtin_amdgcn_raw_buffer_load_b64(f.rsrc, (int)base, (int)(f.nx * 4u), 0);
    const auto q2 = __builtin_amdgcn_raw_buffer_load_b64(f.rsrc, (int)base, (int)(f.nxny * 4u), 0);
    const auto q3 = __builtin_amdgcn_raw_buffer_load_b64(f.rsrc, (int)base, (int)((f.nxny + f.nx) * 4u), 0);
    st.raw[u][0] = __uint_as_float(q0[0]); st.raw[u][1] = __uint_as_float(q0[1]);
    st.raw[u][2] = __uint_as_float(q1[0]); st.raw[u][3] = __uint_as_float(q1[1]);
    st.raw[u][4] = __uint_as_float(q2[0]); st.raw[u][5] = __uint_as_float(q2[1]);
    st.raw[u][6] = __uint_as_float(q3[0]); st.raw[u][7] = __uint_as_float(q3[1]);
  }
}

// accumulators of one lane: slots 0..17 = aC[k][d], 18 = aT, 19 = aF, 20 = aK
struct AccReg {
  float v[21];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 21; ++i) v[i] = 0.0f;
  }
  __device__ __forceinline__ void add(int i, float x) { v[i] += x; }
  __device__ __forceinline__ float get(int i) const { return v[i]; }
};
// the same in LDS (slot-major, one column per lane: conflict-free); ds_add_f32 to the lane's own cells keeps
// program order, so the sums are as reproducible as in registers
struct AccLds {
  float *base;  // &lds[lane]
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < 21; ++i) base[i * kWave] = 0.0f;
  }
  __device__ __forceinline__ void add(int i, float x) {
    __hip_atomic_fetch_add(base + i * kWave, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ float get(int i) const { return base[i * kWave]; }
};

template <int U, class Acc>
__device__ __forceinline__ void consume(const Field &f, const float (&c)[6][3], int r, int L, int it0, int ns,
                                        float inv_ns, float dt, float safe, float w2, float w3, const Stage<U> &st,
                                        Acc &A) {
  float vd[U];
  bool viol = false;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const float *q = st.raw[u];
    const float fx = st.fr[u][0], fy = st.fr[u][1], fz = st.fr[u][2];
    const float c00 = q[0] + fx * (q[1] - q[0]), c10 = q[2] + fx * (q[3] - q[2]);
    const float c01 = q[4] + fx * (q[5] - q[4]), c11 = q[6] + fx * (q[7] - q[6]);
    const float c0 = c00 + fy * (c10 - c00), c1 = c01 + fy * (c11 - c01);
    const float dist = st.inside[u] ? c0 + fz * (c1 - c0) : 10000.0f;
    vd[u] = safe - dist;
    const int j = r + (it0 + u) * L;
    if (j < ns && (st.vv[u] > 0.0f || vd[u] > 0.0f)) viol = true;
  }
  if (viol) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = r + (it0 + u) * L;
      if (j >= ns) continue;
      const float s = st.s[u];
      const float omg = (j == 0 || j == ns - 1) ? 0.5f : 1.0f;
      const float s2 = s * s, s3 = s2 * s, s4 = s2 * s2, s5 = s4 * s;
      float p[3], vel[3];
      horner_pv(c, s, p, vel);
      if (st.vv[u] > 0.0f) {
        const float vq = st.vv[u];
        A.add(19, omg * dt * vq * vq * vq);
        float av = 0.0f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const float acc = 2.0f * c[2][d] + s * (6.0f * c[3][d] + s * (12.0f * c[4][d] + s * (20.0f * c[5][d])));
          av += acc * vel[d];
        }
        const float dK = 3.0f * dt * omg * vq * vq;
        const float b1[6] = {0.0f, 1.0f, 2.0f * s, 3.0f * s2, 4.0f * s3, 5.0f * s4};
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          const float uu = w2 * dK * 2.0f * vel[d];
#pragma unroll
          for (int k = 1; k < 6; ++k) A.add(k * 3 + d, b1[k] * uu);
        }
        A.add(18, w2 * (omg * vq * vq * vq * inv_ns + dK * 2.0f * av * (float)j * inv_ns));
      }
      if (vd[u] > 0.0f) {
        const float *q = st.raw[u];
        const float fx = st.fr[u][0], fy = st.fr[u][1], fz = st.fr[u][2];
        const float dx00 = q[1] - q[0], dx10 = q[3] - q[2], dx01 = q[5] - q[4], dx11 = q[7] - q[6];
        const float c00 = q[0] + fx * dx00, c10 = q[2] + fx * dx10, c01 = q[4] + fx * dx01, c11 = q[6] + fx * dx11;
        const float c0 = c00 + fy * (c10 - c00), c1 = c01 + fy * (c11 - c01);
        const float dx0 = dx00 + fy * (dx10 - dx00), dx1 = dx01 + fy * (dx11 - dx01);
        const float dy0 = c10 - c00, dy1 = c11 - c01;
        float g[3];
        g[0] = (dx0 + fz * (dx1 - dx0)) * f.inv_res;
        g[1] = (dy0 + fz * (dy1 - dy0)) * f.inv_res;
        g[2] = (c1 - c0) * f.inv_res;
        const float vq = vd[u];
        A.add(20, omg * dt * vq * vq * vq);
        const float dK = 3.0f * dt * omg * vq * vq;
        const float b0[6] = {1.0f, s, s2, s3, s4, s5};
        float gv = 0.0f;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          gv += g[d] * vel[d];
          const float uu = -(w3 * dK * g[d]);
#pragma unroll
          for (int k = 0; k < 6; ++k) A.add(k * 3 + d, b0[k] * uu);
        }
        A.add(18, w3 * (omg * vq * vq * vq * inv_ns + dK * (-gv) * (float)j * inv_ns));
      }
    }
  }
}

// sum over the L lanes of a piece, result valid in the piece's first lane
__device__ __forceinline__ float fold_dpp(float v, int L, int r) {
  if (L <= 4) {
    float acc = v;
    for (int i = 1; i < L; ++i) acc = v + dpp_f<0x130>(acc);  // wave_shl:1
    return acc;
  }
  for (int sft = 1; sft < L; sft <<= 1) {
    const float o = __shfl_down(v, sft, kWave);
    if (r + sft < L) v += o;
  }
  return v;
}

template <int U, int OCC, bool PIPE, bool LDSACC>
__global__ __launch_bounds__(kWave, OCC) void sample_v2(int B, int M, DevParams prm, Map3D map,
                                                       const double *__restrict__ coeffs, const double *__restrict__ ts,
                                                       double *__restrict__ costs2, double *__restrict__ grad_C,
                                                       double *__restrict__ grad_T) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const int lane = lane_id();
  int L = kWave / M;
  L = L < 1 ? 1 : L;
  const int piece = (lane * ((65536 + L - 1) / L)) >> 16;
  const int r = lane - piece * L;
  const bool act = piece < M;
  const double T = act ? ts[(size_t)b * M + piece] : 1.0;
  const int ns = act ? (int)(T / prm.delta_t) : 0;
  float c[6][3];
  {
    const double2 *src = reinterpret_cast<const double2 *>(coeffs + ((size_t)b * 6 * M + 6 * (act ? piece : 0)) * 3);
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const double2 v = src[q];
      const int e0 = 2 * q, e1 = 2 * q + 1;
      c[e0 / 3][e0 % 3] = act ? (float)v.x : 0.0f;
      c[e1 / 3][e1 % 3] = act ? (float)v.y : 0.0f;
    }
  }
  Field f;
  f.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(map.data), 0, (int)map.bytes, 0x00020000);
  f.inv_res = (float)(1.0 / map.res);
  f.off[0] = (float)(-map.ox / map.res - 0.5);
  f.off[1] = (float)(-map.oy / map.res - 0.5);
  f.off[2] = (float)(-map.oz / map.res - 0.5);
  f.hi[0] = (float)map.nx - 0.5f; f.hi[1] = (float)map.ny - 0.5f; f.hi[2] = (float)map.nz - 0.5f;
  f.nm2[0] = map.nx - 2; f.nm2[1] = map.ny - 2; f.nm2[2] = map.nz - 2;
  f.nx = (unsigned)map.nx; f.ny = (unsigned)map.ny; f.nxny = (unsigned)map.nx * (unsigned)map.ny;

  const int iters = (prm.dbg & 1) ? 0 : ((prm.dbg & 2) ? min(2, wave_max_nonneg((ns + L - 1) / L)) : wave_max_nonneg((ns + L - 1) / L));
  const float dt = (float)prm.delta_t, vmax2 = (float)(prm.v_max * prm.v_max), safe = (float)prm.safe_dis;
  const float w2 = (float)prm.w[2], w3 = (float)prm.w[3];
  const float inv_ns = ns > 0 ? 1.0f / (float)ns : 0.0f;
  __shared__ float acc_lds[LDSACC ? 21 * kWave : 1];
  typedef typename std::conditional<LDSACC, AccLds, AccReg>::type Acc;
  Acc A;
  if constexpr (LDSACC) A.base = acc_lds + lane;
  A.zero();

  if (PIPE) {
    Stage<U> s0, s1;
    if (iters > 0) stage<U>(f, c, r, L, 0, prm.delta_t, vmax2, s0);
    for (int it0 = 0; it0 < iters; it0 += 2 * U) {
      const bool more1 = it0 + U < iters;
      if (more1) stage<U>(f, c, r, L, it0 + U, prm.delta_t, vmax2, s1);
      consume<U, Acc>(f, c, r, L, it0, ns, inv_ns, dt, safe, w2, w3, s0, A);
      if (more1) {
        if (it0 + 2 * U < iters) stage<U>(f, c, r, L, it0 + 2 * U, prm.delta_t, vmax2, s0);
        consume<U, Acc>(f, c, r, L, it0 + U, ns, inv_ns, dt, safe, w2, w3, s1, A);
      }
    }
  } else {
    for (int it0 = 0; it0 < iters; it0 += U) {
      Stage<U> s0;
      stage<U>(f, c, r, L, it0, prm.delta_t, vmax2, s0);
      consume<U, Acc>(f, c, r, L, it0, ns, inv_ns, dt, safe, w2, w3, s0, A);
    }
  }

  float gc[6][3];
#pragma unroll
  for (int k = 0; k < 6; ++k)
#pragma unroll
    for (int d = 0; d < 3; ++d) gc[k][d] = fold_dpp(A.get(k * 3 + d), L, r);
  const float gT = fold_dpp(A.get(18), L, r), pf = fold_dpp(A.get(19), L, r), pk = fold_dpp(A.get(20), L, r);
  const bool first = act && r == 0;
  const double cf = wave_sum(first ? (double)pf : 0.0), ck = wave_sum(first ? (double)pk : 0.0);
  if (first) {
    double2 *dst = reinterpret_cast<double2 *>(grad_C + ((size_t)b * 6 * M + 6 * piece) * 3);
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int e0 = 2 * q, e1 = 2 * q + 1;
      dst[q] = make_double2((double)gc[e0 / 3][e0 % 3], (double)gc[e1 / 3][e1 % 3]);
    }
    grad_T[(size_t)b * M + piece] = (double)gT;
  }
  if (lane == 0) {
    costs2[(size_t)b * 2 + 0] = cf;
    costs2[(size_t)b * 2 + 1] = ck;
  }
}

}  // namespace

typedef void (*kern_t)(int, int, DevParams, Map3D, const double *, const double *, double *, double *, double *);

extern "C" int probe_variants() { return 10; }

extern "C" const char *probe_name(int v) {
  static const char *names[] = {"U1 occ4 reg",  "U2 occ3 reg",      "U2 occ4 lds",      "U1 occ4 lds",      "U1 occ4 pipe lds",
                                "U2 occ3 lds",  "U2 occ3 pipe lds", "U2 occ4 pipe lds", "U4 occ3 lds", "U4 occ4 lds"};
  return names[v];
}

// returns mean microseconds per launch over `reps` launches (HIP events on a stream of its own), < 0 on error
extern "C" double probe_run(int variant, int B, int M, const double *prm9 /* v_max,T_min,T_max,safe,dt,w0..w3 */,
                            const void *field, int nx, int ny, int nz, double res, const double *origin,
                            const double *coeffs, const double *ts, double *costs2, double *gC, double *gT, int reps) {
  DevParams p{};
  p.dbg = variant >> 8;
  variant &= 0xff;
  p.v_max = prm9[0]; p.T_min = prm9[1]; p.T_max = prm9[2]; p.safe_dis = prm9[3]; p.delta_t = prm9[4];
  for (int k = 0; k < 4; ++k) p.w[k] = prm9[5 + k];
  p.derive();
  Map3D m{};
  m.data = field; m.nx = nx; m.ny = ny; m.nz = nz; m.layout = 0; m.res = res;
  m.ox = origin[0]; m.oy = origin[1]; m.oz = origin[2];
  m.bytes = (unsigned)((size_t)nx * ny * nz * 4 + 256);
  m.derive();
  kern_t k = nullptr;
  switch (variant) {
    case 0: k = sample_v2<1, 4, false, false>; break;
    case 1: k = sample_v2<2, 3, false, false>; break;
    case 2: k = sample_v2<2, 4, false, true>; break;
    case 3: k = sample_v2<1, 4, false, true>; break;
    case 4: k = sample_v2<1, 4, true, true>; break;
    case 5: k = sample_v2<2, 3, false, true>; break;
    case 6: k = sample_v2<2, 3, true, true>; break;
    case 7: k = sample_v2<2, 4, true, true>; break;
    case 8: k = sample_v2<4, 3, false, true>; break;
    case 9: k = sample_v2<4, 4, false, true>; break;
    default: return -1.0;
  }
  hipStream_t s;
  if (hipStreamCreate(&s) != hipSuccess) return -2.0;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(B), dim3(kWave), 0, s, B, M, p, m, coeffs, ts, costs2, gC, gT);
  hipStreamSynchronize(s);
  float ms = 0.f;
  hipError_t e = hipSuccess;
  if (p.dbg & 4) {  // one event pair per launch, like the library's profiling scope
    for (int i = 0; i < reps; ++i) {
      hipEventRecord(e0, s);
      hipLaunchKernelGGL(k, dim3(B), dim3(kWave), 0, s, B, M, p, m, coeffs, ts, costs2, gC, gT);
      hipEventRecord(e1, s);
      e = hipStreamSynchronize(s);
      float one = 0.f;
      hipEventElapsedTime(&one, e0, e1);
      ms += one;
    }
  } else {
    hipEventRecord(e0, s);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(B), dim3(kWave), 0, s, B, M, p, m, coeffs, ts, costs2, gC, gT);
    hipEventRecord(e1, s);
    e = hipStreamSynchronize(s);
    hipEventElapsedTime(&ms, e0, e1);
  }
  hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(s);
  if (e != hipSuccess) return -3.0;
  return 1e3 * ms / reps;
}
