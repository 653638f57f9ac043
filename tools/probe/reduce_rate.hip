// Cost of a wave-wide fp32 sum on gfx950, as throughput (8 independent sums in flight) and inside a dependent chain, for
// w = 1..3 wavefronts per SIMD; with and without independent v_fma_f32 work beside it (does the matrix pipe overlap?).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/reduce_rate.hip -o tools/probe/_build/reduce_rate
// Variants:
//   0  six DPP adds (row_shr 1,2,4,8, row_bcast15, row_bcast31) + v_readlane 63   (csrc/neo_device.hpp wave_sum)
//   1  v_mfma_f32_16x16x4_f32 (ones x v: lane l gets v[l%16] + v[l%16+16] + v[l%16+32] + v[l%16+48]) + four DPP adds
//      (row_ror 8,4,2,1): total in every lane, no v_readlane
//   2  four DPP adds (row_ror) + v_permlane16_swap + add + v_permlane32_swap + add: total in every lane
//   3  four DPP adds (row_ror) + the MFMA last (rows summed by the matrix pipe)
//   4  variant 0 written in assembly: the row broadcasts as one v_add_f32_dpp under a row mask each (7 instructions, not 9)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dppf(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL, int RM>
__device__ __forceinline__ float dppf_any(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, RM, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ float dppf_ror(float v) {  // rotations have no invalid lanes
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), CTRL, 0xf, 0xf, false));
}

template <int KIND>
__device__ __forceinline__ float reduce(float v) {
  if constexpr (KIND == 0) {
    v += dppf<0x111>(v);
    v += dppf<0x112>(v);
    v += dppf<0x114>(v);
    v += dppf<0x118>(v);
    v += dppf_any<0x142, 0xa>(v);
    v += dppf_any<0x143, 0xc>(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  } else if constexpr (KIND == 1) {
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    const f4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, v, z, 0, 0, 0);
    float s = d[0];
    s += dppf_ror<0x128>(s);
    s += dppf_ror<0x124>(s);
    s += dppf_ror<0x122>(s);
    s += dppf_ror<0x121>(s);
    return s;
  } else if constexpr (KIND == 2) {
    float s = v;
    s += dppf_ror<0x128>(s);
    s += dppf_ror<0x124>(s);
    s += dppf_ror<0x122>(s);
    s += dppf_ror<0x121>(s);
    {
      const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
      s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
      const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
      s = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    return s;
  } else if constexpr (KIND == 3) {
    float s = v;
    s += dppf_ror<0x128>(s);
    s += dppf_ror<0x124>(s);
    s += dppf_ror<0x122>(s);
    s += dppf_ror<0x121>(s);
    const f4 z = {0.f, 0.f, 0.f, 0.f};
    const f4 d = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, s, z, 0, 0, 0);
    return d[0];
  } else {
    // variant 0 with the two row broadcasts as ONE v_add_f32_dpp each (rows the mask leaves out keep their value): the
    // builtin cannot express "add under a row mask", the compiler emits v_mov_b32_dpp + v_add_f32
    asm("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"
        "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  }
}

// CHAIN = false: 8 independent sums per step; true: each sum feeds the next (the two-loop recursion's shape).
// FILL = independent v_fma_f32 per sum beside it.
template <int KIND, bool CHAIN, int FILL>
__global__ __launch_bounds__(64) void k(float *out, int iters, float seed) {
  float a[8], f[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed * 1e-3f + (float)(threadIdx.x & 7) * 1e-4f + i; f[i] = seed + i; }
  const float m = 0.999f, c = 0.001f;
  float carry = seed;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (CHAIN) {
        carry = reduce<KIND>(__builtin_fmaf(a[i], carry, c)) * 1e-3f;
      } else {
        a[i] = reduce<KIND>(a[i]) * (1.0f / 64.0f) + c * (float)threadIdx.x;
      }
#pragma unroll
      for (int q = 0; q < FILL; ++q) f[(i + q) & 7] = __builtin_fmaf(f[(i + q) & 7], m, c);
    }
  }
  float s = carry;
  for (int i = 0; i < 8; ++i) s += a[i] + f[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

__global__ void check(float *out) {
  const float v = (float)threadIdx.x;
  out[threadIdx.x] = reduce<0>(v);
  out[64 + threadIdx.x] = reduce<1>(v);
  out[128 + threadIdx.x] = reduce<2>(v);
  out[192 + threadIdx.x] = reduce<3>(v);
  out[256 + threadIdx.x] = reduce<4>(v);
}

template <int KIND, bool CHAIN, int FILL>
void run(const char *name) {
  float *out;
  hipMalloc(&out, 1024 * 8 * 64 * sizeof(float));
  int dev; hipGetDevice(&dev);
  int clk; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, dev);
  for (int w : {1, 2, 3}) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND, CHAIN, FILL><<<1024 * w, 64>>>(out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND, CHAIN, FILL><<<1024 * w, 64>>>(out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double sums_per_simd = (double)w * iters * 8;
    printf("%-28s %s fill %2d  %d wave(s)/SIMD: %7.3f ms, %6.1f cycles per sum per SIMD (at %.0f MHz)\n", name,
           CHAIN ? "chain" : "indep", FILL, w, ms, ms * 1e-3 * clk * 1e3 / sums_per_simd, clk / 1e3);
  }
  hipFree(out);
}

template <int KIND>
void all(const char *name) {
  run<KIND, false, 0>(name);
  run<KIND, false, 16>(name);
  run<KIND, true, 0>(name);
  run<KIND, true, 16>(name);
}

int main() {
  // correctness of the variants first: sum of the lane numbers = 2016 in the lanes that hold the result
  {
    float *out; hipMalloc(&out, 5 * 64 * sizeof(float));
    check<<<1, 64>>>(out);
    float h[5 * 64]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    for (int kd = 0; kd < 5; ++kd) printf("variant %d: lane 0 %.1f  lane 17 %.1f  lane 63 %.1f (2016)\n", kd, h[kd * 64], h[kd * 64 + 17], h[kd * 64 + 63]);
    hipFree(out);
  }
  run<0, false, 16>("fill only reference: see fill 0 rows");
  all<0>("dpp6+readlane");
  all<1>("mfma+dpp4");
  all<2>("dpp4+swap16+swap32");
  all<3>("dpp4+mfma");
  all<4>("dpp6 (asm, fused bcast)+readlane");
  return 0;
}
