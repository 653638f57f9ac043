// VALU issue rate on gfx950: cycles per wave64 instruction and SIMD, for w = 1..4 wavefronts per SIMD, independent
// instruction streams (8 accumulators).  hipcc --offload-arch=gfx950 -O3 tools/probe/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ __launch_bounds__(64) void k(float *out, int iters, float seed) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 8];
  float a[8];
  f2 p[8];
  double d[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f2{a[i], a[i] + 1}; d[i] = a[i]; }
  const float m = 0.999f, c = 0.001f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (KIND == 0) a[i] = __builtin_fmaf(a[i], m, c);
        if (KIND == 1) p[i] = __builtin_elementwise_fma(p[i], f2{m, m}, f2{c, c});
        if (KIND == 2) d[i] = __builtin_fma(d[i], (double)m, (double)c);
        if (KIND == 3) a[i] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[i]), 0x111, 0xf, 0xf, true));
        if (KIND == 4) a[i] = __builtin_amdgcn_rcpf(a[i]);
        if (KIND == 5) a[i] += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a[(i + 1) & 7]), (i * 7 + r) & 63));  // readlane + add
        if (KIND == 6) a[i] = __int_as_float(__builtin_amdgcn_ds_bpermute((threadIdx.x + i + 1) << 2, __float_as_int(a[i])));
        if (KIND == 7) a[i] = (a[(i + 3) & 7] > m) ? a[i] + c : a[i];      // compare + select + add
        if (KIND == 8) {  // neighbour exchange through LDS: one ds_write_b32 + one ds_read_b32 (other lane)
          lds[threadIdx.x] = a[i];
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
          a[i] = lds[(threadIdx.x + i + 1) & 63];
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        }
        if (KIND == 9 && i < 2) {  // four values at once: ds_write_b128 + ds_read_b128
          *reinterpret_cast<f4 *>(lds + threadIdx.x * 4 + (i & 1) * 256) = f4{a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]};
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
          const f4 v = *reinterpret_cast<const f4 *>(lds + ((threadIdx.x + 3) & 63) * 4 + (i & 1) * 256);
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
          a[4 * i] = v.x; a[4 * i + 1] = v.y; a[4 * i + 2] = v.z; a[4 * i + 3] = v.w;
        }
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int KIND>
void run(const char *name) {
  float *out;
  hipMalloc(&out, 1024 * 8 * 64 * sizeof(float));
  int dev; hipGetDevice(&dev);
  int clk; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, dev);
  for (int w : {1, 2, 3, 4, 8}) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<1024 * w, 64>>>(out, 10, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<1024 * w, 64>>>(out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)w * iters * 64;
    printf("%-16s %d wave(s)/SIMD: %.3f ms, %.2f cycles per wave-instruction per SIMD at %.0f MHz\n", name, w, ms,
           ms * 1e-3 * clk * 1e3 / instr_per_simd, clk / 1e3);
  }
  hipFree(out);
}
int main() {
  run<0>("v_fma_f32");
  run<1>("v_pk_fma_f32");
  run<2>("v_fma_f64");
  run<3>("v_add_f32_dpp");
  run<4>("v_rcp_f32");
  run<5>("readlane+add");
  run<6>("ds_bpermute");
  run<7>("cmp+cndmask+add");
  run<8>("lds wr+rd b32");
  run<9>("lds wr+rd b128 (x1/4 instr)");
  return 0;
}
