// VALU issue cost of the integer / conversion / select instructions the kernels' index arithmetic is made of (gfx950): cycles
// per wave64 instruction and SIMD at 1, 2, 4 wavefronts per SIMD, 8 independent chains, each instruction through inline
// assembly so that the named opcode is what runs.  Round 6: floor / clamp / brick address written as single instructions made
// the ESDF-lookup kernel 10 % SLOWER with 8 % fewer instructions -- which of them is the expensive one?
//   hipcc --offload-arch=gfx950 -O3 tools/probe/valu_rate2.hip -o valu_rate2 && ./valu_rate2
#include <hip/hip_runtime.h>
#include <cstdio>

#define OP1(name, text)                                                                          \
  struct name {                                                                                  \
    static constexpr const char *label = #name;                                                  \
    static __device__ __forceinline__ void step(unsigned &a, unsigned b, unsigned c) {           \
      asm volatile(text : "+v"(a) : "v"(b), "v"(c));                                             \
    }                                                                                            \
  };
OP1(v_fma_f32, "v_fma_f32 %0, %0, %1, %2")
OP1(v_add_u32, "v_add_u32 %0, %0, %1")
OP1(v_and_b32, "v_and_b32 %0, %0, %1")
OP1(v_lshrrev_b32, "v_lshrrev_b32 %0, 1, %0")
OP1(v_lshl_add_u32, "v_lshl_add_u32 %0, %0, 2, %1")
OP1(v_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2")
OP1(v_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
OP1(v_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
OP1(v_floor_f32, "v_floor_f32 %0, %0")
OP1(v_cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
OP1(v_cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
OP1(v_cvt_flr_i32_f32, "v_cvt_flr_i32_f32 %0, %0")
OP1(v_max_i32, "v_max_i32 %0, %0, %1")
OP1(v_med3_i32, "v_med3_i32 %0, %0, %1, %2")
OP1(v_med3_f32, "v_med3_f32 %0, %0, %1, %2")
OP1(v_min3_f32, "v_min3_f32 %0, %0, %1, %2")
OP1(v_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc")
OP1(v_cmp_lt_f32, "v_cmp_lt_f32 vcc, %0, %1")
OP1(v_cndmask_sgpr_mask, "v_cndmask_b32 %0, %0, %1, s[20:21]")
OP1(v_cmp_then_cndmask, "v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc")
OP1(v_cmp_sgpr_then_cndmask, "v_cmp_lt_f32 s[20:21], %1, %2\n\tv_cndmask_b32 %0, %0, %1, s[20:21]")
OP1(v_add_f32_dpp, "v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
OP1(v_mov_b32_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0")
OP1(v_readlane, "v_readlane_b32 s20, %0, 5")
OP1(v_fmac_f32, "v_fmac_f32 %0, %1, %2")
OP1(v_mul_f32, "v_mul_f32 %0, %0, %1")
OP1(v_fma_f32_sgpr, "v_fma_f32 %0, %0, s20, %2")
OP1(v_max_f32, "v_max_f32 %0, %0, %1")
OP1(v_mul_f32_literal, "v_mul_f32 %0, 0x40a00000, %0")
OP1(v_mul_f32_inline_const, "v_mul_f32 %0, 2.0, %0")
OP1(v_fma_f32_inline_const, "v_fma_f32 %0, %0, 2.0, %2")
OP1(v_fmac_f32_sgpr, "v_fmac_f32 %0, s20, %1")
OP1(v_add_f32_sgpr, "v_add_f32 %0, s20, %0")
OP1(v_mul_f32_sgpr, "v_mul_f32 %0, s20, %0")
OP1(v_add_f32, "v_add_f32 %0, %0, %1")
OP1(v_cmp_lt_i32, "v_cmp_lt_i32 vcc, %0, %1")
OP1(v_min_f32, "v_min_f32 %0, %0, %1")
OP1(v_cvt_f32_f64ish, "v_cvt_f32_u32 %0, %0")
OP1(v_mov_b32, "v_mov_b32 %0, %1")
OP1(v_mbcnt_lo, "v_mbcnt_lo_u32_b32 %0, %1, %0")
OP1(v_readfirstlane, "v_readfirstlane_b32 s20, %0")
OP1(v_sub_f32_clamp, "v_sub_f32 %0, %0, %1 clamp")

template <class OP>
__global__ __launch_bounds__(64) void k(unsigned *out, int iters, unsigned seed) {
  unsigned a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
  unsigned b = seed * 3 + 1, c = seed + 7;
  unsigned long long wide[8];  // (v_mad_u64_u32 writes a register pair)
  for (int i = 0; i < 8; ++i) wide[i] = a[i];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) OP::step(a[i], b, c);
  }
  unsigned s = 0;
  for (int i = 0; i < 8; ++i) s += a[i] + (unsigned)wide[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <>
__global__ __launch_bounds__(64) void k<v_mad_u64_u32>(unsigned *out, int iters, unsigned seed) {
  unsigned long long a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + i + threadIdx.x;
  unsigned b = seed * 3 + 1, c = seed + 7;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
  }
  unsigned s = 0;
  for (int i = 0; i < 8; ++i) s += (unsigned)a[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <class OP>
void run() {
  unsigned *out;
  (void)hipMalloc(&out, 1024 * 8 * 64 * sizeof(unsigned));
  int dev, clk;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, dev);
  printf("%-20s", OP::label);
  for (int w : {1, 2, 4}) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP><<<1024 * w, 64>>>(out, 10, 1u);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<OP><<<1024 * w, 64>>>(out, iters, 1u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("  %d/SIMD %5.2f", w, ms * 1e-3 * clk * 1e3 / ((double)w * iters * 64));
  }
  printf("   cycles per wave-instruction per SIMD at %d MHz (nominal)\n", clk / 1000);
  (void)hipFree(out);
}
int main() {
  run<v_fma_f32>(); run<v_add_u32>(); run<v_and_b32>(); run<v_lshrrev_b32>(); run<v_lshl_add_u32>(); run<v_mad_u32_u24>();
  run<v_mul_lo_u32>(); run<v_mad_u64_u32>(); run<v_floor_f32>(); run<v_cvt_i32_f32>(); run<v_cvt_f32_i32>(); run<v_cvt_flr_i32_f32>();
  run<v_max_i32>(); run<v_med3_i32>(); run<v_med3_f32>(); run<v_min3_f32>(); run<v_cndmask_b32>(); run<v_cmp_lt_f32>();
  run<v_mov_b32>(); run<v_mbcnt_lo>(); run<v_readfirstlane>(); run<v_sub_f32_clamp>();
  printf("-- selects: the first row above reads VCC that no instruction of the loop writes; as the kernels use them:\n");
  run<v_cndmask_sgpr_mask>(); run<v_cmp_then_cndmask>(); run<v_cmp_sgpr_then_cndmask>();
  printf("-- lane operations and the fp32 pipe:\n");
  run<v_add_f32_dpp>(); run<v_mov_b32_dpp>(); run<v_readlane>(); run<v_fmac_f32>(); run<v_mul_f32>(); run<v_fma_f32_sgpr>(); run<v_max_f32>();
  printf("-- constants and scalar operands on the fp32 pipe:\n");
  run<v_mul_f32_literal>(); run<v_mul_f32_inline_const>(); run<v_fma_f32_inline_const>(); run<v_fmac_f32_sgpr>();
  run<v_add_f32_sgpr>(); run<v_mul_f32_sgpr>(); run<v_add_f32>(); run<v_cmp_lt_i32>(); run<v_min_f32>();
  run<v_cvt_f32_f64ish>();
  return 0;
}
