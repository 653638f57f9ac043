// sector_probe.hip -- does the L2 ask the fabric ONCE for a 128-byte line whose two 64-byte halves are requested by two
// different load instructions of one wavefront (round 6: the corner-brick lookup's four 8-byte loads straddle the halves)?
// Every lane reads two 8-byte words of a pseudo-random 128-byte line of a buffer of `mb` megabytes:
//   mode 0: bytes 0 and 16 (same half)    mode 1: bytes 0 and 64 (the two halves, two instructions)
//   mode 2: bytes 0 and 64, but the second load only after the first has returned (dependent)
//   mode 3: ONE 16-byte load per lane, even lanes byte 0, odd lanes byte 64 of the line of lane & ~1 (both halves in one instruction)
//   ./sector_probe <mb> <waves> <rounds> <mode>;  counters: rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_MISS_sum TCC_HIT_sum TCP_TCC_READ_REQ_sum
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(64, 4) void probe(const float *buf, unsigned nline, int rounds, int mode, float *out) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(buf), 0, (int)(nline * 128u), 0x00020000);
  const unsigned lane = threadIdx.x, wave = blockIdx.x;
  unsigned h = wave * 2654435761u + lane * 40503u + 12345u;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    h = h * 1664525u + 1013904223u;
    unsigned line = (h >> 4) % nline;
    float v;
    if (mode == 3) {
      const unsigned l2 = __shfl(line, lane & ~1u, 64);
      const auto a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(l2 * 128u + (lane & 1u) * 64u), 0, 0);
      v = __uint_as_float(a[0]) + __uint_as_float(a[3]);
    } else {
      const auto a = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(line * 128u), 0, 0);
      unsigned off2 = line * 128u + (mode == 0 ? 16u : 64u);
      if (mode == 2) off2 += (__uint_as_float(a[0]) != 12345.678f ? 0u : 4u);
      const auto b = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)off2, 0, 0);
      v = __uint_as_float(a[0]) + __uint_as_float(a[1]) + __uint_as_float(b[0]) + __uint_as_float(b[1]);
    }
    acc += v;
    h ^= (unsigned)(v != 12345.678f ? 0u : 1u);  // the next address waits for this data (value never matches)
  }
  out[wave * 64 + lane] = acc;
}

int main(int argc, char **argv) {
  const size_t mb = argc > 1 ? atol(argv[1]) : 432;
  const int waves = argc > 2 ? atoi(argv[2]) : 4096, rounds = argc > 3 ? atoi(argv[3]) : 13, mode = argc > 4 ? atoi(argv[4]) : 0;
  const unsigned nline = (unsigned)(mb * 1024 * 1024 / 128);
  float *buf, *out;
  hipMalloc(&buf, (size_t)nline * 128);
  hipMemset(buf, 0, (size_t)nline * 128);
  hipMalloc(&out, (size_t)waves * 64 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(probe, dim3(waves), dim3(64), 0, 0, buf, nline, rounds, mode, out);
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe, dim3(waves), dim3(64), 0, 0, buf, nline, rounds, mode, out);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double us = 1e3 * ms / reps, lk = (double)waves * 64 * rounds / (mode == 3 ? 2 : 1);
  printf("{\"buffer_mb\": %zu, \"waves\": %d, \"rounds\": %d, \"mode\": %d, \"lines_touched\": %.0f, \"kernel_us\": %.2f, \"lines_per_us\": %.1f}\n",
         mb, waves, rounds, mode, lk, us, lk / us);
  return 0;
}
