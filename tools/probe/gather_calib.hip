// gather_calib.hip -- calibration of FETCH_SIZE and of the achievable rate for the ESDF kernel's ACCESS SHAPE
// (MI355X_MICROARCH.md: "other access widths are uncalibrated: calibrate on a known byte count in your own access
// pattern").  Every lane reads one 32-byte record (two adjacent buffer_load_dwordx4, as Lookup3D::load does in the
// yz-quad layout) at a pseudo-random 32-byte-aligned offset of a buffer of `mb` megabytes, `rounds` times with a
// dependent address chain (like the sample loop: next address only after the data arrived) or without.
//   ./gather_calib <mb> <waves> <rounds> <dependent 0|1> <stride_records: 0 = random, k = lane l reads record base + l * k>
// prints useful bytes, time, GB/s; run under `rocprofv3 --pmc FETCH_SIZE` / TCC_HIT_sum TCC_MISS_sum for the counters.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(64, 4) void gather(const float *buf, unsigned nrec, int rounds, int dependent, unsigned stride,
                                                float *out) {
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(buf), 0, (int)(nrec * 32u), 0x00020000);
  const unsigned lane = threadIdx.x, wave = blockIdx.x;
  unsigned h = wave * 2654435761u + lane * 40503u + 12345u;
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r) {
    h = h * 1664525u + 1013904223u;
    unsigned rec;
    if (stride == 0) {
      rec = (h >> 4) % nrec;
    } else {
      const unsigned base = ((wave * 7919u + r * 104729u) * 64u) % (nrec - 64u * stride);
      rec = base + lane * stride;
    }
    const auto lo = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(rec * 32u), 0, 0);
    const auto hi = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(rec * 32u + 16u), 0, 0);
    const float v = __uint_as_float(lo[0]) + __uint_as_float(lo[3]) + __uint_as_float(hi[1]) + __uint_as_float(hi[2]);
    acc += v;
    if (dependent) h ^= (unsigned)(v != 12345.678f ? 0u : 1u);  // the next address waits for this data (value never matches)
  }
  out[wave * 64 + lane] = acc;
}

int main(int argc, char **argv) {
  const size_t mb = argc > 1 ? atol(argv[1]) : 432;
  const int waves = argc > 2 ? atoi(argv[2]) : 4096, rounds = argc > 3 ? atoi(argv[3]) : 13, dep = argc > 4 ? atoi(argv[4]) : 1;
  const unsigned stride = argc > 5 ? atoi(argv[5]) : 0;
  const unsigned nrec = (unsigned)(mb * 1024 * 1024 / 32);
  float *buf, *out;
  hipMalloc(&buf, (size_t)nrec * 32);
  hipMemset(buf, 0, (size_t)nrec * 32);
  hipMalloc(&out, (size_t)waves * 64 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gather, dim3(waves), dim3(64), 0, 0, buf, nrec, rounds, dep, stride, out);
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gather, dim3(waves), dim3(64), 0, 0, buf, nrec, rounds, dep, stride, out);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double us = 1e3 * ms / reps, bytes = (double)waves * 64 * rounds * 32;
  printf("{\"buffer_mb\": %zu, \"waves\": %d, \"rounds\": %d, \"dependent\": %d, \"stride_records\": %u, \"lookups\": %.0f, "
         "\"useful_bytes\": %.0f, \"kernel_us\": %.2f, \"useful_GBps\": %.1f, \"lookups_per_us\": %.1f}\n",
         mb, waves, rounds, dep, stride, bytes / 32, bytes, us, bytes / us / 1e3, bytes / 32 / us);
  return 0;
}
