// probe: do 8-byte buffer loads at 4-byte-aligned offsets return the right data on this GPU?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* p, int n, const int* idx, float* out, int m) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, n * 4, 0x00020000);
  auto v = __builtin_amdgcn_raw_buffer_load_b64(r, idx[i] * 4, 0, 0);
  out[2 * i] = __uint_as_float(v[0]);
  out[2 * i + 1] = __uint_as_float(v[1]);
}
int main() {
  const int n = 1 << 20, m = 4096;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  std::vector<int> idx(m);
  for (int i = 0; i < m; ++i) idx[i] = (i * 7919 + 1) % (n - 1);   // odd and even
  float *d, *o; int* di;
  hipMalloc(&d, n * 4); hipMalloc(&o, m * 8); hipMalloc(&di, m * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemcpy(di, idx.data(), m * 4, hipMemcpyHostToDevice);
  k<<<m / 256, 256>>>(d, n, di, o, m);
  hipError_t e = hipDeviceSynchronize();
  std::vector<float> r(2 * m);
  hipMemcpy(r.data(), o, m * 8, hipMemcpyDeviceToHost);
  int bad = 0, odd = 0;
  for (int i = 0; i < m; ++i) { odd += idx[i] & 1; if (r[2 * i] != (float)idx[i] || r[2 * i + 1] != (float)(idx[i] + 1)) ++bad; }
  printf("sync=%d odd=%d bad=%d\n", (int)e, odd, bad);
  return bad != 0;
}
