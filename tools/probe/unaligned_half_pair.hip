// probe: does a 4-byte raw buffer load at a 2-byte-aligned offset return the right two fp16 values on this GPU?
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <vector>
__global__ void k(const __half* p, int n, const int* idx, float* out, int m, unsigned soff) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, n * 2, 0x00020000);
  const unsigned v = __builtin_amdgcn_raw_buffer_load_b32(r, idx[i] * 2, (int)soff, 0);
  out[2 * i] = __half2float(__ushort_as_half((unsigned short)(v & 0xffffu)));
  out[2 * i + 1] = __half2float(__ushort_as_half((unsigned short)(v >> 16)));
}
int main() {
  const int n = 1 << 20, m = 4096;
  std::vector<__half> h(n);
  for (int i = 0; i < n; ++i) h[i] = __float2half((float)(i % 2048));
  std::vector<int> idx(m);
  for (int i = 0; i < m; ++i) idx[i] = (i * 7919 + 1) % (n - 4096);   // odd and even
  __half* d; float* o; int* di;
  hipMalloc(&d, n * 2); hipMalloc(&o, m * 8); hipMalloc(&di, m * 4);
  hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
  hipMemcpy(di, idx.data(), m * 4, hipMemcpyHostToDevice);
  int bad_total = 0;
  for (unsigned soff : {0u, 2u * 301u, 2u * 300u}) {   // odd and even scalar offsets (in elements: 301, 300)
    k<<<m / 256, 256>>>(d, n, di, o, m, soff);
    hipError_t e = hipDeviceSynchronize();
    std::vector<float> r(2 * m);
    hipMemcpy(r.data(), o, m * 8, hipMemcpyDeviceToHost);
    int bad = 0, odd = 0;
    for (int i = 0; i < m; ++i) {
      const int e0 = idx[i] + soff / 2;
      odd += e0 & 1;
      if (r[2 * i] != (float)(e0 % 2048) || r[2 * i + 1] != (float)((e0 + 1) % 2048)) ++bad;
    }
    printf("soff=%u sync=%d odd=%d bad=%d\n", soff, (int)e, odd, bad);
    bad_total += bad;
  }
  return bad_total != 0;
}
