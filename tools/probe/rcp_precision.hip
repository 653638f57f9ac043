// precision of v_rcp_f64 on gfx950, raw and after one / two Newton steps (run on a GPU box):
//   hipcc --offload-arch=gfx950 -O2 tools/probe/rcp_precision.hip -o /tmp/rcp && /tmp/rcp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double *x, double *r0, double *r1, double *r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = x[i];
  double r = __builtin_amdgcn_rcp(d);
  r0[i] = r;
  r = fma(fma(-d, r, 1.0), r, r);
  r1[i] = r;
  r = fma(fma(-d, r, 1.0), r, r);
  r2[i] = r;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), a(n), b(n), c(n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> u(-12.0, 12.0);
  for (auto &v : x) v = std::pow(10.0, u(g)) * ((g() & 1) ? 1 : -1);
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / (long double)x[i];
    e0 = std::fmax(e0, (double)fabsl(((long double)a[i] - t) / t));
    e1 = std::fmax(e1, (double)fabsl(((long double)b[i] - t) / t));
    e2 = std::fmax(e2, (double)fabsl(((long double)c[i] - t) / t));
  }
  printf("max relative error of v_rcp_f64: raw %.3e, one Newton step %.3e, two %.3e (2^-53 = 1.1e-16)\n", e0, e1, e2);
  return 0;
}
