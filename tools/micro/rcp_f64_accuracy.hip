// v_rcp_f64 accuracy on gfx950: max relative error of the raw instruction and after one / two Newton steps.
// hipcc --offload-arch=gfx950 -O3 tools/micro/rcp_f64_accuracy.hip -o /tmp/rcp_acc && /tmp/rcp_acc
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double* d, double* r0, double* r1, double* r2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = d[i];
  double r = __builtin_amdgcn_rcp(x);
  r0[i] = r;
  r = fma(r, fma(-x, r, 1.0), r);
  r1[i] = r;
  r = fma(r, fma(-x, r, 1.0), r);
  r2[i] = r;
}
int main() {
  const int n = 1 << 22;
  std::vector<double> h(n), o0(n), o1(n), o2(n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> m(1.0, 2.0);
  std::uniform_int_distribution<int> e(-300, 300);
  for (int i = 0; i < n; ++i) h[i] = std::ldexp(m(g), e(g)) * ((i & 1) ? -1.0 : 1.0);
  double *d, *a, *b, *c;
  hipMalloc(&d, n * 8); hipMalloc(&a, n * 8); hipMalloc(&b, n * 8); hipMalloc(&c, n * 8);
  hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, a, b, c, n);
  hipMemcpy(o0.data(), a, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(o1.data(), b, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(o2.data(), c, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / static_cast<long double>(h[i]);
    e0 = std::fmax(e0, static_cast<double>(fabsl((o0[i] - t) / t)));
    e1 = std::fmax(e1, static_cast<double>(fabsl((o1[i] - t) / t)));
    e2 = std::fmax(e2, static_cast<double>(fabsl((o2[i] - t) / t)));
  }
  std::printf("{\"v_rcp_f64_max_rel_err\": %.3e, \"after_1_newton\": %.3e, \"after_2_newton\": %.3e, \"eps\": %.3e}\n", e0, e1, e2, 1.11e-16);
  return 0;
}
