// A/B of the two 128 x 128 top-block kernels of the blocked LDL^T (csrc/ldlt_blocked.h, csrc/ldlt_top_mfma.h):
// same input, compares the factor, the packed operand copy and the inertia counts, times both with HIP events.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dnlp_amd/csrc tools/micro/top128_ab.hip -o gpurun_out/top128_ab -lhiprtc
#include "ldlt_blocked.h"
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
using namespace dnlp;
int main(int argc, char** argv) {
  const int ld = 640, j0 = 128, n = 128;
  const int kind = argc > 1 ? atoi(argv[1]) : 0;     // 0 positive definite, 1 quasi-definite (64 negative), 2 with a zero pivot
  std::mt19937_64 g(7);
  std::normal_distribution<double> N(0.0, 1.0);
  std::vector<double> h(static_cast<size_t>(ld) * ld, NAN);   // everything outside the lower triangle of the block is poison
  std::vector<double> B(n * n);
  for (auto& v : B) v = N(g);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0;
      for (int k = 0; k < n; ++k) s += B[i * n + k] * B[j * n + k];
      if (kind >= 1 && i >= 64 && j >= 64) s = -s - (i == j ? 1.0 : 0.0);
      if (kind >= 1 && i >= 64 && j < 64) s = B[i * n + j];
      if (i == j) s += (kind >= 1 && i >= 64) ? -1.0 : 1.0;
      h[(j0 + i) + static_cast<size_t>(j0 + j) * ld] = s;
    }
  if (kind == 2) {
    for (int i = 0; i < n; ++i) h[(j0 + i) + static_cast<size_t>(j0) * ld] = 0.0;   // first pivot exactly zero
  }
  double *dA, *dA0, *dL;
  LdltInfo* info;
  hipMalloc(&dA, h.size() * 8); hipMalloc(&dA0, h.size() * 8); hipMalloc(&dL, LD_TOP_WS * 8); hipMalloc(&info, sizeof(LdltInfo));
  hipMemcpy(dA0, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  std::vector<double> out[2], lt[2];
  LdltInfo inf[2];
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int v = 0; v < 2; ++v) {
    double tsum = 0, tmin = 1e9;
    const int reps = 50;
    for (int r = 0; r < reps + 5; ++r) {
      hipMemcpy(dA, dA0, h.size() * 8, hipMemcpyDeviceToDevice);
      hipMemset(info, 0, sizeof(LdltInfo));
      hipMemset(dL, 0, LD_TOP_WS * 8);
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      if (v == 0) hipLaunchKernelGGL(ldlt_top128_kernel, dim3(1), dim3(LD_TOP_THREADS), 0, 0, dA, (i64)ld, j0, info, 1e-300, dL);
      else hipLaunchKernelGGL(ldlt_top128_mfma_kernel, dim3(1), dim3(LD_TOPM_THREADS), 0, 0, dA, (i64)ld, j0, info, 1e-300, dL);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (r >= 5) { tsum += ms; tmin = std::fmin(tmin, ms); }
    }
    out[v].resize(h.size()); lt[v].resize(LD_TOP_WS);
    hipMemcpy(out[v].data(), dA, h.size() * 8, hipMemcpyDeviceToHost);
    hipMemcpy(lt[v].data(), dL, LD_TOP_WS * 8, hipMemcpyDeviceToHost);
    hipMemcpy(&inf[v], info, sizeof(LdltInfo), hipMemcpyDeviceToHost);
    printf("%s: mean %.2f us, min %.2f us (event pairs around one launch), nneg %d nzero %d fail %d\n",
           v ? "mfma blocks" : "4x4 tiles  ", tsum / reps * 1e3, tmin * 1e3, inf[v].nneg, inf[v].nzero, inf[v].fail);
  }
  double dmax = 0, lmax = 0, scale = 0;
  int touched = 0;
  for (size_t i = 0; i < h.size(); ++i) {
    const int r = static_cast<int>(i % ld) - j0, c = static_cast<int>(i / ld) - j0;
    const bool in = r >= 0 && r < n && c >= 0 && c <= r;
    if (in) { dmax = std::fmax(dmax, std::fabs(out[0][i] - out[1][i])); scale = std::fmax(scale, std::fabs(out[0][i])); }
    else if (!(out[1][i] != out[1][i])) ++touched;              // the poison must still be there
  }
  for (int i = 0; i < LD_TOP_WS; ++i) lmax = std::fmax(lmax, std::fabs(lt[0][i] - lt[1][i]));
  // reconstruction error of the new factor: || L D L^T - A || over the lower triangle
  double rec = 0, an = 0;
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0;
      for (int k = 0; k <= j; ++k) {
        const double lik = i == k ? 1.0 : out[1][(j0 + i) + static_cast<size_t>(j0 + k) * ld];
        const double ljk = j == k ? 1.0 : out[1][(j0 + j) + static_cast<size_t>(j0 + k) * ld];
        s += lik * ljk * out[1][(j0 + k) + static_cast<size_t>(j0 + k) * ld];
      }
      double a = h[(j0 + i) + static_cast<size_t>(j0 + j) * ld];
      if (kind == 2 && j == 0 && i == 0) a = 1e-300;
      rec = std::fmax(rec, std::fabs(s - a)); an = std::fmax(an, std::fabs(a));
    }
  printf("max |factor difference| %.3e (scale %.3e), packed copy %.3e, reconstruction %.3e of %.3e, writes outside the triangle %d\n",
         dmax, scale, lmax, rec, an, touched);
  const bool ok = (kind == 2 || (dmax <= 1e-9 * scale && lmax <= 1e-9 * scale && rec <= 1e-10 * an)) && touched == 0 &&
                  inf[0].nneg == inf[1].nneg && inf[0].nzero == inf[1].nzero && inf[0].fail == inf[1].fail;
  printf(ok ? "OK\n" : "MISMATCH\n");
  return ok ? 0 : 1;
}
