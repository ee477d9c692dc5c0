// One 256-thread workgroup doing the right-looking rank-1 updates of an order-n dense LDL^T in global memory (the n > 256
// branch of BlockExecT::ldlt_factor, exec_block.h): what does the loop nest cost by itself?  v0 = one entry per lane and
// trip, v1 = eight rows per lane in flight.   hipcc --offload-arch=gfx950 -O3 block_rank1.hip -o bin/block_rank1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)
template <int V> __global__ void __launch_bounds__(256) k(double* Aall, int n, long ld) {
  double* A = Aall + static_cast<long>(blockIdx.x) * ld * n;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int kk = 0; kk < n; ++kk) {
    double* Ak = A + static_cast<long>(kk) * ld;
    const double inv = 1.0 / Ak[kk];
    if (V == 0) {
      for (int j = kk + 1 + wv; j < n; j += 4) {
        const double wj = Ak[j] * inv;
        if (wj == 0.0) continue;
        double* Aj = A + static_cast<long>(j) * ld;
        for (int i = j + lane; i < n; i += 64) Aj[i] -= Ak[i] * wj;
      }
    } else {
      for (int g0 = (kk + 1) & ~511; g0 < n; g0 += 512) {
        double akr[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { const int i = g0 + lane + 64 * r; akr[r] = (i > kk && i < n) ? Ak[i] : 0.0; }
        const int jend = n < g0 + 512 ? n : g0 + 512;
        for (int j = kk + 1 + wv; j < jend; j += 4) {
          const double wj = Ak[j] * inv;
          if (wj == 0.0) continue;
          double* Aj = A + static_cast<long>(j) * ld;
          double t[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) { const int i = g0 + lane + 64 * r; t[r] = Aj[(i >= j && i < n) ? i : j]; }
#pragma unroll
          for (int r = 0; r < 8; ++r) { const int i = g0 + lane + 64 * r; if (i >= j && i < n) Aj[i] = t[r] - akr[r] * wj; }
        }
      }
    }
    __syncthreads();
    for (int i = kk + 1 + tid; i < n; i += 256) Ak[i] *= inv;
    __syncthreads();
  }
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 600, B = argc > 2 ? std::atoi(argv[2]) : 4;
  const long ld = n + 8;
  std::vector<double> h(static_cast<size_t>(ld) * n);
  for (int c = 0; c < n; ++c) for (int r = 0; r < n; ++r) h[r + c * ld] = (r == c) ? 2.0 * n : 1.0 / (1.0 + ((r * 7 + c * 13) % 11));
  double* A;
  CK(hipMalloc(&A, sizeof(double) * ld * n * B));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int v = 0; v < 2; ++v) {
    for (int b = 0; b < B; ++b) CK(hipMemcpy(A + static_cast<long>(b) * ld * n, h.data(), sizeof(double) * ld * n, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0));
    if (v == 0) hipLaunchKernelGGL(k<0>, dim3(B), dim3(256), 0, 0, A, n, ld); else hipLaunchKernelGGL(k<1>, dim3(B), dim3(256), 0, 0, A, n, ld);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("order %d, %d instances, variant %d: %.2f ms per factorisation\n", n, B, v, ms);
  }
  return 0;
}
