// Host cost of a small device-to-device copy: hipMemcpyAsync against a plain copy kernel, and hipMemsetAsync
// against a fill kernel (the interior-point loop issues ~80 of them per iteration on vectors of a few KB).
// hipcc --offload-arch=gfx950 -O2 d2d_launch_cost.hip -o d2d_launch_cost && ./d2d_launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void copy8(double* __restrict__ d, const double* __restrict__ s, long n) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i < n) d[i] = s[i];
}
__global__ void fill8(double* __restrict__ d, long n) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i < n) d[i] = 0.0;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (long n : {300L, 10000L, 1000000L}) {
    double *a, *b;
    hipMalloc(&a, n * 8); hipMalloc(&b, n * 8);
    hipMemset(a, 0, n * 8);
    const int reps = 2000;
    for (int mode = 0; mode < 4; ++mode) {
      hipStreamSynchronize(st);
      const double t0 = now();
      for (int r = 0; r < reps; ++r) {
        if (mode == 0) hipMemcpyAsync(b, a, n * 8, hipMemcpyDeviceToDevice, st);
        else if (mode == 1) hipLaunchKernelGGL(copy8, dim3((n + 255) / 256), dim3(256), 0, st, b, a, n);
        else if (mode == 2) hipMemsetAsync(b, 0, n * 8, st);
        else hipLaunchKernelGGL(fill8, dim3((n + 255) / 256), dim3(256), 0, st, b, n);
      }
      const double t1 = now();
      hipStreamSynchronize(st);
      const double t2 = now();
      const char* nm[] = {"hipMemcpyAsync d2d", "copy kernel", "hipMemsetAsync", "fill kernel"};
      std::printf("n=%8ld %-20s issue %.2f us/op, issue+drain %.2f us/op\n", n, nm[mode], (t1 - t0) / reps * 1e6, (t2 - t0) / reps * 1e6);
    }
    hipFree(a); hipFree(b);
  }
  return 0;
}
