// What one hop of a dataflow chain between workgroups costs on gfx950 (round 4: the one-launch triangular sweeps of
// csrc/ldlt_blocked.h).  Workgroup j waits for 128 doubles published by workgroup j - 1 (data-as-flag, the exchange
// buffer preset to all-ones), does the sweep's reduction skeleton (two products through LDS) and publishes its own.
//   stride S : only workgroups with blockIdx % S == 0 take part (S = 8: the chain stays on one XCD)
//   tiles  T : 0 none; 1: workgroup j streams j tiles of 128 KB (one ahead), as the forward sweep does
//   split  P : the tiles of a block row are shared by P workgroups (k mod P), partial sums through a second buffer
//   hipcc --offload-arch=gfx950 -O3 chain_hop.hip -o bin/chain_hop && bin/chain_hop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)
constexpr unsigned long long kEmpty = 0xFFFFFFFFFFFFFFFFull;
constexpr int H = 128, T = 1024;
struct Tile { double v[16]; };
__device__ inline void tile_load(Tile& t, const double* __restrict__ M, long ld, int tid) {
  const int ch = __builtin_amdgcn_readfirstlane(tid >> 7);
  const double* col = M + static_cast<long>(16 * ch) * ld + (tid & (H - 1));
#pragma unroll
  for (int c = 0; c < 16; ++c) t.v[c] = col[c * ld];
}
__device__ inline void tile_vec(const Tile& t, const double* v, double* part, int tid) {
  const int ch = __builtin_amdgcn_readfirstlane(tid >> 7);
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int c = 0; c < 16; c += 2) { s0 = fma(t.v[c], v[16 * ch + c], s0); s1 = fma(t.v[c + 1], v[16 * ch + c + 1], s1); }
  part[ch * H + (tid & (H - 1))] = s0 + s1;
  __syncthreads();
}
__device__ inline double part_sum(const double* part, int i) { double s = 0.0;
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) s += part[ch * H + i];
  return s; }
template <int SCOPE> __device__ inline void fetch(const double* src, double* dst, int tid) {
  __syncthreads();
  if (tid < H) {
    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(src) + tid;
    unsigned long long v;
    unsigned spins = 0;
    while ((v = __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE)) == kEmpty && ++spins < (1u << 18)) __builtin_amdgcn_s_sleep(1);     // (gives up: a stale line must not hang the box)
    dst[tid] = __longlong_as_double(static_cast<long long>(v));
  }
  __syncthreads();
}
__device__ inline void publish(double* dst, double v, int tid) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst) + tid, static_cast<unsigned long long>(__double_as_longlong(v)), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
// P workgroups per block row: member q of row i takes the tiles k with k % P == q; the member with (i - 1) % P == q (it has
// the LAST tile) is the row's owner: it adds the other members' partial sums (fixed order) and publishes y_i.
template <int SCOPE> __global__ void __launch_bounds__(T) chain(const double* __restrict__ A, long ld, double* xch, double* pex, int stride, int tiles,
                                                                int P, int nrow) {
  __shared__ double y[H], acc[H], part[T];
  if (blockIdx.x % stride) return;
  const int tid = threadIdx.x, w = blockIdx.x / stride, i = w / P, q = w % P;
  if (i >= nrow) return;
  const int owner = (i > 0) ? (i - 1) % P : 0;
  if (tid < H) acc[tid] = 1.0;
  Tile cur, nxt, ti;
  tile_load(ti, A, ld, tid);
  for (int c = 0; c < 16; ++c) ti.v[c] *= 1e-3;
  __syncthreads();
  int k = q;
  if (tiles && k < i) tile_load(cur, A + static_cast<long>(i) * H + static_cast<long>(k) * H * ld, ld, tid);
  else cur = ti;
  for (; k < i; k += P) {
    if (tiles && k + P < i) tile_load(nxt, A + static_cast<long>(i) * H + static_cast<long>(k + P) * H * ld, ld, tid);
    else nxt = ti;
    fetch<SCOPE>(xch + k * H, y, tid);
    tile_vec(cur, y, part, tid);
    if (tid < H) acc[tid] -= 1e-3 * part_sum(part, tid);
    __syncthreads();
    cur = nxt;
  }
  if (q != owner) {                                   // partial sums of this member
    if (tid < H) publish(pex + (static_cast<long>(i) * P + q) * H, acc[tid] - 1.0, tid);
    return;
  }
  for (int o = 0; o < P; ++o) {
    if (o == owner) continue;
    fetch<SCOPE>(pex + (static_cast<long>(i) * P + o) * H, y, tid);
    if (tid < H) acc[tid] += y[tid];
    __syncthreads();
  }
  tile_vec(ti, acc, part, tid);
  if (tid < H) publish(xch + i * H, part_sum(part, tid) + 1.0, tid);
}
int main(int argc, char** argv) {
  const int nrow = argc > 1 ? std::atoi(argv[1]) : 86;
  const long n = static_cast<long>(nrow) * H, ld = n + (argc > 2 ? std::atoi(argv[2]) : 8);
  double *A, *xch, *pex;
  CK(hipMalloc(&A, sizeof(double) * ld * n));
  CK(hipMalloc(&xch, sizeof(double) * n));
  CK(hipMalloc(&pex, sizeof(double) * n * 8));
  std::vector<double> h(static_cast<size_t>(ld) * 64);
  for (size_t k = 0; k < h.size(); ++k) h[k] = 1e-3 * static_cast<double>(k % 17);
  for (long c = 0; c < n; c += 64) CK(hipMemcpy(A + c * ld, h.data(), sizeof(double) * std::min<long>(64, n - c) * ld, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, int scope, int stride, int tiles, int P) {
    float best = 1e30f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipMemset(xch, 0xFF, sizeof(double) * n));
      CK(hipMemset(pex, 0xFF, sizeof(double) * n * 8));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      const unsigned grid = static_cast<unsigned>(nrow * P * stride);
      if (scope == 0) hipLaunchKernelGGL(chain<__HIP_MEMORY_SCOPE_AGENT>, dim3(grid), dim3(T), 0, 0, A, ld, xch, pex, stride, tiles, P, nrow);
      else hipLaunchKernelGGL(chain<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(grid), dim3(T), 0, 0, A, ld, xch, pex, stride, tiles, P, nrow);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    double chk; CK(hipMemcpy(&chk, xch + (n - 1), 8, hipMemcpyDeviceToHost));
    std::printf("%-44s rows %d: %8.1f us = %6.2f us per hop   (check %.6g)\n", name, nrow, best * 1e3, best * 1e3 / nrow, chk);
  };
  run("chain only, consecutive workgroups", 0, 1, 0, 1);
  run("chain only, one XCD (stride 8)", 0, 8, 0, 1);
  run("chain only, one XCD, workgroup-scope loads", 1, 8, 0, 1);
  run("with tiles, consecutive", 0, 1, 1, 1);
  run("with tiles, one XCD", 0, 8, 1, 1);
  run("with tiles, 2 workgroups per row", 0, 1, 1, 2);
  run("with tiles, 3 workgroups per row", 0, 1, 1, 3);
  run("with tiles, 4 workgroups per row", 0, 1, 1, 4);
  run("with tiles, 2 per row on 2 XCDs (stride 4)", 0, 4, 1, 2);
  return 0;
}
