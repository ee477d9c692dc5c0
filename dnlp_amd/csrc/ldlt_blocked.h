// Blocked right-looking LDL^T (no pivoting) for large dense KKT matrices on gfx950.
//
//   for each outer panel of NB columns:
//     for each inner block of nb columns inside the panel:
//       ldlt_diag_kernel   — nb x nb diagonal block factored in LDS by one workgroup
//       ldlt_trsm_kernel   — rows below: X = A21 L11^-T (= L21 D) by per-row forward
//                            substitution (one row per lane, coalesced down the columns);
//                            stores L21 in place and W21 = L21 D in the panel workspace
//       gemm_nt_update     — rest of the panel:   A[r, c] -= W[r, :] . L[c, :]
//     gemm_nt_update       — trailing matrix (Schur complement): A22 -= W21 L21^T, lower
//                            tiles only.  This is the n^3/3 part and the ONLY place MFMA is
//                            used: v_mfma_f64_16x16x4_f64, 128x128 block tile, 64x64 per
//                            wave (4x4 MFMA tiles, 128 accumulator VGPRs), operands staged
//                            global -> registers -> LDS with 16-B lanes, LDS rows padded by
//                            16 doubles so the two k-rows a 32-lane group reads hit disjoint
//                            banks.
//   The MFMA computes D[row=j][col=i] so that lane&15 walks the contiguous (row) direction
//   of the column-major C tile in the read-modify-write epilogue.
//
// Inertia is read off D (Sylvester); zero / tiny pivots are counted and replaced so the
// caller's regularisation loop (WB Algorithm IC) can react.
#pragma once
#include "exec_hip.h"

namespace dnlp {

typedef double mfma_d4 __attribute__((ext_vector_type(4)));

constexpr int LD_NB = 256;     // outer panel width
constexpr int LD_nb = 32;      // inner block
constexpr int GM_BM = 128, GM_BN = 128, GM_BK = 16, GM_PAD = 16;

struct LdltInfo { int nneg, nzero, fail, pad; double dmax; };

// ---- diagonal block ----------------------------------------------------------------------
__global__ void __launch_bounds__(256) ldlt_diag_kernel(double* A, i64 ld, int j0, int jb, LdltInfo* info,
                                                        double tiny) {
  __shared__ double S[LD_nb][LD_nb + 1];
  __shared__ double dinv;
  const int tid = threadIdx.x;
  for (int e = tid; e < jb * jb; e += 256) {
    const int r = e % jb, c = e / jb;
    S[r][c] = (r >= c) ? A[(j0 + r) + static_cast<i64>(j0 + c) * ld] : 0.0;
  }
  __syncthreads();
  for (int k = 0; k < jb; ++k) {
    if (tid == 0) {
      double d = S[k][k];
      if (!(d == d)) { info->fail = 1; d = 1.0; }
      if (fabs(d) <= tiny) { info->nzero += 1; d = (d < 0.0 ? -tiny : tiny); if (d == 0.0) d = 1e-300; }
      if (d < 0.0) info->nneg += 1;
      S[k][k] = d;
      dinv = 1.0 / d;
    }
    __syncthreads();
    const double di = dinv;
    // trailing update of the block with the unscaled column, then scale the column
    const int t = jb - k - 1;
    for (int e = tid; e < t * t; e += 256) {
      const int r = k + 1 + e % t, c = k + 1 + e / t;
      if (r >= c) S[r][c] -= S[r][k] * S[c][k] * di;
    }
    __syncthreads();
    for (int r = k + 1 + tid; r < jb; r += 256) S[r][k] *= di;
    __syncthreads();
  }
  for (int e = tid; e < jb * jb; e += 256) {
    const int r = e % jb, c = e / jb;
    if (r >= c) A[(j0 + r) + static_cast<i64>(j0 + c) * ld] = S[r][c];
  }
}

// ---- panel rows: forward substitution per row ----------------------------------------------
__global__ void __launch_bounds__(256) ldlt_trsm_kernel(double* A, i64 ld, int j0, int jb, int n,
                                                        double* Wp, i64 ldw, int wcol0) {
  __shared__ double Lb[LD_nb][LD_nb + 1];
  __shared__ double dv[LD_nb];
  const int tid = threadIdx.x;
  for (int e = tid; e < jb * jb; e += 256) {
    const int r = e % jb, c = e / jb;
    Lb[r][c] = (r > c) ? A[(j0 + r) + static_cast<i64>(j0 + c) * ld] : 0.0;
  }
  if (tid < jb) dv[tid] = A[(j0 + tid) + static_cast<i64>(j0 + tid) * ld];
  __syncthreads();
  const i64 r = static_cast<i64>(j0) + jb + static_cast<i64>(blockIdx.x) * 256 + tid;
  if (r >= n) return;
  double x[LD_nb];
#pragma unroll
  for (int c = 0; c < LD_nb; ++c) x[c] = (c < jb) ? A[r + static_cast<i64>(j0 + c) * ld] : 0.0;
#pragma unroll
  for (int c = 1; c < LD_nb; ++c) {
    if (c < jb) {
      double s = x[c];
#pragma unroll
      for (int t = 0; t < c; ++t) s -= x[t] * Lb[c][t];
      x[c] = s;
    }
  }
#pragma unroll
  for (int c = 0; c < LD_nb; ++c) {
    if (c < jb) {
      Wp[r + static_cast<i64>(wcol0 + c) * ldw] = x[c];
      A[r + static_cast<i64>(j0 + c) * ld] = x[c] / dv[c];
    }
  }
}

// ---- C -= W L^T on FP64 MFMA ---------------------------------------------------------------
__global__ void __launch_bounds__(256, 2) gemm_nt_update(double* __restrict__ C, i64 ldc,
                                                         const double* __restrict__ W, i64 ldw,
                                                         const double* __restrict__ L, i64 ldl, int M,
                                                         int Nc, int Kd, int lower, int ntm, int vec_ok) {
  const int tm = blockIdx.x % ntm, tn = blockIdx.x / ntm;
  if (lower && (tm * GM_BM + GM_BM - 1 < tn * GM_BN)) return;
  __shared__ double Ws[GM_BK][GM_BM + GM_PAD];
  __shared__ double Ls[GM_BK][GM_BN + GM_PAD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  mfma_d4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = mfma_d4{0.0, 0.0, 0.0, 0.0};
  // staging map: this thread moves rows (2*(tid&63), +1) of k-rows (tid>>6) + 4q, q = 0..3
  const int srow = 2 * (tid & 63), sk = tid >> 6;
  const i64 gi = static_cast<i64>(tm) * GM_BM + srow;      // W row
  const i64 gj = static_cast<i64>(tn) * GM_BN + srow;      // L row
  double2 rw[4], rl[4];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = kt * GM_BK + sk + 4 * q;
      double2 vw = {0.0, 0.0}, vl = {0.0, 0.0};
      if (k < Kd) {
        const double* pw = W + gi + static_cast<i64>(k) * ldw;
        const double* pl = L + gj + static_cast<i64>(k) * ldl;
        if (vec_ok && gi + 1 < M) vw = *reinterpret_cast<const double2*>(pw);
        else { if (gi < M) vw.x = pw[0]; if (gi + 1 < M) vw.y = pw[1]; }
        if (vec_ok && gj + 1 < Nc) vl = *reinterpret_cast<const double2*>(pl);
        else { if (gj < Nc) vl.x = pl[0]; if (gj + 1 < Nc) vl.y = pl[1]; }
      }
      rw[q] = vw;
      rl[q] = vl;
    }
  };
  const int nkt = (Kd + GM_BK - 1) / GM_BK;
  load_tile(0);
  for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<double2*>(&Ws[sk + 4 * q][srow]) = rw[q];
      *reinterpret_cast<double2*>(&Ls[sk + 4 * q][srow]) = rl[q];
    }
    __syncthreads();
    if (kt + 1 < nkt) load_tile(kt + 1);
#pragma unroll
    for (int kk = 0; kk < GM_BK; kk += 4) {
      double a[4], b[4];
      const int kr = kk + (lane >> 4), lc = lane & 15;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = Ls[kr][wn * 64 + t * 16 + lc];
        b[t] = Ws[kr][wm * 64 + t * 16 + lc];
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ni], b[mi], acc[ni][mi], 0, 0, 0);
    }
    __syncthreads();
  }
  // epilogue: D[row = j][col = i]; lane&15 walks i (contiguous), (lane>>4) + 4r walks j
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const i64 i = static_cast<i64>(tm) * GM_BM + wm * 64 + mi * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const i64 j = static_cast<i64>(tn) * GM_BN + wn * 64 + ni * 16 + (lane >> 4) + 4 * r;
        if (i < M && j < Nc && (!lower || i >= j)) C[i + j * ldc] -= acc[ni][mi][r];
      }
    }
}

// ---- triangular solves with the unit-lower factor (blocks of LD_nb) --------------------------
// forward, diagonal block: one wave; lane t owns b[j0+t]
__global__ void __launch_bounds__(64) ldlt_fwd_diag(const double* A, i64 ld, int j0, int jb, double* b) {
  const int t = threadIdx.x;
  double v = (t < jb) ? b[j0 + t] : 0.0;
  for (int k = 0; k < jb; ++k) {
    const double yk = __shfl(v, k, 64);
    if (t > k && t < jb) v -= A[(j0 + t) + static_cast<i64>(j0 + k) * ld] * yk;
  }
  if (t < jb) b[j0 + t] = v;
}
// forward, rows below the block: b[r] -= sum_t A[r, j0+t] y[t]
__global__ void __launch_bounds__(256) ldlt_fwd_update(const double* __restrict__ A, i64 ld, int j0, int jb,
                                                       int n, double* __restrict__ b) {
  __shared__ double y[LD_nb];
  if (threadIdx.x < jb) y[threadIdx.x] = b[j0 + threadIdx.x];
  __syncthreads();
  const i64 r = static_cast<i64>(j0) + jb + static_cast<i64>(blockIdx.x) * 256 + threadIdx.x;
  if (r >= n) return;
  double s = 0.0;
#pragma unroll 8
  for (int t = 0; t < jb; ++t) s += A[r + static_cast<i64>(j0 + t) * ld] * y[t];
  b[r] -= s;
}
__global__ void __launch_bounds__(256) ldlt_diag_scale(const double* A, i64 ld, int n, double* b) {
  const i64 i = static_cast<i64>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) b[i] /= A[i + i * ld];
}
// backward, rows below the block: acc[t] += sum_r A[r, j0+t] x[r]   (acc zeroed by the caller)
__global__ void __launch_bounds__(256) ldlt_bwd_update(const double* __restrict__ A, i64 ld, int j0, int jb,
                                                       int n, const double* __restrict__ b, double* acc,
                                                       int rows_per_block) {
  __shared__ double red[4][LD_nb];
  const i64 rbeg = static_cast<i64>(j0) + jb + static_cast<i64>(blockIdx.x) * rows_per_block;
  i64 rend = rbeg + rows_per_block;
  if (rend > n) rend = n;
  double s[LD_nb];
#pragma unroll
  for (int t = 0; t < LD_nb; ++t) s[t] = 0.0;
  for (i64 r = rbeg + threadIdx.x; r < rend; r += 256) {
    const double xr = b[r];
#pragma unroll
    for (int t = 0; t < LD_nb; ++t)
      if (t < jb) s[t] += A[r + static_cast<i64>(j0 + t) * ld] * xr;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < LD_nb; ++t) {
    const double w = wave_sum(s[t]);
    if (lane == 0) red[wid][t] = w;
  }
  __syncthreads();
  if (threadIdx.x < jb) {
    const double tot = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    unsafeAtomicAdd(&acc[threadIdx.x], tot);
  }
}
// backward, diagonal block: x_t = (b_t - acc_t) - sum_{t' > t} L[t', t] x_t'
__global__ void __launch_bounds__(64) ldlt_bwd_diag(const double* A, i64 ld, int j0, int jb, double* b,
                                                    double* acc) {
  const int t = threadIdx.x;
  double v = (t < jb) ? b[j0 + t] - acc[t] : 0.0;
  for (int k = jb - 1; k >= 0; --k) {
    const double xk = __shfl(v, k, 64);
    if (t < k) v -= A[(j0 + k) + static_cast<i64>(j0 + t) * ld] * xk;
  }
  if (t < jb) { b[j0 + t] = v; acc[t] = 0.0; }
}

struct BlockedLdlt {
  HipExec* ex = nullptr;
  i64 n = 0, ld = 0, ldw = 0;
  double* Wp = nullptr;
  LdltInfo* info = nullptr;
  double* acc = nullptr;
  double last_update_seconds = 0.0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool time_updates = false;

  void init(HipExec* e, i64 n_, i64 ld_) {
    ex = e; n = n_; ld = ld_;
    ldw = (n + 7) / 8 * 8;
    Wp = ex->alloc<double>(static_cast<size_t>(ldw) * LD_NB);
    info = ex->alloc<LdltInfo>(1);
    acc = ex->alloc<double>(LD_nb);
    DNLP_HIP_CHECK(hipEventCreate(&ev0));
    DNLP_HIP_CHECK(hipEventCreate(&ev1));
  }

  void gemm(double* C, const double* W, const double* L, i64 ldl, int M, int Nc, int Kd, int lower) {
    if (M <= 0 || Nc <= 0 || Kd <= 0) return;
    const int ntm = (M + GM_BM - 1) / GM_BM, ntn = (Nc + GM_BN - 1) / GM_BN;
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const int vec_ok = al(W) && al(L) && (ldw % 2 == 0) && (ldl % 2 == 0);
    hipLaunchKernelGGL(gemm_nt_update, dim3(static_cast<unsigned>(ntm) * ntn), dim3(256), 0, ex->stream, C, ld, W, ldw,
                       L, ldl, M, Nc, Kd, lower, ntm, vec_ok);
  }

  bool factor(double* A, int* nneg, int* nzero) {
    LdltInfo z;
    std::memset(&z, 0, sizeof z);
    DNLP_HIP_CHECK(hipMemcpyAsync(info, &z, sizeof z, hipMemcpyHostToDevice, ex->stream));
    DNLP_HIP_CHECK(hipStreamSynchronize(ex->stream));
    const double tiny = 1e-300;
    float upd_ms = 0.f;
    const int ni = static_cast<int>(n);
    for (int K0 = 0; K0 < ni; K0 += LD_NB) {
      const int KB = std::min(LD_NB, ni - K0);
      for (int j0 = K0; j0 < K0 + KB; j0 += LD_nb) {
        const int jb = std::min(LD_nb, K0 + KB - j0);
        hipLaunchKernelGGL(ldlt_diag_kernel, dim3(1), dim3(256), 0, ex->stream, A, ld, j0, jb, info, tiny);
        const int r0 = j0 + jb;
        if (r0 >= ni) continue;
        const int rows = ni - r0;
        hipLaunchKernelGGL(ldlt_trsm_kernel, dim3((rows + 255) / 256), dim3(256), 0, ex->stream, A, ld, j0, jb, ni, Wp,
                           ldw, j0 - K0);
        const int nc = K0 + KB - r0;
        if (nc > 0)
          gemm(A + r0 + static_cast<i64>(r0) * ld, Wp + r0 + static_cast<i64>(j0 - K0) * ldw,
               A + r0 + static_cast<i64>(j0) * ld, ld, rows, nc, jb, 1);
      }
      const int r1 = K0 + KB;
      if (r1 < ni) {
        if (time_updates) DNLP_HIP_CHECK(hipEventRecord(ev0, ex->stream));
        gemm(A + r1 + static_cast<i64>(r1) * ld, Wp + r1, A + r1 + static_cast<i64>(K0) * ld, ld, ni - r1, ni - r1, KB, 1);
        if (time_updates) {
          DNLP_HIP_CHECK(hipEventRecord(ev1, ex->stream));
          DNLP_HIP_CHECK(hipEventSynchronize(ev1));
          float ms = 0.f;
          DNLP_HIP_CHECK(hipEventElapsedTime(&ms, ev0, ev1));
          upd_ms += ms;
        }
      }
    }
    LdltInfo out;
    DNLP_HIP_CHECK(hipMemcpyAsync(&out, info, sizeof out, hipMemcpyDeviceToHost, ex->stream));
    DNLP_HIP_CHECK(hipStreamSynchronize(ex->stream));
    *nneg = out.nneg;
    *nzero = out.nzero;
    last_update_seconds = upd_ms * 1e-3;
    return out.fail == 0;
  }

  void solve(const double* A, double* b) {
    const int ni = static_cast<int>(n);
    for (int j0 = 0; j0 < ni; j0 += LD_nb) {
      const int jb = std::min(LD_nb, ni - j0);
      hipLaunchKernelGGL(ldlt_fwd_diag, dim3(1), dim3(64), 0, ex->stream, A, ld, j0, jb, b);
      const int rows = ni - j0 - jb;
      if (rows > 0)
        hipLaunchKernelGGL(ldlt_fwd_update, dim3((rows + 255) / 256), dim3(256), 0, ex->stream, A, ld, j0, jb, ni, b);
    }
    hipLaunchKernelGGL(ldlt_diag_scale, dim3((ni + 255) / 256), dim3(256), 0, ex->stream, A, ld, ni, b);
    ex->zero(acc, sizeof(double) * LD_nb);
    const int nblk = (ni + LD_nb - 1) / LD_nb;
    for (int bi = nblk - 1; bi >= 0; --bi) {
      const int j0 = bi * LD_nb, jb = std::min(LD_nb, ni - j0);
      const int rows = ni - j0 - jb;
      if (rows > 0) {
        const int rpb = 4096;
        hipLaunchKernelGGL(ldlt_bwd_update, dim3((rows + rpb - 1) / rpb), dim3(256), 0, ex->stream, A, ld, j0, jb, ni, b,
                           acc, rpb);
      }
      hipLaunchKernelGGL(ldlt_bwd_diag, dim3(1), dim3(64), 0, ex->stream, A, ld, j0, jb, b, acc);
    }
  }
};

inline void HipExec::ldlt_prepare(LdltWork& w, i64 n, i64 ld, bool pivoted) {
  if (pivoted) {
    w.st = alloc<BkState>(1);
  } else {
    w.blocked = new BlockedLdlt();
    w.blocked->init(this, n, ld);
  }
}
inline bool HipExec::ldlt_factor(LdltWork& w, double* A, i64 n, i64 ld, i32* ipiv, bool pivoted, int* nneg, int* nzero) {
  if (pivoted) return bk_factor(w, A, n, ld, ipiv, nneg, nzero);
  return w.blocked->factor(A, nneg, nzero);
}
inline void HipExec::ldlt_solve(LdltWork& w, const double* A, i64 n, i64 ld, const i32* ipiv, bool pivoted, double* b) {
  if (pivoted) {
    hipLaunchKernelGGL(bk_solve_kernel, dim3(1), dim3(BK_T), 0, stream, A, static_cast<int>(n), ld, ipiv, b);
  } else {
    w.blocked->solve(A, b);
  }
}

// ---- seeded dense symmetric test matrix (BASELINE config C4) --------------------------------
__device__ inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ inline double u01(uint64_t h) { return static_cast<double>(h >> 11) * (1.0 / 9007199254740992.0); }

__global__ void __launch_bounds__(256) gen_vec_kernel(double* v, i64 n, uint64_t seed) {
  const i64 i = static_cast<i64>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) v[i] = 2.0 * u01(splitmix64(seed ^ (0xA5A5A5A5ull + static_cast<uint64_t>(i) * 0x100000001B3ull))) - 1.0;
}
// A[i,j] = noise(min(i,j), max(i,j)) in [-1,1) + spike * v_i v_j
__global__ void __launch_bounds__(256) gen_sym_kernel(double* A, i64 n, i64 ld, uint64_t seed, double spike,
                                                      const double* v, i64 nrb) {
  const i64 c = blockIdx.x / nrb, rb = blockIdx.x % nrb;
  const i64 r = rb * 256 + threadIdx.x;
  if (r >= n) return;
  const uint64_t lo = static_cast<uint64_t>(r < c ? r : c), hi = static_cast<uint64_t>(r < c ? c : r);
  const double noise = 2.0 * u01(splitmix64(seed + lo * 0x9E3779B97F4A7C15ull + splitmix64(hi))) - 1.0;
  A[r + c * ld] = noise + spike * v[r] * v[c];
}

}  // namespace dnlp
