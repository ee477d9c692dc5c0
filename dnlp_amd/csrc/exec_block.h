// Execution space of the batched solver: ONE WORKGROUP = ONE PROBLEM INSTANCE.
//
// BASELINE config C5: thousands of small parametrised NLPs that share one
// tape structure.  Driving each of them from the host through HipExec costs a stream
// synchronisation per scalar (hundreds per interior-point iteration), so the batch path runs the
// SAME single-source classes (model.h, kkt_dense.h, ipm_core.h) entirely inside one kernel:
// every workgroup of 256 lanes (4 wavefronts) executes the interior-point control flow
// redundantly and uniformly, E::map is a workgroup-strided loop, E::sum/max/min are
// wavefront-shuffle + LDS reductions whose result every lane receives, and the dense KKT
// system of the instance is factorised by the workgroup itself (Bunch-Kaufman, DSYTF2
// semantics, same pivoting rule as the chip-wide kernels of exec_hip.h).  Vectors live in a
// per-workgroup slab of global memory (L2 resident at these sizes); the KKT matrix is placed in
// LDS when it fits (160 KB per CU on gfx950).  No host round trip happens between the launch
// and the final status write.
//
// Rules that make the redundant control flow legal: every scalar that steers a branch comes
// from a reduction (identical in all lanes) or from uniform loads; every map / reduction /
// copy ends in a workgroup barrier; "control space" memory is global memory written by lane 0
// or written identically by all lanes.
#pragma once
#include <hip/hip_runtime.h>

#include "atom_math.h"
#include "exec.h"
#include "wave_ops.h"
#include "bk_panel.h"
#include "sparse_ldl.h"

namespace dnlp {


// LDS-address-space pointer: ds_read / ds_write instead of flat accesses when the KKT matrix of
// the instance sits in LDS
typedef __attribute__((address_space(3))) double lds_double;


// Bunch-Kaufman LDL^T (DSYTF2 semantics, lower) by ONE wavefront, no workgroup barrier: the
// pivot column is cached in registers (R rows per lane, n <= 64 R), the trailing update walks
// the columns with the multiplier broadcast from LDS, all-zero multipliers (sparse KKT
// systems) skip their column.  AP is `double*` or `lds_double*`; piv is the pivot vector in LDS.
template <int R, class AP>
__device__ bool bk_factor_wave(AP A, int n, int ld, int* piv, int* nneg_out, int* nzero_out) {
  const int lane = threadIdx.x & 63;
  const double alpha = 0.6403882032022076;   // (1 + sqrt(17)) / 8
  int k = 0, nneg = 0, nzero = 0;
  while (k < n) {
    AP Ak = A + k * ld;
    double v = -1.0;
    int idx = n;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int i = lane + 64 * r;
      if (i > k && i < n) { const double a = fabs(Ak[i]); if (a > v) { v = a; idx = i; } }
    }
    wave_all_argmax(v, idx);
    double colmax = v;
    int imax = idx;
    if (colmax < 0.0) { colmax = 0.0; imax = k; }
    const double absakk = fabs(Ak[k]);
    if (!(absakk == absakk) || !(colmax == colmax)) return false;
    int kstep = 1, kp = k;
    bool zero_piv = false;
    if (fmax(absakk, colmax) == 0.0) {
      zero_piv = true;
    } else if (absakk < alpha * colmax) {
      double rv = 0.0;
      for (int j = k + lane; j < imax; j += 64) rv = fmax(rv, fabs(A[imax + j * ld]));
      for (int i = imax + 1 + lane; i < n; i += 64) rv = fmax(rv, fabs(A[i + imax * ld]));
      rv = wave_all_max(rv);
      const double rowmax = rv;
      const double aii = fabs(A[imax + imax * ld]);
      if (absakk >= alpha * colmax * (colmax / rowmax)) kp = k;
      else if (aii >= alpha * rowmax) kp = imax;
      else { kp = imax; kstep = 2; }
    }
    const int kk = k + kstep - 1;
    if (!zero_piv && kp != kk) {
      AP Akk = A + kk * ld;
      AP Akp = A + kp * ld;
      for (int i = kp + 1 + lane; i < n; i += 64) { const double t = Akk[i]; Akk[i] = Akp[i]; Akp[i] = t; }
      for (int j = kk + 1 + lane; j < kp; j += 64) { const double t = Akk[j]; Akk[j] = A[kp + j * ld]; A[kp + j * ld] = t; }
      // standard form: the interchange also moves the rows of the columns already factored,
      // so that P A P^T = L D L^T with one permutation applied before / after the solves
      for (int c = lane; c < k; c += 64) { const double t = A[kk + c * ld]; A[kk + c * ld] = A[kp + c * ld]; A[kp + c * ld] = t; }
      wave_sync();
      if (lane == 0) {
        const double t = Akk[kk];
        Akk[kk] = Akp[kp];
        Akp[kp] = t;
        if (kstep == 2) { const double t2 = Ak[k + 1]; Ak[k + 1] = Ak[kp]; Ak[kp] = t2; }
      }
      wave_sync();
    }
    if (zero_piv) {
      if (lane == 0) { Ak[k] = 1e-20; piv[k] = k + 1; }
      ++nzero;
      wave_sync();
      k += 1;
      continue;
    }
    if (kstep == 1) {
      const double d = Ak[k];
      if (d < 0.0) ++nneg;
      if (fabs(d) < 1e-300) ++nzero;
      if (lane == 0) piv[k] = kp + 1;
      const double inv = 1.0 / d;
      double ak[R];
#pragma unroll
      for (int r = 0; r < R; ++r) { const int i = lane + 64 * r; ak[r] = (i > k && i < n) ? Ak[i] : 0.0; }
      // trailing update, four columns in flight: the columns with a non-zero multiplier are
      // enumerated from ballot masks (scalar unit), the multiplier comes from the owning lane's
      // register — no LDS access steers the control flow
#pragma unroll
      for (int r2 = 0; r2 < R; ++r2) {
        unsigned long long mask = __ballot(ak[r2] != 0.0);
        while (mask) {
          int jj[4];
          double w[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (mask) {
              const int b = __builtin_ctzll(mask);
              mask &= mask - 1;
              jj[u] = b + 64 * r2;
              w[u] = __shfl(ak[r2], b, 64) * inv;
            } else {
              jj[u] = n;            // sentinel: no row satisfies i >= n
              w[u] = 0.0;
            }
          }
          double t[4][R];
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const int i = lane + 64 * r;
              t[u][r] = (i >= jj[u] && i < n) ? A[i + jj[u] * ld] : 0.0;
            }
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const int i = lane + 64 * r;
              if (i >= jj[u] && i < n) A[i + jj[u] * ld] = t[u][r] - ak[r] * w[u];
            }
        }
      }
      wave_sync();
#pragma unroll
      for (int r = 0; r < R; ++r) { const int i = lane + 64 * r; if (i > k && i < n) Ak[i] = ak[r] * inv; }
      wave_sync();
    } else {
      AP Ak1 = Ak + ld;
      double d21 = Ak[k + 1];
      const double d11 = Ak1[k + 1] / d21, d22 = Ak[k] / d21;
      const double tt = 1.0 / (d11 * d22 - 1.0);
      d21 = tt / d21;
      ++nneg;
      if (lane == 0) { piv[k] = -(kp + 1); piv[k + 1] = -(kp + 1); }
      double a0[R], a1[R];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int i = lane + 64 * r;
        const bool in = i > k + 1 && i < n;
        a0[r] = in ? Ak[i] : 0.0;
        a1[r] = in ? Ak1[i] : 0.0;
      }
#pragma unroll
      for (int r2 = 0; r2 < R; ++r2) {
        unsigned long long mask = __ballot(a0[r2] != 0.0 || a1[r2] != 0.0);
        while (mask) {
          int jj[2];
          double w0[2], w1[2];
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            if (mask) {
              const int b = __builtin_ctzll(mask);
              mask &= mask - 1;
              jj[u] = b + 64 * r2;
              const double ajk = __shfl(a0[r2], b, 64), ajk1 = __shfl(a1[r2], b, 64);
              w0[u] = d21 * (d11 * ajk - ajk1);
              w1[u] = d21 * (d22 * ajk1 - ajk);
            } else {
              jj[u] = n;
              w0[u] = w1[u] = 0.0;
            }
          }
          double t[2][R];
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const int i = lane + 64 * r;
              t[u][r] = (i >= jj[u] && i < n) ? A[i + jj[u] * ld] : 0.0;
            }
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < R; ++r) {
              const int i = lane + 64 * r;
              if (i >= jj[u] && i < n) A[i + jj[u] * ld] = t[u][r] - (a0[r] * w0[u] + a1[r] * w1[u]);
            }
        }
      }
      wave_sync();
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int i = lane + 64 * r;
        if (i > k + 1 && i < n) {
          Ak[i] = d21 * (d11 * a0[r] - a1[r]);
          Ak1[i] = d21 * (d22 * a1[r] - a0[r]);
        }
      }
      wave_sync();
    }
    k += kstep;
  }
  *nneg_out = nneg;
  *nzero_out = nzero;
  return true;
}

// ---- solves with a standard-form factor (bk_factor_wave), right-hand side in registers -------
template <int R> __device__ inline double lane_get(const double (&x)[R], int idx) {
  double o = 0.0;
#pragma unroll
  for (int r = 0; r < R; ++r) { const double t = readlane_d(x[r], idx & 63); if ((idx >> 6) == r) o = t; }
  return o;
}
template <int R> __device__ inline int lane_geti(const int (&x)[R], int idx) {
  int o = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int t = __builtin_amdgcn_readlane(x[r], __builtin_amdgcn_readfirstlane(idx & 63));
    if ((idx >> 6) == r) o = t;
  }
  return o;
}
template <int R> __device__ inline void lane_set(double (&x)[R], int idx, double val, int lane) {
#pragma unroll
  for (int r = 0; r < R; ++r) if ((idx >> 6) == r && lane == (idx & 63)) x[r] = val;
}
template <int R> __device__ inline void lane_swap(double (&x)[R], int a, int b, int lane) {
  const double va = lane_get<R>(x, a), vb = lane_get<R>(x, b);
  lane_set<R>(x, a, vb, lane);
  lane_set<R>(x, b, va, lane);
}

// x = A^-1 b for P A P^T = L D L^T (unit lower L in standard form, D with 1x1 / 2x2 blocks,
// piv as DSYTF2 records it).  One wavefront; lane l holds entries l, l+64, ...; the chain of n
// dependent steps is readlane -> fma, the columns (forward) and rows (backward) of L stream
// from memory one step ahead of their use.
template <int R, class AP>
__device__ void bk_solve_reg(AP A, int n, int ld, const int* piv, double* v) {
  const int lane = threadIdx.x & 63;
  double x[R], dg[R], sd[R];
  int pr[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int i = lane + 64 * r;
    x[r] = i < n ? v[i] : 0.0;
    pr[r] = i < n ? piv[i] : 1;
    dg[r] = i < n ? A[i + i * ld] : 1.0;                 // D diagonal
    sd[r] = (i + 1 < n) ? A[i + 1 + i * ld] : 0.0;       // D sub-diagonal (2x2 blocks)
  }
  // P b
  for (int k = 0; k < n;) {
    const int p = lane_geti<R>(pr, k);
    if (p > 0) { if (p - 1 != k) lane_swap<R>(x, k, p - 1, lane); k += 1; }
    else { if (-p - 1 != k + 1) lane_swap<R>(x, k + 1, -p - 1, lane); k += 2; }
  }
  // L y = P b, then z = D^-1 y on the fly
  {
    double cur[R], nxt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { const int i = lane + 64 * r; cur[r] = (i > 0 && i < n) ? A[i] : 0.0; }
    for (int k = 0; k < n;) {
      const int p = lane_geti<R>(pr, k);
#pragma unroll
      for (int r = 0; r < R; ++r) { const int i = lane + 64 * r; nxt[r] = (k + 1 < n && i > k + 1 && i < n) ? A[i + (k + 1) * ld] : 0.0; }
      if (p > 0) {
        const double yk = lane_get<R>(x, k);
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] -= cur[r] * yk;          // cur is zero on rows <= k
        lane_set<R>(x, k, yk / lane_get<R>(dg, k), lane);
#pragma unroll
        for (int r = 0; r < R; ++r) cur[r] = nxt[r];
        k += 1;
      } else {
        const double yk = lane_get<R>(x, k), yk1 = lane_get<R>(x, k + 1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int i = lane + 64 * r;
          if (i > k + 1) x[r] -= cur[r] * yk + nxt[r] * yk1;      // cur holds the sub-diagonal at row k+1: skip it
        }
        const double akm1k = lane_get<R>(sd, k);
        const double akm1 = lane_get<R>(dg, k) / akm1k, ak = lane_get<R>(dg, k + 1) / akm1k;
        const double denom = akm1 * ak - 1.0, bkm1 = yk / akm1k, bkk = yk1 / akm1k;
        lane_set<R>(x, k, (ak * bkm1 - bkk) / denom, lane);
        lane_set<R>(x, k + 1, (akm1 * bkk - bkm1) / denom, lane);
#pragma unroll
        for (int r = 0; r < R; ++r) { const int i = lane + 64 * r; cur[r] = (k + 2 < n && i > k + 2 && i < n) ? A[i + (k + 2) * ld] : 0.0; }
        k += 2;
      }
    }
  }
  // L^T w = z in axpy form: row k of L streams in, entries c < k of x are updated
  {
    double cur[R], nxt[R];
#pragma unroll
    for (int r = 0; r < R; ++r) { const int c = lane + 64 * r; cur[r] = (c < n - 1) ? A[(n - 1) + c * ld] : 0.0; }
    for (int k = n - 1; k >= 0;) {
      const int p = lane_geti<R>(pr, k);
#pragma unroll
      for (int r = 0; r < R; ++r) { const int c = lane + 64 * r; nxt[r] = (k >= 1 && c < k - 1) ? A[(k - 1) + c * ld] : 0.0; }
      if (p > 0) {
        const double xk = lane_get<R>(x, k);
#pragma unroll
        for (int r = 0; r < R; ++r) x[r] -= cur[r] * xk;          // cur is zero on entries >= k
#pragma unroll
        for (int r = 0; r < R; ++r) cur[r] = nxt[r];
        k -= 1;
      } else {
        // block (k-1, k): nxt was loaded for c < k-1, cur for c < k (its entry k-1 is the D sub-diagonal)
        const double xk = lane_get<R>(x, k), xk1 = lane_get<R>(x, k - 1);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          const int c = lane + 64 * r;
          if (c < k - 1) x[r] -= cur[r] * xk + nxt[r] * xk1;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) { const int c = lane + 64 * r; cur[r] = (k >= 2 && c < k - 2) ? A[(k - 2) + c * ld] : 0.0; }
        k -= 2;
      }
    }
  }
  // P^T w: the interchanges in reverse order
  for (int k = n - 1; k >= 0;) {
    const int p = lane_geti<R>(pr, k);
    if (p > 0) { if (p - 1 != k) lane_swap<R>(x, k, p - 1, lane); k -= 1; }
    else { if (-p - 1 != k) lane_swap<R>(x, k, -p - 1, lane); k -= 2; }
  }
#pragma unroll
  for (int r = 0; r < R; ++r) { const int i = lane + 64 * r; if (i < n) v[i] = x[r]; }
}

// DSYTRS (lower) by ONE wavefront; right-hand side v and pivots piv in LDS.
template <class AP>
__device__ void bk_solve_wave(AP A, int n, int ld, const int* piv, double* v) {
  const int lane = threadIdx.x & 63;
  int k = 0;
  while (k < n) {
    AP Ak = A + k * ld;
    const int p = piv[k];
    if (p > 0) {
      const int kp = p - 1;
      if (lane == 0 && kp != k) { const double t = v[k]; v[k] = v[kp]; v[kp] = t; }
      wave_sync();
      const double bk = v[k];
      if (bk != 0.0)
        for (int i = k + 1 + lane; i < n; i += 64) v[i] -= Ak[i] * bk;
      if (lane == 0) v[k] = bk / Ak[k];
      k += 1;
    } else {
      AP Ak1 = Ak + ld;
      const int kp = -p - 1;
      if (lane == 0 && kp != k + 1) { const double t = v[k + 1]; v[k + 1] = v[kp]; v[kp] = t; }
      wave_sync();
      const double bk = v[k], bk1 = v[k + 1];
      for (int i = k + 2 + lane; i < n; i += 64) v[i] -= Ak[i] * bk + Ak1[i] * bk1;
      if (lane == 0) {
        const double akm1k = Ak[k + 1];
        const double akm1 = Ak[k] / akm1k, ak = Ak1[k + 1] / akm1k;
        const double denom = akm1 * ak - 1.0, bkm1 = bk / akm1k, bkk = bk1 / akm1k;
        v[k] = (ak * bkm1 - bkk) / denom;
        v[k + 1] = (akm1 * bkk - bkm1) / denom;
      }
      k += 2;
    }
    wave_sync();
  }
  k = n - 1;
  while (k >= 0) {
    const bool one = piv[k] > 0;
    AP Ak = A + k * ld;
    AP Akm = one ? Ak : Ak - ld;
    double s0 = 0.0, s1 = 0.0;
    for (int i = k + 1 + lane; i < n; i += 64) {
      const double bi = v[i];
      s0 += Ak[i] * bi;
      if (!one) s1 += Akm[i] * bi;
    }
    s0 = wave_all_sum(s0);
    if (!one) s1 = wave_all_sum(s1);
    if (lane == 0) {
      v[k] -= s0;
      if (!one) v[k - 1] -= s1;
      const int kp = (one ? piv[k] : -piv[k]) - 1;
      if (kp != k) { const double t = v[k]; v[k] = v[kp]; v[kp] = t; }
    }
    wave_sync();
    k -= one ? 1 : 2;
  }
}

// NT lanes work on one instance: 64 (one wavefront; small instances, several per CU) or 256.
// PK (packed, NT = 64 only): the wavefront is one of several in a workgroup, each with its own instance (batch.h
// batch_solve_packed_kernel: the instances of a compute unit share ONE LDS copy of the index arrays, which leaves room
// for every vector of all of them): lanes are tid() & 63, a "barrier" orders this wavefront's memory operations
// only (wavefronts of a workgroup run different instances and never wait for each other), and every vector the solver
// objects allocate before freeze_vectors() lives in LDS — the execution space says so (vectors_in_lds) and the solver's
// element accesses become LDS instructions (exec.h VecP).
template <int NT, bool PK = false>
struct BlockExecT {
  static_assert(!PK || NT == 64, "packed instances are one wavefront each");
  static constexpr bool vectors_in_lds = PK;
  __device__ static int tid() { return PK ? static_cast<int>(threadIdx.x & 63u) : static_cast<int>(threadIdx.x); }
  bool vec_frozen = false;          // PK: the solver's vectors are all allocated (in LDS); later allocations may go to the global slab
  __device__ void freeze_vectors() { vec_frozen = true; }
  static constexpr int kBatchThreads = NT;
  static constexpr bool is_device = false;          // model.h: generic lambda form of the flat sweep
  static constexpr bool has_log = false;
  static constexpr bool has_host_control = false;
  static constexpr bool objects_in_lds = true;      // batch.h places every solver object in LDS (DNLP_THIS_IN_LDS)
  static constexpr bool has_condensed_ls = false;
  static constexpr int kFilterCap = 32;
  struct Log { __device__ void append(const Log&) {} };
  struct FlatTableT {};
  struct LdltWork {
    int expect_neg = -1; bool time_updates = false; bool padded = false; bool standard = false;
    // orders 257 .. 2048 on four wavefronts: the panel-blocked factorisation of bk_panel.h (see bk_panels below)
    bool panels = false;
    BkState* st = nullptr; BkPanelSwaps* swaps = nullptr; double* bk_w = nullptr; i32* bk_perm = nullptr; i32* bk_dtype = nullptr;
  };
  static constexpr int kPanelMaxOrder = 2048;

  // per-workgroup bump allocators: global slab and (for the KKT matrix) LDS
  char* ws = nullptr;
  size_t ws_cap = 0, ws_off = 0;
  char* lds_pool = nullptr;
  size_t lds_cap = 0, lds_off = 0;
  int overflow = 0;
  int lds_mode = 0;          // 0: nothing in LDS, 1: the KKT matrix, 2: the vectors, 3: both
  // reduction scratch in LDS: two alternating buffers (one barrier per reduction)
  double* red = nullptr;     // 2 x 4 doubles
  int* redi = nullptr;       // 2 x 4 ints
  int parity = 0;
  // LDS staging of the right-hand side and the pivot vector for the single-wavefront solve
  double* vec = nullptr;     // kWaveSolveMax doubles
  int* piv = nullptr;        // kWaveSolveMax ints
  static constexpr int kWaveSolveMax = 512;

  __device__ BlockExecT(char* slab, size_t cap, char* lds, size_t ldscap, double* r, int* ri, double* v, int* pv)
      : ws(slab), ws_cap(cap), lds_pool(lds), lds_cap(ldscap), red(r), redi(ri), vec(v), piv(pv) {}

  __device__ void barrier() { if constexpr (PK) wave_sync(); else __syncthreads(); }

  template <class T> __device__ T* alloc(size_t n) {
    // (PK: 16-byte granules — four instances' vectors have to fit the LDS of a compute unit, and ~80 allocations of
    //  400-byte vectors lose 2.3 KB per instance to 64-byte granules)
    constexpr size_t gran = PK ? 16 : 64;
    size_t bytes = ((n ? n : 1) * sizeof(T) + gran - 1) & ~(gran - 1);
    char* p;
    const bool big = bytes >= 16384;
    bool in_lds = (big ? (lds_mode & 1) : (lds_mode & 2)) && lds_off + bytes <= lds_cap;
    if constexpr (PK) {
      in_lds = !big && lds_off + bytes <= lds_cap;
      // a solver vector outside LDS — too large for the pool or one of the 'big' ones that go to the slab — would break
      // vectors_in_lds (32-bit LDS pointers): the launch reports -198 and the host takes the regular kernel
      if (!in_lds && !vec_frozen) { overflow = 1; return reinterpret_cast<T*>(lds_pool); }
    }
    if (in_lds) {
      // LDS first: every map / reduction of the interior-point loop is one dependent memory
      // round trip, ~0.1 us in LDS against ~1 us in L2.  With lds_all the whole working set of
      // a small instance (vectors + KKT matrix) lives in LDS; otherwise only the KKT matrix does.
      p = lds_pool + lds_off;
      lds_off += bytes;
    } else {
      if (ws_off + bytes > ws_cap) { overflow = 1; return reinterpret_cast<T*>(ws); }
      p = ws + ws_off;
      ws_off += bytes;
    }
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
    for (size_t i = tid(); i < bytes / 8; i += kBatchThreads) q[i] = 0ull;
    barrier();
    return reinterpret_cast<T*>(p);
  }
  template <class T> __device__ T* ctl_alloc(size_t n) { return alloc<T>(n); }

  __device__ void copy(void* dst, const void* src, size_t bytes) {
    if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | bytes) & 7) == 0) {
      unsigned long long* d = static_cast<unsigned long long*>(dst);
      const unsigned long long* s = static_cast<const unsigned long long*>(src);
      for (size_t i = tid(); i < bytes / 8; i += kBatchThreads) d[i] = s[i];
    } else {
      char* d = static_cast<char*>(dst);
      const char* s = static_cast<const char*>(src);
      for (size_t i = tid(); i < bytes; i += kBatchThreads) d[i] = s[i];
    }
    barrier();
  }
  __device__ void h2d(void* dst, const void* src, size_t bytes) { copy(dst, src, bytes); }
  __device__ void d2h(void* dst, const void* src, size_t bytes) { copy(dst, src, bytes); }
  __device__ void d2d(void* dst, const void* src, size_t bytes) { copy(dst, src, bytes); }
  __device__ void zero(void* p, size_t bytes) {
    if (((reinterpret_cast<uintptr_t>(p) | bytes) & 7) == 0) {
      unsigned long long* d = static_cast<unsigned long long*>(p);
      for (size_t i = tid(); i < bytes / 8; i += kBatchThreads) d[i] = 0ull;
    } else {
      char* d = static_cast<char*>(p);
      for (size_t i = tid(); i < bytes; i += kBatchThreads) d[i] = 0;
    }
    barrier();
  }
  __device__ void sync() {}

  template <class F> __device__ void map(i64 n, F f) {
    for (i64 i = tid(); i < n; i += kBatchThreads) f(i);
    barrier();
  }

  // mode 0 sum, 1 max (NaN -> +inf), 2 min (NaN -> -inf): same conventions as HipExec::reduce
  template <int MODE, class F> __device__ double reduce(i64 n, F f) {
    double acc = MODE == 0 ? 0.0 : -kInf;
    for (i64 i = tid(); i < n; i += kBatchThreads) {
      double v = f(i);
      if (MODE == 0) acc += v;
      else if (MODE == 1) acc = fmax(acc, v != v ? kInf : v);
      else { v = v != v ? -kInf : v; acc = fmax(acc, -v); }
    }
    acc = MODE == 0 ? wave_all_sum(acc) : wave_all_max(acc);
    if constexpr (NT == 64) return MODE == 2 ? -acc : acc;     // the butterfly left the result in every lane
    double* buf = red + 4 * parity;
    parity ^= 1;
    if ((tid() & 63) == 0) buf[tid() >> 6] = acc;
    barrier();
    double r = buf[0];
    for (int k = 1; k < kBatchThreads / 64; ++k) r = MODE == 0 ? r + buf[k] : fmax(r, buf[k]);
    return MODE == 2 ? -r : r;
  }
  template <class F> __device__ double sum(i64 n, F f) { return reduce<0>(n, f); }
  template <class F> __device__ double max(i64 n, F f) { return reduce<1>(n, f); }
  template <class F> __device__ double min(i64 n, F f) { return reduce<2>(n, f); }
  // NM maxima and NS sums in one pass over the data
  template <int NM, int NS, class F> __device__ RMulti reduce_multi(i64 n, F f) {
    RMulti r;
#pragma unroll
    for (int k = 0; k < 4; ++k) { r.mx[k] = -kInf; r.sm[k] = 0.0; }
    for (i64 i = tid(); i < n; i += kBatchThreads) {
      const RMulti v = f(i);
#pragma unroll
      for (int k = 0; k < NM; ++k) r.mx[k] = fmax(r.mx[k], v.mx[k] != v.mx[k] ? kInf : v.mx[k]);
#pragma unroll
      for (int k = 0; k < NS; ++k) r.sm[k] += v.sm[k];
    }
#pragma unroll
    for (int k = 0; k < NM; ++k) r.mx[k] = wave_all_max(r.mx[k]);
#pragma unroll
    for (int k = 0; k < NS; ++k) r.sm[k] = wave_all_sum(r.sm[k]);
    if constexpr (NT == 64) return r;
    // several wavefronts: one value at a time through the two-slot staging buffer of reduce()
#pragma unroll
    for (int k = 0; k < NM + NS; ++k) {
      double* buf = red + 4 * parity;
      parity ^= 1;
      const double mine = k < NM ? r.mx[k] : r.sm[k - NM];
      if ((tid() & 63) == 0) buf[tid() >> 6] = mine;
      barrier();
      double t = buf[0];
      for (int w = 1; w < kBatchThreads / 64; ++w) t = k < NM ? fmax(t, buf[w]) : t + buf[w];
      if (k < NM) r.mx[k] = t; else r.sm[k - NM] = t;
    }
    return r;
  }
  // two minima in one pass (NaN -> -inf, as min)
  template <class F> __device__ D2 min2(i64 n, F f) {
    double a0 = -kInf, a1 = -kInf;                   // min via max of the negatives
    for (i64 i = tid(); i < n; i += kBatchThreads) {
      const D2 v = f(i);
      a0 = fmax(a0, v.first != v.first ? kInf : -v.first);
      a1 = fmax(a1, v.second != v.second ? kInf : -v.second);
    }
    a0 = wave_all_max(a0);
    a1 = wave_all_max(a1);
    if constexpr (NT == 64) return D2{-a0, -a1};
    double* buf = red + 4 * parity;
    parity ^= 1;
    if ((tid() & 63) == 0) buf[tid() >> 6] = a0;
    barrier();
    double r0 = buf[0];
    for (int k = 1; k < kBatchThreads / 64; ++k) r0 = fmax(r0, buf[k]);
    double* buf1 = red + 4 * parity;
    parity ^= 1;
    if ((tid() & 63) == 0) buf1[tid() >> 6] = a1;
    barrier();
    double r1 = buf1[0];
    for (int k = 1; k < kBatchThreads / 64; ++k) r1 = fmax(r1, buf1[k]);
    return D2{-r0, -r1};
  }

  // block-wide argmax of v (first index wins ties, as IDAMAX); every lane gets the result
  __device__ void argmax(double v, int idx, double& outv, int& outi) {
    wave_all_argmax(v, idx);
    if constexpr (NT == 64) { outv = v; outi = idx; return; }
    double* bv = red + 4 * parity;
    int* bi = redi + 4 * parity;
    parity ^= 1;
    if ((tid() & 63) == 0) { bv[tid() >> 6] = v; bi[tid() >> 6] = idx; }
    barrier();
    outv = bv[0];
    outi = bi[0];
    for (int k = 1; k < kBatchThreads / 64; ++k)
      if (bv[k] > outv || (bv[k] == outv && bi[k] < outi)) { outv = bv[k]; outi = bi[k]; }
  }

  // static-pattern sparse LDL^T (sparse_ldl.h): the lanes of this instance are the policy
  struct Par {
    BlockExecT* ex;
    __device__ int lanes() const { return NT; }
    __device__ int lane() const { return BlockExecT::tid(); }
    __device__ void sync() const { ex->barrier(); }
    // (PK: the factor values, the work array and the right-hand side are solver vectors: LDS)
    template <class T> __device__ T* vec(T* p) const { DNLP_VEC_IN_LDS(BlockExecT, p); return p; }
    __device__ double sum(double v) const {
      v = wave_all_sum(v);
      if constexpr (NT == 64) return v;
      double* buf = ex->red + 4 * ex->parity;
      ex->parity ^= 1;
      if ((BlockExecT::tid() & 63) == 0) buf[BlockExecT::tid() >> 6] = v;
      ex->barrier();
      double r = buf[0];
      for (int k = 1; k < NT / 64; ++k) r += buf[k];
      return r;
    }
  };
  __device__ bool sparse_factor(const SparsePlan& pl, double* vals, double* w, int* nneg, int* nzero) {
    return sparse_ldl_factor(pl, vals, w, nneg, nzero, Par{this});
  }
  __device__ void sparse_solve(const SparsePlan& pl, const double* vals, double* x) { sparse_ldl_solve(pl, vals, x, Par{this}); }

  // facilities of the large dense path that a batch instance never uses (the host rejects
  // tapes with dense quad_form blocks before launching)
  __device__ void gemv_sym(i64, const double*, i64, const double*, double*) { __builtin_trap(); }
  __device__ void orthogonalize(int, const double*, i64, double*, double*) { __builtin_trap(); }
  __device__ void dense_block_add(double*, i64, i64, const double*, i64, i64, double, bool) { __builtin_trap(); }
  __device__ void ldlt_prepare(LdltWork& lw, i64 n, i64, bool) {
    if constexpr (NT == 256 && !PK) {
      if (n > 256 && n <= kPanelMaxOrder) {
        lw.panels = true;
        lw.st = alloc<BkState>(1);
        lw.swaps = alloc<BkPanelSwaps>(1);
        lw.bk_w = alloc<double>(static_cast<size_t>((n + 7) / 8 * 8) * 16);
        lw.bk_perm = alloc<i32>(static_cast<size_t>(n));
        lw.bk_dtype = alloc<i32>(static_cast<size_t>(n));
      }
    }
  }

  // out = J v / J^T v / sym(H) v through the tape's index by output (tape.h CooIdx): a lane owns an output and sums
  // its segment serially, in storage order; an output with a long segment (a variable under many rows: both position
  // coordinates of the localization example sit under all fifty range rows) is summed by ALL lanes and the fixed
  // reduction tree of reduce().  No atomics: the same bits on every run.
  __device__ void coo_gather(const CooIdx& ix, const double* a, const double* v, double* out) {
    DNLP_VEC_IN_LDS(BlockExecT, a); DNLP_VEC_IN_LDS(BlockExecT, v); DNLP_VEC_IN_LDS(BlockExecT, out);   // (values, operand and result are solver vectors)
    const i32* ptr = ix.ptr;
    const i32 *ent = ix.ent, *src = ix.src;
    for (i64 g = tid(); g < ix.nout; g += kBatchThreads) {
      const i64 p0 = ptr[g], p1 = ptr[g + 1];
      if (p1 - p0 > CooIdx::kHeavy) continue;
      double s = 0.0;
      for (i64 p = p0; p < p1; ++p) s += a[ent[p]] * v[src[p]];
      out[g] = s;                    // (assigned: no zeroing pass before the product)
    }
    for (i64 h = 0; h < ix.nheavy; ++h) {
      const i64 g = ix.heavy[h];
      const i64 p0 = ptr[g];
      const double s = reduce<0>(ptr[g + 1] - p0, [=] __device__(i64 q) { return a[ent[p0 + q]] * v[src[p0 + q]]; });
      if (tid() == 0) out[g] = s;
    }
    barrier();
  }
  // scatter form with atomics: only for patterns the tape did not index
  __device__ void coo_mult(i64 nnz, const i32* r, const i32* c, const double* a, const double* v, double* out, bool trans) {
    for (i64 p = tid(); p < nnz; p += kBatchThreads) {
      if (trans) unsafeAtomicAdd(&out[c[p]], a[p] * v[r[p]]);
      else unsafeAtomicAdd(&out[r[p]], a[p] * v[c[p]]);
    }
    barrier();
  }
  __device__ void coo_sym_mult(i64 nnz, const i32* r, const i32* c, const double* a, const double* v, double* out) {
    for (i64 p = tid(); p < nnz; p += kBatchThreads) {
      const double av = a[p];
      unsafeAtomicAdd(&out[r[p]], av * v[c[p]]);
      if (r[p] != c[p]) unsafeAtomicAdd(&out[c[p]], av * v[r[p]]);
    }
    barrier();
  }

  // ---- Bunch-Kaufman LDL^T by one workgroup (DSYTF2 semantics, lower storage) -------------
  // Same pivot rule, interchanges, multipliers and inertia count as bk_pivot_kernel /
  // bk_update_kernel of exec_hip.h and as the DSYTF2 restatement of the test oracle.
  // ---- orders 257 .. 2048, four wavefronts: panel-blocked Bunch-Kaufman (bk_panel.h) ----------------------------
  // The unblocked loop below touches the whole trailing matrix once per column from one workgroup (36 M global
  // read-modify-writes at order 600: 0.12 s per factorisation, a 64-start best_of on a dense order-600 problem took twice
  // as long batched as one by one).  Here the instance's workgroup runs the host-driven path's panel body (W = L D of 16
  // columns in registers, the DSYTF2 pivot choices) and then the rank-16 trailing update itself: every trailing entry is
  // touched once per PANEL.  The factor comes out in the fully permuted form (interchanges applied to earlier columns
  // too); bk_panels_solve below is the matching solve (32-column blocks, the vector in LDS).
  // (the LDS arrays of the three phases — update strips, permutation pass, solve — are never live together: one pool)
  static constexpr int kBkPoolDoubles = kPanelMaxOrder + 32 * 33 + 8;
  __device__ static double* bk_pool() {
    __shared__ __attribute__((aligned(16))) double pool[kBkPoolDoubles];
    return pool;
  }
  template <int ROWS, int NBP>
  __device__ void bk_panels(LdltWork& lw, double* A, int n, i64 ld, i32* ipiv) {
    double (*Ls)[NBP + 1] = reinterpret_cast<double (*)[NBP + 1]>(bk_pool());
    const int tid = BlockExecT::tid();
    const i64 ldw = (n + 7) / 8 * 8;
    BkState* st = lw.st;
    BkPanelSwaps* swaps = lw.swaps;
    const double* Wg = lw.bk_w;
    for (;;) {
      bk_panel_body<ROWS, NBP, NT>(A, n, ld, ipiv, st, lw.bk_w, ldw, swaps);
      __syncthreads();
      const int k0 = st->kp, cnt = st->kstep, pending = st->pending, fail = st->fail, knext = st->k, ns = swaps->count;
      if (!pending || cnt <= 0) break;
      // the panel's interchanges on the rows of the columns of EARLIER panels (one lane per column, the swaps in order)
      if (ns > 0)
        for (int q = tid; q < k0; q += NT) {
          double* col = A + static_cast<i64>(q) * ld;
          for (int e = 0; e < ns; ++e) {
            const int ra = swaps->rows[2 * e], rb = swaps->rows[2 * e + 1];
            const double t = col[ra];
            col[ra] = col[rb];
            col[rb] = t;
          }
        }
      const int j0 = k0 + cnt;
      if (!fail && j0 < n) {
        // A22 -= W21 L21^T: a lane keeps the W rows it owns for the whole update, 16-column strips of multipliers in LDS
        double w[ROWS][NBP];
#pragma unroll
        for (int s2 = 0; s2 < ROWS; ++s2) {
          const int r = tid + s2 * NT;
#pragma unroll
          for (int i = 0; i < NBP; ++i) w[s2][i] = (r >= j0 && r < n) ? Wg[r + static_cast<i64>(i) * ldw] : 0.0;
        }
        for (int cstart = j0; cstart < n; cstart += 16) {
          for (int e = tid; e < 16 * NBP; e += NT) {
            const int cc = e % 16, i = e / 16;
            Ls[cc][i] = (cstart + cc < n && i < cnt) ? A[(cstart + cc) + static_cast<i64>(k0 + i) * ld] : 0.0;
          }
          __syncthreads();
#pragma unroll
          for (int s2 = 0; s2 < ROWS; ++s2) {
            const int r = tid + s2 * NT;
            if (r >= cstart && r < n) {
              double* row = A + r + static_cast<i64>(cstart) * ld;
              double cur[16];
#pragma unroll
              for (int cc = 0; cc < 16; ++cc) cur[cc] = (cstart + cc <= r) ? row[static_cast<i64>(cc) * ld] : 0.0;
#pragma unroll
              for (int cc = 0; cc < 16; ++cc) {
                double sacc = 0.0;
#pragma unroll
                for (int i = 0; i < NBP; ++i) sacc += w[s2][i] * Ls[cc][i];
                if (cstart + cc <= r) row[static_cast<i64>(cc) * ld] = cur[cc] - sacc;
              }
            }
          }
          __syncthreads();
        }
      }
      __syncthreads();
      if (fail || knext >= n) break;
    }
  }
  __device__ bool bk_panels_factor(LdltWork& lw, double* A, int n, i64 ld, i32* ipiv, int* nneg_out, int* nzero_out) {
    static_assert(3 * kPanelMaxOrder * sizeof(int) <= kBkPoolDoubles * sizeof(double), "permutation pass does not fit the pool");
    int* sp = reinterpret_cast<int*>(bk_pool());
    int *sdt = sp + kPanelMaxOrder, *spv = sdt + kPanelMaxOrder;
    const int tid = BlockExecT::tid();
    if (tid == 0) {
      BkState z;
      z.k = z.kstep = z.kp = z.pending = z.nneg = z.nzero = z.fail = z.pad = 0;
      z.d11 = z.d22 = z.d21 = 0.0;
      *lw.st = z;
      lw.swaps->count = 0;
    }
    __syncthreads();
    // rows per lane sized to the order: the register tile W (ROWS x NBP doubles) and the per-column work follow
    if (n <= 2 * NT) bk_panels<2, 16>(lw, A, n, ld, ipiv);
    else if (n <= 3 * NT) bk_panels<3, 16>(lw, A, n, ld, ipiv);
    else if (n <= 4 * NT) bk_panels<4, 8>(lw, A, n, ld, ipiv);
    else if (n <= 6 * NT) bk_panels<6, 8>(lw, A, n, ld, ipiv);
    else bk_panels<8, 8>(lw, A, n, ld, ipiv);
    __syncthreads();
    // the interchanges as ONE permutation and the pivot structure (0: 1x1, 1 / 2: first / second column of a 2x2 block)
    for (int i = tid; i < n; i += NT) { sp[i] = i; spv[i] = ipiv[i]; }
    __syncthreads();
    if (tid == 0) {
      int k = 0;
      while (k < n) {
        if (spv[k] > 0) {
          const int kp = spv[k] - 1;
          if (kp != k) { const int t = sp[k]; sp[k] = sp[kp]; sp[kp] = t; }
          sdt[k] = 0;
          k += 1;
        } else {
          const int kp = -spv[k] - 1;
          if (k + 1 < n && kp != k + 1) { const int t = sp[k + 1]; sp[k + 1] = sp[kp]; sp[kp] = t; }
          sdt[k] = 1;
          if (k + 1 < n) sdt[k + 1] = 2;
          k += 2;
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < n; i += NT) { lw.bk_perm[i] = sp[i]; lw.bk_dtype[i] = sdt[i]; }
    __syncthreads();
    *nneg_out = lw.st->nneg;
    *nzero_out = lw.st->nzero;
    return lw.st->fail == 0;
  }
  // P A P^T = L D L^T x = b with the vector in LDS: 32-column blocks, the diagonal block by one wavefront (lane = row,
  // v_readlane broadcasts), the rows below / the columns' dot products by all four (exec_hip.h bk_solve_kernel's form)
  __device__ void bk_panels_solve(const LdltWork& lw, const double* A, int n, i64 ld, double* b) {
    double* x = bk_pool();
    double (*Lb)[33] = reinterpret_cast<double (*)[33]>(x + kPanelMaxOrder);
    const int tid = BlockExecT::tid(), lane = tid & 63, wave = tid >> 6;
    const i32 *perm = lw.bk_perm, *dtype = lw.bk_dtype;
    for (int i = tid; i < n; i += NT) x[i] = b[perm[i]];
    __syncthreads();
    auto stage_diag = [&](int j0, int jb) {
      for (int e = tid; e < jb * jb; e += NT) {
        const int r = e % jb, c = e / jb;
        double v = (r > c) ? A[(j0 + r) + static_cast<i64>(j0 + c) * ld] : 0.0;
        if (r == c + 1 && dtype[j0 + c] == 1) v = 0.0;          // the off-diagonal of a 2x2 pivot is D, not L
        Lb[r][c] = v;
      }
    };
    for (int j0 = 0; j0 < n; j0 += 32) {                         // forward: L y = P b
      const int jb = n - j0 < 32 ? n - j0 : 32;
      stage_diag(j0, jb);
      __syncthreads();
      if (wave == 0) {
        double y = lane < jb ? x[j0 + lane] : 0.0;
        for (int c = 0; c < jb; ++c) {
          const double yc = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(y), c), __builtin_amdgcn_readlane(__double2loint(y), c));
          if (lane > c && lane < jb) y -= Lb[lane][c] * yc;
        }
        if (lane < jb) x[j0 + lane] = y;
      }
      __syncthreads();
      const int r0 = j0 + jb;
      const bool straddle = dtype[r0 - 1] == 1;                  // 2x2 pivot across the block boundary
      for (int r = r0 + tid; r < n; r += NT) {
        double s = 0.0;
#pragma unroll 8
        for (int c = 0; c < jb; ++c) s += A[r + static_cast<i64>(j0 + c) * ld] * x[j0 + c];
        if (straddle && r == r0) s -= A[r + static_cast<i64>(r0 - 1) * ld] * x[r0 - 1];
        x[r] -= s;
      }
      __syncthreads();
    }
    for (int k = tid; k < n; k += NT) {                          // D z = y
      const int ty = dtype[k];
      if (ty == 0) {
        x[k] /= A[k + static_cast<i64>(k) * ld];
      } else if (ty == 1) {
        const double akm1k = A[k + 1 + static_cast<i64>(k) * ld];
        const double akm1 = A[k + static_cast<i64>(k) * ld] / akm1k, ak = A[k + 1 + static_cast<i64>(k + 1) * ld] / akm1k;
        const double denom = akm1 * ak - 1.0, bkm1 = x[k] / akm1k, bkk = x[k + 1] / akm1k;
        x[k] = (ak * bkm1 - bkk) / denom;
        x[k + 1] = (akm1 * bkk - bkm1) / denom;
      }
    }
    __syncthreads();
    for (int j0 = (n - 1) / 32 * 32; j0 >= 0; j0 -= 32) {       // backward: L^T w = z
      const int jb = n - j0 < 32 ? n - j0 : 32;
      const int r0 = j0 + jb;
      stage_diag(j0, jb);
      const bool straddle = r0 < n && dtype[r0 - 1] == 1;
      for (int c = wave; c < jb; c += NT / 64) {
        double s = 0.0;
        for (int r = r0 + lane; r < n; r += 64) s += A[r + static_cast<i64>(j0 + c) * ld] * x[r];
        s = wave_all_sum(s);
        if (lane == 0) {
          if (straddle && c == jb - 1) s -= A[r0 + static_cast<i64>(r0 - 1) * ld] * x[r0];
          x[j0 + c] -= s;
        }
      }
      __syncthreads();
      if (wave == 0) {
        double w = lane < jb ? x[j0 + lane] : 0.0;
        for (int c = jb - 1; c >= 0; --c) {
          const double wc = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(w), c), __builtin_amdgcn_readlane(__double2loint(w), c));
          if (lane < c) w -= Lb[c][lane] * wc;
        }
        if (lane < jb) x[j0 + lane] = w;
      }
      __syncthreads();
    }
    for (int i = tid; i < n; i += NT) b[perm[i]] = x[i];
    __syncthreads();
  }

  __device__ bool ldlt_factor(LdltWork& lw, double* A, i64 nn, i64 ld, i32* ipiv, bool, int* nneg_out, int* nzero_out) {
    const int n = static_cast<int>(nn), tid = BlockExecT::tid();
    lw.standard = false;
    if constexpr (NT == 256 && !PK) {
      if (lw.panels) {
        const bool ok = bk_panels_factor(lw, A, n, ld, ipiv, nneg_out, nzero_out);
        parity = 0;
        return ok;
      }
    }
    if (n <= 256) {
      // small instance: one wavefront, no workgroup barrier inside; L comes out in standard form
      lw.standard = true;
      if (tid < 64) {
        int nn_ = 0, nz_ = 0;
        bool ok;
        const int ldi = static_cast<int>(ld);
        if (in_lds(A)) {
          lds_double* L = (lds_double*)A;
          if (n <= 64) ok = bk_factor_wave<1>(L, n, ldi, piv, &nn_, &nz_);
          else if (n <= 128) ok = bk_factor_wave<2>(L, n, ldi, piv, &nn_, &nz_);
          else ok = bk_factor_wave<4>(L, n, ldi, piv, &nn_, &nz_);
        } else {
          if (n <= 64) ok = bk_factor_wave<1>(A, n, ldi, piv, &nn_, &nz_);
          else if (n <= 128) ok = bk_factor_wave<2>(A, n, ldi, piv, &nn_, &nz_);
          else ok = bk_factor_wave<4>(A, n, ldi, piv, &nn_, &nz_);
        }
        if (tid == 0) { redi[0] = ok ? 1 : 0; redi[1] = nn_; redi[2] = nz_; }
      }
      barrier();
      const bool ok = redi[0] != 0;
      *nneg_out = redi[1];
      *nzero_out = redi[2];
      for (int i = tid; i < n; i += kBatchThreads) ipiv[i] = piv[i];
      barrier();
      parity = 0;     // redi[0..2] were used outside the alternating scheme: restart it
      return ok;
    }
    const int lane = tid & 63, wv = tid >> 6;
    const double alpha = 0.6403882032022076;   // (1 + sqrt(17)) / 8
    int k = 0, nneg = 0, nzero = 0;
    while (k < n) {
      double* Ak = A + static_cast<i64>(k) * ld;
      double v = -1.0;
      int idx = n;
      for (int i = k + 1 + tid; i < n; i += kBatchThreads) {
        const double a = fabs(Ak[i]);
        if (a > v) { v = a; idx = i; }
      }
      double colmax;
      int imax;
      argmax(v, idx, colmax, imax);
      if (colmax < 0.0) { colmax = 0.0; imax = k; }
      const double absakk = fabs(Ak[k]);
      if (!(absakk == absakk) || !(colmax == colmax)) return false;
      int kstep = 1, kp = k;
      bool zero_piv = false;
      if (fmax(absakk, colmax) == 0.0) {
        zero_piv = true;
      } else if (absakk < alpha * colmax) {
        double rv = 0.0;
        for (int j = k + tid; j < imax; j += kBatchThreads) rv = fmax(rv, fabs(A[imax + static_cast<i64>(j) * ld]));
        for (int i = imax + 1 + tid; i < n; i += kBatchThreads) rv = fmax(rv, fabs(A[i + static_cast<i64>(imax) * ld]));
        double rowmax;
        int dummy;
        argmax(rv, tid, rowmax, dummy);
        const double aii = fabs(A[imax + static_cast<i64>(imax) * ld]);
        if (absakk >= alpha * colmax * (colmax / rowmax)) kp = k;
        else if (aii >= alpha * rowmax) kp = imax;
        else { kp = imax; kstep = 2; }
      }
      const int kk = k + kstep - 1;
      if (!zero_piv && kp != kk) {
        // symmetric interchange of rows/columns kk and kp in the trailing matrix
        double* Akk = A + static_cast<i64>(kk) * ld;
        double* Akp = A + static_cast<i64>(kp) * ld;
        for (int i = kp + 1 + tid; i < n; i += kBatchThreads) { const double t = Akk[i]; Akk[i] = Akp[i]; Akp[i] = t; }
        for (int j = kk + 1 + tid; j < kp; j += kBatchThreads) {
          const double t = Akk[j];
          Akk[j] = A[kp + static_cast<i64>(j) * ld];
          A[kp + static_cast<i64>(j) * ld] = t;
        }
        barrier();
        if (tid == 0) {
          const double t = Akk[kk];
          Akk[kk] = Akp[kp];
          Akp[kp] = t;
          if (kstep == 2) { const double t2 = Ak[k + 1]; Ak[k + 1] = Ak[kp]; Ak[kp] = t2; }
        }
        barrier();
      }
      if (zero_piv) {
        // structurally empty column: unit tiny pivot, reported through nzero
        if (tid == 0) { Ak[k] = 1e-20; ipiv[k] = k + 1; }
        ++nzero;
        barrier();
        k += 1;
        continue;
      }
      if (kstep == 1) {
        const double d = Ak[k];
        if (d < 0.0) ++nneg;
        if (fabs(d) < 1e-300) ++nzero;
        if (tid == 0) ipiv[k] = kp + 1;
        const double inv = 1.0 / d;
        // trailing update: wavefront w owns columns j = k+1+w, k+5+w, ...; lanes walk the rows
        for (int j = k + 1 + wv; j < n; j += kBatchThreads / 64) {
          const double wj = Ak[j] * inv;
          if (wj == 0.0) continue;
          double* Aj = A + static_cast<i64>(j) * ld;
          for (int i = j + lane; i < n; i += 64) Aj[i] -= Ak[i] * wj;
        }
        barrier();
        for (int i = k + 1 + tid; i < n; i += kBatchThreads) Ak[i] *= inv;
        barrier();
      } else {
        double* Ak1 = Ak + ld;
        double d21 = Ak[k + 1];
        const double d11 = Ak1[k + 1] / d21, d22 = Ak[k] / d21;
        const double tt = 1.0 / (d11 * d22 - 1.0);
        d21 = tt / d21;
        ++nneg;
        if (tid == 0) { ipiv[k] = -(kp + 1); ipiv[k + 1] = -(kp + 1); }
        for (int j = k + 2 + wv; j < n; j += kBatchThreads / 64) {
          const double ajk = Ak[j], ajk1 = Ak1[j];
          const double wk = d21 * (d11 * ajk - ajk1), wk1 = d21 * (d22 * ajk1 - ajk);
          double* Aj = A + static_cast<i64>(j) * ld;
          for (int i = j + lane; i < n; i += 64) Aj[i] -= Ak[i] * wk + Ak1[i] * wk1;
        }
        barrier();
        for (int j = k + 2 + tid; j < n; j += kBatchThreads) {
          const double ajk = Ak[j], ajk1 = Ak1[j];
          Ak[j] = d21 * (d11 * ajk - ajk1);
          Ak1[j] = d21 * (d22 * ajk1 - ajk);
        }
        barrier();
      }
      k += kstep;
    }
    *nneg_out = nneg;
    *nzero_out = nzero;
    return true;
  }

  __device__ bool in_lds(const void* p) const {
    const char* c = static_cast<const char*>(p);
    return lds_cap > 0 && c >= lds_pool && c < lds_pool + lds_cap;
  }

  // Small orders: the substitution is a chain of n dependent steps, so workgroup barriers (two
  // to three per column) are the whole cost of the block form.  The right-hand side and the
  // pivot vector are staged in LDS, wavefront 0 walks the columns (bk_solve_wave) and the other
  // wavefronts wait at one barrier.
  __device__ void ldlt_solve_wave(const double* A, int n, i64 ld, const i32* ipiv, double* b, bool standard) {
    const int tid = BlockExecT::tid();
    for (int i = tid; i < n; i += kBatchThreads) { vec[i] = b[i]; piv[i] = ipiv[i]; }
    barrier();
    if (tid < 64) {
      const int ldi = static_cast<int>(ld);
      if (standard) {
        if (in_lds(A)) {
          const lds_double* L = (const lds_double*)A;
          if (n <= 64) bk_solve_reg<1>(L, n, ldi, piv, vec);
          else if (n <= 128) bk_solve_reg<2>(L, n, ldi, piv, vec);
          else bk_solve_reg<4>(L, n, ldi, piv, vec);
        } else {
          if (n <= 64) bk_solve_reg<1>(A, n, ldi, piv, vec);
          else if (n <= 128) bk_solve_reg<2>(A, n, ldi, piv, vec);
          else bk_solve_reg<4>(A, n, ldi, piv, vec);
        }
      } else if (in_lds(A)) {
        bk_solve_wave((const lds_double*)A, n, ldi, piv, vec);
      } else {
        bk_solve_wave(A, n, ldi, piv, vec);
      }
    }
    barrier();
    for (int i = tid; i < n; i += kBatchThreads) b[i] = vec[i];
    barrier();
  }

  // DSYTRS (lower) by one workgroup; b in exec-space memory
  __device__ void ldlt_solve(LdltWork& lw, const double* A, i64 nn, i64 ld, const i32* ipiv, bool, double* b) {
    const int n = static_cast<int>(nn), tid = BlockExecT::tid();
    if constexpr (NT == 256 && !PK) {
      if (lw.panels) { bk_panels_solve(lw, A, n, ld, b); return; }
    }
    if (n <= kWaveSolveMax) { ldlt_solve_wave(A, n, ld, ipiv, b, lw.standard); return; }
    int k = 0;
    while (k < n) {
      const double* Ak = A + static_cast<i64>(k) * ld;
      if (ipiv[k] > 0) {
        const int kp = ipiv[k] - 1;
        if (tid == 0 && kp != k) { const double t = b[k]; b[k] = b[kp]; b[kp] = t; }
        barrier();
        const double bk = b[k];
        for (int i = k + 1 + tid; i < n; i += kBatchThreads) b[i] -= Ak[i] * bk;
        barrier();
        if (tid == 0) b[k] = bk / Ak[k];
        k += 1;
      } else {
        const double* Ak1 = Ak + ld;
        const int kp = -ipiv[k] - 1;
        if (tid == 0 && kp != k + 1) { const double t = b[k + 1]; b[k + 1] = b[kp]; b[kp] = t; }
        barrier();
        const double bk = b[k], bk1 = b[k + 1];
        for (int i = k + 2 + tid; i < n; i += kBatchThreads) b[i] -= Ak[i] * bk + Ak1[i] * bk1;
        barrier();
        if (tid == 0) {
          const double akm1k = Ak[k + 1];
          const double akm1 = Ak[k] / akm1k, ak = Ak1[k + 1] / akm1k;
          const double denom = akm1 * ak - 1.0, bkm1 = bk / akm1k, bkk = bk1 / akm1k;
          b[k] = (ak * bkm1 - bkk) / denom;
          b[k + 1] = (akm1 * bkk - bkm1) / denom;
        }
        k += 2;
      }
      barrier();
    }
    k = n - 1;
    while (k >= 0) {
      const bool one = ipiv[k] > 0;
      const double* Ak = A + static_cast<i64>(k) * ld;
      const double* Akm = one ? Ak : Ak - ld;
      const double s0 = reduce<0>(n - k - 1, [=] __device__(i64 q) { return Ak[k + 1 + q] * b[k + 1 + q]; });
      const double s1 = one ? 0.0 : reduce<0>(n - k - 1, [=] __device__(i64 q) { return Akm[k + 1 + q] * b[k + 1 + q]; });
      if (tid == 0) {
        b[k] -= s0;
        if (!one) b[k - 1] -= s1;
        const int kp = (one ? ipiv[k] : -ipiv[k]) - 1;
        if (kp != k) { const double t = b[k]; b[k] = b[kp]; b[kp] = t; }
      }
      barrier();
      k -= one ? 1 : 2;
    }
  }
};

}  // namespace dnlp
