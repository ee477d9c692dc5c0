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

namespace dnlp {

constexpr int kBatchThreads = 256;

struct BlockExec {
  static constexpr bool is_device = false;          // model.h: generic lambda form of the flat sweep
  static constexpr bool has_log = false;
  static constexpr bool has_host_control = false;
  static constexpr int kFilterCap = 32;
  struct Log { __device__ void append(const Log&) {} };
  struct FlatTableT {};
  struct LdltWork { int expect_neg = -1; bool time_updates = false; bool padded = false; };

  // per-workgroup bump allocators: global slab and (for the KKT matrix) LDS
  char* ws = nullptr;
  size_t ws_cap = 0, ws_off = 0;
  char* lds_pool = nullptr;
  size_t lds_cap = 0, lds_off = 0;
  int overflow = 0;
  // reduction scratch in LDS: two alternating buffers (one barrier per reduction)
  double* red = nullptr;     // 2 x 4 doubles
  int* redi = nullptr;       // 2 x 4 ints
  int parity = 0;

  __device__ BlockExec(char* slab, size_t cap, char* lds, size_t ldscap, double* r, int* ri)
      : ws(slab), ws_cap(cap), lds_pool(lds), lds_cap(ldscap), red(r), redi(ri) {}

  __device__ void barrier() { __syncthreads(); }

  template <class T> __device__ T* alloc(size_t n) {
    size_t bytes = ((n ? n : 1) * sizeof(T) + 63) & ~static_cast<size_t>(63);
    char* p;
    if (bytes >= 16384 && lds_off + bytes <= lds_cap) {
      // the one large block of a small instance is its KKT matrix: keep it in LDS
      p = lds_pool + lds_off;
      lds_off += bytes;
    } else {
      if (ws_off + bytes > ws_cap) { overflow = 1; return reinterpret_cast<T*>(ws); }
      p = ws + ws_off;
      ws_off += bytes;
    }
    unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
    for (size_t i = threadIdx.x; i < bytes / 8; i += kBatchThreads) q[i] = 0ull;
    __syncthreads();
    return reinterpret_cast<T*>(p);
  }
  template <class T> __device__ T* ctl_alloc(size_t n) { return alloc<T>(n); }

  __device__ void copy(void* dst, const void* src, size_t bytes) {
    if (((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | bytes) & 7) == 0) {
      unsigned long long* d = static_cast<unsigned long long*>(dst);
      const unsigned long long* s = static_cast<const unsigned long long*>(src);
      for (size_t i = threadIdx.x; i < bytes / 8; i += kBatchThreads) d[i] = s[i];
    } else {
      char* d = static_cast<char*>(dst);
      const char* s = static_cast<const char*>(src);
      for (size_t i = threadIdx.x; i < bytes; i += kBatchThreads) d[i] = s[i];
    }
    __syncthreads();
  }
  __device__ void h2d(void* dst, const void* src, size_t bytes) { copy(dst, src, bytes); }
  __device__ void d2h(void* dst, const void* src, size_t bytes) { copy(dst, src, bytes); }
  __device__ void d2d(void* dst, const void* src, size_t bytes) { copy(dst, src, bytes); }
  __device__ void zero(void* p, size_t bytes) {
    if (((reinterpret_cast<uintptr_t>(p) | bytes) & 7) == 0) {
      unsigned long long* d = static_cast<unsigned long long*>(p);
      for (size_t i = threadIdx.x; i < bytes / 8; i += kBatchThreads) d[i] = 0ull;
    } else {
      char* d = static_cast<char*>(p);
      for (size_t i = threadIdx.x; i < bytes; i += kBatchThreads) d[i] = 0;
    }
    __syncthreads();
  }
  __device__ void sync() {}

  template <class F> __device__ void map(i64 n, F f) {
    for (i64 i = threadIdx.x; i < n; i += kBatchThreads) f(i);
    __syncthreads();
  }

  // mode 0 sum, 1 max (NaN -> +inf), 2 min (NaN -> -inf): same conventions as HipExec::reduce
  template <int MODE, class F> __device__ double reduce(i64 n, F f) {
    double acc = MODE == 0 ? 0.0 : -kInf;
    for (i64 i = threadIdx.x; i < n; i += kBatchThreads) {
      double v = f(i);
      if (MODE == 0) acc += v;
      else if (MODE == 1) acc = fmax(acc, v != v ? kInf : v);
      else { v = v != v ? -kInf : v; acc = fmax(acc, -v); }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const double other = __shfl_xor(acc, o, 64);
      acc = MODE == 0 ? acc + other : fmax(acc, other);
    }
    double* buf = red + 4 * parity;
    parity ^= 1;
    if ((threadIdx.x & 63) == 0) buf[threadIdx.x >> 6] = acc;
    __syncthreads();
    double r = buf[0];
    for (int k = 1; k < kBatchThreads / 64; ++k) r = MODE == 0 ? r + buf[k] : fmax(r, buf[k]);
    return MODE == 2 ? -r : r;
  }
  template <class F> __device__ double sum(i64 n, F f) { return reduce<0>(n, f); }
  template <class F> __device__ double max(i64 n, F f) { return reduce<1>(n, f); }
  template <class F> __device__ double min(i64 n, F f) { return reduce<2>(n, f); }

  // block-wide argmax of v (first index wins ties, as IDAMAX); every lane gets the result
  __device__ void argmax(double v, int idx, double& outv, int& outi) {
    for (int o = 32; o > 0; o >>= 1) {
      const double ov = __shfl_xor(v, o, 64);
      const int oi = __shfl_xor(idx, o, 64);
      if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    double* bv = red + 4 * parity;
    int* bi = redi + 4 * parity;
    parity ^= 1;
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = v; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    outv = bv[0];
    outi = bi[0];
    for (int k = 1; k < kBatchThreads / 64; ++k)
      if (bv[k] > outv || (bv[k] == outv && bi[k] < outi)) { outv = bv[k]; outi = bi[k]; }
  }

  // facilities of the large dense path that a batch instance never uses (the host rejects
  // tapes with dense quad_form blocks before launching)
  __device__ void gemv_sym(i64, const double*, i64, const double*, double*) { __builtin_trap(); }
  __device__ void orthogonalize(int, const double*, i64, double*, double*) { __builtin_trap(); }
  __device__ void dense_block_add(double*, i64, i64, const double*, i64, i64, double, bool) { __builtin_trap(); }
  __device__ void ldlt_prepare(LdltWork&, i64, i64, bool) {}

  // out (+)= J v / J^T v / sym(H) v from COO triplets; `out` was zeroed by the caller
  __device__ void coo_mult(i64 nnz, const i32* r, const i32* c, const double* a, const double* v, double* out, bool trans) {
    for (i64 p = threadIdx.x; p < nnz; p += kBatchThreads) {
      if (trans) unsafeAtomicAdd(&out[c[p]], a[p] * v[r[p]]);
      else unsafeAtomicAdd(&out[r[p]], a[p] * v[c[p]]);
    }
    __syncthreads();
  }
  __device__ void coo_sym_mult(i64 nnz, const i32* r, const i32* c, const double* a, const double* v, double* out) {
    for (i64 p = threadIdx.x; p < nnz; p += kBatchThreads) {
      const double av = a[p];
      unsafeAtomicAdd(&out[r[p]], av * v[c[p]]);
      if (r[p] != c[p]) unsafeAtomicAdd(&out[c[p]], av * v[r[p]]);
    }
    __syncthreads();
  }

  // ---- Bunch-Kaufman LDL^T by one workgroup (DSYTF2 semantics, lower storage) -------------
  // Same pivot rule, interchanges, multipliers and inertia count as bk_pivot_kernel /
  // bk_update_kernel of exec_hip.h and as the DSYTF2 restatement of the test oracle.
  __device__ bool ldlt_factor(LdltWork&, double* A, i64 nn, i64 ld, i32* ipiv, bool, int* nneg_out, int* nzero_out) {
    const int n = static_cast<int>(nn), tid = threadIdx.x;
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
        __syncthreads();
        if (tid == 0) {
          const double t = Akk[kk];
          Akk[kk] = Akp[kp];
          Akp[kp] = t;
          if (kstep == 2) { const double t2 = Ak[k + 1]; Ak[k + 1] = Ak[kp]; Ak[kp] = t2; }
        }
        __syncthreads();
      }
      if (zero_piv) {
        // structurally empty column: unit tiny pivot, reported through nzero
        if (tid == 0) { Ak[k] = 1e-20; ipiv[k] = k + 1; }
        ++nzero;
        __syncthreads();
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
        __syncthreads();
        for (int i = k + 1 + tid; i < n; i += kBatchThreads) Ak[i] *= inv;
        __syncthreads();
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
        __syncthreads();
        for (int j = k + 2 + tid; j < n; j += kBatchThreads) {
          const double ajk = Ak[j], ajk1 = Ak1[j];
          Ak[j] = d21 * (d11 * ajk - ajk1);
          Ak1[j] = d21 * (d22 * ajk1 - ajk);
        }
        __syncthreads();
      }
      k += kstep;
    }
    *nneg_out = nneg;
    *nzero_out = nzero;
    return true;
  }

  // DSYTRS (lower) by one workgroup; b in exec-space memory
  __device__ void ldlt_solve(LdltWork&, const double* A, i64 nn, i64 ld, const i32* ipiv, bool, double* b) {
    const int n = static_cast<int>(nn), tid = threadIdx.x;
    int k = 0;
    while (k < n) {
      const double* Ak = A + static_cast<i64>(k) * ld;
      if (ipiv[k] > 0) {
        const int kp = ipiv[k] - 1;
        if (tid == 0 && kp != k) { const double t = b[k]; b[k] = b[kp]; b[kp] = t; }
        __syncthreads();
        const double bk = b[k];
        for (int i = k + 1 + tid; i < n; i += kBatchThreads) b[i] -= Ak[i] * bk;
        __syncthreads();
        if (tid == 0) b[k] = bk / Ak[k];
        k += 1;
      } else {
        const double* Ak1 = Ak + ld;
        const int kp = -ipiv[k] - 1;
        if (tid == 0 && kp != k + 1) { const double t = b[k + 1]; b[k + 1] = b[kp]; b[kp] = t; }
        __syncthreads();
        const double bk = b[k], bk1 = b[k + 1];
        for (int i = k + 2 + tid; i < n; i += kBatchThreads) b[i] -= Ak[i] * bk + Ak1[i] * bk1;
        __syncthreads();
        if (tid == 0) {
          const double akm1k = Ak[k + 1];
          const double akm1 = Ak[k] / akm1k, ak = Ak1[k + 1] / akm1k;
          const double denom = akm1 * ak - 1.0, bkm1 = bk / akm1k, bkk = bk1 / akm1k;
          b[k] = (ak * bkm1 - bkk) / denom;
          b[k + 1] = (akm1 * bkk - bkm1) / denom;
        }
        k += 2;
      }
      __syncthreads();
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
      __syncthreads();
      k -= one ? 1 : 2;
    }
  }
};

}  // namespace dnlp
