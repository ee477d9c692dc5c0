// TEST INFRASTRUCTURE — host execution space for the single-source solver core.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build or call this.
// It instantiates dnlp_amd/csrc/{model.h, ipm_core.h, kkt_dense.h} — the same algorithm text
// the MI355X library compiles with hipcc — as plain host loops, so that the restated
// interior-point algorithm (Waechter & Biegler 2006; IPOPT is absent from the reference tree)
// can be pinned against the reference tests' known optima without a GPU, and so the HIP
// kernels have an independent scalar implementation to be compared with on the GPU box.
// The dense symmetric-indefinite factorisation below restates LAPACK's DSYTF2/DSYTRS
// (Bunch-Kaufman partial pivoting, lower storage), the role MUMPS plays for IPOPT.
#pragma once
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../dnlp_amd/csrc/exec.h"
#include "../dnlp_amd/csrc/fused_obj.h"
#include "../dnlp_amd/csrc/sparse_ldl.h"

#include <dlfcn.h>
#include <string>

namespace dnlp {

// Optional LAPACK backend for the dense symmetric-indefinite factorisation of the CPU baseline
// (bench.py cpu_baseline): DSYTRF / DSYTRS (blocked Bunch-Kaufman, the dense counterpart of the
// MA27 / MUMPS factorisation IPOPT uses) resolved at run time from the OpenBLAS that ships inside
// the scipy wheel of this image (Fortran interface, 32-bit integers, symbol prefix scipy_).
// orc_use_lapack(path) switches every later pivoted ldlt_factor / ldlt_solve of this process to it;
// the restated DSYTF2 below stays the default so that the parity tests do not depend on a BLAS.
struct HostLapack {
  using sytrf_t = void (*)(const char*, const int*, double*, const int*, int*, double*, const int*, int*);
  using sytrs_t = void (*)(const char*, const int*, const int*, const double*, const int*, const int*, double*,
                           const int*, int*);
  using gemm_t = void (*)(const char*, const char*, const int*, const int*, const int*, const double*, const double*,
                          const int*, const double*, const int*, const double*, double*, const int*);
  using trsm_t = void (*)(const char*, const char*, const char*, const char*, const int*, const int*, const double*,
                          const double*, const int*, double*, const int*);
  sytrf_t sytrf = nullptr;
  sytrs_t sytrs = nullptr;
  gemm_t gemm = nullptr;                 // DGEMM / DTRSM of the same library: the blocked unpivoted LDL^T below
  trsm_t trsm = nullptr;
  bool blocked_unpivoted = false;        // orc_use_blocked_ldlt: unpivoted factorisations of order >= 512 go through them
  std::vector<double> wpanel, wpanel2;
  double last_phases[4] = {0.0, 0.0, 0.0, 0.0};   // blocked LDL^T, last factorisation: diagonal blocks, DTRSM, W copy, DGEMM (seconds)
  int (*get_threads)() = nullptr;
  void (*set_threads)(int) = nullptr;
  std::vector<double> work;
  std::vector<int> ipiv;
  static HostLapack& get() { static HostLapack L; return L; }
  bool load(const char* path) {
    void* h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!h) return false;
    for (const char* pre : {"scipy_", ""}) {
      const std::string a = std::string(pre) + "dsytrf_", b = std::string(pre) + "dsytrs_";
      sytrf = reinterpret_cast<sytrf_t>(dlsym(h, a.c_str()));
      sytrs = reinterpret_cast<sytrs_t>(dlsym(h, b.c_str()));
      if (sytrf && sytrs) {
        gemm = reinterpret_cast<gemm_t>(dlsym(h, (std::string(pre) + "dgemm_").c_str()));
        trsm = reinterpret_cast<trsm_t>(dlsym(h, (std::string(pre) + "dtrsm_").c_str()));
        get_threads = reinterpret_cast<int (*)()>(dlsym(h, (std::string(pre) + "openblas_get_num_threads").c_str()));
        set_threads = reinterpret_cast<void (*)(int)>(dlsym(h, (std::string(pre) + "openblas_set_num_threads").c_str()));
        return true;
      }
    }
    sytrf = nullptr; sytrs = nullptr;
    return false;
  }
};

struct HostExec : HostControlled {
  static constexpr bool is_device = false;
  struct FlatTableT {};
  struct LdltWork { std::vector<double> d; int expect_neg = -1; bool time_updates = false; bool padded = false; };
  void ldlt_stats(LdltWork&, double* out5) { for (int k = 0; k < 5; ++k) out5[k] = 0.0; }
  explicit HostExec(int device = 0) { (void)device; }

  template <class T> T* alloc(size_t n) {
    T* p = static_cast<T*>(std::calloc(n ? n : 1, sizeof(T)));
    if (!p) throw std::bad_alloc();
    return p;
  }
  void release(void* p) { std::free(p); }
  void h2d(void* dst, const void* src, size_t bytes) { if (bytes) std::memcpy(dst, src, bytes); }
  void d2h(void* dst, const void* src, size_t bytes) { if (bytes) std::memcpy(dst, src, bytes); }
  void d2d(void* dst, const void* src, size_t bytes) { if (bytes) std::memmove(dst, src, bytes); }
  void zero(void* p, size_t bytes) { if (bytes) std::memset(p, 0, bytes); }
  void sync() {}

  template <class F> void map(i64 n, F f) {
#pragma omp parallel for schedule(static) if (n > 16384)
    for (i64 i = 0; i < n; ++i) f(i);
  }
  template <class F> double sum(i64 n, F f) {
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static) if (n > 16384)
    for (i64 i = 0; i < n; ++i) s += f(i);
    return s;
  }
  template <class F> double max(i64 n, F f) {
    double s = -kInf;
    bool nan = false;
    for (i64 i = 0; i < n; ++i) { double v = f(i); if (v != v) nan = true; if (v > s) s = v; }
    return nan ? std::nan("") : s;
  }
  template <class F> double min(i64 n, F f) {
    double s = kInf;
    bool nan = false;
    for (i64 i = 0; i < n; ++i) { double v = f(i); if (v != v) nan = true; if (v < s) s = v; }
    return nan ? std::nan("") : s;
  }

  template <int NM, int NS, class F> RMulti reduce_multi(i64 n, F f) {
    RMulti r;
    for (int k = 0; k < 4; ++k) { r.mx[k] = -kInf; r.sm[k] = 0.0; }
    for (i64 i = 0; i < n; ++i) {
      const RMulti v = f(i);
      for (int k = 0; k < NM; ++k) r.mx[k] = std::fmax(r.mx[k], v.mx[k] != v.mx[k] ? kInf : v.mx[k]);
      for (int k = 0; k < NS; ++k) r.sm[k] += v.sm[k];
    }
    return r;
  }
  template <class F> D2 min2(i64 n, F f) {
    D2 s{kInf, kInf};
    bool nan = false;
    for (i64 i = 0; i < n; ++i) {
      const D2 v = f(i);
      if (v.first != v.first || v.second != v.second) nan = true;
      if (v.first < s.first) s.first = v.first;
      if (v.second < s.second) s.second = v.second;
    }
    return nan ? D2{std::nan(""), std::nan("")} : s;
  }

  // static-pattern sparse LDL^T (csrc/sparse_ldl.h): the single-source routine with one lane
  bool sparse_factor(const SparsePlan& pl, double* vals, double* w, int* nneg, int* nzero) {
    return sparse_ldl_factor(pl, vals, w, nneg, nzero, SeqPar());
  }
  void sparse_solve(const SparsePlan& pl, const double* vals, double* x) { sparse_ldl_solve(pl, vals, x, SeqPar()); }

  // fused element program (csrc/fused_obj.h): sequential host loop, local slots
  double fused_eval(const FusedSlotProg& P, const double* x, const double* consts, double* grad) {
    double f = 0.0, slots[256];
    const bool valid[1] = {true};
    for (i64 i = 0; i < P.nelem; ++i)
      f += fused_elements<1>(P, i, 0, valid, x, consts, [&](int k, int) -> double& { return slots[k]; },
                             [&](i64 idx, double v) { grad[idx] += v; });
    return f;
  }

  // (generated kernels exist only in the HIP space)
  bool fused_generated_eval(const std::vector<FusedSlotProg>&, const double*, const double*, double*, i64, double&) { return false; }

  struct LbfgsResult { int status = -199, iterations = 0, evaluations = 0, slots = 0; double f = 0.0, gnorm = 0.0, seconds = 0.0; bool persistent = false; };
  bool lbfgs_generated_solve(const std::vector<FusedSlotProg>&, const double*, double, i64, double*, int, double, int, LbfgsResult&) { return false; }

  // c = V^T w for k stored vectors; out = sum_q c_q V_q
  void vt_dot(int k, const double* V, i64 N, const double* w, double* c_host) {
    for (int q = 0; q < k; ++q) {
      double s = 0.0;
      for (i64 i = 0; i < N; ++i) s += V[static_cast<i64>(q) * N + i] * w[i];
      c_host[q] = s;
    }
  }
  void v_comb(int k, const double* V, i64 N, const double* c_host, double* out) {
    for (i64 i = 0; i < N; ++i) {
      double s = 0.0;
      for (int q = 0; q < k; ++q) s += c_host[q] * V[static_cast<i64>(q) * N + i];
      out[i] = s;
    }
  }
  // Gram-Schmidt step against k stored vectors: c = V^T w, w -= V c
  void orthogonalize(int k, const double* V, i64 N, double* w, double* c_host) {
    for (int q = 0; q < k; ++q) {
      double s = 0.0;
      for (i64 i = 0; i < N; ++i) s += V[static_cast<i64>(q) * N + i] * w[i];
      c_host[q] = s;
    }
    for (int q = 0; q < k; ++q)
      for (i64 i = 0; i < N; ++i) w[i] -= c_host[q] * V[static_cast<i64>(q) * N + i];
  }
  // out = P u for a symmetric column-major matrix
  void gemv_sym(i64 n, const double* P, i64 ld, const double* u, double* out) {
    // P symmetric: row i of P u is the dot product of column i with u (contiguous reads)
    std::vector<double> acc(static_cast<size_t>(n), 0.0);
#pragma omp parallel for schedule(static) if (n > 512)
    for (i64 i = 0; i < n; ++i) {
      const double* col = P + i * ld;
      double s = 0.0;
      for (i64 j = 0; j < n; ++j) s += col[j] * u[j];
      acc[static_cast<size_t>(i)] = s;
    }
    std::memcpy(out, acc.data(), sizeof(double) * static_cast<size_t>(n));
  }
  // out += A v (or A^T v) for COO entries
  void coo_mult(i64 nnz, const i32* r, const i32* c, const double* a, const double* v, double* out, bool trans) {
    for (i64 p = 0; p < nnz; ++p) {
      if (trans) out[c[p]] += a[p] * v[r[p]]; else out[r[p]] += a[p] * v[c[p]];
    }
  }
  // out = (pattern product) through the tape's index by output: every output sums its segment in storage order and is
  // ASSIGNED (an output without entries becomes 0): no zeroing pass before the product
  void coo_gather(const CooIdx& ix, const double* a, const double* v, double* out) {
    for (i64 g = 0; g < ix.nout; ++g) {
      double s = 0.0;
      for (i64 p = ix.ptr[g]; p < ix.ptr[g + 1]; ++p) s += a[ix.ent[p]] * v[ix.src[p]];
      out[g] = s;
    }
  }
  // out += S v for a symmetric matrix given by its lower-triangle COO entries
  void coo_sym_mult(i64 nnz, const i32* r, const i32* c, const double* a, const double* v, double* out) {
    for (i64 p = 0; p < nnz; ++p) {
      out[r[p]] += a[p] * v[c[p]];
      if (r[p] != c[p]) out[c[p]] += a[p] * v[r[p]];
    }
  }
  // K[x0+r, x0+c] (+)= w * P[r, c] on the lower triangle
  void dense_block_add(double* K, i64 ldk, i64 x0, const double* P, i64 ldp, i64 nb, double w, bool set) {
    for (i64 c = 0; c < nb; ++c)
      for (i64 r = c; r < nb; ++r) {
        double& dst = K[(x0 + r) + (x0 + c) * ldk];
        dst = (set ? 0.0 : dst) + w * P[r + c * ldp];
      }
  }

  void ldlt_prepare(LdltWork&, i64, i64, bool) {}

  // Bunch-Kaufman (DSYTF2, lower) when pivoted; plain right-looking LDL^T otherwise.
  bool ldlt_factor(LdltWork&, double* A, i64 n, i64 ld, i32* ipiv, bool pivoted, int* nneg, int* nzero) {
    *nneg = 0; *nzero = 0;
    HostLapack& LP = HostLapack::get();
    if (pivoted && LP.sytrf) {
      const int nn = static_cast<int>(n), lda = static_cast<int>(ld);
      int info = 0, lwork = -1;
      double wq = 0.0;
      for (i64 j = 0; j < n; ++j) if (!(A[j + j * ld] == A[j + j * ld])) return false;
      LP.sytrf("L", &nn, A, &lda, ipiv, &wq, &lwork, &info);
      lwork = static_cast<int>(wq) > nn ? static_cast<int>(wq) : nn;
      if (static_cast<int>(LP.work.size()) < lwork) LP.work.resize(static_cast<size_t>(lwork));
      LP.sytrf("L", &nn, A, &lda, ipiv, LP.work.data(), &lwork, &info);
      if (info < 0) return false;
      // inertia from D (Sylvester): 1x1 blocks by sign, a 2x2 block has one eigenvalue of each sign
      for (i64 k = 0; k < n;) {
        if (ipiv[k] > 0) {
          const double d = A[k + k * ld];
          if (!(d == d)) return false;
          if (d == 0.0) { (*nzero)++; A[k + k * ld] = 1e-20; }
          else if (d < 0.0) (*nneg)++;
          k += 1;
        } else { (*nneg)++; k += 2; }
      }
      return true;
    }
    auto a = [&](i64 i, i64 j) -> double& { return A[i + j * ld]; };
    double amax = 0.0;
    for (i64 j = 0; j < n; ++j) amax = std::fmax(amax, std::fabs(a(j, j)));
    const double tiny = 1e-14 * std::fmax(amax, 1e-300) * 0 + 1e-300;
    if (!pivoted && LP.blocked_unpivoted && LP.gemm && LP.trsm && n >= 512) {
      // CPU baseline (bench.py): the device's algorithm on the host's BLAS — right-looking blocked LDL^T, the panel's
      // diagonal block unblocked, the rows below by DTRSM (A21 L11^-T = L21 D = W), the trailing matrix
      // C -= W L21^T by one DGEMM per block column of the lower triangle, all cores.
      const i64 NB = n >= 4000 ? 512 : 256;
      const double one = 1.0, mone = -1.0;
      // DNLP_HOST_LDLT_TIMING=1: seconds per phase of this factorisation on stderr (where the host's time goes)
      static const bool timing = std::getenv("DNLP_HOST_LDLT_TIMING") != nullptr;     // (bench.py sets it: the phase split is part of the baseline's line)
      auto clk = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
      double t_diag = 0.0, t_trsm = 0.0, t_copy = 0.0, t_gemm = 0.0, t_mark = timing ? clk() : 0.0;
      auto lap = [&](double& acc) { if (timing) { const double t = clk(); acc += t - t_mark; t_mark = t; } };
      for (i64 K0 = 0; K0 < n; K0 += NB) {
        if (timing) t_mark = clk();
        const i64 KB = std::min<i64>(NB, n - K0);
        // the panel's diagonal block, itself blocked by 64 columns (unblocked it was 45 Mflop of scalar code per
        // 512-column panel, more time than the DGEMMs of the whole factorisation on a 256-core host): 64 x 64 diagonal
        // pieces unblocked, the rows of the block below them by DTRSM, the rest of the block by DGEMM
        const i64 IB = 64;
        for (i64 k0 = K0; k0 < K0 + KB; k0 += IB) {
          const i64 kb = std::min<i64>(IB, K0 + KB - k0), e0 = k0 + kb, er = K0 + KB - e0;
          for (i64 k = k0; k < e0; ++k) {
            double d = a(k, k);
            if (!(d == d)) return false;
            if (std::fabs(d) <= tiny) { (*nzero)++; d = (d < 0 ? -1.0 : 1.0) * 1e-20; a(k, k) = d; }
            if (d < 0) (*nneg)++;
            ipiv[k] = static_cast<i32>(k + 1);
            const double inv = 1.0 / d;
            for (i64 j = k + 1; j < e0; ++j) {
              const double lj = A[j + k * ld] * inv;
              if (lj == 0.0) continue;
              for (i64 i = j; i < e0; ++i) A[i + j * ld] -= A[i + k * ld] * lj;
            }
            for (i64 i = k + 1; i < e0; ++i) a(i, k) *= inv;
          }
          if (er <= 0) break;
          const int mi2 = static_cast<int>(er), ki2 = static_cast<int>(kb), ldi2 = static_cast<int>(ld);
          LP.trsm("R", "L", "T", "U", &mi2, &ki2, &one, A + k0 + k0 * ld, &ldi2, A + e0 + k0 * ld, &ldi2);   // W = L D
          if (LP.wpanel2.size() < static_cast<size_t>(er * kb)) LP.wpanel2.resize(static_cast<size_t>(er * kb));
          double* W2 = LP.wpanel2.data();
          for (i64 c = 0; c < kb; ++c) {
            const double inv = 1.0 / A[(k0 + c) + (k0 + c) * ld];
            double* col = A + e0 + (k0 + c) * ld;
            double* wc = W2 + c * er;
            for (i64 i = 0; i < er; ++i) { wc[i] = col[i]; col[i] *= inv; }
          }
          LP.gemm("N", "T", &mi2, &mi2, &ki2, &mone, W2, &mi2, A + e0 + k0 * ld, &ldi2, &one, A + e0 + e0 * ld, &ldi2);
        }
        lap(t_diag);
        const i64 r1 = K0 + KB, rows = n - r1;
        if (rows <= 0) break;
        const int mi = static_cast<int>(rows), ki = static_cast<int>(KB), ldi = static_cast<int>(ld);
        LP.trsm("R", "L", "T", "U", &mi, &ki, &one, A + K0 + K0 * ld, &ldi, A + r1 + K0 * ld, &ldi);   // A21 <- W = L21 D
        lap(t_trsm);
        if (LP.wpanel.size() < static_cast<size_t>(rows * KB)) LP.wpanel.resize(static_cast<size_t>(rows * KB));
        double* W = LP.wpanel.data();
#pragma omp parallel for schedule(static)
        for (i64 c = 0; c < KB; ++c) {
          const double inv = 1.0 / A[(K0 + c) + (K0 + c) * ld];
          double* col = A + r1 + (K0 + c) * ld;
          double* wc = W + c * rows;
          for (i64 i = 0; i < rows; ++i) { wc[i] = col[i]; col[i] *= inv; }
        }
        // few, large DGEMMs: block columns of a quarter of the trailing order (a call per 256 columns spends its time in
        // the thread pool's hand-offs on a 256-core host: 82 GFLOP/s at n = 1e4 against 3.2 TFLOP/s of plain DGEMM); the
        // part of a block above the diagonal is computed and ignored (<= 1/8 more flops)
        lap(t_copy);
        const i64 CB = std::max<i64>(NB, (rows / 4 + NB - 1) / NB * NB);
        for (i64 J = 0; J < rows; J += CB) {
          const int jb = static_cast<int>(std::min<i64>(CB, rows - J)), mr = static_cast<int>(rows - J);
          LP.gemm("N", "T", &mr, &jb, &ki, &mone, W + J, &mi, A + (r1 + J) + K0 * ld, &ldi, &one,
                  A + (r1 + J) + (r1 + J) * ld, &ldi);
        }
        lap(t_gemm);
      }
      LP.last_phases[0] = t_diag; LP.last_phases[1] = t_trsm; LP.last_phases[2] = t_copy; LP.last_phases[3] = t_gemm;
      if (timing && std::getenv("DNLP_HOST_LDLT_TIMING")[0] == '2')
        std::fprintf(stderr, "[host ldlt] n %lld: diagonal blocks %.3f s, dtrsm %.3f s, W copy %.3f s, dgemm %.3f s\n",
                     static_cast<long long>(n), t_diag, t_trsm, t_copy, t_gemm);
      return true;
    }
    if (!pivoted) {
      for (i64 k = 0; k < n; ++k) {
        double d = a(k, k);
        if (!(d == d)) return false;
        if (std::fabs(d) <= tiny) { (*nzero)++; d = (d < 0 ? -1.0 : 1.0) * 1e-20; a(k, k) = d; }
        if (d < 0) (*nneg)++;
        ipiv[k] = static_cast<i32>(k + 1);
        const double inv = 1.0 / d;
#pragma omp parallel for schedule(dynamic, 16) if (n - k > 256)
        for (i64 j = k + 1; j < n; ++j) {
          const double wj = A[j + k * ld];
          if (wj == 0.0) continue;
          const double lj = wj * inv;
          double* cj = A + j * ld;
          const double* ck = A + k * ld;
          for (i64 i = j; i < n; ++i) cj[i] -= ck[i] * lj;
        }
        for (i64 i = k + 1; i < n; ++i) a(i, k) *= inv;
      }
      return true;
    }
    const double alpha = (1.0 + std::sqrt(17.0)) / 8.0;
    i64 k = 0;
    while (k < n) {
      int kstep = 1;
      i64 kp = k;
      const double absakk = std::fabs(a(k, k));
      i64 imax = k;
      double colmax = 0.0;
      for (i64 i = k + 1; i < n; ++i) { double v = std::fabs(a(i, k)); if (v > colmax) { colmax = v; imax = i; } }
      if (!(absakk == absakk) || !(colmax == colmax)) return false;
      if (std::fmax(absakk, colmax) == 0.0) {
        (*nzero)++;
        ipiv[k] = static_cast<i32>(k + 1);
        a(k, k) = 1e-20;   // keep the solve finite; the caller regularises and refactors
        k += 1;
        continue;
      }
      if (absakk >= alpha * colmax) {
        kp = k;
      } else {
        double rowmax = 0.0;
        for (i64 j = k; j < imax; ++j) rowmax = std::fmax(rowmax, std::fabs(a(imax, j)));
        for (i64 i = imax + 1; i < n; ++i) rowmax = std::fmax(rowmax, std::fabs(a(i, imax)));
        if (absakk >= alpha * colmax * (colmax / rowmax)) kp = k;
        else if (std::fabs(a(imax, imax)) >= alpha * rowmax) kp = imax;
        else { kp = imax; kstep = 2; }
      }
      const i64 kk = k + kstep - 1;
      if (kp != kk) {
        for (i64 i = kp + 1; i < n; ++i) std::swap(a(i, kk), a(i, kp));
        for (i64 j = kk + 1; j < kp; ++j) std::swap(a(j, kk), a(kp, j));
        std::swap(a(kk, kk), a(kp, kp));
        if (kstep == 2) std::swap(a(k + 1, k), a(kp, k));
      }
      if (kstep == 1) {
        const double d = a(k, k);
        if (d < 0) (*nneg)++;
        if (std::fabs(d) < 1e-300) (*nzero)++;
        const double d11 = 1.0 / d;
        for (i64 j = k + 1; j < n; ++j) {
          const double wj = a(j, k) * d11;
          if (wj != 0.0) for (i64 i = j; i < n; ++i) a(i, j) -= a(i, k) * wj;
        }
        for (i64 i = k + 1; i < n; ++i) a(i, k) *= d11;
        ipiv[k] = static_cast<i32>(kp + 1);
      } else {
        (*nneg)++;   // a Bunch-Kaufman 2x2 pivot has one positive and one negative eigenvalue
        if (k < n - 2) {
          double d21 = a(k + 1, k);
          const double d11 = a(k + 1, k + 1) / d21, d22 = a(k, k) / d21;
          const double tt = 1.0 / (d11 * d22 - 1.0);
          d21 = tt / d21;
          for (i64 j = k + 2; j < n; ++j) {
            const double wk = d21 * (d11 * a(j, k) - a(j, k + 1));
            const double wkp1 = d21 * (d22 * a(j, k + 1) - a(j, k));
            for (i64 i = j; i < n; ++i) a(i, j) -= a(i, k) * wk + a(i, k + 1) * wkp1;
            a(j, k) = wk;
            a(j, k + 1) = wkp1;
          }
        }
        ipiv[k] = ipiv[k + 1] = static_cast<i32>(-(kp + 1));
      }
      k += kstep;
    }
    return true;
  }

  // DSYTRS (lower) / plain L D L^T solve, in place
  void ldlt_solve(LdltWork&, const double* A, i64 n, i64 ld, const i32* ipiv, bool pivoted, double* b) {
    auto a = [&](i64 i, i64 j) -> double { return A[i + j * ld]; };
    HostLapack& LP = HostLapack::get();
    if (pivoted && LP.sytrs) {
      const int nn = static_cast<int>(n), lda = static_cast<int>(ld), one = 1;
      int info = 0;
      LP.sytrs("L", &nn, &one, A, &lda, ipiv, b, &nn, &info);
      return;
    }
    if (!pivoted) {
      for (i64 k = 0; k < n; ++k) { const double bk = b[k]; if (bk != 0.0) for (i64 i = k + 1; i < n; ++i) b[i] -= a(i, k) * bk; }
      for (i64 k = 0; k < n; ++k) b[k] /= a(k, k);
      for (i64 k = n - 1; k >= 0; --k) { double s = b[k]; for (i64 i = k + 1; i < n; ++i) s -= a(i, k) * b[i]; b[k] = s; }
      return;
    }
    i64 k = 0;
    while (k < n) {
      if (ipiv[k] > 0) {
        const i64 kp = ipiv[k] - 1;
        if (kp != k) std::swap(b[k], b[kp]);
        const double bk = b[k];
        for (i64 i = k + 1; i < n; ++i) b[i] -= a(i, k) * bk;
        b[k] = bk / a(k, k);
        k += 1;
      } else {
        const i64 kp = -ipiv[k] - 1;
        if (kp != k + 1) std::swap(b[k + 1], b[kp]);
        const double bk = b[k], bk1 = b[k + 1];
        for (i64 i = k + 2; i < n; ++i) b[i] -= a(i, k) * bk + a(i, k + 1) * bk1;
        const double akm1k = a(k + 1, k), akm1 = a(k, k) / akm1k, ak = a(k + 1, k + 1) / akm1k;
        const double denom = akm1 * ak - 1.0, bkm1 = bk / akm1k, bkk = bk1 / akm1k;
        b[k] = (ak * bkm1 - bkk) / denom;
        b[k + 1] = (akm1 * bkk - bkm1) / denom;
        k += 2;
      }
    }
    k = n - 1;
    while (k >= 0) {
      if (ipiv[k] > 0) {
        double s = b[k];
        for (i64 i = k + 1; i < n; ++i) s -= a(i, k) * b[i];
        b[k] = s;
        const i64 kp = ipiv[k] - 1;
        if (kp != k) std::swap(b[k], b[kp]);
        k -= 1;
      } else {
        double s0 = b[k], s1 = b[k - 1];
        for (i64 i = k + 1; i < n; ++i) { s0 -= a(i, k) * b[i]; s1 -= a(i, k - 1) * b[i]; }
        b[k] = s0;
        b[k - 1] = s1;
        const i64 kp = -ipiv[k] - 1;
        if (kp != k) std::swap(b[k], b[kp]);
        k -= 2;
      }
    }
  }
};

}  // namespace dnlp
