// Interior-point filter line-search loop for  min f(x)  s.t.  cl <= g(x) <= cu,  lb <= x <= ub.
//
// Replaces the reference's dispatch into third-party IPOPT (`cyipopt.Problem(...).solve(x0)`,
// cvxpy/reductions/solvers/nlp_solvers/ipopt_nlpif.py:140-170).  IPOPT's source is not part of
// the reference tree; the algorithm restated here is the published one — A. Waechter and
// L. T. Biegler, "On the implementation of an interior-point filter line-search algorithm for
// large-scale nonlinear programming", Math. Prog. 106(1), 2006 (cited below as WB and by
// equation / section number) — with IPOPT's documented default constants and the option
// defaults the reference sets (ipopt_nlpif.py:153-160: tol 1e-7, bound_relax_factor 0,
// exact Hessian, least-squares multiplier initialisation).
//
// Single source over an execution space E (exec.h): vectors live in E's memory, every
// per-element formula is a DNLP_HD lambda, scalars come back through E::sum/max/min.  The
// KKT object K owns assembly, factorisation (with inertia) and solves.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "model.h"
#include "ipm_options.h"

namespace dnlp {

struct IpmStats {
  int iterations = 0;
  int factorizations = 0;
  int skipped_factorizations = 0;
  double wall = 0.0, t_eval = 0.0, t_factor = 0.0, t_solve = 0.0, t_assemble = 0.0;
  double final_mu = 0.0, inf_pr = 0.0, inf_du = 0.0, cmpl = 0.0, nlp_error = 0.0;
  double last_delta_w = 0.0;
};

// Per-iteration callback, the role of Oracles.intermediate (nlp_solver.py:423-427; cyipopt calls it
// once per iteration with IPOPT's eleven intermediate_callback values).  A zero return stops the
// solve with User_Requested_Stop, as IPOPT does for `false`.
typedef int (*IntermediateCb)(int alg_mod, int iter_count, double obj_value, double inf_pr, double inf_du, double mu,
                              double d_norm, double regularization_size, double alpha_du, double alpha_pr,
                              int ls_trials, void* user_data);

// (see exec.h: DNLP_THIS_IN_LDS) — at the top of every member function of the in-kernel instantiation
#define DNLP_IPM_LDS() do { DNLP_THIS_IN_LDS(E); DNLP_PTR_IN_LDS(E, ex_); DNLP_PTR_IN_LDS(E, md_); DNLP_PTR_IN_LDS(E, kkt_); } while (0)

// (in the in-kernel space step() is inlined into its three call sites: as a called function it saves and restores ~107
//  callee-saved VGPRs per call — 27 KB per wavefront each way — that its caller never uses)
#if DNLP_DEVICE_PASS
#define DNLP_STEP_INLINE __attribute__((always_inline))
#else
#define DNLP_STEP_INLINE
#endif

template <class E, class K>
class Ipm {
 public:
  IntermediateCb intermediate_cb = nullptr;   // host-driven spaces only
  void* intermediate_user = nullptr;
  DNLP_HD Ipm(E* ex, Model<E>* model, K* kkt) : ex_(ex), md_(model), kkt_(kkt) {}
#if !DNLP_DEVICE_PASS
  ~Ipm() { delete lm_; }
#endif

  IpmOptions opt;
  IpmStats stats;
  typename E::Log iterlog;              // iteration table (host / HIP spaces); empty in the batch kernel

  // ---- problem data / iterate (exec space) ------------------------------------
  i64 N = 0, m = 0;
  using V = VecP<E>;                  // (exec.h: a double* that knows when it is in LDS)
  V x, s, y, zL, zU, vL, vU;
  V xL, xU, sL, sU;                   // scaled slack bounds
  V eqmask;                           // 1.0 for equality rows
  V grad, g, jv, sg;
  V dx, ds, dy, dzL, dzU, dvL, dvU;
  V xt, st, gt;
  V rhs, sol, res, cor, Sx, Dd, Ss;
  V rx, rs, rp, tN, tM, csoc;
  double* rowmax_tmp_ = nullptr;      // 64 partial row maxima per constraint row (gradient-based scaling of long rows)
  V aff[7], cen[7];                   // mu-oracle directions
  V zeroM;
  double sf = 1.0;
  double f = 0.0;                     // scaled objective at x
  double mu = 0.1, tau = 0.99;
  double delta_w_last = 0.0;
  // warm-start multipliers in the user's (unscaled) units, exec space; null = not provided
  const double *ws_mult_g = nullptr, *ws_mult_xL = nullptr, *ws_mult_xU = nullptr;
  bool initialized = false;
  int status = Internal_Error;
  int iter = 0;
  // filter (WB section 2.3): fixed capacity, the oldest entry is evicted when full
  static constexpr int kFilterCap = E::kFilterCap;
  double filt_th[kFilterCap], filt_ph[kFilterCap];
  int nfilt = 0;
  double theta_max = 1e4, theta_min = 1e-4;
  int acceptable_count = 0;
  double last_obj = 0.0;
  bool fixed_mode = false;            // adaptive strategy: currently in monotone (fixed) mode
  double kkt_hist[4];                 // kkt-error globalisation: last 4 free-mode values
  int n_hist = 0;

  // ------------------------------------------------------------------------------
  template <class T> DNLP_HD T* A(i64 n) { return ex_->template alloc<T>(static_cast<size_t>(n > 0 ? n : 1)); }

  DNLP_HD void allocate() {
    DNLP_IPM_LDS();
    N = md_->N(); m = md_->m();
    x = A<double>(N); s = A<double>(m); y = A<double>(m); zL = A<double>(N); zU = A<double>(N);
    vL = A<double>(m); vU = A<double>(m); xL = A<double>(N); xU = A<double>(N); sL = A<double>(m);
    sU = A<double>(m); eqmask = A<double>(m); grad = A<double>(N); g = A<double>(m);
    jv = A<double>(md_->t.nnzJ); sg = A<double>(m); dx = A<double>(N); ds = A<double>(m); dy = A<double>(m);
    dzL = A<double>(N); dzU = A<double>(N); dvL = A<double>(m); dvU = A<double>(m); xt = A<double>(N);
    st = A<double>(m); gt = A<double>(m); rhs = A<double>(N + m); sol = A<double>(N + m);
    res = A<double>(N + m); cor = A<double>(N + m); Sx = A<double>(N); Dd = A<double>(m); Ss = A<double>(m);
    rx = A<double>(N); rs = A<double>(m); rp = A<double>(m); tN = A<double>(N); tM = A<double>(m);
    csoc = A<double>(m);
    zeroM = A<double>(m);
    fixmask = A<double>(N);
    ex_->zero(zeroM, sizeof(double) * static_cast<size_t>(m));
    const i64 sz[7] = {N, m, m, N, N, m, m};
    for (int k = 0; k < 7; ++k) { aff[k] = A<double>(sz[k]); cen[k] = A<double>(sz[k]); }
  }

  template <class... Args> DNLP_HD void logf(const char* fmt, Args... args) {
#if !DNLP_DEVICE_PASS
    if constexpr (E::has_log) {
      char buf[512];
      std::snprintf(buf, sizeof buf, fmt, args...);
      if (opt.print_level >= 5) { std::fputs(buf, stdout); std::fputc('\n', stdout); std::fflush(stdout); }
      iterlog.lines.emplace_back(buf);
    }
#endif
  }
  DNLP_HD void filter_clear() { nfilt = 0; }
  DNLP_HD void filter_add(double th, double ph) {
    DNLP_IPM_LDS();
    if (nfilt == kFilterCap) {
      for (int k = 1; k < nfilt; ++k) { filt_th[k - 1] = filt_th[k]; filt_ph[k - 1] = filt_ph[k]; }
      --nfilt;
    }
    filt_th[nfilt] = th; filt_ph[nfilt] = ph; ++nfilt;
  }


  // ---- limited-memory quasi-Newton Hessian (hessian_approximation = limited-memory) ---------------------------
  // The role of IPOPT's LimMemQuasiNewtonUpdater + LowRankAugSystemSolver (the reference passes the option through,
  // ipopt_nlpif.py:153-168; cvxpy/tests/NLP_tests/test_entropy_related.py:40 uses it).  No second derivatives are
  // evaluated.  The Hessian of the Lagrangian is replaced by the BFGS matrix of the last k <= max_history pairs
  //   s = x+ - x,   y = grad_x L(x+, lambda+) - grad_x L(x, lambda+)
  // in the compact form of Byrd, Nocedal and Schnabel:
  //   B = sigma I - Psi M^{-1} Psi^T,   Psi = [sigma S, Y],   M = [[sigma S^T S, L], [L^T, -D]],
  // sigma = s^T y / s^T s of the newest pair (IPOPT's "scalar1"), L the strictly lower part of S^T Y, D its diagonal.
  // A pair with s^T y <= sqrt(eps) |s| |y| is skipped; max_skipping skips in a row drop the history.  The KKT matrix
  // is K = K0 - [Psi; 0] M^{-1} [Psi; 0]^T with K0 the matrix of the DIAGONAL Hessian sigma I (+ Sigma + delta_w):
  // K0 is factorised by the usual path (static-pattern sparse / dense), and a solve is Sherman-Morrison-Woodbury,
  //   K^{-1} r = K0^{-1} r + Z (M - Psi^T Z_x)^{-1} Psi^T (K0^{-1} r)_x,   Z = K0^{-1} [Psi; 0]   (2k solves per factorisation).
  // B is positive definite by construction, so K has the inertia (N, m, 0) exactly when K0 has it (both need only J of
  // full row rank): the inertia test reads the factorisation of K0.  Host-driven spaces only: the batch kernel keeps
  // the exact Hessian and says so (stats[23]).
  static constexpr int kLmMaxH = 12;
  struct LmState {
    int hist = 6, alloc_hist = 0, k = 0, skipped = 0;
    int updates = 0, skips_total = 0, resets = 0;
    int slot[kLmMaxH];                          // chronological position (0 = oldest) -> physical row
    double sigma = 1.0;
    double SS[kLmMaxH][kLmMaxH], SY[kLmMaxH][kLmMaxH];    // s_a^T s_b and s_a^T y_b by physical row
    double *S = nullptr, *Y = nullptr;          // hist x N each (exec space)
    double *xold = nullptr, *gold = nullptr, *jold = nullptr, *sv = nullptr, *yv = nullptr;
    double* Z = nullptr;                        // 2 hist x (N + m)
    double* tmp = nullptr;                      // N + m
    double Mm[4 * kLmMaxH * kLmMaxH], Mlu[4 * kLmMaxH * kLmMaxH], Clu[4 * kLmMaxH * kLmMaxH];
    int Mpiv[2 * kLmMaxH], Cpiv[2 * kLmMaxH];
    bool c_ready = false, c_ok = false, have_old = false;
  };
  LmState* lm_ = nullptr;
  DNLP_HD bool lm_on() const {
    if constexpr (E::has_host_control) return lm_ != nullptr && opt.hessian_approximation == 1;
    else return false;
  }
  // LU with partial pivoting of a small row-major matrix held by the control thread
  DNLP_HD static bool lm_lu(double* Am, int n, int* piv) {
    for (int c = 0; c < n; ++c) {
      int p = c;
      for (int r = c + 1; r < n; ++r) if (fabs(Am[r * n + c]) > fabs(Am[p * n + c])) p = r;
      piv[c] = p;
      if (!(fabs(Am[p * n + c]) > 0.0)) return false;
      if (p != c) for (int q = 0; q < n; ++q) { const double t = Am[c * n + q]; Am[c * n + q] = Am[p * n + q]; Am[p * n + q] = t; }
      for (int r = c + 1; r < n; ++r) {
        const double l = Am[r * n + c] / Am[c * n + c];
        Am[r * n + c] = l;
        for (int q = c + 1; q < n; ++q) Am[r * n + q] -= l * Am[c * n + q];
      }
    }
    return true;
  }
  DNLP_HD static void lm_lu_solve(const double* Am, int n, const int* piv, double* b) {
    for (int c = 0; c < n; ++c) { const double t = b[c]; b[c] = b[piv[c]]; b[piv[c]] = t; }     // (whole rows were swapped: P first)
    for (int c = 0; c < n; ++c) for (int r = c + 1; r < n; ++r) b[r] -= Am[r * n + c] * b[c];
    for (int c = n - 1; c >= 0; --c) { for (int q = c + 1; q < n; ++q) b[c] -= Am[c * n + q] * b[q]; b[c] /= Am[c * n + c]; }
  }
  DNLP_HD void lm_begin() {
    DNLP_IPM_LDS();
    if constexpr (E::has_host_control) {
      if (opt.hessian_approximation != 1) { kkt_->skip_hessian = false; return; }
      if (!lm_) lm_ = new LmState();
      LmState& L = *lm_;
      const int h = opt.limited_memory_max_history < 1 ? 1 : (opt.limited_memory_max_history > kLmMaxH ? kLmMaxH : opt.limited_memory_max_history);
      if (!L.S || h > L.alloc_hist) {
        L.S = A<double>(static_cast<i64>(h) * N); L.Y = A<double>(static_cast<i64>(h) * N);
        L.Z = A<double>(static_cast<i64>(2 * h) * (N + m));
        if (!L.xold) {
          L.xold = A<double>(N); L.gold = A<double>(N); L.jold = A<double>(md_->t.nnzJ); L.sv = A<double>(N); L.yv = A<double>(N);
          L.tmp = A<double>(N + m);
        }
        L.alloc_hist = h;
      }
      L.hist = h;
      lm_reset();
      if (ladder_rung_ == 0) L.updates = L.skips_total = L.resets = 0;
      kkt_->skip_hessian = true;
    }
  }
  DNLP_HD void lm_reset() {
    DNLP_IPM_LDS();
    if (!lm_) return;
    lm_->k = 0; lm_->skipped = 0; lm_->sigma = 1.0; lm_->c_ready = false; lm_->have_old = false;
  }
  // column c of Psi: row pointer and scale
  DNLP_HD const double* lm_col(int c, double& scale) const {
    const LmState& L = *lm_;
    if (c < L.k) { scale = L.sigma; return L.S + static_cast<i64>(L.slot[c]) * N; }
    scale = 1.0;
    return L.Y + static_cast<i64>(L.slot[c - L.k]) * N;
  }
  DNLP_HD void lm_psi_dots(const double* v, double* t) {
    DNLP_IPM_LDS();
    const int n2 = 2 * lm_->k;
    for (int c = 0; c < n2; ++c) {
      double sc;
      const double* row = lm_col(c, sc);
      t[c] = sc * ex_->sum(N, [=] DNLP_HD(i64 j) { return row[j] * v[j]; });
    }
  }
  // the point the next pair is measured from: x, grad f and the Jacobian values of the current iterate
  DNLP_HD void lm_save_point() {
    DNLP_IPM_LDS();
    LmState& L = *lm_;
    ex_->d2d(L.xold, x, sizeof(double) * static_cast<size_t>(N));
    ex_->d2d(L.gold, grad, sizeof(double) * static_cast<size_t>(N));
    ex_->d2d(L.jold, jv, sizeof(double) * static_cast<size_t>(md_->t.nnzJ));
    L.have_old = true;
  }
  // after an accepted step: the new pair (both gradients of the Lagrangian with the NEW multipliers)
  DNLP_HD void lm_update() {
    DNLP_IPM_LDS();
    LmState& L = *lm_;
    if (!L.have_old) return;
    L.have_old = false;
    md_->jac_tmult(jv, y, tN);
    md_->jac_tmult(L.jold, y, L.tmp);
    {
      double *sv = L.sv, *yv = L.yv;
      const double *xx = x, *xo = L.xold, *gn = grad, *go = L.gold, *jn = tN, *jo = L.tmp, *fm = fixmask;
      ex_->map(N, [=] DNLP_HD(i64 j) {
        const bool fx = fm[j] != 0.0;
        sv[j] = fx ? 0.0 : xx[j] - xo[j];
        yv[j] = fx ? 0.0 : (gn[j] - go[j]) + (jn[j] - jo[j]);
      });
    }
    const double *sv = L.sv, *yv = L.yv;
    const double sts = ex_->sum(N, [=] DNLP_HD(i64 j) { return sv[j] * sv[j]; });
    const double sty = ex_->sum(N, [=] DNLP_HD(i64 j) { return sv[j] * yv[j]; });
    const double yty = ex_->sum(N, [=] DNLP_HD(i64 j) { return yv[j] * yv[j]; });
    const double eps = 2.220446049250313e-16;
    if (!(sty > std::sqrt(eps) * std::sqrt(sts) * std::sqrt(yty)) || !std::isfinite(sty) || !std::isfinite(yty)) {
      ++L.skips_total;
      if (++L.skipped >= opt.limited_memory_max_skipping) {
        logf("   limited-memory: %d updates skipped in a row (s^T y = %.3e): history dropped", L.skipped, sty);
        lm_reset();
        ++L.resets;
      }
      return;
    }
    L.skipped = 0;
    if (L.k == L.hist) { for (int a = 1; a < L.k; ++a) L.slot[a - 1] = L.slot[a]; --L.k; }
    int phys = 0;
    for (;; ++phys) { bool used = false; for (int a = 0; a < L.k; ++a) used = used || L.slot[a] == phys; if (!used) break; }
    ex_->d2d(L.S + static_cast<i64>(phys) * N, sv, sizeof(double) * static_cast<size_t>(N));
    ex_->d2d(L.Y + static_cast<i64>(phys) * N, yv, sizeof(double) * static_cast<size_t>(N));
    for (int a = 0; a < L.k; ++a) {
      const int pa = L.slot[a];
      const double *sa = L.S + static_cast<i64>(pa) * N, *ya = L.Y + static_cast<i64>(pa) * N;
      L.SS[pa][phys] = L.SS[phys][pa] = ex_->sum(N, [=] DNLP_HD(i64 j) { return sa[j] * sv[j]; });
      L.SY[pa][phys] = ex_->sum(N, [=] DNLP_HD(i64 j) { return sa[j] * yv[j]; });
      L.SY[phys][pa] = ex_->sum(N, [=] DNLP_HD(i64 j) { return sv[j] * ya[j]; });
    }
    L.SS[phys][phys] = sts;
    L.SY[phys][phys] = sty;
    L.slot[L.k++] = phys;
    L.sigma = fmin(fmax(sty / sts, 1e-8), 1e8);
    ++L.updates;
    // M = [[sigma S^T S, L], [L^T, -D]] in chronological order
    const int k = L.k, n2 = 2 * k;
    for (int a = 0; a < k; ++a)
      for (int b = 0; b < k; ++b) {
        const int pa = L.slot[a], pb = L.slot[b];
        L.Mm[a * n2 + b] = L.sigma * L.SS[pa][pb];
        const double lab = a > b ? L.SY[pa][pb] : 0.0, lba = b > a ? L.SY[pb][pa] : 0.0;
        L.Mm[a * n2 + (k + b)] = lab;
        L.Mm[(k + a) * n2 + b] = lba;
        L.Mm[(k + a) * n2 + (k + b)] = a == b ? -L.SY[pa][pa] : 0.0;
      }
    for (int q = 0; q < n2 * n2; ++q) L.Mlu[q] = L.Mm[q];
    if (!lm_lu(L.Mlu, n2, L.Mpiv)) {
      logf("   limited-memory: the middle matrix is singular: history dropped");
      lm_reset();
      ++L.resets;
    }
    L.c_ready = false;
  }
  // out = B v
  DNLP_HD void lm_hess_mult(const double* v, double* out) {
    DNLP_IPM_LDS();
    LmState& L = *lm_;
    const double sg0 = L.sigma;
    ex_->map(N, [=] DNLP_HD(i64 j) { out[j] = sg0 * v[j]; });
    if (L.k == 0) return;
    const int n2 = 2 * L.k;
    double t[2 * kLmMaxH];
    lm_psi_dots(v, t);
    lm_lu_solve(L.Mlu, n2, L.Mpiv, t);
    for (int c = 0; c < n2; ++c) {
      double sc;
      const double* row = lm_col(c, sc);
      const double u = sc * t[c];
      ex_->map(N, [=] DNLP_HD(i64 j) { out[j] -= u * row[j]; });
    }
  }
  // Z = K0^{-1} [Psi; 0] and the LU factors of M - Psi^T Z_x, once per factorisation of K0
  DNLP_HD void lm_prepare() {
    DNLP_IPM_LDS();
    LmState& L = *lm_;
    L.c_ready = true;
    L.c_ok = true;
    if (L.k == 0) return;
    const int n2 = 2 * L.k;
    const i64 NM = N + m, NN = N;
    for (int c = 0; c < n2; ++c) {
      double sc;
      const double* row = lm_col(c, sc);
      double* tp = L.tmp;
      ex_->map(NM, [=] DNLP_HD(i64 j) { tp[j] = j < NN ? sc * row[j] : 0.0; });
      kkt_->solve(L.tmp, L.Z + static_cast<i64>(c) * NM);
    }
    for (int a = 0; a < n2; ++a) {
      double sa;
      const double* ra = lm_col(a, sa);
      for (int b = a; b < n2; ++b) {
        const double* zb = L.Z + static_cast<i64>(b) * NM;
        const double d = sa * ex_->sum(N, [=] DNLP_HD(i64 j) { return ra[j] * zb[j]; });
        L.Clu[a * n2 + b] = L.Mm[a * n2 + b] - d;
        L.Clu[b * n2 + a] = L.Mm[b * n2 + a] - d;
      }
    }
    bool fin = true;
    for (int q = 0; q < n2 * n2; ++q) fin = fin && std::isfinite(L.Clu[q]);
    L.c_ok = fin && lm_lu(L.Clu, n2, L.Cpiv);
    if (!L.c_ok) logf("   limited-memory: the capacitance matrix of the low-rank solve is singular");
  }
  // out = K^{-1} r through the factorisation of K0
  DNLP_HD void kkt_solve(const double* r, double* out) {
    DNLP_IPM_LDS();
    kkt_->solve(r, out);
    if constexpr (E::has_host_control) {
      if (!lm_on() || lm_->k == 0) return;
      LmState& L = *lm_;
      if (!L.c_ready) lm_prepare();
      if (!L.c_ok) return;                 // (the refinement on the true operator reports the bad solve)
      const int n2 = 2 * L.k;
      const i64 NM = N + m;
      double t[2 * kLmMaxH];
      lm_psi_dots(out, t);
      lm_lu_solve(L.Clu, n2, L.Cpiv, t);
      for (int c = 0; c < n2; ++c) {
        const double u = t[c];
        const double* zc = L.Z + static_cast<i64>(c) * NM;
        ex_->map(NM, [=] DNLP_HD(i64 j) { out[j] += u * zc[j]; });
      }
    }
  }

  // ---- evaluation helpers (scaled problem) --------------------------------------
  // f~(xp), g~(xp) -> returns false on non-finite values
  DNLP_HD bool eval_fg(const double* xp, double& fval, double* gout, bool check = true) {
    DNLP_IPM_LDS();
    double t0 = now_sec();
    md_->sweep(xp, false);
    fval = sf * md_->eval_f_after_sweep();
    md_->eval_g_after_sweep(gout);
    const double* sgp = sg;
    ex_->map(m, [=] DNLP_HD(i64 i) { gout[i] *= sgp[i]; });
    // NaN/inf detector (check = false: the caller folds it into its own pass, see measures())
    double chk = check ? ex_->sum(m, [=] DNLP_HD(i64 i) { return gout[i] - gout[i]; }) : 0.0;
    stats.t_eval += now_sec() - t0;
    return std::isfinite(fval) && chk == 0.0;
  }
  // gradient and Jacobian values at the point of the last sweep (scaled)
  DNLP_HD void eval_derivs_after_sweep() {
    DNLP_IPM_LDS();
    jty_valid_ = false;
    double t0 = now_sec();
    md_->eval_grad_after_sweep(grad);
    md_->eval_jac_after_sweep(jv);
    const double sff = sf;
    double* gr = grad;
    ex_->map(N, [=] DNLP_HD(i64 j) { gr[j] *= sff; });
    const double* sgp = sg;
    const i32* jr = md_->t.jac_rows;
    double* jvv = jv;
    ex_->map(md_->t.nnzJ, [=] DNLP_HD(i64 p) { jvv[p] *= sgp[jr[p]]; });
    stats.t_eval += now_sec() - t0;
  }
  DNLP_HD void eval_hessian() {
    DNLP_IPM_LDS();
    if (lm_on()) return;                 // limited-memory mode: no second derivatives
    double t0 = now_sec();
    const double* sgp = sg;
    const double* yy = y;
    double* lam = tM;
    ex_->map(m, [=] DNLP_HD(i64 i) { lam[i] = sgp[i] * yy[i]; });
    md_->eval_hess(x, sf, lam);
    stats.t_eval += now_sec() - t0;
  }

  // ---- initialisation (WB section 3.6) -------------------------------------------
  // x0_ctl: control-space pointer (host memory for the host / HIP spaces)
  DNLP_HD int begin(const double* x0_ctl) {
    DNLP_IPM_LDS();
    double t_start = now_sec();
    if (!x) allocate();
    lm_begin();
    const TapeView& T = md_->t;
    const double inf = opt.nlp_inf, brf = opt.bound_relax_factor;
    const bool warm = opt.warm_start != 0 && ws_mult_g != nullptr && ws_mult_xL != nullptr && ws_mult_xU != nullptr;
    const double k1 = warm ? opt.warm_start_bound_push : opt.bound_push, k2 = warm ? opt.warm_start_bound_frac : opt.bound_frac;
    const i64 NN = N;
    // variable bounds: optional relaxation (IPOPT bound_relax_factor; the reference sets it to 0),
    // |bound| >= nlp_inf means "no bound"
    {
      const double *lb = T.d_lb, *ub = T.d_ub, *cl = T.d_cl, *cu = T.d_cu;
      double *l = xL, *u = xU, *sl = sL, *su = sU;
      ex_->map(N, [=] DNLP_HD(i64 j) {
        double a = lb[j], b = ub[j];
        if (brf > 0) {
          if (a > -inf) a -= fmin(brf * fmax(1.0, fabs(a)), 1e-3);
          if (b < inf) b += fmin(brf * fmax(1.0, fabs(b)), 1e-3);
        }
        l[j] = a <= -inf ? -kInf : a;
        u[j] = b >= inf ? kInf : b;
      });
      ex_->map(m, [=] DNLP_HD(i64 i) {
        sl[i] = cl[i] <= -inf ? -kInf : cl[i];
        su[i] = cu[i] >= inf ? kInf : cu[i];
      });
      const double badx = ex_->max(N, [=] DNLP_HD(i64 j) { return l[j] > u[j] ? 1.0 : 0.0; });
      const double badc = m ? ex_->max(m, [=] DNLP_HD(i64 i) { return sl[i] > su[i] ? 1.0 : 0.0; }) : 0.0;
      if (badx > 0.0 || badc > 0.0) return status = Invalid_Option;
    }
    // scaling at the user's starting point (IPOPT gradient-based scaling)
    sf = 1.0;
    {
      double* sgp = sg;
      ex_->map(m, [=] DNLP_HD(i64 i) { sgp[i] = 1.0; });
    }
    ex_->h2d(x, x0_ctl, sizeof(double) * static_cast<size_t>(N));
    if (opt.nlp_scaling) {
      const double *xx = x, *l = xL, *u = xU;
      double* px = xt;
      ex_->map(N, [=] DNLP_HD(i64 j) { px[j] = push_into_bounds1(xx[j], l[j], u[j], k1, k2); });
      md_->sweep(xt, false);
      md_->eval_grad_after_sweep(grad);
      md_->eval_jac_after_sweep(jv);
      const double* gr = grad;
      const double gmax = ex_->max(N, [=] DNLP_HD(i64 j) { return fabs(gr[j]); });
      const double smax = opt.nlp_scaling_max_gradient;
      if (std::isfinite(gmax) && gmax > smax) sf = std::max(smax / gmax, 1e-8);
      if (m > 0) {
        const i64* rp_ = T.jac_rowptr;
        const double* jvv = jv;
        double* sgp = sg;
        if (T.nnzJ > 32 * m) {
          // long rows (a dense constraint block: 1e4 entries per row at BASELINE C3): 64 strided partial maxima
          // per row, then one pass over the partials -- one lane walking a whole row took 1.8 ms there
          if (!rowmax_tmp_) rowmax_tmp_ = A<double>(64 * m);
          double* part = rowmax_tmp_;
          ex_->map(64 * m, [=] DNLP_HD(i64 q) {
            const i64 i = q >> 6, lane = q & 63;
            double rmax = 0.0;
            bool fin = true;
            for (i64 p = rp_[i] + lane; p < rp_[i + 1]; p += 64) { const double a = fabs(jvv[p]); if (!(a <= kInf) || a == kInf) fin = false; if (a > rmax) rmax = a; }
            part[q] = fin ? rmax : -1.0;
          });
          ex_->map(m, [=] DNLP_HD(i64 i) {
            double rmax = 0.0;
            bool fin = true;
            for (int k = 0; k < 64; ++k) { const double a = part[64 * i + k]; if (a < 0.0) fin = false; if (a > rmax) rmax = a; }
            sgp[i] = (fin && rmax > smax) ? fmax(smax / rmax, 1e-8) : 1.0;
          });
        } else {
          ex_->map(m, [=] DNLP_HD(i64 i) {
            double rmax = 0.0;
            bool fin = true;
            for (i64 p = rp_[i]; p < rp_[i + 1]; ++p) { const double a = fabs(jvv[p]); if (!(a <= kInf) || a == kInf) fin = false; if (a > rmax) rmax = a; }
            sgp[i] = (fin && rmax > smax) ? fmax(smax / rmax, 1e-8) : 1.0;
          });
        }
      }
    }
    // scaled constraint bounds, equality mask, counts
    {
      double *sl = sL, *su = sU, *eq = eqmask;
      const double* sgp = sg;
      ex_->map(m, [=] DNLP_HD(i64 i) {
        eq[i] = (sl[i] == su[i]) ? 1.0 : 0.0;
        sl[i] *= sgp[i];
        su[i] *= sgp[i];
      });
      const double *l = xL, *u = xU;
      const i64 n_eq = m ? static_cast<i64>(ex_->sum(m, [=] DNLP_HD(i64 i) { return eq[i]; })) : 0;
      const i64 n_free = static_cast<i64>(ex_->sum(N, [=] DNLP_HD(i64 j) { return l[j] == u[j] ? 0.0 : 1.0; }));
      if (n_eq > n_free) return status = Not_Enough_Degrees_Of_Freedom;
      n_eq_ = n_eq;
      kkt_->n_fixed = N - n_free;
    }
    // start point pushed into the bounds; fixed variables (lb == ub) sit on their value, keep no
    // bound multipliers (their bounds are dropped) and are pinned in the KKT system
    {
      double *xx = x, *l = xL, *u = xU, *fm = fixmask;
      ex_->map(N, [=] DNLP_HD(i64 j) {
        const bool fx = l[j] == u[j];
        xx[j] = fx ? l[j] : push_into_bounds1(xx[j], l[j], u[j], k1, k2);
        fm[j] = fx ? 1.0 : 0.0;
        if (fx) { l[j] = -kInf; u[j] = kInf; }
      });
    }
    nb_cache_ = -1;
    // evaluate at the pushed point
    if (!eval_fg(x, f, g)) return status = Invalid_Number_Detected;
    eval_derivs_after_sweep();
    // slacks: s = d(x) pushed into [sL, sU]; equality rows carry their right-hand side
    {
      double* ss = s;
      const double *gg = g, *sl = sL, *su = sU, *eq = eqmask;
      ex_->map(m, [=] DNLP_HD(i64 i) { ss[i] = eq[i] != 0.0 ? sl[i] : push_into_bounds1(gg[i], sl[i], su[i], k1, k2); });
    }
    // bound multipliers
    {
      const double zi = opt.bound_mult_init_val;
      const double *l = xL, *u = xU, *sl = sL, *su = sU, *eq = eqmask;
      double *a = zL, *b = zU, *c = vL, *d = vU;
      ex_->map(N, [=] DNLP_HD(i64 j) { a[j] = (l[j] > -kInf) ? zi : 0.0; b[j] = (u[j] < kInf) ? zi : 0.0; });
      ex_->map(m, [=] DNLP_HD(i64 i) {
        c[i] = (eq[i] == 0.0 && sl[i] > -kInf) ? zi : 0.0;
        d[i] = (eq[i] == 0.0 && su[i] < kInf) ? zi : 0.0;
      });
    }
    (void)NN;
    mu = opt.mu_init;
    tau = std::max(0.99, 1.0 - mu);
    jty_valid_ = false;
    ex_->zero(y, sizeof(double) * static_cast<size_t>(m));
    if (warm) {
      // given multipliers -> scaled problem: y~ = y sf / sg, z~ = z sf; bound multipliers of the
      // inequality slacks from the sign convention  -y - vL + vU = 0; all kept >= mult_bound_push
      const double sff = sf, mp = opt.warm_start_mult_bound_push;
      const double *wy = ws_mult_g, *wl = ws_mult_xL, *wu = ws_mult_xU, *sgp = sg;
      const double *l = xL, *u = xU, *sl = sL, *su = sU, *eq = eqmask;
      double *yy = y, *a = zL, *b = zU, *c = vL, *d = vU;
      ex_->map(m, [=] DNLP_HD(i64 i) {
        const double yi = wy[i] * sff / sgp[i];
        yy[i] = yi;
        const bool in = eq[i] == 0.0;
        c[i] = (in && sl[i] > -kInf) ? fmax(-yi, mp) : 0.0;
        d[i] = (in && su[i] < kInf) ? fmax(yi, mp) : 0.0;
      });
      ex_->map(N, [=] DNLP_HD(i64 j) {
        a[j] = (l[j] > -kInf) ? fmax(wl[j] * sff, mp) : 0.0;
        b[j] = (u[j] < kInf) ? fmax(wu[j] * sff, mp) : 0.0;
      });
    } else if (m > 0 && opt.least_square_init_duals >= 0) {
      init_multipliers_ls();
    }
    filter_clear();
    double th0 = theta_at(g, s);
    theta_max = 1e4 * std::max(1.0, th0);
    theta_min = 1e-4 * std::max(1.0, th0);
    iter = 0;
    acceptable_count = 0;
    delta_w_last = 0.0;
    dc_fixed_count_ = 0; dc_fixed_last_ = false; always_dc_ = false;
    lan_warm_ = false; lan_width_ = 0.0;
    resto_stationary_ = false; resto_theta_ = 0.0;
    tiny_streak_ = 0;
    e_cached_valid_ = false;
    fixed_mode = false;
    n_hist = 0;
    initialized = true;
    status = Internal_Error;
    stats = IpmStats();
    t_begin_ = t_start;
    logf("iter    objective    inf_pr   inf_du lg(mu)  ||d||  lg(rg) alpha_du alpha_pr  ls");
    return 0;
  }

  // least-squares equality multipliers (WB eq. (36)); discarded above constr_mult_init_max
  DNLP_HD void init_multipliers_ls() {
    DNLP_IPM_LDS();
    jty_valid_ = false;                 // (y is about to change)
    const double* eq = eqmask;
    if constexpr (E::has_host_control) if (!kkt_->pivoted && m <= 8) {
      // Few rows on a large system: the (1,1) block of the least-squares system is the identity,
      // so it condenses to the m x m system (J J^T + D) y = J r_x - r_y — no O(n^3) work.
      double* r = rhs;
      const double *gr = grad, *a = zL, *b = zU, *c = vL, *d = vU, *fm = fixmask;
      ex_->map(N, [=] DNLP_HD(i64 j) { r[j] = fm[j] != 0.0 ? 0.0 : -(gr[j] - a[j] + b[j]); });
      double G[64] = {0.0}, bb[8] = {0.0}, heq[8], hc[8], hd[8];
      ex_->d2h(heq, eqmask, sizeof(double) * static_cast<size_t>(m));
      ex_->d2h(hc, vL, sizeof(double) * static_cast<size_t>(m));
      ex_->d2h(hd, vU, sizeof(double) * static_cast<size_t>(m));
      if (!lanQ) { lanQ = A<double>(static_cast<i64>(8) * N); }
      for (i64 i = 0; i < m; ++i) {
        double* e = tM;
        ex_->map(m, [=] DNLP_HD(i64 q) { e[q] = (q == i) ? 1.0 : 0.0; });
        double* qi = lanQ + i * N;
        md_->jac_tmult(jv, tM, qi);
        ex_->map(N, [=] DNLP_HD(i64 j) { if (fm[j] != 0.0) qi[j] = 0.0; });
      }
      for (i64 i = 0; i < m; ++i) {
        const double* qi = lanQ + i * N;
        for (i64 k = 0; k <= i; ++k) {
          const double* qk = lanQ + k * N;
          const double v = ex_->sum(N, [=] DNLP_HD(i64 j) { return qi[j] * qk[j]; });
          G[static_cast<size_t>(i * m + k)] = G[static_cast<size_t>(k * m + i)] = v;
        }
        const double ry = (heq[static_cast<size_t>(i)] == 0.0) ? -(-hc[static_cast<size_t>(i)] + hd[static_cast<size_t>(i)]) : 0.0;
        bb[static_cast<size_t>(i)] = ex_->sum(N, [=] DNLP_HD(i64 j) { return qi[j] * r[j]; }) - ry;
        G[static_cast<size_t>(i * m + i)] += (heq[static_cast<size_t>(i)] == 0.0) ? 1.0 : 0.0;
      }
      // dense Gaussian elimination with partial pivoting on the m x m system
      bool ok = true;
      const int mm = static_cast<int>(m);
      for (int k = 0; k < mm && ok; ++k) {
        int piv = k;
        for (int i = k + 1; i < mm; ++i) if (std::fabs(G[i * mm + k]) > std::fabs(G[piv * mm + k])) piv = i;
        if (std::fabs(G[piv * mm + k]) < 1e-14) { ok = false; break; }
        if (piv != k) { for (int j = 0; j < mm; ++j) std::swap(G[k * mm + j], G[piv * mm + j]); std::swap(bb[k], bb[piv]); }
        for (int i = k + 1; i < mm; ++i) {
          const double f = G[i * mm + k] / G[k * mm + k];
          for (int j = k; j < mm; ++j) G[i * mm + j] -= f * G[k * mm + j];
          bb[i] -= f * bb[k];
        }
      }
      if (ok) {
        double ymax = 0.0;
        for (int i = mm - 1; i >= 0; --i) {
          double v = bb[i];
          for (int j = i + 1; j < mm; ++j) v -= G[i * mm + j] * bb[j];
          bb[i] = v / G[i * mm + i];
          ymax = std::fmax(ymax, std::fabs(bb[i]));
        }
        if (std::isfinite(ymax) && ymax <= opt.constr_mult_init_max)
          ex_->h2d(y, bb, sizeof(double) * static_cast<size_t>(m));
        return;
      }
    }
    double *sx = Sx, *dd = Dd;
    ex_->map(N, [=] DNLP_HD(i64 j) { sx[j] = 1.0; });
    ex_->map(m, [=] DNLP_HD(i64 i) { dd[i] = (eq[i] == 0.0) ? 1.0 : 0.0; });
    if constexpr (E::has_condensed_ls) if (!kkt_->pivoted && !kkt_->sparse && m > 8 && 4 * m <= N) {
      // A moderate number of rows on a large dense system (BASELINE C3: m = 1e3, N = 1e4): the identity (1,1)
      // block condenses the least-squares system to S y = r_y - J r_x with S = -(D + J J^T), one MFMA pass over
      // J and an order-m factorisation instead of an O((N + m)^3) one.
      double* r = rhs;
      const double *gr = grad, *a = zL, *b = zU, *c = vL, *d = vU, *fm = fixmask;
      const i64 NN = N;
      ex_->map(N, [=] DNLP_HD(i64 j) { r[j] = fm[j] != 0.0 ? 0.0 : -(gr[j] - a[j] + b[j]); });
      md_->jac_mult(jv, rhs, sol);                                 // sol[0..m) = J r_x
      double* t = sol;
      ex_->map(m, [=] DNLP_HD(i64 i) { t[i] = ((eq[i] == 0.0) ? -(-c[i] + d[i]) : 0.0) - t[i]; });
      if (ex_->condensed_ls(N, m, md_->t.nnzJ, md_->t.jac_rows, md_->t.jac_cols, jv, fixmask, Dd, sol, sol + m)) {
        const double* so = sol + m;
        const double ymax = ex_->max(m, [=] DNLP_HD(i64 i) { return fabs(so[i]); });
        if (std::isfinite(ymax) && ymax <= opt.constr_mult_init_max) {
          double* yy = y;
          ex_->map(m, [=] DNLP_HD(i64 i) { yy[i] = so[i]; });
        }
        (void)NN;
        return;
      }
    }
    int nneg = 0, nzero = 0;
    md_->clear_dense_w();
    ex_->zero(md_->Hs, sizeof(double) * static_cast<size_t>(md_->t.nnzH));
    bool ok = kkt_->assemble_factor(*md_, jv, Sx, Dd, fixmask, 0.0, &nneg, &nzero);
    stats.factorizations++;
    if (!ok || nzero > 0 || nneg != m) {
      // retry once with a tiny dual regularisation for rank-deficient Jacobians
      ex_->map(m, [=] DNLP_HD(i64 i) { dd[i] += 1e-8; });
      ok = kkt_->assemble_factor(*md_, jv, Sx, Dd, fixmask, 0.0, &nneg, &nzero);
      stats.factorizations++;
      if (!ok) return;
    }
    double* r = rhs;
    const double *gr = grad, *a = zL, *b = zU, *c = vL, *d = vU;
    const i64 NN = N;
    ex_->map(N, [=] DNLP_HD(i64 j) { r[j] = -(gr[j] - a[j] + b[j]); });
    ex_->map(m, [=] DNLP_HD(i64 i) { r[NN + i] = (eq[i] == 0.0) ? -(-c[i] + d[i]) : 0.0; });
    kkt_->solve(rhs, sol);
    const double* so = sol;
    double ymax = ex_->max(m, [=] DNLP_HD(i64 i) { return fabs(so[NN + i]); });
    if (std::isfinite(ymax) && ymax <= opt.constr_mult_init_max) {
      double* yy = y;
      ex_->map(m, [=] DNLP_HD(i64 i) { yy[i] = so[NN + i]; });
    }
  }

  // ---- measures -------------------------------------------------------------------
  // primal residual: g - cl for equalities, g - s for inequalities
  DNLP_HD double theta_at(const double* gg, const double* ss) {
    DNLP_IPM_LDS();
    const double *eq = eqmask, *sl = sL;
    return ex_->sum(m, [=] DNLP_HD(i64 i) { return fabs(eq[i] != 0.0 ? gg[i] - sl[i] : gg[i] - ss[i]); });
  }
  DNLP_HD double barrier_at(double fv, const double* xx, const double* ss, double muv) {
    DNLP_IPM_LDS();
    const double *l = xL, *u = xU, *sl = sL, *su = sU, *eq = eqmask;
    const double kd = opt.kappa_d;
    double bx = ex_->sum(N, [=] DNLP_HD(i64 j) {
      double v = 0.0;
      const bool hl = l[j] > -kInf, hu = u[j] < kInf;
      if (hl) v -= log(xx[j] - l[j]);
      if (hu) v -= log(u[j] - xx[j]);
      if (hl && !hu) v += kd * (xx[j] - l[j]);
      if (hu && !hl) v += kd * (u[j] - xx[j]);
      return v;
    });
    double bs = ex_->sum(m, [=] DNLP_HD(i64 i) {
      if (eq[i] != 0.0) return 0.0;
      double v = 0.0;
      const bool hl = sl[i] > -kInf, hu = su[i] < kInf;
      if (hl) v -= log(ss[i] - sl[i]);
      if (hu) v -= log(su[i] - ss[i]);
      if (hl && !hu) v += kd * (ss[i] - sl[i]);
      if (hu && !hl) v += kd * (su[i] - ss[i]);
      return v;
    });
    return fv + muv * (bx + bs);
  }

  // theta, the barrier terms and the NaN / inf detector of g in ONE pass over [variables | constraint rows]
  // (theta_at + barrier_at + the check of eval_fg are four reductions)
  struct Measures { double theta, phi, chk; };
  DNLP_HD Measures measures(double fv, const double* gg, const double* xx, const double* ss, double muv) {
    DNLP_IPM_LDS();
    const double *l = xL, *u = xU, *sl = sL, *su = sU, *eq = eqmask;
    const double kd = opt.kappa_d;
    const i64 NN = N;
    const RMulti R = ex_->template reduce_multi<0, 3>(N + m, [=] DNLP_HD(i64 k) -> RMulti {
      RMulti v;
      v.mx[0] = v.mx[1] = v.mx[2] = v.mx[3] = 0.0;
      v.sm[0] = v.sm[1] = v.sm[2] = v.sm[3] = 0.0;
      double bt = 0.0;
      if (k < NN) {
        const i64 j = k;
        const bool hl = l[j] > -kInf, hu = u[j] < kInf;
        if (hl) bt -= log(xx[j] - l[j]);
        if (hu) bt -= log(u[j] - xx[j]);
        if (hl && !hu) bt += kd * (xx[j] - l[j]);
        if (hu && !hl) bt += kd * (u[j] - xx[j]);
      } else {
        const i64 i = k - NN;
        v.sm[0] = fabs(eq[i] != 0.0 ? gg[i] - sl[i] : gg[i] - ss[i]);
        v.sm[2] = gg[i] - gg[i];
        if (eq[i] == 0.0) {
          const bool hl = sl[i] > -kInf, hu = su[i] < kInf;
          if (hl) bt -= log(ss[i] - sl[i]);
          if (hu) bt -= log(su[i] - ss[i]);
          if (hl && !hu) bt += kd * (ss[i] - sl[i]);
          if (hu && !hl) bt += kd * (su[i] - ss[i]);
        }
      }
      v.sm[1] = bt;
      return v; });
    return Measures{R.sm[0], fv + muv * R.sm[1], R.sm[2]};
  }

  // J^T y of the current (Jacobian values, multipliers), kept in tN: the dual residuals, the three barrier_terms of a
  // free-mode iteration and the directional derivative of the line search all need it, and it changes only when the
  // point is accepted (eval_derivs_after_sweep) or the multipliers are reset — one product per iteration instead of
  // five.  (kkt_residual's own J^T v goes to xt, which is free whenever a system is solved.)
  DNLP_HD const double* jty() {
    DNLP_IPM_LDS();
    if (!jty_valid_) { md_->jac_tmult(jv, y, tN); jty_valid_ = true; }
    return tN;
  }
  // dual residuals rx = grad + J^T y - zL + zU ; rs = -y - vL + vU (inequality rows)
  DNLP_HD void dual_residuals() {
    DNLP_IPM_LDS();
    const double* jt = jty();
    double *r = rx, *q = rs;
    const double *gr = grad, *a = zL, *b = zU, *c = vL, *d = vU, *yy = y, *eq = eqmask, *fm = fixmask;
    const i64 NN = N;
    ex_->map(N + m, [=] DNLP_HD(i64 k) {
      if (k < NN) { const i64 j = k; r[j] = fm[j] != 0.0 ? 0.0 : gr[j] + jt[j] - a[j] + b[j]; }
      else { const i64 i = k - NN; q[i] = (eq[i] == 0.0) ? -yy[i] - c[i] + d[i] : 0.0; } });
  }

  struct Err { double dual, primal, cmpl, sd, sc, total, primal_unscaled; };

  // WB eq. (5): scaled optimality error for barrier parameter muv
  DNLP_HD Err error(double muv) {
    DNLP_IPM_LDS();
    dual_residuals();
    const double *r = rx, *q = rs, *gg = g, *ss = s, *eq = eqmask, *sl = sL, *su = sU, *l = xL, *u = xU,
                 *xx = x, *a = zL, *b = zU, *c = vL, *d = vU, *yy = y, *sgp = sg;
    // every norm of WB eq. (5) in ONE pass over [variables | constraint rows]: maxima = dual residual,
    // primal residual, complementarity, unscaled primal residual (the convergence test's); sums = |y| + |v|
    // and |z|.  (Nine separate reductions before: nine launches and read-backs on the host-driven space.)
    const i64 NN = N;
    const RMulti R = ex_->template reduce_multi<4, 2>(N + m, [=] DNLP_HD(i64 k) -> RMulti {
      RMulti v;
      v.mx[0] = v.mx[1] = v.mx[2] = v.mx[3] = 0.0;
      v.sm[0] = v.sm[1] = v.sm[2] = v.sm[3] = 0.0;
      if (k < NN) {
        const i64 j = k;
        v.mx[0] = fabs(r[j]);
        double cv = 0.0;
        if (l[j] > -kInf) cv = fmax(cv, fabs((xx[j] - l[j]) * a[j] - muv));
        if (u[j] < kInf) cv = fmax(cv, fabs((u[j] - xx[j]) * b[j] - muv));
        v.mx[2] = cv;
        v.sm[1] = fabs(a[j]) + fabs(b[j]);
      } else {
        const i64 i = k - NN;
        v.mx[0] = fabs(q[i]);
        const double pr = fabs(eq[i] != 0.0 ? gg[i] - sl[i] : gg[i] - ss[i]);
        v.mx[1] = pr;
        v.mx[3] = pr / sgp[i];
        double cv = 0.0;
        if (eq[i] == 0.0) {
          if (sl[i] > -kInf) cv = fmax(cv, fabs((ss[i] - sl[i]) * c[i] - muv));
          if (su[i] < kInf) cv = fmax(cv, fabs((su[i] - ss[i]) * d[i] - muv));
        }
        v.mx[2] = cv;
        v.sm[0] = fabs(yy[i]) + fabs(c[i]) + fabs(d[i]);
      }
      return v; });
    Err e;
    e.dual = std::max(R.mx[0], 0.0);
    e.primal = m ? std::max(R.mx[1], 0.0) : 0.0;
    e.cmpl = std::max(R.mx[2], 0.0);
    e.primal_unscaled = m ? std::max(R.mx[3], 0.0) : 0.0;
    const double smax = 100.0;
    const double sy = R.sm[0], sz = R.sm[1];
    i64 nb = n_bound_mults();
    e.sd = std::max(smax, (sy + sz) / std::max<double>(1.0, static_cast<double>(m + nb))) / smax;
    e.sc = std::max(smax, sz / std::max<double>(1.0, static_cast<double>(nb))) / smax;
    e.total = std::max(std::max(e.dual / e.sd, e.primal), e.cmpl / e.sc);
    return e;
  }

  DNLP_HD i64 n_bound_mults() {
    DNLP_IPM_LDS();
    if (nb_cache_ >= 0) return nb_cache_;
    const double *l = xL, *u = xU, *sl = sL, *su = sU, *eq = eqmask;
    double c1 = ex_->sum(N, [=] DNLP_HD(i64 j) { return (l[j] > -kInf ? 1.0 : 0.0) + (u[j] < kInf ? 1.0 : 0.0); });
    double c2 = m ? ex_->sum(m, [=] DNLP_HD(i64 i) {
      return eq[i] != 0.0 ? 0.0 : (sl[i] > -kInf ? 1.0 : 0.0) + (su[i] < kInf ? 1.0 : 0.0); }) : 0.0;
    nb_cache_ = static_cast<i64>(c1 + c2);
    return nb_cache_;
  }

  // ---- search direction (WB section 2.2 / 3.1) ---------------------------------------
  // Builds Sigma and the barrier right-hand sides for barrier parameter muv.
  DNLP_HD void barrier_terms(double muv) {
    DNLP_IPM_LDS();
    const double* jt = jty();
    double *sx = Sx, *sS = Ss, *r = rx, *q = rs, *p = rp;
    const double *l = xL, *u = xU, *sl = sL, *su = sU, *eq = eqmask, *xx = x, *ss = s, *a = zL, *b = zU,
                 *c = vL, *d = vU, *gr = grad, *yy = y, *gg = g, *fm = fixmask;
    const double kd = opt.kappa_d;
    const i64 NN = N;
    ex_->map(N + m, [=] DNLP_HD(i64 k) {
      if (k < NN) {
      const i64 j = k;
      double sig = 0.0, gphi = gr[j];
      const bool hl = l[j] > -kInf, hu = u[j] < kInf;
      if (hl) { sig += a[j] / (xx[j] - l[j]); gphi -= muv / (xx[j] - l[j]); }
      if (hu) { sig += b[j] / (u[j] - xx[j]); gphi += muv / (u[j] - xx[j]); }
      if (hl && !hu) gphi += kd * muv;
      if (hu && !hl) gphi -= kd * muv;
      sx[j] = sig;
      r[j] = fm[j] != 0.0 ? 0.0 : gphi + jt[j];             // grad_x phi + J^T y
      return;
      }
      const i64 i = k - NN;
      if (eq[i] != 0.0) { sS[i] = 0.0; q[i] = 0.0; p[i] = gg[i] - sl[i]; return; }
      double sig = 0.0, gphi = 0.0;
      const bool hl = sl[i] > -kInf, hu = su[i] < kInf;
      if (hl) { sig += c[i] / (ss[i] - sl[i]); gphi -= muv / (ss[i] - sl[i]); }
      if (hu) { sig += d[i] / (su[i] - ss[i]); gphi += muv / (su[i] - ss[i]); }
      if (hl && !hu) gphi += kd * muv;
      if (hu && !hl) gphi -= kd * muv;
      sS[i] = sig;
      q[i] = gphi - yy[i];                                  // grad_s phi - y
      p[i] = gg[i] - ss[i];
    });
  }

  // inertia-correcting factorisation of the reduced KKT matrix (WB Algorithm IC)
  // Provable lower bound on the primal regularisation delta_w (large unpivoted systems).
  //
  // All-equality case with few rows (BASELINE C4: m = 1): K = [H~ + dw I, J^T; J, 0] has inertia
  // (N, m, 0) iff the reduced Hessian Z^T (H~ + dw I) Z is positive definite, i.e. iff
  // dw > -lambda_min(P H~ P | range P), P the orthogonal projector onto null(J).  Ritz values of
  // a k-step Lanczos run on P H~ P (full reorthogonalisation; the Krylov space is independent of
  // the shift dw) bound lambda_min from above, so every dw < -theta_1 is certain to fail.
  // General case: H~ is a principal submatrix of K, so by Cauchy interlacing dw < -theta_(m+1)
  // of H~ itself is certain to fail.  Either way the doomed O(n^3) factorisations are skipped at
  // the price of k Hessian-vector products (k HBM sweeps).  Returns (lower bound, spectral width).
  DNLP_HD D2 lanczos_delta_lower_bound() {
    DNLP_IPM_LDS();
    if constexpr (!E::has_host_control) {
      return {0.0, 0.0};
    } else {
    const int kmax = 24, qmax = 8;
    if (N < 4 * kmax) return {0.0, 0.0};
    const double* eqm = eqmask;
    const bool all_eq = m > 0 && ex_->sum(m, [=] DNLP_HD(i64 i) { return eqm[i]; }) == static_cast<double>(m);
    const bool projected = all_eq && m <= qmax;
    if (!projected && m + 1 > kmax - 2) return {0.0, 0.0};
    if (!lanV) { lanV = A<double>(static_cast<i64>(kmax + 1) * N); lanW = A<double>(N); lanQ = A<double>(static_cast<i64>(qmax) * N); lanY = A<double>(N); }
    int nq = 0;
    if (projected) {
      // orthonormal basis of the row space of J (modified Gram-Schmidt on J^T e_i)
      for (i64 i = 0; i < m; ++i) {
        double* e = tM;
        ex_->map(m, [=] DNLP_HD(i64 r) { e[r] = (r == i) ? 1.0 : 0.0; });
        double* q = lanQ + static_cast<i64>(nq) * N;
        md_->jac_tmult(jv, tM, q);
        for (int pass = 0; pass < 2; ++pass)
          for (int c = 0; c < nq; ++c) {
            const double* qc = lanQ + static_cast<i64>(c) * N;
            const double d = ex_->sum(N, [=] DNLP_HD(i64 j) { return q[j] * qc[j]; });
            ex_->map(N, [=] DNLP_HD(i64 j) { q[j] -= d * qc[j]; });
          }
        const double nr = std::sqrt(ex_->sum(N, [=] DNLP_HD(i64 j) { return q[j] * q[j]; }));
        if (nr > 1e-14) { ex_->map(N, [=] DNLP_HD(i64 j) { q[j] /= nr; }); ++nq; }
      }
    }
    double cbuf[32];
    auto project = [&](double* v) { ex_->orthogonalize(nq, lanQ, N, v, cbuf); };
    double al[24], be[24];
    int nal = 0, nbe = 0;
    // wanted-th smallest eigenvalue of the leading kk x kk part of the tridiagonal T by Sturm bisection
    const int want = projected ? 1 : static_cast<int>(m) + 1;
    double width = 0.0;
    auto ritz = [&](int kk) {
      double lo = al[0], hi = al[0];
      for (int i = 0; i < kk; ++i) {
        const double r = (i > 0 ? std::fabs(be[i - 1]) : 0.0) + (i < nbe && i + 1 < kk ? std::fabs(be[i]) : 0.0);
        lo = std::min(lo, al[i] - r);
        hi = std::max(hi, al[i] + r);
      }
      width = hi - lo;
      auto count_below = [&](double t) {
        int c = 0;
        double d = 1.0;
        for (int i = 0; i < kk; ++i) {
          const double b2 = (i > 0) ? be[i - 1] * be[i - 1] : 0.0;
          d = (al[i] - t) - (i > 0 ? b2 / d : 0.0);
          if (d == 0.0) d = 1e-300;
          if (d < 0.0) ++c;
        }
        return c;
      };
      for (int it = 0; it < 100; ++it) {
        const double mid = 0.5 * (lo + hi);
        if (count_below(mid) >= want) hi = mid; else lo = mid;
      }
      return hi;
    };
    // Warm start (projected case): the previous call's Ritz vector of the smallest eigenvalue —
    // between interior-point iterations the extreme eigenvector barely moves, so the Krylov space
    // built on it contains the new one after a few steps instead of 24 (every Ritz value is an
    // upper bound of lambda_min whatever the start vector, so the bound stays certified).  The run
    // stops when the Ritz value has settled to 1e-4 of the spectral width seen by the first, cold run.
    const bool warm = projected && lan_warm_;
    double* v0 = lanV;
    {
      // deterministic vector with components in every direction (cold start; small admixture when warm)
      const double* yw = lanY;
      const double mix = (warm ? 1e-3 : 1.0) / std::sqrt(static_cast<double>(N)), wy = warm ? 1.0 : 0.0;   // (no member access inside the lambda)
      ex_->map(N, [=] DNLP_HD(i64 j) {
        v0[j] = mix * (1.0 + 0.5 * sin(1.0 + 0.37 * static_cast<double>(j % 1009))) + (wy != 0.0 ? wy * yw[j] : 0.0); });
      project(v0);
      const double nrm = std::sqrt(ex_->sum(N, [=] DNLP_HD(i64 j) { return v0[j] * v0[j]; }));
      if (!(nrm > 0.0)) return {0.0, 0.0};
      ex_->map(N, [=] DNLP_HD(i64 j) { v0[j] /= nrm; });
    }
    double theta_prev = 0.0;
    for (int k = 0; k < kmax; ++k) {
      double* vk = lanV + static_cast<i64>(k) * N;
      double* w = lanW;
      md_->hess_mult(vk, w);
      const double *sx = Sx, *fm = fixmask;
      ex_->map(N, [=] DNLP_HD(i64 j) { w[j] = fm[j] != 0.0 ? 0.0 : w[j] + sx[j] * vk[j]; });
      project(w);
      const double a = ex_->sum(N, [=] DNLP_HD(i64 j) { return w[j] * vk[j]; });
      al[nal++] = a;
      // full reorthogonalisation against all previous vectors (classical Gram-Schmidt, twice):
      // two sweeps of w per pass instead of one reduction per stored vector
      for (int pass = 0; pass < 2; ++pass) ex_->orthogonalize(k + 1, lanV, N, w, cbuf);
      const double b = std::sqrt(ex_->sum(N, [=] DNLP_HD(i64 j) { return w[j] * w[j]; }));
      if (!(b > 1e-12 * (std::fabs(a) + 1.0)) || k + 1 == kmax) break;
      if (warm && nal >= 3) {
        const double th = ritz(nal);
        if (std::fabs(th - theta_prev) <= 1e-4 * lan_width_) break;
        theta_prev = th;
      } else if (warm) {
        theta_prev = ritz(nal);
      }
      be[nbe++] = b;
      double* vn = lanV + static_cast<i64>(k + 1) * N;
      ex_->map(N, [=] DNLP_HD(i64 j) { vn[j] = w[j] / b; });
    }
    const int kk = nal;
    if (kk < want) return {0.0, 0.0};
    const double theta = ritz(kk);
    if (!warm) lan_width_ = width;
    width = std::max(width, lan_width_);
    if (projected) {
      // Ritz vector for the next call: s from the three-term recurrence of (T - theta I) s = 0
      double sv[24];
      sv[0] = 1.0;
      for (int i = 0; i + 1 < kk; ++i) {
        const double bi = (i < nbe && be[i] != 0.0) ? be[i] : 1e-300;
        sv[i + 1] = ((theta - al[i]) * sv[i] - (i > 0 ? be[i - 1] * sv[i - 1] : 0.0)) / bi;
      }
      double nn = 0.0;
      for (int i = 0; i < kk; ++i) nn += sv[i] * sv[i];
      nn = std::sqrt(nn);
      bool fin = nn > 0.0 && std::isfinite(nn);
      if (fin) {
        for (int i = 0; i < kk; ++i) sv[i] /= nn;
        ex_->v_comb(kk, lanV, N, sv, lanY);
        lan_warm_ = true;
      } else {
        lan_warm_ = false;
      }
    }
    return {theta < 0.0 ? -theta : 0.0, width};
    }
  }

  // one factorisation attempt: 0 ok, 1 wrong inertia, 2 singular
  DNLP_HD int try_factor(double dw, double dc) {
    DNLP_IPM_LDS();
    int nneg = 0, nzero = 0;
    double* dd = Dd;
    const double *sS = Ss, *eq = eqmask;
    ex_->map(m, [=] DNLP_HD(i64 i) { dd[i] = dc + (eq[i] == 0.0 ? 1.0 / fmax(sS[i] + dw, 1e-20) : 0.0); });
    double t0 = now_sec();
    // (limited-memory mode: K0 carries the diagonal sigma I of the BFGS matrix; the low-rank part enters in kkt_solve)
    double dwx = dw;
    if constexpr (E::has_host_control) if (lm_on()) { dwx = dw + lm_->sigma; lm_->c_ready = false; }
    bool ok = kkt_->assemble_factor(*md_, jv, Sx, Dd, fixmask, dwx, &nneg, &nzero);
    stats.t_factor += now_sec() - t0;
    stats.factorizations++;
    last_nneg_ = nneg;
    if (!ok) return 2;
    if (nzero > 0) return 2;
    return nneg == m ? 0 : 1;
  }

  DNLP_HD bool factor_with_inertia(double& delta_w, double& delta_c) {
    DNLP_IPM_LDS();
    const double dw_min = 1e-20, dw_0 = 1e-4, dw_max = opt.max_hessian_perturbation, dc_bar = 1e-8, kwp = 8.0, kwpb = 100.0, kwm = 1.0 / 3.0, kc = 0.25;
    delta_w = 0.0; delta_c = 0.0;
    // large unpivoted systems: skip regularisation values that are provably too small
    double dw_lb = 0.0, dw_first = 0.0;
    bool have_lb = false;
    auto get_lb = [&]() {
      const D2 lbw = lanczos_delta_lower_bound();
      dw_lb = lbw.first;
      // first trial above the bound: Ritz values converge from above, so add 5% and 2% of the
      // spectral width (an over-regularisation of that size is harmless, a failed attempt is not)
      dw_first = dw_lb > 0.0 ? 1.05 * dw_lb + 0.02 * lbw.second : 0.0;
      have_lb = true;
    };
    const bool use_lb = opt.lanczos_inertia_bound && !kkt_->pivoted && (N + m) >= opt.lanczos_min_n && !lm_on();
    auto attempt = [&](double dw, double dc) -> int {
      if (use_lb && have_lb && dw < dw_lb) { stats.skipped_factorizations++; return 1; }
      double t0 = now_sec();
      int r = try_factor(dw, dc);
      if (opt.print_level >= 6) logf("   factor attempt delta_w=%.4e delta_c=%.2e -> %d (%.2fs) lanczos_lb=%.4e", dw, dc, r, now_sec() - t0, dw_lb);
      return r;
    };
    if (use_lb && (iter == 0 || delta_w_used_last_iter_)) get_lb();
    // Degenerate Jacobian (IPOPT's PDPerturbationHandler heuristic): a rank-deficient J shows up as
    // a zero pivot or — with static pivots, where the equality-row pivots shrink like a^2 / delta_w
    // but never vanish — as an inertia that no delta_w repairs, while the dual regularisation
    // delta_c alone does.  After three such iterations delta_c is applied from the first attempt on.
    const double dc_val = dc_bar * std::pow(mu, kc);
    auto dc_fixed = [&]() {
      dc_fixed_last_ = true;
      if (++dc_fixed_count_ >= 3 && !always_dc_) {
        always_dc_ = true;
        logf("   Jacobian treated as degenerate: delta_c is applied in every factorisation from here on");
      }
    };
    int r = attempt(0.0, always_dc_ ? dc_val : 0.0);
    if (always_dc_) delta_c = dc_val;
    // Static pivots (sparse KKT) that are singular without any regularisation in consecutive
    // iterations are structural (free variables without curvature next to unregularised equality
    // rows): where the dense matrix is affordable the instance switches to Bunch-Kaufman pivoting
    // for good instead of regularising every step.
    if (r == 2 && !always_dc_ && kkt_->can_fallback() && (!opt.lazy_dense_fallback || ladder_rung_ > 0)) {
      ++sparse_singular_streak_;
      // small systems switch at once (the dense factorisation costs nothing there); larger ones only
      // when the singularity persists beyond the first iteration (multipliers start at zero)
      if ((N + m) <= 512 || (iter >= 1 && sparse_singular_streak_ >= 2)) {
        kkt_->fallback_to_dense();
        logf("   static pivot sequence singular in consecutive iterations: Bunch-Kaufman from here on");
        r = attempt(0.0, 0.0);
      }
    } else {
      sparse_singular_streak_ = 0;
    }
    if (r == 0) { delta_w_used_last_iter_ = false; if (!always_dc_) dc_fixed_last_ = false; return true; }
    if (use_lb && !have_lb) get_lb();
    delta_w_used_last_iter_ = true;
    if (r == 2) delta_c = dc_val;
    delta_w = (delta_w_last == 0.0) ? dw_0 : std::max(dw_min, kwm * delta_w_last);
    if (have_lb && delta_w < dw_first) delta_w = dw_first;   // first value not provably hopeless
    // singular with delta_w = 0 (or the wrong inertia that delta_c repaired in the previous
    // iteration): first try the dual regularisation alone
    if (delta_c == 0.0 ? (r == 2 || (r == 1 && dc_fixed_last_)) : (r == 2 && !always_dc_)) {
      int r2 = attempt(0.0, dc_val);
      if (r2 == 0) { delta_c = dc_val; delta_w = 0.0; if (r == 1) dc_fixed(); return true; }
      if (r == 1) dc_fixed_last_ = false;
    }
    const double dw_start = delta_w;
    int wrong_no_dc = 0, nneg_seen = -1;
    for (int k = 0; k < 100; ++k) {
      int r2 = attempt(delta_w, delta_c);
      if (r2 == 0) { delta_w_last = delta_w; if (!always_dc_) dc_fixed_last_ = false; return true; }
      if (r2 == 2 && delta_c == 0.0) delta_c = dc_val;
      // Three growing delta_w that leave the SAME wrong number of negative pivots (negative curvature
      // would shed them one by one as delta_w grows; an abandoned attempt reports no usable count):
      // suspect the Jacobian and try the dual regularisation.
      if (r2 == 1 && delta_c == 0.0) {
        if (last_nneg_ < 1000000 && (nneg_seen < 0 || last_nneg_ == nneg_seen)) ++wrong_no_dc; else wrong_no_dc = 0;
        nneg_seen = last_nneg_ < 1000000 ? last_nneg_ : -1;
        if (wrong_no_dc >= 3) {
          int r3 = attempt(0.0, dc_val);
          if (r3 == 0) { delta_c = dc_val; delta_w = 0.0; dc_fixed(); return true; }
          r3 = attempt(dw_start, dc_val);
          if (r3 == 0) { delta_c = dc_val; delta_w = delta_w_last = dw_start; dc_fixed(); return true; }
          delta_c = dc_val;                       // keep it and go on growing delta_w
        } else if (delta_w > 1e20) {
          delta_c = dc_val;                       // backstop: nothing helped up to 1e20
          delta_w = dw_start;
          continue;
        }
      }
      // with a certified lower bound in hand the trial is already in the right decade: grow gently
      if (have_lb && dw_lb > 0.0 && delta_w <= 64.0 * dw_lb) delta_w *= 2.0;
      else delta_w = (delta_w_last == 0.0) ? kwpb * delta_w : kwp * delta_w;
      if (delta_w > dw_max) return false;
    }
    return false;
  }

  // out = rhsv - K v and, in the same pass, max |out| and max |v| (the refinement's residual, its norm and the
  // solution norm: one pass over [variables | constraint rows] after the three products instead of two maps and a reduction)
  DNLP_HD RMulti kkt_residual(const double* v, double dw, const double* rhsv, double* out) {
    DNLP_IPM_LDS();
    bool quasi = false;
    if constexpr (E::has_host_control) if (lm_on()) { lm_hess_mult(v, out); quasi = true; }
    if (!quasi) md_->hess_mult(v, out);
    md_->jac_tmult(jv, v + N, xt);
    md_->jac_mult(jv, v, tM);
    const double *sx = Sx, *jt = xt, *jx = tM, *dd = Dd, *fm = fixmask;
    const i64 NN = N;
    return ex_->template reduce_multi<2, 0>(N + m, [=] DNLP_HD(i64 k) -> RMulti {
      double kv;
      if (k < NN) kv = fm[k] != 0.0 ? v[k] : out[k] + (sx[k] + dw) * v[k] + jt[k];
      else kv = jx[k - NN] - dd[k - NN] * v[k];
      const double r = rhsv[k] - kv;
      out[k] = r;
      RMulti w;
      w.mx[0] = fabs(r); w.mx[1] = fabs(v[k]); w.mx[2] = w.mx[3] = 0.0;
      w.sm[0] = w.sm[1] = w.sm[2] = w.sm[3] = 0.0;
      return w; });
  }

  // solve K sol = rhs with iterative refinement on the unfactored operator
  DNLP_HD bool solve_refined(double dw) {
    DNLP_IPM_LDS();
    double t0 = now_sec();
    kkt_solve(rhs, sol);
    const double* rr = rhs;
    double rn = ex_->max(N + m, [=] DNLP_HD(i64 i) { return fabs(rr[i]); });
    double best = kInf;
    bool fresh = false;          // last_ratio_ already belongs to the current sol (loop left right after its residual)
    for (int it = 0; it < opt.max_refine; ++it) {
      const RMulti Rn = kkt_residual(sol, dw, rhs, res);
      double en = Rn.mx[0], sn = Rn.mx[1];
      // (IPOPT's ComputeResidualRatio caps the solution norm at 1e6 x the right-hand side's; tried here in round 3:
      //  near a stationary point of the barrier problem the right-hand side is tiny against a legitimate solution,
      //  every solve is then "singular" and delta_w runs away — 1569 iterations on one of 8192 localization
      //  instances — so the plain ratio stays)
      double ratio = en / (std::max(rn, 1e-300) + sn);
      if (!std::isfinite(en)) { stats.t_solve += now_sec() - t0; return false; }
      last_ratio_ = std::isfinite(ratio) ? ratio : kInf;
      fresh = true;
      if (it >= opt.min_refine && ratio <= 1e-10) break;
      if (en >= best * 0.999 && it >= opt.min_refine) break;   // no further progress
      best = std::min(best, en);
      kkt_solve(res, cor);
      double* sw = sol;
      const double* co = cor;
      ex_->map(N + m, [=] DNLP_HD(i64 i) { sw[i] += co[i]; });
      fresh = false;
    }
    if (!fresh) {
      const RMulti Rn = kkt_residual(sol, dw, rhs, res);
      const double en = Rn.mx[0], sn = Rn.mx[1];
      last_ratio_ = en / (std::max(rn, 1e-300) + sn);
      if (!std::isfinite(last_ratio_)) last_ratio_ = kInf;
    }
    stats.t_solve += now_sec() - t0;
    if (opt.print_level >= 7) logf("   solve: |rhs| %.3e residual ratio %.3e", rn, last_ratio_);
    return true;
  }

  // direction for barrier parameter muv with primal residual `pres` (rp or the SOC one)
  // `into`: the seven arrays the direction is written to (null: dx, ds, dy, dzL, dzU, dvL, dvU) — the mu oracle lets its
  // two solves write straight into aff[] / cen[] instead of copying seven arrays after each
  DNLP_HD bool compute_direction(double muv, const double* pres, double dw, bool centering = false, const V* into = nullptr) {
    DNLP_IPM_LDS();
    double* r = rhs;
    const double *rxx = rx, *q = rs, *sS = Ss, *eq = eqmask;
    const i64 NN = N;
    ex_->map(N + m, [=] DNLP_HD(i64 k) {
      if (k < NN) r[k] = -rxx[k];
      else { const i64 i = k - NN; r[k] = -pres[i] - (eq[i] == 0.0 ? q[i] / (sS[i] + dw) : 0.0); } });
    if (!solve_refined(dw)) return false;
    const double* so = sol;
    double *ddx = into ? static_cast<double*>(into[0]) : static_cast<double*>(dx), *dds = into ? static_cast<double*>(into[1]) : static_cast<double*>(ds),
           *ddy = into ? static_cast<double*>(into[2]) : static_cast<double*>(dy);
    // bound multiplier steps (WB eq. (12)); one pass over [variables | constraint rows]
    const double *l = xL, *u = xU, *sl = sL, *su = sU, *xx = x, *ss = s, *a = zL, *b = zU, *c = vL, *d = vU;
    double *da = into ? static_cast<double*>(into[3]) : static_cast<double*>(dzL), *db = into ? static_cast<double*>(into[4]) : static_cast<double*>(dzU),
           *dc = into ? static_cast<double*>(into[5]) : static_cast<double*>(dvL), *dd2 = into ? static_cast<double*>(into[6]) : static_cast<double*>(dvU);
    const double keep = centering ? 0.0 : 1.0;   // the centering direction has no "- z" term
    ex_->map(N + m, [=] DNLP_HD(i64 k) {
      if (k < NN) {
        const i64 j = k;
        const double dxj = so[j];
        ddx[j] = dxj;
        da[j] = (l[j] > -kInf) ? (muv - a[j] * dxj) / (xx[j] - l[j]) - keep * a[j] : 0.0;
        db[j] = (u[j] < kInf) ? (muv + b[j] * dxj) / (u[j] - xx[j]) - keep * b[j] : 0.0;
      } else {
        const i64 i = k - NN;
        const bool in = eq[i] == 0.0;
        const double dsi = in ? (so[NN + i] - q[i]) / (sS[i] + dw) : 0.0;
        ddy[i] = so[NN + i];
        dds[i] = dsi;
        dc[i] = (in && sl[i] > -kInf) ? (muv - c[i] * dsi) / (ss[i] - sl[i]) - keep * c[i] : 0.0;
        dd2[i] = (in && su[i] < kInf) ? (muv + d[i] * dsi) / (su[i] - ss[i]) - keep * d[i] : 0.0;
      } });
    return true;
  }

  // fraction-to-boundary step sizes (WB eq. (15))
  DNLP_HD double max_step_primal(double tauv) {
    DNLP_IPM_LDS();
    const double *l = xL, *u = xU, *sl = sL, *su = sU, *xx = x, *ss = s, *ddx = dx, *dds = ds, *eq = eqmask;
    double ax = ex_->min(N, [=] DNLP_HD(i64 j) {
      double a = 1.0;
      if (l[j] > -kInf && ddx[j] < 0.0) a = fmin(a, -tauv * (xx[j] - l[j]) / ddx[j]);
      if (u[j] < kInf && ddx[j] > 0.0) a = fmin(a, tauv * (u[j] - xx[j]) / ddx[j]);
      return a;
    });
    double as = m ? ex_->min(m, [=] DNLP_HD(i64 i) {
      double a = 1.0;
      if (eq[i] != 0.0) return a;
      if (sl[i] > -kInf && dds[i] < 0.0) a = fmin(a, -tauv * (ss[i] - sl[i]) / dds[i]);
      if (su[i] < kInf && dds[i] > 0.0) a = fmin(a, tauv * (su[i] - ss[i]) / dds[i]);
      return a;
    }) : 1.0;
    return std::min(1.0, std::min(ax, as));
  }
  DNLP_HD double max_step_dual(double tauv) {
    DNLP_IPM_LDS();
    const double *a = zL, *b = zU, *c = vL, *d = vU, *da = dzL, *db = dzU, *dc = dvL, *dd2 = dvU;
    double az = ex_->min(N, [=] DNLP_HD(i64 j) {
      double t = 1.0;
      if (da[j] < 0.0) t = fmin(t, -tauv * a[j] / da[j]);
      if (db[j] < 0.0) t = fmin(t, -tauv * b[j] / db[j]);
      return t;
    });
    double av = m ? ex_->min(m, [=] DNLP_HD(i64 i) {
      double t = 1.0;
      if (dc[i] < 0.0) t = fmin(t, -tauv * c[i] / dc[i]);
      if (dd2[i] < 0.0) t = fmin(t, -tauv * d[i] / dd2[i]);
      return t;
    }) : 1.0;
    return std::min(1.0, std::min(az, av));
  }

  // both fraction-to-boundary step sizes in one pass over [variables | constraint rows]
  DNLP_HD D2 max_steps(double tauv) {
    DNLP_IPM_LDS();
    const double *l = xL, *u = xU, *sl = sL, *su = sU, *xx = x, *ss = s, *ddx = dx, *dds = ds, *eq = eqmask;
    const double *a = zL, *b = zU, *c = vL, *d = vU, *da = dzL, *db = dzU, *dc = dvL, *dd2 = dvU;
    const i64 NN = N;
    const D2 r = ex_->min2(N + m, [=] DNLP_HD(i64 k) -> D2 {
      double tp = 1.0, td = 1.0;
      if (k < NN) {
        const i64 j = k;
        if (l[j] > -kInf && ddx[j] < 0.0) tp = fmin(tp, -tauv * (xx[j] - l[j]) / ddx[j]);
        if (u[j] < kInf && ddx[j] > 0.0) tp = fmin(tp, tauv * (u[j] - xx[j]) / ddx[j]);
        if (da[j] < 0.0) td = fmin(td, -tauv * a[j] / da[j]);
        if (db[j] < 0.0) td = fmin(td, -tauv * b[j] / db[j]);
      } else {
        const i64 i = k - NN;
        if (eq[i] == 0.0) {
          if (sl[i] > -kInf && dds[i] < 0.0) tp = fmin(tp, -tauv * (ss[i] - sl[i]) / dds[i]);
          if (su[i] < kInf && dds[i] > 0.0) tp = fmin(tp, tauv * (su[i] - ss[i]) / dds[i]);
        }
        if (dc[i] < 0.0) td = fmin(td, -tauv * c[i] / dc[i]);
        if (dd2[i] < 0.0) td = fmin(td, -tauv * d[i] / dd2[i]);
      }
      return D2{tp, td}; });
    return D2{std::min(1.0, r.first), std::min(1.0, r.second)};
  }

  DNLP_HD bool filter_ok(double th, double ph) const {
    const double gth = 1e-5, gph = 1e-8;
    for (int k = 0; k < nfilt; ++k)
      if (!(th <= (1.0 - gth) * filt_th[k] || ph <= filt_ph[k] - gph * filt_th[k])) return false;
    return true;
  }

  // ---- convergence tests (IPOPT OptimalityErrorConvergenceCheck) ----------------------
  DNLP_HD int check_convergence(const Err& e0) {
    DNLP_IPM_LDS();
    double unsc_du = e0.dual / sf;
    double unsc_pr = e0.primal_unscaled;          // computed in the same pass as the scaled norms (error())
    double unsc_co = e0.cmpl / sf;
    stats.inf_pr = unsc_pr; stats.inf_du = unsc_du; stats.cmpl = unsc_co; stats.nlp_error = e0.total;
    if (e0.total <= opt.tol && unsc_du <= opt.dual_inf_tol && unsc_pr <= opt.constr_viol_tol &&
        unsc_co <= opt.compl_inf_tol)
      return Solve_Succeeded;
    bool acc = e0.total <= opt.acceptable_tol && unsc_du <= opt.acceptable_dual_inf_tol &&
               unsc_pr <= opt.acceptable_constr_viol_tol && unsc_co <= opt.acceptable_compl_inf_tol;
    acceptable_count = acc ? acceptable_count + 1 : 0;
    if (opt.acceptable_iter > 0 && acceptable_count >= opt.acceptable_iter) return Solved_To_Acceptable_Level;
    return 99;
  }

  // ---- one interior-point iteration; returns IPOPT status or 99 to continue -----------
  DNLP_STEP_INLINE DNLP_HD int step() {
    DNLP_IPM_LDS();
    if (!initialized) return status = Internal_Error;
    // (the optimality error of this point was already computed for the log line that closed the
    // previous iteration: nine reductions saved per iteration)
    Err e0 = e_cached_valid_ ? e_cached_ : error(0.0);
    e_cached_valid_ = false;
    // convergence is tested before the limits, as IPOPT does: a point that converges exactly at
    // max_iter is a success
    int cv = check_convergence(e0);
    if (iter == 0 && !notify(e0, 0.0, 0.0, 0.0, 0.0, 0)) return status = User_Requested_Stop;
    if (cv != 99) return status = cv;
    if (iter >= opt.max_iter) return status = Maximum_Iterations_Exceeded;
    if (now_sec() - t_begin_ > opt.max_wall_time) return status = Maximum_WallTime_Exceeded;
    {
      const double* xx = x;
      double xm = ex_->max(N, [=] DNLP_HD(i64 j) { return fabs(xx[j]); });
      if (!(xm <= opt.diverging_iterates_tol)) return status = Diverging_Iterates;
    }
    // barrier parameter update: monotone / fixed-mode decisions need no linear algebra;
    // the free-mode oracle needs the factorisation and runs after it
    const bool want_oracle = update_mu(e0);
    // Hessian of the Lagrangian at (x, y)
    eval_hessian();
    if constexpr (E::has_host_control) if (lm_on()) lm_save_point();
    barrier_terms(mu);
    double dw = 0.0, dc = 0.0;
    if (!factor_with_inertia(dw, dc)) return status = Error_In_Step_Computation;
    stats.last_delta_w = dw;
    bool have_dir = false;
    if (want_oracle) have_dir = quality_function_mu(dw);
    if (!have_dir) {
      barrier_terms(mu);
      if (!compute_direction(mu, rp, dw)) return status = Error_In_Step_Computation;
    }
    // IPOPT's "pretend singular" safeguard (PDFullSpaceSolver, residual_ratio_singular = 1e-5):
    // a factorisation with the right inertia whose refined solution still has a large residual
    // is treated as singular and the system is perturbed (delta_c, then growing delta_w).
    for (int tries = 0; tries < 6 && last_ratio_ > 1e-5; ++tries) {
      if (dc == 0.0) dc = 1e-8 * std::pow(mu, 0.25);
      dw = (dw == 0.0) ? ((delta_w_last == 0.0) ? 1e-4 : std::max(1e-20, delta_w_last / 3.0)) : 8.0 * dw;
      int r = try_factor(dw, dc);
      while (r != 0 && dw < opt.max_hessian_perturbation) { dw *= 8.0; r = try_factor(dw, dc); }
      if (r != 0) return status = Error_In_Step_Computation;
      delta_w_last = dw;
      stats.last_delta_w = dw;
      barrier_terms(mu);
      if (!compute_direction(mu, rp, dw)) return status = Error_In_Step_Computation;
    }
    // ---- backtracking filter line search (WB Algorithm A, steps A-5) ----
    const D2 steps = max_steps(tau);
    double a_max = steps.first;
    double a_z = steps.second;
    const Measures mk = measures(f, g, x, s, mu);
    const double theta_k = mk.theta;
    const double phi_k = mk.phi;
    double gphid;   // directional derivative of the barrier function
    {
      const double *rxx = rx, *ddx = dx, *q = rs, *yy = y, *dds = ds, *eq = eqmask;
      // rx = grad_x phi + J^T y
      const double* jt = jty();
      const i64 NN = N;
      gphid = ex_->sum(N + m, [=] DNLP_HD(i64 k) {
        if (k < NN) return (rxx[k] - jt[k]) * ddx[k];
        const i64 i = k - NN;
        return eq[i] == 0.0 ? (q[i] + yy[i]) * dds[i] : 0.0; });
    }
    const double g_th = 1e-5, g_ph = 1e-8, dlt = 1.0, s_th = 1.1, s_ph = 2.3, eta = 1e-8, g_al = 0.05;
    double a_min;
    if (gphid < 0.0) {
      a_min = std::min(g_th, g_ph * theta_k / (-gphid));
      if (theta_k <= theta_min) a_min = std::min(a_min, dlt * std::pow(theta_k, s_th) / std::pow(-gphid, s_ph));
      a_min *= g_al;
    } else {
      a_min = g_al * g_th;
    }
    const double macheps = 2.220446049250313e-16;
    auto le = [&](double a, double b, double base) { return a - b <= 10.0 * macheps * std::fabs(base); };
    double alpha = a_max;
    bool accepted = false, ftype = false;
    int ls = 0;
    bool soc_tried = false;
    double th_t = 0.0, ph_t = 0.0, f_t = 0.0;
    while (true) {
      ++ls;
      trial_point(alpha);
      bool fin = eval_fg(xt, f_t, gt, false);
      if (fin) {
        const Measures mt = measures(f_t, gt, xt, st, mu);
        th_t = mt.theta;
        ph_t = mt.phi;
        fin = mt.chk == 0.0 && std::isfinite(th_t) && std::isfinite(ph_t);
      }
      if (fin && th_t <= theta_max && filter_ok(th_t, ph_t)) {
        bool sw = gphid < 0.0 && alpha * std::pow(-gphid, s_ph) > dlt * std::pow(theta_k, s_th);
        if (theta_k <= theta_min && sw) {
          if (le(ph_t, phi_k + eta * alpha * gphid, phi_k)) { accepted = true; ftype = true; }
        } else {
          if (le(th_t, (1.0 - g_th) * theta_k, theta_k) || le(ph_t, phi_k - g_ph * theta_k, phi_k)) accepted = true;
        }
      }
      if (accepted) break;
      // second-order correction on the first trial (WB section 2.4)
      if (ls == 1 && !soc_tried && fin && th_t >= theta_k && opt.max_soc > 0 && m > 0) {
        soc_tried = true;
        if (second_order_correction(alpha, dw, theta_k, phi_k, gphid, th_t, ph_t, f_t, ftype)) {
          accepted = true;
          a_z = max_step_dual(tau);
          break;
        }
        // restore the plain direction
        compute_direction(mu, rp, dw);
      }
      alpha *= 0.5;
      if (alpha < a_min || ls > 60) break;
    }
    double alpha_used = alpha;
    if (!accepted) {
      if (opt.restoration && restoration_phase(theta_k)) {
        // restoration produced a new (x, s) acceptable to the filter; multipliers reset
        if constexpr (E::has_host_control) if (lm_on()) lm_reset();
        ++iter;
        Err e = error(0.0);
        stats.iterations = iter;
        if (!notify(e, 0.0, 0.0, 0.0, 0.0, -1)) return status = User_Requested_Stop;
        return 99;
      }
      // tiny steps with tiny infeasibility: report the search-direction status
      const double* ddx = dx;
      double dn = ex_->max(N, [=] DNLP_HD(i64 j) { return fabs(ddx[j]); });
      if (dn < 1e-12 && theta_k < opt.constr_viol_tol) return status = Search_Direction_Becomes_Too_Small;
      // "infeasible" needs evidence: the restoration must have ended at a (numerically) stationary
      // point of the constraint violation that is still above constr_viol_tol; anything else is a
      // failure of the restoration itself, as IPOPT reports it
      return status = (resto_stationary_ && resto_theta_ > opt.constr_viol_tol) ? Infeasible_Problem_Detected
                                                                               : Restoration_Failed;
    }
    // filter augmentation (WB eq. (22)) unless f-type step with Armijo
    if (!ftype) filter_add((1.0 - g_th) * theta_k, phi_k - g_ph * theta_k);
    // accept the trial point
    const double* ddx = dx;
    double dnorm = ex_->max(N, [=] DNLP_HD(i64 j) { return fabs(ddx[j]); });
    accept_trial(alpha_used, a_z, f_t);
    if constexpr (E::has_host_control) if (lm_on()) lm_update();
    ++iter;
    stats.iterations = iter;
    // Stall guard (deviation, feeds the retry ladder): forty consecutive accepted steps that each keep
    // less than 1e-3 of the Newton step (the fraction-to-boundary rule pinning the iterate against its
    // bounds while the direction is huge) make no progress that max_iter could wait for — observed on one
    // of 8192 circle-packing instances in free-mu mode: alpha_pr = 1e-4, ||d|| ~ 500 for 3000 iterations,
    // while the monotone rung solves the same instance in 23.  Reported as IPOPT's tiny-step status.
    // IPOPT has no such rule, and long runs of short fraction-to-boundary steps are legitimate on badly
    // scaled problems: the guard is active only where a rung of the retry ladder can take the run over
    // (inside solve() with adaptive_fallback on; option stall_guard = yes / no overrides), and forty such
    // steps in a row end the run only if, over the whole streak, the violation did not fall by a tenth and
    // the objective did not fall by a hundredth (it oscillates by ~0.5 % in the runs this is for: a decrease
    // that small is not progress); a streak that did make progress starts over.
    const bool guard_on = opt.stall_guard == 1 ||
                          (opt.stall_guard < 0 && in_solve_ && opt.adaptive_fallback && ladder_rung_ < 2);
    if (guard_on && alpha_used <= kStallAlpha) {
      if (tiny_streak_ == 0) { streak_theta0_ = theta_k; streak_f0_ = f; }
      ++tiny_streak_;
      if (tiny_streak_ >= kStallSteps) {
        const bool progress = th_t < 0.9 * streak_theta0_ ||
                              (th_t <= streak_theta0_ && f_t < streak_f0_ - 1e-2 * fmax(1.0, fabs(streak_f0_)));
        if (progress) tiny_streak_ = 0;
      }
    } else {
      tiny_streak_ = 0;
    }
    if (tiny_streak_ >= kStallSteps) {
      logf("stall guard: %d accepted steps with alpha_pr <= %.0e and no progress in theta or f (iteration %d)", kStallSteps, kStallAlpha, iter);
      return status = Search_Direction_Becomes_Too_Small;
    }
    Err e = error(0.0);
    e_cached_ = e;
    e_cached_valid_ = true;
    if (!notify(e, dnorm, dw, a_z, alpha_used, ls)) return status = User_Requested_Stop;
    return 99;
  }

  DNLP_HD void trial_point(double alpha) {
    DNLP_IPM_LDS();
    double *a = xt, *b = st;
    const double *xx = x, *ss = s, *ddx = dx, *dds = ds, *eq = eqmask, *sl = sL;
    const i64 NN = N;
    ex_->map(N + m, [=] DNLP_HD(i64 k) {
      if (k < NN) a[k] = xx[k] + alpha * ddx[k];
      else { const i64 i = k - NN; b[i] = eq[i] != 0.0 ? sl[i] : ss[i] + alpha * dds[i]; } });
  }

  DNLP_HD void accept_trial(double alpha, double a_z, double f_new) {
    DNLP_IPM_LDS();
    double *xx = x, *ss = s, *yy = y, *a = zL, *b = zU, *c = vL, *d = vU;
    const double *nx = xt, *ns = st, *ddy = dy, *da = dzL, *db = dzU, *dc = dvL, *dd2 = dvU;
    double* gg = g;
    const double* gn = gt;
    const i64 NN = N;
    ex_->map(N + m, [=] DNLP_HD(i64 k) {
      if (k < NN) { const i64 j = k; xx[j] = nx[j]; a[j] += a_z * da[j]; b[j] += a_z * db[j]; }
      else { const i64 i = k - NN; ss[i] = ns[i]; yy[i] += alpha * ddy[i]; c[i] += a_z * dc[i]; d[i] += a_z * dd2[i]; gg[i] = gn[i]; } });
    f = f_new;
    // derivatives at the accepted point: the last sweep was the accepted trial only if no
    // later trial was evaluated, so sweep again (cheap relative to the factorisation)
    md_->sweep(x, false);
    eval_derivs_after_sweep();
    reset_bound_multipliers();
  }

  // WB eq. (16): keep z within [mu/(kS (x-l)), kS mu/(x-l)]
  DNLP_HD void reset_bound_multipliers() {
    DNLP_IPM_LDS();
    const double kS = 1e10, muv = mu;
    const double *l = xL, *u = xU, *sl = sL, *su = sU, *xx = x, *ss = s, *eq = eqmask;
    double *a = zL, *b = zU, *c = vL, *d = vU;
    const i64 NN = N;
    ex_->map(N + m, [=] DNLP_HD(i64 k) {
      if (k < NN) {
        const i64 j = k;
        if (l[j] > -kInf) { double t = xx[j] - l[j]; a[j] = fmax(fmin(a[j], kS * muv / t), muv / (kS * t)); }
        if (u[j] < kInf) { double t = u[j] - xx[j]; b[j] = fmax(fmin(b[j], kS * muv / t), muv / (kS * t)); }
        return;
      }
      const i64 i = k - NN;
      if (eq[i] != 0.0) return;
      if (sl[i] > -kInf) { double t = ss[i] - sl[i]; c[i] = fmax(fmin(c[i], kS * muv / t), muv / (kS * t)); }
      if (su[i] < kInf) { double t = su[i] - ss[i]; d[i] = fmax(fmin(d[i], kS * muv / t), muv / (kS * t)); }
    });
  }

  // WB section 2.4: up to max_soc corrections c_soc = alpha*c_soc + c(x + alpha d)
  DNLP_HD bool second_order_correction(double alpha, double dw, double theta_k, double phi_k, double gphid,
                               double& th_t, double& ph_t, double& f_t, bool& ftype) {
    const double k_soc = 0.99, g_th = 1e-5, g_ph = 1e-8, dlt = 1.0, s_th = 1.1, s_ph = 2.3, eta = 1e-8;
    const double macheps = 2.220446049250313e-16;
    auto le = [&](double a, double b, double base) { return a - b <= 10.0 * macheps * std::fabs(base); };
    double* cs = csoc;
    const double *p = rp, *gg = gt, *ss = st, *eq = eqmask, *sl = sL;
    // c_soc = alpha * c(x_k) + c(x_k + alpha d)
    ex_->map(m, [=] DNLP_HD(i64 i) { cs[i] = alpha * p[i] + (eq[i] != 0.0 ? gg[i] - sl[i] : gg[i] - ss[i]); });
    double th_old = th_t;
    for (int k = 0; k < opt.max_soc; ++k) {
      if (!compute_direction(mu, csoc, dw)) return false;
      double a_soc = max_step_primal(tau);
      trial_point(a_soc);
      double fv;
      if (!eval_fg(xt, fv, gt)) return false;
      double th = theta_at(gt, st), ph = barrier_at(fv, xt, st, mu);
      if (!std::isfinite(th) || !std::isfinite(ph)) return false;
      if (th <= theta_max && filter_ok(th, ph)) {
        bool sw = gphid < 0.0 && alpha * std::pow(-gphid, s_ph) > dlt * std::pow(theta_k, s_th);
        bool ok = false;
        if (theta_k <= theta_min && sw) {
          if (le(ph, phi_k + eta * alpha * gphid, phi_k)) { ok = true; ftype = true; }
        } else if (le(th, (1.0 - g_th) * theta_k, theta_k) || le(ph, phi_k - g_ph * theta_k, phi_k)) {
          ok = true;
        }
        if (ok) { th_t = th; ph_t = ph; f_t = fv; return true; }
      }
      if (th > k_soc * th_old) return false;
      th_old = th;
      const double *gg2 = gt, *ss2 = st;
      ex_->map(m, [=] DNLP_HD(i64 i) { cs[i] = a_soc * cs[i] + (eq[i] != 0.0 ? gg2[i] - sl[i] : gg2[i] - ss2[i]); });
    }
    return false;
  }

  // ---- barrier parameter strategies --------------------------------------------------
  DNLP_HD double avg_complementarity() {
    DNLP_IPM_LDS();
    const double *l = xL, *u = xU, *sl = sL, *su = sU, *xx = x, *ss = s, *a = zL, *b = zU, *c = vL, *d = vU, *eq = eqmask;
    i64 nb = n_bound_mults();
    if (nb == 0) return 0.0;
    const i64 NN = N;
    const double tot = ex_->sum(N + m, [=] DNLP_HD(i64 k) {
      double v = 0.0;
      if (k < NN) {
        const i64 j = k;
        if (l[j] > -kInf) v += (xx[j] - l[j]) * a[j];
        if (u[j] < kInf) v += (u[j] - xx[j]) * b[j];
      } else {
        const i64 i = k - NN;
        if (eq[i] != 0.0) return v;
        if (sl[i] > -kInf) v += (ss[i] - sl[i]) * c[i];
        if (su[i] < kInf) v += (su[i] - ss[i]) * d[i];
      }
      return v; });
    return tot / static_cast<double>(nb);
  }

  // Smallest barrier parameter.  IPOPT's monotone update never goes below min(tol, compl_inf_tol) /
  // (barrier_tol_factor + 1) (IpMonotoneMuUpdate); its adaptive update uses mu_min — 1e-11 by default, lowered to
  // half of min(tol, compl_inf_tol) when that is smaller — in the free AND in the fixed mode (IpAdaptiveMuUpdate):
  // the published logs of the adaptive default end at lg(mu) = -11 (phase_retrieval.ipynb:112-115), which a floor of
  // tol / 11 cannot reach — rounds 1-2 applied the monotone floor to both, and solutions stopped at an objective
  // error of (active bounds) x 9e-9 where IPOPT's stop at (active bounds) x 1e-11.
  DNLP_HD double mu_floor_now() const {
    const double t = std::min(opt.tol, opt.compl_inf_tol);
    if (opt.mu_strategy == 0) return std::max(opt.mu_min, t / 11.0);
    return std::min(opt.mu_min, 0.5 * t);
  }

  DNLP_HD void monotone_update() {
    DNLP_IPM_LDS();
    const double k_eps = 10.0, k_mu = 0.2, th_mu = 1.5;
    const double mu_floor = mu_floor_now();
    for (int k = 0; k < 50; ++k) {
      Err e = error(mu);
      if (e.total <= k_eps * mu && mu > mu_floor) {
        double nm = std::max(mu_floor, std::min(k_mu * mu, std::pow(mu, th_mu)));
        if (nm >= mu) break;
        mu = nm;
        tau = std::max(0.99, 1.0 - mu);
        filter_clear();
      } else {
        break;
      }
    }
  }

  // Adaptive strategy (IPOPT mu_strategy=adaptive, the reference's default) in its
  // documented LOQO-oracle form with the kkt-error globalisation: free mode picks
  // mu = sigma * avg_compl with the LOQO centrality heuristic; if the KKT error fails to
  // decrease by the factor 0.9999 relative to the last 4 free-mode iterates the algorithm
  // falls back to the monotone mode until it does.
  // Adaptive strategy (IPOPT mu_strategy=adaptive, the reference's default): free mode chooses
  // mu with the quality-function oracle (IPOPT's default mu_oracle; Nocedal, Waechter, Waltz,
  // "Adaptive barrier update strategies for nonlinear interior methods", SIOPT 19(4), 2009);
  // globalisation = kkt-error: if the KKT error fails to decrease by the factor 0.9999
  // relative to the last 4 free-mode iterates the algorithm runs the monotone (fixed) mode,
  // started at 0.8 * average complementarity, until it does.
  // Returns true when the free-mode oracle must be run after the factorisation.
  DNLP_HD void hist_push(double v) {
    DNLP_IPM_LDS();
    if (n_hist == 4) { for (int k = 1; k < 4; ++k) kkt_hist[k - 1] = kkt_hist[k]; --n_hist; }
    kkt_hist[n_hist++] = v;
  }
  DNLP_HD bool update_mu(const Err& e0) {
    DNLP_IPM_LDS();
    if (n_bound_mults() == 0) { tau = 0.99; return false; }   // no barrier terms at all
    if (opt.mu_strategy == 0) { monotone_update(); return false; }
    const double mu_floor = mu_floor_now();
    double kkt = e0.dual + e0.primal + e0.cmpl;
    if (!fixed_mode) {
      bool ok = n_hist == 0;
      for (int k = 0; k < n_hist; ++k) if (kkt <= 0.9999 * kkt_hist[k]) ok = true;
      if (ok) {
        hist_push(kkt);
      } else {
        fixed_mode = true;
        mu = std::max(mu_floor, std::min(0.8 * avg_complementarity(), 1e5));
        tau = std::max(0.99, 1.0 - mu);
        filter_clear();
      }
    } else {
      bool ok = false;
      for (int k = 0; k < n_hist; ++k) if (kkt <= 0.9999 * kkt_hist[k]) ok = true;
      if (ok || n_hist == 0) {
        fixed_mode = false;
        hist_push(kkt);
      }
    }
    if (fixed_mode) { monotone_update(); return false; }
    return true;
  }

  // Quality-function oracle.  With the KKT matrix factored: affine-scaling direction (mu = 0)
  // and centering direction (unit mu) -> direction(mu) = aff + mu * cen; the quality function
  // is the linearised KKT error after the fraction-to-boundary step; golden section in
  // log(sigma), mu = sigma * average complementarity.  On success the search direction for the
  // chosen mu is already in dx..dvU and rx/rs hold the residuals for that mu.
  DNLP_HD bool quality_function_mu(double dw) {
    DNLP_IPM_LDS();
    const double avg = avg_complementarity();
    const i64 nb = n_bound_mults();
    if (!(avg > 0.0) || nb == 0) return false;
    const double mu_floor = mu_floor_now();
    barrier_terms(0.0);
    const double *rxx = rx, *rss = rs, *rpp = rp;
    const i64 NN0 = N;
    const RMulti R2 = ex_->template reduce_multi<0, 2>(N + m, [=] DNLP_HD(i64 k) -> RMulti {
      RMulti v;
      v.mx[0] = v.mx[1] = v.mx[2] = v.mx[3] = 0.0;
      v.sm[0] = v.sm[1] = v.sm[2] = v.sm[3] = 0.0;
      if (k < NN0) v.sm[0] = rxx[k] * rxx[k];
      else { const i64 i = k - NN0; v.sm[0] = rss[i] * rss[i]; v.sm[1] = rpp[i] * rpp[i]; }
      return v; });
    const double nd2 = R2.sm[0], np2 = m ? R2.sm[1] : 0.0;
    if (!compute_direction(0.0, rp, dw, false, aff)) return false;
    // a poorly solved affine system must not hide behind a well solved centring system (the step is
    // aff + mu cen, mu tiny: its accuracy is the affine solve's): the "pretend singular" test that follows
    // the oracle sees the worse of the two residual ratios
    const double ratio_aff = last_ratio_;
    {
      // centering right-hand side: derivative of the barrier terms w.r.t. mu
      double *r = rx, *q = rs;
      const double *l = xL, *u = xU, *sl = sL, *su = sU, *eq = eqmask, *xx = x, *ss = s, *fm = fixmask;
      const double kd = opt.kappa_d;
      const i64 NN1 = N;
      ex_->map(N + m, [=] DNLP_HD(i64 k) {
        if (k < NN1) {
          const i64 j = k;
          double c = 0.0;
          const bool hl = l[j] > -kInf, hu = u[j] < kInf;
          if (hl) c -= 1.0 / (xx[j] - l[j]);
          if (hu) c += 1.0 / (u[j] - xx[j]);
          if (hl && !hu) c += kd;
          if (hu && !hl) c -= kd;
          r[j] = fm[j] != 0.0 ? 0.0 : c;
          return;
        }
        const i64 i = k - NN1;
        double c = 0.0;
        if (eq[i] == 0.0) {
          const bool hl = sl[i] > -kInf, hu = su[i] < kInf;
          if (hl) c -= 1.0 / (ss[i] - sl[i]);
          if (hu) c += 1.0 / (su[i] - ss[i]);
          if (hl && !hu) c += kd;
          if (hu && !hl) c -= kd;
        }
        q[i] = c;
      });
    }
    if (!compute_direction(1.0, zeroM, dw, true, cen)) return false;
    if (ratio_aff > last_ratio_) last_ratio_ = ratio_aff;
    const i64 n_ineq = m - n_eq_;                      // (counted once in begin())
    const double n_dual = static_cast<double>(N + n_ineq), n_pri = static_cast<double>(m > 0 ? m : 1);
    const double *ax = aff[0], *as = aff[1], *aa = aff[3], *ab = aff[4], *ac = aff[5], *ad = aff[6];
    const double *cx = cen[0], *cs = cen[1], *ca = cen[3], *cb = cen[4], *cc = cen[5], *cd = cen[6];
    const double *l = xL, *u = xU, *sl = sL, *su = sU, *eq = eqmask, *xx = x, *ss = s, *a = zL, *b = zU,
                 *c = vL, *d = vU;
    auto qf = [&](double sigma) -> double {
      const double mus = sigma * avg;
      const double tv = std::max(0.99, 1.0 - mus);
      // ONE pass for both fraction-to-boundary step sizes and one for the complementarity term, over
      // the concatenated index range [variables | constraint rows] (six reductions per evaluation
      // before; an evaluation runs ~17 times per iteration and, on the host-driven space, every
      // reduction is a launch plus a host round trip)
      const i64 NN = N;
      const D2 al = ex_->min2(N + m, [=] DNLP_HD(i64 k) -> D2 {
        double tp = 1.0, td = 1.0;
        if (k < NN) {
          const i64 j = k;
          const double dxx = ax[j] + mus * cx[j];
          if (l[j] > -kInf && dxx < 0.0) tp = fmin(tp, -tv * (xx[j] - l[j]) / dxx);
          if (u[j] < kInf && dxx > 0.0) tp = fmin(tp, tv * (u[j] - xx[j]) / dxx);
          const double da = aa[j] + mus * ca[j], db = ab[j] + mus * cb[j];
          if (da < 0.0) td = fmin(td, -tv * a[j] / da);
          if (db < 0.0) td = fmin(td, -tv * b[j] / db);
        } else {
          const i64 i = k - NN;
          if (eq[i] == 0.0) {
            const double dss = as[i] + mus * cs[i];
            if (sl[i] > -kInf && dss < 0.0) tp = fmin(tp, -tv * (ss[i] - sl[i]) / dss);
            if (su[i] < kInf && dss > 0.0) tp = fmin(tp, tv * (su[i] - ss[i]) / dss);
          }
          const double dc = ac[i] + mus * cc[i], dd2 = ad[i] + mus * cd[i];
          if (dc < 0.0) td = fmin(td, -tv * c[i] / dc);
          if (dd2 < 0.0) td = fmin(td, -tv * d[i] / dd2);
        }
        return D2{tp, td}; });
      const double apv = std::min(1.0, al.first), adv = std::min(1.0, al.second);
      const double comp = ex_->sum(N + m, [=] DNLP_HD(i64 k) {
        double v = 0.0;
        if (k < NN) {
          const i64 j = k;
          const double dxx = ax[j] + mus * cx[j];
          if (l[j] > -kInf) { const double t = (xx[j] - l[j] + apv * dxx) * (a[j] + adv * (aa[j] + mus * ca[j])); v += t * t; }
          if (u[j] < kInf) { const double t = (u[j] - xx[j] - apv * dxx) * (b[j] + adv * (ab[j] + mus * cb[j])); v += t * t; }
        } else {
          const i64 i = k - NN;
          if (eq[i] != 0.0) return v;
          const double dss = as[i] + mus * cs[i];
          if (sl[i] > -kInf) { const double t = (ss[i] - sl[i] + apv * dss) * (c[i] + adv * (ac[i] + mus * cc[i])); v += t * t; }
          if (su[i] < kInf) { const double t = (su[i] - ss[i] - apv * dss) * (d[i] + adv * (ad[i] + mus * cd[i])); v += t * t; }
        }
        return v; });
      return (1.0 - adv) * (1.0 - adv) * nd2 / n_dual + (1.0 - apv) * (1.0 - apv) * np2 / n_pri +
             comp / static_cast<double>(nb);
    };
    // search interval (IPOPT: sigma_min 1e-6, sigma_max 1e2, 8 section steps, tol 1e-2)
    const double mu_max = opt.mu_max_fact * avg;
    double s_lo = std::max(1e-6, mu_floor / avg), s_up = std::min(1e2, mu_max / avg);
    double sigma;
    if (s_lo >= s_up) {
      sigma = s_lo;
    } else {
      const double q1 = qf(1.0), s1m = 1.0 - 1e-2, q1m = qf(std::max(s_lo, s1m));
      double lo, up;
      if (q1m > q1 && s_up > 1.0) { lo = 1.0; up = s_up; } else { lo = s_lo; up = std::min(std::max(s_lo, s1m), s_up); }
      // golden section in log(sigma) over [lo, up] (IPOPT's search), then IPOPT's end-point check
      const double gr = 0.5 * (3.0 - std::sqrt(5.0));
      double fsel = 0.0;
      bool endpoint = false;
      auto section = [&](double slo, double sup) {
        double la = std::log(slo), lb = std::log(std::max(sup, slo * (1 + 1e-12)));
        double m1 = la + gr * (lb - la), m2 = lb - gr * (lb - la);
        double f1 = qf(std::exp(m1)), f2 = qf(std::exp(m2));
        for (int it = 0; it < 8 && (lb - la) > 1e-2 * std::fabs(lb) + 1e-12; ++it) {
          if (f1 > f2) { la = m1; m1 = m2; f1 = f2; m2 = lb - gr * (lb - la); f2 = qf(std::exp(m2)); }
          else { lb = m2; m2 = m1; f2 = f1; m1 = la + gr * (lb - la); f1 = qf(std::exp(m1)); }
        }
        double sg = std::exp(f1 < f2 ? m1 : m2);
        fsel = std::min(f1, f2);
        const double qlo = qf(slo), qup = qf(sup);
        endpoint = false;
        if (qlo < fsel && qlo <= qup) { sg = slo; fsel = qlo; endpoint = true; }
        else if (qup < fsel) { sg = sup; fsel = qup; endpoint = true; }
        return sg;
      };
      sigma = section(lo, up);
      // An end point that beats the section's interior result means the quality function is not
      // unimodal on [lo, up] (observed on AC power flow: q(1e-6) = 655, q(1e-3) = 837, q(0.1) = 5.9,
      // q(0.99) = 309: the section settles in the left basin, the check returns sigma = 0.99 and mu
      // shrinks by 1 % per iteration for hundreds of iterations).  Only then (deviation from IPOPT's
      // search, which stops here): a coarse log-spaced scan brackets the best basin and the section
      // is repeated inside that bracket.
      if (endpoint && up > lo * 10.0) {
        const double grid[6] = {lo, 1e-4, 1e-2, 1e-1, 0.5, up};
        double gs[6], gq[6];
        int ng = 0;
        for (int k = 0; k < 6; ++k) {
          if (grid[k] < lo || grid[k] > up || (ng > 0 && grid[k] <= gs[ng - 1])) continue;
          gs[ng] = grid[k]; gq[ng] = qf(grid[k]); ++ng;
        }
        int best = 0;
        for (int k = 1; k < ng; ++k) if (gq[k] < gq[best]) best = k;
        if (gq[best] < fsel && best > 0 && best + 1 < ng) {
          const double f0 = fsel, s0 = sigma;
          sigma = section(gs[best - 1], gs[best + 1]);
          if (!(fsel < f0)) sigma = s0;
        }
      }
    }
    double nm = std::max(mu_floor, std::min(sigma * avg, mu_max));
    if (!std::isfinite(nm)) return false;
    mu = nm;
    tau = std::max(0.99, 1.0 - mu);
    filter_clear();                   // free mode: every iteration is a new barrier problem
    barrier_terms(mu);                // residuals for the chosen mu (line search, SOC)
    const double muv = mu;
    {
      // direction(mu) = aff + mu cen, all seven arrays in one pass over [variables | constraint rows]
      double *o0 = dx, *o1 = ds, *o2 = dy, *o3 = dzL, *o4 = dzU, *o5 = dvL, *o6 = dvU;
      const double *ay = aff[2], *cy = cen[2];
      const i64 NN2 = N;
      ex_->map(N + m, [=] DNLP_HD(i64 k) {
        if (k < NN2) {
          o0[k] = ax[k] + muv * cx[k]; o3[k] = aa[k] + muv * ca[k]; o4[k] = ab[k] + muv * cb[k];
        } else {
          const i64 i = k - NN2;
          o1[i] = as[i] + muv * cs[i]; o2[i] = ay[i] + muv * cy[i]; o5[i] = ac[i] + muv * cc[i]; o6[i] = ad[i] + muv * cd[i];
        } });
    }
    return true;
  }

  // ---- feasibility restoration (simplified form of WB section 3.3) ---------------------
  // Minimises the constraint violation with damped Gauss-Newton steps in (x, s) that keep
  // the iterate strictly inside its bounds, until the point is acceptable to the filter and
  // reduces the violation by the factor 0.9.
  DNLP_HD bool restoration_phase(double theta_k) {
    DNLP_IPM_LDS();
    const double phi_k = barrier_at(f, x, s, mu);
    filter_add((1.0 - 1e-5) * theta_k, phi_k - 1e-8 * theta_k);
    double th_cur = theta_k;
    double zeta = std::sqrt(mu);
    resto_stationary_ = false;
    resto_theta_ = theta_k;
    for (int it = 0; it < 100; ++it) {
      // Gauss-Newton system: [zeta*I  J^T; J  -(1 + [ineq]... )]  on the residual r = c(x, s)
      double *sx = Sx, *dd = Dd;
      const double* eq = eqmask;
      const double zz = zeta;
      ex_->map(N, [=] DNLP_HD(i64 j) { sx[j] = zz; });
      // minimise 1/2|r + J dx - ds|^2 + zeta/2 |dx|^2 + zeta/2 |ds|^2  ->  D = 1 (+ 1/zeta for slacks)
      ex_->map(m, [=] DNLP_HD(i64 i) { dd[i] = 1.0 + (eq[i] == 0.0 ? 1.0 / zz : 0.0); });
      md_->clear_dense_w();
      ex_->zero(md_->Hs, sizeof(double) * static_cast<size_t>(md_->t.nnzH));
      int nneg = 0, nzero = 0;
      if (!kkt_->assemble_factor(*md_, jv, Sx, Dd, fixmask, 0.0, &nneg, &nzero)) return false;
      stats.factorizations++;
      double* r = rhs;
      const double *gg = g, *ss = s, *sl = sL;
      const i64 NN = N;
      ex_->map(N, [=] DNLP_HD(i64 j) { r[j] = 0.0; });
      ex_->map(m, [=] DNLP_HD(i64 i) { r[NN + i] = -(eq[i] != 0.0 ? gg[i] - sl[i] : gg[i] - ss[i]); });
      kkt_->solve(rhs, sol);
      const double* so = sol;
      double *ddx = dx, *dds = ds;
      ex_->map(N, [=] DNLP_HD(i64 j) { ddx[j] = so[j]; });
      // ds = -lambda/zeta with lambda = so[N+i]:  from zeta ds - (-lambda) ... ds = so[N+i]/zeta
      ex_->map(m, [=] DNLP_HD(i64 i) { dds[i] = eq[i] == 0.0 ? so[NN + i] / zz : 0.0; });
      double a = max_step_primal(tau);
      bool moved = false;
      for (int bt = 0; bt < 30; ++bt) {
        trial_point(a);
        double fv;
        if (eval_fg(xt, fv, gt)) {
          double th = theta_at(gt, st);
          if (std::isfinite(th) && th < (1.0 - 1e-4 * a) * th_cur) {
            ex_->d2d(x, xt, sizeof(double) * static_cast<size_t>(N));
            ex_->d2d(s, st, sizeof(double) * static_cast<size_t>(m));
            ex_->d2d(g, gt, sizeof(double) * static_cast<size_t>(m));
            f = fv;
            th_cur = th;
            moved = true;
            break;
          }
        }
        a *= 0.5;
      }
      // no damped Gauss-Newton step of any length reduces the violation: a stationary point of it
      if (!moved) { zeta *= 10.0; if (zeta > 1e8) { resto_stationary_ = true; resto_theta_ = th_cur; return false; } continue; }
      resto_theta_ = th_cur;
      md_->sweep(x, false);
      eval_derivs_after_sweep();
      double ph = barrier_at(f, x, s, mu);
      if (th_cur <= 0.9 * theta_k && th_cur <= theta_max && filter_ok(th_cur, ph)) {
        // reset multipliers as IPOPT does after restoration
        ex_->zero(y, sizeof(double) * static_cast<size_t>(m));
        jty_valid_ = false;
        const double zi = 1.0;
        const double *l = xL, *u = xU, *su = sU;
        double *za = zL, *zb = zU, *c = vL, *d = vU;
        ex_->map(N, [=] DNLP_HD(i64 j) { za[j] = (l[j] > -kInf) ? zi : 0.0; zb[j] = (u[j] < kInf) ? zi : 0.0; });
        ex_->map(m, [=] DNLP_HD(i64 i) {
          c[i] = (eq[i] == 0.0 && sl[i] > -kInf) ? zi : 0.0;
          d[i] = (eq[i] == 0.0 && su[i] < kInf) ? zi : 0.0; });
        if (m > 0) init_multipliers_ls();
        return true;
      }
      if (th_cur < 1e-13) return false;
    }
    return false;
  }

  // A rung of the retry ladder that converged did so in monotone mode, whose barrier floor (tol / 10)
  // leaves the point a factor ~1e3 less accurate than the free mode normally ends (its last Newton
  // steps overshoot the tolerance).  A few more iterations toward tol / 1000 make up for that; if they
  // do not get there the better of the two points that satisfies the requested tolerance is kept.
  DNLP_HD void polish() {
    DNLP_IPM_LDS();
    double* sv[7] = {x, s, y, zL, zU, vL, vU};
    const i64 sz[7] = {N, m, m, N, N, m, m};
    for (int k = 0; k < 7; ++k) ex_->d2d(aff[k], sv[k], sizeof(double) * static_cast<size_t>(sz[k]));   // aff: free in monotone mode
    const double tol0 = opt.tol, mu0 = mu, tau0 = tau;
    const int maxit0 = opt.max_iter;
    opt.tol = tol0 * 1e-3;
    opt.max_iter = iter + 12;
    while (step() == 99) {}
    opt.tol = tol0;
    opt.max_iter = maxit0;
    if (status == Solve_Succeeded) return;
    const Err e = error(0.0);
    if (check_convergence(e) == Solve_Succeeded) { status = Solve_Succeeded; return; }
    for (int k = 0; k < 7; ++k) ex_->d2d(sv[k], aff[k], sizeof(double) * static_cast<size_t>(sz[k]));
    mu = mu0; tau = tau0;
    // objective, constraint values, derivatives and the reported infeasibilities all refer to the
    // restored point, not to the abandoned polish iterate
    eval_fg(x, f, g);
    eval_derivs_after_sweep();
    e_cached_valid_ = false;
    acceptable_count = 0;
    check_convergence(error(0.0));
    status = Solve_Succeeded;
  }

  // ---- driver -------------------------------------------------------------------------
  DNLP_HD int solve(const double* x0_ctl) {
    DNLP_IPM_LDS();
    const double t_all = now_sec();
    in_solve_ = true;
    int rc = begin(x0_ctl);
    if (rc != 0) { in_solve_ = false; return rc; }
    while (true) {
      int r = step();
      if (r != 99) break;
    }
    // Robustness ladder (documented deviation): a run that ends in a step / restoration failure,
    // an infeasibility verdict or diverging iterates is repeated from the same start, first in
    // monotone mode (when it was adaptive), then in monotone mode with a ten times larger initial
    // barrier (>= 1): nonconvex canonical forms whose objective is unbounded off the feasible
    // set are path-sensitive, and a stronger barrier keeps the early iterates centred.
    if (opt.adaptive_fallback) {
      const int strategy0 = opt.mu_strategy;
      const double mu_init0 = opt.mu_init;
      const int max_iter0 = opt.max_iter;
      const double max_wall0 = opt.max_wall_time;
      for (int rung = 1; rung <= 2; ++rung) {
        const bool failed = status == Infeasible_Problem_Detected || status == Restoration_Failed ||
                            status == Error_In_Step_Computation || status == Search_Direction_Becomes_Too_Small ||
                            status == Diverging_Iterates;
        if (!failed) break;
        if (rung == 1 && strategy0 != 1) continue;       // already monotone: straight to rung 2
        const int it_first = iter;
        // the iteration and wall-clock budgets are the caller's for the whole solve, not per rung
        if (it_first >= max_iter0) { status = Maximum_Iterations_Exceeded; break; }
        if (now_sec() - t_all > max_wall0) { status = Maximum_WallTime_Exceeded; break; }
        opt.max_iter = max_iter0 - it_first;
        opt.max_wall_time = max_wall0 - (now_sec() - t_all);
        ladder_rung_ = rung;
        opt.mu_strategy = 0;
        if (rung == 2) opt.mu_init = mu_init0 * 10.0 > 1.0 ? mu_init0 * 10.0 : 1.0;
        logf("run ended with status %d after %d iterations: restarting in monotone mode, mu_init %.1e", status, iter,
             opt.mu_init);
        typename E::Log keep = iterlog;
        rc = begin(x0_ctl);
        if (rc == 0) {
          while (true) {
            int r = step();
            if (r != 99) break;
          }
        }
        if (status == Solve_Succeeded) polish();
        iter += it_first;
        stats.iterations = iter;
        keep.append(iterlog);
        iterlog = keep;
      }
      opt.mu_strategy = strategy0;
      opt.mu_init = mu_init0;
      opt.max_iter = max_iter0;
      opt.max_wall_time = max_wall0;
      ladder_rung_ = 0;
    }
    in_solve_ = false;
    stats.wall = now_sec() - t_all;
    stats.final_mu = mu;
    if constexpr (E::has_host_control)
      if (lm_on()) logf("limited-memory quasi-Newton: %d pairs accepted, %d skipped, %d history resets; no second derivatives evaluated",
                        lm_->updates, lm_->skips_total, lm_->resets);
    return status;
  }

  // unscaled results into exec-space arrays (any may be null); `res` / `cor` are free between steps
  DNLP_HD void extract_exec(double* xo, double* mult_g, double* mult_xL, double* mult_xU, double* gout) {
    DNLP_IPM_LDS();
    const double sff = sf;
    const double *xx = x, *yy = y, *sgp = sg, *a = zL, *b = zU, *gg = g;
    if (xo) ex_->map(N, [=] DNLP_HD(i64 j) { xo[j] = xx[j]; });
    if (mult_g) ex_->map(m, [=] DNLP_HD(i64 i) { mult_g[i] = yy[i] * sgp[i] / sff; });
    if (gout) ex_->map(m, [=] DNLP_HD(i64 i) { gout[i] = gg[i] / sgp[i]; });
    if (mult_xL) ex_->map(N, [=] DNLP_HD(i64 j) { mult_xL[j] = a[j] / sff; });
    if (mult_xU) ex_->map(N, [=] DNLP_HD(i64 j) { mult_xU[j] = b[j] / sff; });
  }
  DNLP_HD double objective_unscaled() const { return f / sf; }

  // unscaled results to control-space buffers (any may be null)
  DNLP_HD void extract(double* xo, double* obj, double* mult_g, double* mult_xL, double* mult_xU, double* gout) {
    DNLP_IPM_LDS();
    if (xo) ex_->d2h(xo, x, sizeof(double) * static_cast<size_t>(N));
    if (obj) *obj = objective_unscaled();
    extract_exec(nullptr, mult_g ? res : nullptr, mult_xL ? cor : nullptr, nullptr, gout ? cor + N : nullptr);
    if (mult_g) ex_->d2h(mult_g, res, sizeof(double) * static_cast<size_t>(m));
    if (gout) ex_->d2h(gout, cor + N, sizeof(double) * static_cast<size_t>(m));
    if (mult_xL) ex_->d2h(mult_xL, cor, sizeof(double) * static_cast<size_t>(N));
    if (mult_xU) {
      extract_exec(nullptr, nullptr, nullptr, cor, nullptr);
      ex_->d2h(mult_xU, cor, sizeof(double) * static_cast<size_t>(N));
    }
  }

  // iteration log line + the user's intermediate callback; false = the user asked to stop
  DNLP_HD bool notify(const Err& e, double dnorm, double dw, double a_du, double a_pr, int ls) {
    DNLP_IPM_LDS();
    log_iter(e, dnorm, dw, a_du, a_pr, ls);
#if !DNLP_DEVICE_PASS
    if constexpr (E::has_host_control) {
      if (intermediate_cb)
        return intermediate_cb(ls < 0 ? 1 : 0, iter, f / sf, e.primal, e.dual, mu, dnorm, dw, a_du, a_pr, ls < 0 ? 0 : ls,
                               intermediate_user) != 0;
    }
#endif
    return true;
  }

  DNLP_HD void log_iter(const Err& e, double dnorm, double dw, double a_du, double a_pr, int ls) {
    DNLP_IPM_LDS();
#if !DNLP_DEVICE_PASS
    if constexpr (E::has_log) {
      char rg[16];
      if (dw > 0) std::snprintf(rg, sizeof rg, "%5.1f", std::log10(dw)); else std::snprintf(rg, sizeof rg, "    -");
      logf("%4d %14.7e %8.2e %8.2e %5.1f %8.2e %s %8.2e %8.2e %3d%s", iter, f / sf, e.primal, e.dual,
           std::log10(std::max(mu, 1e-300)), dnorm, rg, a_du, a_pr, ls, ls < 0 ? "r" : "");
    }
#else
    (void)e; (void)dnorm; (void)dw; (void)a_du; (void)a_pr; (void)ls;
#endif
  }

  V fixmask;

 private:
  E* ex_;
  Model<E>* md_;
  K* kkt_;
  i64 nb_cache_ = -1;
  double last_ratio_ = 0.0;
  bool delta_w_used_last_iter_ = false;
  int sparse_singular_streak_ = 0;
  Err e_cached_;                    // optimality error of the current point (valid between two step() calls)
  bool e_cached_valid_ = false;
  int last_nneg_ = 0;               // negative pivots reported by the last factorisation attempt
  int ladder_rung_ = 0;             // 0: first run; 1, 2: rungs of the retry ladder
  static constexpr double kStallAlpha = 1e-2;   // stall guard: a step that keeps at most this much of the Newton step ...
  static constexpr int kStallSteps = 30;        // ... this many times in a row, without progress (see step())
  bool jty_valid_ = false;          // tN holds J^T y of the current Jacobian values and multipliers (jty())
  int tiny_streak_ = 0;             // consecutive accepted steps with alpha_pr <= kStallAlpha (stall guard)
  double streak_theta0_ = 0.0, streak_f0_ = 0.0;   // violation / objective when the current streak began
  bool in_solve_ = false;           // inside solve() (the retry ladder exists) as opposed to begin() / step() calls
  i64 n_eq_ = 0;                    // equality rows (fixed at begin())
  bool resto_stationary_ = false;   // the last restoration ended where no step reduces the violation
  double resto_theta_ = 0.0;        // violation where the last restoration ended
  int dc_fixed_count_ = 0;          // iterations whose wrong inertia the dual regularisation alone repaired
  bool dc_fixed_last_ = false, always_dc_ = false;
  double *lanV = nullptr, *lanW = nullptr, *lanQ = nullptr, *lanY = nullptr;
  bool lan_warm_ = false;           // lanY holds the previous call's smallest Ritz vector
  double lan_width_ = 0.0;          // spectral width seen by the last cold Lanczos run
  double t_begin_ = 0.0;
};

}  // namespace dnlp
