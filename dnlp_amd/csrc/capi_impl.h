// Implementation of the C ABI of include/dnlp_hip.h over an execution space E.
// libdnlp_hip.so instantiates it with HipExec (capi.hip).  The test oracle instantiates the
// same text with its host space under the prefix orc_ (oracle/oracle_lib.cpp).
#pragma once
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "ipm_core.h"
#include "kkt_dense.h"
#include "lbfgs_core.h"

namespace dnlp {

inline std::string& tls_error() {
  static thread_local std::string e;
  return e;
}

template <class E>
struct ProblemT {
  E ex;
  std::unique_ptr<TapeBlob> blob;
  Model<E> model;
  DenseKkt<E> kkt;
  std::unique_ptr<Ipm<E, DenseKkt<E>>> ipm;
  std::unique_ptr<ReducedLbfgs<E>> lbfgs;
  FusedObjective<E> fused;
  std::shared_ptr<void> batch_state;   // exec-space specific batched-solve state (capi.hip)
  bool use_fused = true;
  bool lbfgs_device_loop = true;            // option lbfgs_device_loop=no keeps the host-driven L-BFGS loop
  bool sparse_tail = true;                  // option sparse_dense_tail: the chain of one-block levels over a dense separator
                                            // is factorised as a dense matrix (sparse_plan.h: choose_tail)
  bool kkt_paired = true;                   // option kkt_paired: rotated static pairs + unpivoted LDL^T for dense patterns
  i64 paired_min_n = 1024;                  // (below it the one-workgroup-panel Bunch-Kaufman costs 1-5 ms and never loses digits)
  std::unique_ptr<SparsePlanHost> batch_plan_full;   // (capi.hip batch_runner: a full plan when sparse_plan has a dense tail)
  std::vector<double> plan_jabs;            // |Jacobian| at the start point (steers the static pairing)
  bool exact_hessian_substituted = false;   // hessian_approximation=limited-memory was requested
  int lbfgs_history = 10;
  IpmOptions opt;
  i64 pivot_max_n = 2048;
  i64 optimistic_min_n = static_cast<i64>(1) << 40;   // option kkt_optimistic_min_n: dense orders above it start unpivoted (kkt_dense.h); default off
  double *dx = nullptr, *dlam = nullptr, *dg = nullptr, *dgrad = nullptr, *djac = nullptr, *dh = nullptr;
  bool swept = false, kkt_ready = false, time_kernels = false;
  double *ws_g = nullptr, *ws_l = nullptr, *ws_u = nullptr;     // warm-start multipliers (exec space)
  double *ws_buf_g = nullptr, *ws_buf_l = nullptr, *ws_buf_u = nullptr;
  int linear_solver = 0;            // 0 auto, 1 dense, 2 sparse (static-pattern LDL^T)
  // totals over every dnlp_ipm_begin / dnlp_ipm_step of this handle (bench.py: a timed region may
  // span restarts, and begin() resets the per-solve statistics)
  long cum_iterations = 0, cum_factorizations = 0, cum_begins = 0;
  double cum_begin_seconds = 0.0;
  IntermediateCb intermediate_cb = nullptr;   // dnlp_set_intermediate_cb
  void* intermediate_user = nullptr;
  SparsePlanHost sparse_plan;
  double plan_seconds = 0.0;      // host time of the symbolic analysis (build_sparse_plan)
  bool sparse_planned = false, use_sparse = false;

  explicit ProblemT(int device) : ex(device) {}
  // generated kernels exist in the HIP space only
  template <class X = E> auto set_fused_codegen(bool on) -> decltype(std::declval<X&>().fused_codegen, void()) { ex.fused_codegen = on; }
  void set_fused_codegen(...) {}
  ~ProblemT() { model.destroy(); }

  void create(const void* data, size_t len) {
    blob.reset(new TapeBlob(data, len));
    finish_create();
  }
  void create(const TapeArrayDesc* arrays, int n) {
    blob.reset(new TapeBlob(arrays, n));
    finish_create();
  }
  void finish_create() {
    model.init(&ex, *blob);
    fused.load(&ex, *blob);
    ex.sync();          // every upload out of the caller's memory has landed
    blob.reset();       // (a view of the caller's memory: not kept)
    const auto& t = model.t;
    dx = ex.template alloc<double>(static_cast<size_t>(t.N));
    dlam = ex.template alloc<double>(static_cast<size_t>(t.m + 1));
    dg = ex.template alloc<double>(static_cast<size_t>(t.m + 1));
    dgrad = ex.template alloc<double>(static_cast<size_t>(t.N));
    djac = ex.template alloc<double>(static_cast<size_t>(t.nnzJ + 1));
    dh = ex.template alloc<double>(static_cast<size_t>(t.nnzH + 1));
  }

  void load_x(const double* x, int new_x, bool need_h = false) {
    (void)need_h;
    if (new_x || !swept) {
      ex.h2d(dx, x, sizeof(double) * static_cast<size_t>(model.t.N));
      model.sweep(dx, false);
      swept = true;
    }
  }

  // Decide dense vs sparse KKT once per handle: the plan is built when the pattern is sparse enough
  // to be worth it, and used when its factor is a small fraction of the dense triangle.
  void plan_linear_solver() {
    if (sparse_planned) return;
    sparse_planned = true;
    use_sparse = false;
    const auto& t = *model.owner;
    const double n = static_cast<double>(t.N + t.m);
    if (t.nblk > 0 || t.ndense > 0 || n < 2) return;
    // a forced-dense handle needs the analysis only for the static pairing of kkt_dense.h (ensure_ipm's conditions):
    // outside them the Jacobian sweep and the whole update program would be built for nothing
    if (linear_solver == 1) {
      const i64 nn = t.N + t.m;
      const bool pairing = E::has_host_control && kkt_paired && nn >= paired_min_n && nn <= 16384 && optimistic_min_n > nn;
      if (!pairing) return;
    }
    // Jacobian magnitudes at the tape's start point steer the static 2x2 pairing away from
    // couplings that vanish there
    std::vector<double>& jabs = plan_jabs;
    jabs.assign(static_cast<size_t>(t.nnzJ), 0.0);
    if (t.nnzJ > 0) {
      ex.h2d(dx, t.h_x0.data(), sizeof(double) * static_cast<size_t>(t.N));
      model.sweep(dx, false);
      model.eval_jac_after_sweep(djac);
      ex.d2h(jabs.data(), djac, sizeof(double) * static_cast<size_t>(t.nnzJ));
      for (double& v : jabs) v = std::isfinite(v) ? std::fabs(v) : 0.0;
      swept = false;
    }
    const double pattern = static_cast<double>(t.nnzH + t.nnzJ) + n;
    if (linear_solver == 0 && pattern > 0.1 * 0.5 * n * n) return;        // already dense (the pairing alone: ensure_ipm)
    const double t_plan0 = now_sec();
    build_sparse_plan(t, sparse_plan, opt.bound_relax_factor > 0.0, &jabs, E::has_host_control && sparse_tail);
    plan_seconds = now_sec() - t_plan0;
    if (std::getenv("DNLP_TIME_PLAN"))
      std::fprintf(stderr, "[dnlp] sparse plan: order %.0f, %zu triples, fill ratio %.4f, %.3f s on the host\n", n,
                   static_cast<size_t>(sparse_plan.ntrip), sparse_plan.fill_ratio, plan_seconds);
    // a long update program (dense-ish fill) is walked by one workgroup: it must be clearly cheaper
    // than the chip-wide dense factorisation (n^3/3 flops at MFMA rate; phase retrieval, order
    // 1472 with 3.6e6 triples, stays dense — a 7e5-order chain with 8e5 triples is sparse)
    const double triples = static_cast<double>(static_cast<size_t>(sparse_plan.ntrip));
    use_sparse = linear_solver == 2 ||
                 (sparse_plan.fill_ratio <= 0.3 && (triples <= 4e5 || triples * 1000.0 <= n * n * n / 3.0));
    if (linear_solver == 1) use_sparse = false;      // forced dense (the plan still serves the static pairing of kkt_dense.h)
  }

  void ensure_ipm() {
    if (!kkt_ready) {
      plan_linear_solver();
      kkt.pivot_max_n = pivot_max_n;
      kkt.optimistic_min_n = optimistic_min_n;
      if (use_sparse) {
        kkt.init_sparse(&ex, model.t.N, model.t.m, sparse_plan.upload(&ex, sparse_plan.tail_n > 0));
        kkt.fallback_max_n = linear_solver == 2 ? 0 : 2048;      // forced sparse never falls back
      }
      else {
        // dense pattern with a static pairing: the rotated-pair unpivoted factorisation (kkt_dense.h) instead of
        // Bunch-Kaufman, for orders where the chip-wide / host blocked LDL^T pays (option kkt_paired = no disables)
        const auto& tp = *model.owner;
        const i64 nn = tp.N + tp.m;
        bool done = false;
        if constexpr (E::has_host_control) {
          if (kkt_paired && tp.nblk == 0 && tp.ndense == 0 && nn >= paired_min_n && nn <= 16384 && optimistic_min_n > nn) {
            std::vector<i32> perm, pair_pos;
            const double t0 = now_sec();
            static_pivot_order(sparse_plan, nn, perm, pair_pos);
            if (std::getenv("DNLP_TIME_PLAN"))
              std::fprintf(stderr, "[dnlp] static pivot order: order %lld, %zu pairs, %.3f s on the host\n", (long long)nn, pair_pos.size(), now_sec() - t0);
            if (!perm.empty() && !pair_pos.empty()) { kkt.init_paired(&ex, tp.N, tp.m, perm, pair_pos); done = true; }
          }
        }
        if (!done) kkt.init(&ex, model.t.N, model.t.m);
      }
      kkt_ready = true;
    }
    kkt.lw.time_updates = time_kernels;
    if (!ipm) ipm.reset(new Ipm<E, DenseKkt<E>>(&ex, &model, &kkt));
    ipm->opt = opt;
    ipm->ws_mult_g = ws_g; ipm->ws_mult_xL = ws_l; ipm->ws_mult_xU = ws_u;
    ipm->intermediate_cb = intermediate_cb; ipm->intermediate_user = intermediate_user;
  }

  // back to the defaults of a fresh handle (a cached handle is reused by the next solve of the same
  // problem: options of the previous call must not leak into it).  The linear-solver choice stays: it
  // is fixed once the KKT object exists.
  void reset_options() {
    opt = IpmOptions();
    pivot_max_n = 2048;
    optimistic_min_n = static_cast<i64>(1) << 40;
    use_fused = true;
    lbfgs_device_loop = true;
    lbfgs_history = 10;
    time_kernels = false;
    exact_hessian_substituted = false;
    set_fused_codegen(true);
  }

  int set_option(const std::string& k, const std::string& v) {
    auto num = [&]() { return std::strtod(v.c_str(), nullptr); };
    auto yes = [&]() { return v == "yes" || v == "1" || v == "true" || v == "True"; };
    if (k == "tol") opt.tol = num();
    else if (k == "max_iter") opt.max_iter = static_cast<int>(num());
    else if (k == "mu_strategy") {
      if (v == "adaptive") opt.mu_strategy = 1;
      else if (v == "monotone") opt.mu_strategy = 0;
      else return -12;
    } else if (k == "mu_init") opt.mu_init = num();
    else if (k == "bound_relax_factor") opt.bound_relax_factor = num();
    else if (k == "bound_push") opt.bound_push = num();
    else if (k == "bound_frac") opt.bound_frac = num();
    else if (k == "hessian_approximation") {
      // "limited-memory": IPOPT's quasi-Newton interior-point mode, no second derivatives (ipm_core.h: lm_*)
      if (v != "exact" && v != "limited-memory") return -12;
      opt.hessian_approximation = (v == "limited-memory") ? 1 : 0;
      // (the in-kernel solver has no quasi-Newton mode: a batch launch with this option is refused with -12,
      //  capi.hip; the front-end takes the host-driven loop for a single solve)
      exact_hessian_substituted = false;
    }
    else if (k == "derivative_test") { /* accepted, unused: oracles are exact by construction */ }
    else if (k == "least_square_init_duals") opt.least_square_init_duals = yes() ? 1 : 0;
    else if (k == "print_level") opt.print_level = static_cast<int>(num());
    else if (k == "dual_inf_tol") opt.dual_inf_tol = num();
    else if (k == "constr_viol_tol") opt.constr_viol_tol = num();
    else if (k == "compl_inf_tol") opt.compl_inf_tol = num();
    else if (k == "acceptable_tol") opt.acceptable_tol = num();
    else if (k == "acceptable_iter") opt.acceptable_iter = static_cast<int>(num());
    else if (k == "acceptable_constr_viol_tol") opt.acceptable_constr_viol_tol = num();
    else if (k == "acceptable_dual_inf_tol") opt.acceptable_dual_inf_tol = num();
    else if (k == "acceptable_compl_inf_tol") opt.acceptable_compl_inf_tol = num();
    else if (k == "nlp_scaling_method") opt.nlp_scaling = (v == "none") ? 0 : 1;
    else if (k == "nlp_scaling_max_gradient") opt.nlp_scaling_max_gradient = num();
    else if (k == "max_wall_time" || k == "max_cpu_time") opt.max_wall_time = num();
    else if (k == "max_hessian_perturbation") opt.max_hessian_perturbation = num();
    else if (k == "max_soc") opt.max_soc = static_cast<int>(num());
    else if (k == "constr_mult_init_max") opt.constr_mult_init_max = num();
    else if (k == "bound_mult_init_val") opt.bound_mult_init_val = num();
    else if (k == "warm_start_init_point") opt.warm_start = yes() ? 1 : 0;
    else if (k == "warm_start_bound_push") opt.warm_start_bound_push = num();
    else if (k == "warm_start_bound_frac") opt.warm_start_bound_frac = num();
    else if (k == "warm_start_mult_bound_push") opt.warm_start_mult_bound_push = num();
    else if (k == "kkt_pivot_max_n") pivot_max_n = std::min<i64>(static_cast<i64>(num()), E::kPivotedMaxOrder);
    else if (k == "kkt_optimistic_min_n") optimistic_min_n = static_cast<i64>(num());
    else if (k == "kkt_paired") kkt_paired = yes();
    else if (k == "sparse_dense_tail") sparse_tail = yes();
    else if (k == "kkt_paired_min_n") paired_min_n = static_cast<i64>(num());
    else if (k == "lazy_dense_fallback") opt.lazy_dense_fallback = yes() ? 1 : 0;
    else if (k == "restoration") opt.restoration = yes() ? 1 : 0;
    else if (k == "time_kernels") time_kernels = yes();
    else if (k == "lbfgs_history") lbfgs_history = static_cast<int>(num());
    else if (k == "limited_memory_max_history") { lbfgs_history = static_cast<int>(num()); opt.limited_memory_max_history = static_cast<int>(num()); }
    else if (k == "limited_memory_max_skipping") opt.limited_memory_max_skipping = static_cast<int>(num());
    else if (k == "fused_objective") use_fused = yes();
    else if (k == "fused_codegen") set_fused_codegen(yes());
    else if (k == "lbfgs_device_loop") lbfgs_device_loop = yes();
    else if (k == "adaptive_fallback") opt.adaptive_fallback = yes() ? 1 : 0;
    else if (k == "stall_guard") opt.stall_guard = (v == "auto") ? -1 : yes() ? 1 : 0;
    else if (k == "lanczos_inertia_bound") opt.lanczos_inertia_bound = yes() ? 1 : 0;
    else if (k == "lanczos_min_n") opt.lanczos_min_n = static_cast<int>(num());
    else if (k == "linear_solver") {
      // IPOPT's names (mumps, ma27, ...) are accepted and mean "your choice"; dense / sparse force a path
      linear_solver = (v == "dense") ? 1 : (v == "sparse") ? 2 : 0;
      if (kkt_ready) return -12;      // fixed once the KKT object exists
    }
    else if (k == "sb" || k == "print_user_options" || k == "print_timing_statistics")
      { /* IPOPT options with no counterpart here: accepted and ignored */ }
    else return -12;   // Invalid_Option, as IPOPT reports unknown names
    return 0;
  }
};

}  // namespace dnlp

// ------------------------------------------------------------------------------------------
// extern "C" surface generator.  PFX = dnlp_ for the product, orc_ for the oracle.
#define DNLP_CAT2(a, b) a##b
#define DNLP_CAT(a, b) DNLP_CAT2(a, b)
#define DNLP_TRY(...)                                                   \
  try { __VA_ARGS__ } catch (const std::exception& e) { dnlp::tls_error() = e.what(); return -199; } \
  catch (...) { dnlp::tls_error() = "unknown exception"; return -199; }

#define DNLP_DEFINE_CAPI(PFX, EXEC, HANDLE)                                                          \
  struct HANDLE : dnlp::ProblemT<EXEC> { using dnlp::ProblemT<EXEC>::ProblemT; };                    \
  using DNLP_CAT(PFX, problem_t) = dnlp::ProblemT<EXEC>;                                             \
  extern "C" {                                                                                       \
  const char* DNLP_CAT(PFX, last_error)(void) { return dnlp::tls_error().c_str(); }                  \
  HANDLE* DNLP_CAT(PFX, create)(const void* blob, size_t len, int device) {                            \
    try {                                                                                            \
      auto* p = new HANDLE(device);                                                \
      p->create(blob, len);                                                                          \
      return p;                                                                                      \
    } catch (const std::exception& e) { dnlp::tls_error() = e.what(); return nullptr; }              \
  }                                                                                                  \
  HANDLE* DNLP_CAT(PFX, create_arrays)(const dnlp::TapeArrayDesc* arrays, int n_arrays, int device) { \
    try {                                                                                            \
      auto* p = new HANDLE(device);                                                                  \
      p->create(arrays, n_arrays);                                                                   \
      return p;                                                                                      \
    } catch (const std::exception& e) { dnlp::tls_error() = e.what(); return nullptr; }              \
  }                                                                                                  \
  void DNLP_CAT(PFX, destroy)(HANDLE* vp) { delete vp; }                                                \
  int DNLP_CAT(PFX, bind_dense)(HANDLE* vp, int cid, const double* dptr, int64_t ld) {                 \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    auto& t = *p->model.owner;                                                                       \
    if (cid < 0 || cid >= static_cast<int>(t.h_dense_ptr.size())) { dnlp::tls_error() = "bad constant id"; return -1; } \
    t.h_dense_ptr[cid] = dptr; t.h_dense_ld[cid] = ld; return 0;                                     \
  }                                                                                                  \
  int DNLP_CAT(PFX, dims)(HANDLE* vp, int64_t* n, int64_t* m, int64_t* nj, int64_t* nh) {              \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    if (n) *n = p->model.t.N; if (m) *m = p->model.t.m;                                              \
    if (nj) *nj = p->model.t.nnzJ; if (nh) *nh = p->model.t.coo_complete ? p->model.t.nnzH : -1;     \
    return 0;                                                                                        \
  }                                                                                                  \
  int DNLP_CAT(PFX, bounds)(HANDLE* vp, double* lb, double* ub, double* cl, double* cu, double* x0) {  \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    auto& t = *p->model.owner;                                                                       \
    auto cp = [](double* d, const std::vector<double>& s) { if (d && !s.empty()) std::memcpy(d, s.data(), s.size() * 8); }; \
    cp(lb, t.h_lb); cp(ub, t.h_ub); cp(cl, t.h_cl); cp(cu, t.h_cu); cp(x0, t.h_x0); return 0;        \
  }                                                                                                  \
  int DNLP_CAT(PFX, eval_f)(HANDLE* vp, const double* x, int new_x, double* f) {                       \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    DNLP_TRY(p->load_x(x, new_x); *f = p->model.eval_f_after_sweep(); return 0;)                     \
  }                                                                                                  \
  int DNLP_CAT(PFX, eval_grad_f)(HANDLE* vp, const double* x, int new_x, double* grad) {               \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    DNLP_TRY(p->load_x(x, new_x); p->model.eval_grad_after_sweep(p->dgrad);                          \
             p->ex.d2h(grad, p->dgrad, 8 * static_cast<size_t>(p->model.t.N)); return 0;)           \
  }                                                                                                  \
  int DNLP_CAT(PFX, eval_g)(HANDLE* vp, const double* x, int new_x, double* g) {                       \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    DNLP_TRY(p->load_x(x, new_x); p->model.eval_g_after_sweep(p->dg);                                \
             p->ex.d2h(g, p->dg, 8 * static_cast<size_t>(p->model.t.m)); return 0;)                  \
  }                                                                                                  \
  int DNLP_CAT(PFX, eval_jac_g)(HANDLE* vp, const double* x, int new_x, int32_t* iRow, int32_t* jCol, double* vals) { \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    auto& t = *p->model.owner;                                                                       \
    DNLP_TRY(                                                                                        \
      if (!vals) {                                                                                   \
        if (iRow) std::memcpy(iRow, t.h_jac_rows.data(), 4 * t.h_jac_rows.size());                   \
        if (jCol) std::memcpy(jCol, t.h_jac_cols.data(), 4 * t.h_jac_cols.size());                   \
        return 0;                                                                                    \
      }                                                                                              \
      p->load_x(x, new_x); p->model.eval_jac_after_sweep(p->djac);                                   \
      p->ex.d2h(vals, p->djac, 8 * static_cast<size_t>(t.nnzJ)); return 0;)                          \
  }                                                                                                  \
  int DNLP_CAT(PFX, eval_h)(HANDLE* vp, const double* x, int new_x, double sigma, const double* lambda, \
                            int new_lambda, int32_t* iRow, int32_t* jCol, double* vals) {            \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    auto& t = *p->model.owner;                                                                       \
    (void)new_lambda;                                                                                \
    DNLP_TRY(                                                                                        \
      if (!t.coo_complete) { dnlp::tls_error() = "Hessian too large for COO output; use dnlp_solve"; return -1; } \
      if (!vals) {                                                                                   \
        if (iRow) std::memcpy(iRow, t.h_hess_rows.data(), 4 * t.h_hess_rows.size());                 \
        if (jCol) std::memcpy(jCol, t.h_hess_cols.data(), 4 * t.h_hess_cols.size());                 \
        return 0;                                                                                    \
      }                                                                                              \
      (void)new_x;                                                                                   \
      p->ex.h2d(p->dx, x, 8 * static_cast<size_t>(t.N));                                             \
      if (t.m) p->ex.h2d(p->dlam, lambda, 8 * static_cast<size_t>(t.m));                             \
      p->model.eval_hess(p->dx, sigma, p->dlam); p->swept = true;                                    \
      p->model.hess_coo(p->dh);                                                                      \
      p->ex.d2h(vals, p->dh, 8 * static_cast<size_t>(t.nnzH)); return 0;)                            \
  }                                                                                                  \
  int DNLP_CAT(PFX, set_option)(HANDLE* vp, const char* k, const char* v) {                            \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    int rc = p->set_option(k, v);                                                                    \
    if (rc) dnlp::tls_error() = std::string("invalid option ") + k + "=" + v;                        \
    return rc;                                                                                       \
  }                                                                                                  \
  int DNLP_CAT(PFX, reset_options)(HANDLE* vp) {                                                     \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    p->reset_options(); return 0;                                                                    \
  }                                                                                                  \
  int DNLP_CAT(PFX, set_intermediate_cb)(HANDLE* vp, dnlp::IntermediateCb cb, void* user) {          \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    p->intermediate_cb = cb; p->intermediate_user = user;                                            \
    if (p->ipm) { p->ipm->intermediate_cb = cb; p->ipm->intermediate_user = user; }                  \
    return 0;                                                                                        \
  }                                                                                                  \
  int DNLP_CAT(PFX, ipm_begin)(HANDLE* vp, const double* x0) {                                         \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    DNLP_TRY(p->ensure_ipm(); p->swept = false; const double t0 = dnlp::now_sec();                   \
             const int rc = p->ipm->begin(x0); p->ex.sync();                                         \
             p->cum_factorizations += p->ipm->stats.factorizations;                                  \
             p->cum_begin_seconds += dnlp::now_sec() - t0; p->cum_begins += 1; return rc;)           \
  }                                                                                                  \
  int DNLP_CAT(PFX, ipm_step)(HANDLE* vp, int max_steps, int* done) {                                  \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    /* counts interior-point iterations that were actually carried out (ipm->iter advanced): the  \
       step() call that only detects convergence / a limit and returns does not count */            \
    DNLP_TRY(int r = 99; const int it0 = p->ipm->iter, f0 = p->ipm->stats.factorizations;            \
             while (r == 99 && p->ipm->iter - it0 < max_steps) r = p->ipm->step();                   \
             p->ex.sync(); const int k = p->ipm->iter - it0;                                         \
             p->cum_iterations += k; p->cum_factorizations += p->ipm->stats.factorizations - f0;     \
             if (done) *done = k; return r;)                                                         \
  }                                                                                                  \
  int DNLP_CAT(PFX, ipm_finish)(HANDLE* vp, double* x, double* obj, double* g, double* mg, double* mxl, \
                                double* mxu, int* iters) {                                           \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    DNLP_TRY(p->ipm->extract(x, obj, mg, mxl, mxu, g);                                               \
             if (iters) *iters = p->ipm->iter; return p->ipm->status;)                               \
  }                                                                                                  \
  int DNLP_CAT(PFX, solve)(HANDLE* vp, double* x, double* obj, double* g, double* mg, double* mxl,     \
                           double* mxu, int* iters) {                                                \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    DNLP_TRY(p->ensure_ipm(); p->swept = false;                                                      \
             int st = p->ipm->solve(x);                                                              \
             if (p->ipm->initialized) p->ipm->extract(x, obj, mg, mxl, mxu, g);                      \
             if (iters) *iters = p->ipm->iter; return st;)                                           \
  }                                                                                                  \
  int DNLP_CAT(PFX, solve_reduced)(HANDLE* vp, double* x, double* obj, int* iters, int* evals, double* gnorm) { \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    DNLP_TRY(if (!p->lbfgs) p->lbfgs.reset(new dnlp::ReducedLbfgs<EXEC>(&p->ex, &p->model));           \
             p->lbfgs->tol = p->opt.tol; p->lbfgs->max_iter = p->opt.max_iter > 3000 ? p->opt.max_iter : 20000; \
             p->lbfgs->history = p->lbfgs_history; p->lbfgs->print_level = p->opt.print_level;         \
             p->lbfgs->fused = &p->fused; p->lbfgs->use_fused = p->use_fused;                        \
             p->lbfgs->allow_device_loop = p->lbfgs_device_loop;                                     \
             p->swept = false;                                                                       \
             int st = p->lbfgs->solve(x);                                                            \
             p->lbfgs->extract(x, obj);                                                              \
             if (iters) *iters = p->lbfgs->iterations; if (evals) *evals = p->lbfgs->evaluations;    \
             if (gnorm) *gnorm = p->lbfgs->gnorm_final; return st;)                                  \
  }                                                                                                  \
  int DNLP_CAT(PFX, reduced_info)(HANDLE* vp, double* out, int n) {                                  \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    double v[6] = {0, 0, 0, 0, 0, 0};                                                                \
    if (p->lbfgs) { v[0] = p->lbfgs->device_loop_used ? 1.0 : 0.0; v[1] = p->lbfgs->device_seconds;  \
                    v[2] = p->lbfgs->device_slots; v[3] = p->lbfgs->fused_used ? 1.0 : 0.0;          \
                    v[4] = p->lbfgs->wall; v[5] = p->lbfgs->device_persistent ? 1.0 : 0.0; }                \
    for (int i = 0; i < n && i < 6; ++i) out[i] = v[i];                                              \
    return 0;                                                                                        \
  }                                                                                                  \
  int DNLP_CAT(PFX, eval_fused)(HANDLE* vp, const double* xfree, double* f, double* grad) {            \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    DNLP_TRY(if (!p->fused.present) { dnlp::tls_error() = "no fused objective program in this tape"; return -11; } \
             const size_t nf = static_cast<size_t>(p->fused.nfree);                                   \
             double* dxf = p->ex.template alloc<double>(nf); double* dgf = p->ex.template alloc<double>(nf); \
             p->ex.h2d(dxf, xfree, 8 * nf); *f = p->fused.eval(dxf, dgf); p->ex.d2h(grad, dgf, 8 * nf); \
             p->ex.release(dxf); p->ex.release(dgf); return 0;)                                      \
  }                                                                                                  \
  int DNLP_CAT(PFX, set_warm_start)(HANDLE* vp, const double* mg, const double* mxl, const double* mxu) { \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    DNLP_TRY(const size_t N = static_cast<size_t>(p->model.t.N), m = static_cast<size_t>(p->model.t.m); \
             if (!mg || !mxl || !mxu) { p->ws_g = p->ws_l = p->ws_u = nullptr; return 0; }            \
             if (!p->ws_buf_g) { p->ws_buf_g = p->ex.template alloc<double>(m ? m : 1);               \
               p->ws_buf_l = p->ex.template alloc<double>(N); p->ws_buf_u = p->ex.template alloc<double>(N); } \
             p->ex.h2d(p->ws_buf_g, mg, 8 * m); p->ex.h2d(p->ws_buf_l, mxl, 8 * N); p->ex.h2d(p->ws_buf_u, mxu, 8 * N); \
             p->ws_g = p->ws_buf_g; p->ws_l = p->ws_buf_l; p->ws_u = p->ws_buf_u; return 0;)            \
  }                                                                                                  \
  int DNLP_CAT(PFX, kkt_info)(HANDLE* vp, int64_t* out) {                                               \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    DNLP_TRY(p->plan_linear_solver();                                                                \
             out[0] = p->use_sparse ? 1 : 0; out[1] = p->sparse_plan.nnzL; out[2] = p->sparse_plan.nblk(); \
             out[3] = p->sparse_plan.maxs; out[4] = p->sparse_plan.n_pairs;                          \
             out[5] = static_cast<int64_t>(p->sparse_plan.ntrip);                                 \
             out[6] = static_cast<int64_t>(p->sparse_plan.lev_off.size()) - 1;                              \
             out[7] = (!p->use_sparse && p->model.t.N + p->model.t.m <= p->pivot_max_n) ? 1 : 0; return 0;)  \
  }                                                                                                  \
  int64_t DNLP_CAT(PFX, kkt_tail_nodes)(HANDLE* vp) {                                                  \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    DNLP_TRY(p->plan_linear_solver(); return p->use_sparse ? static_cast<int64_t>(p->sparse_plan.tail_n) : 0;)     \
  }                                                                                                  \
  int DNLP_CAT(PFX, kkt_mode)(HANDLE* vp) {                                                            \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    if (!p->kkt_ready) return -1;                                                                    \
    return p->kkt.sparse ? 0 : p->kkt.paired ? 3 : p->kkt.paired_factorizations > 0 ? 4 : p->kkt.pivoted ? 1 : 2;                           \
  }                                                                                                  \
  int DNLP_CAT(PFX, get_stats)(HANDLE* vp, double* s, int n) {                                         \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    if (!p->ipm) return -1;                                                                          \
    const auto& st = p->ipm->stats;                                                                  \
    double k3[5] = {0, 0, 0, 0, 0}; p->ex.ldlt_stats(p->kkt.lw, k3);                                                   \
    double v[24] = {double(st.iterations), double(st.factorizations), st.wall, st.t_eval, st.t_factor, \
                    st.t_solve, p->ipm->mu, st.inf_pr, st.inf_du, st.cmpl, st.nlp_error,              \
                    st.last_delta_w, p->ipm->sf, k3[0], k3[1], k3[2],                                \
                    double(p->cum_iterations), double(p->cum_factorizations), double(p->cum_begins), \
                    p->cum_begin_seconds, double(st.skipped_factorizations), k3[3], k3[4],                  \
                    p->exact_hessian_substituted ? 1.0 : 0.0};       \
    for (int i = 0; i < n && i < 24; ++i) s[i] = v[i];                                               \
    return 0;                                                                                        \
  }                                                                                                  \
  size_t DNLP_CAT(PFX, get_log)(HANDLE* vp, char* buf, size_t cap) {                                   \
    auto* p = static_cast<DNLP_CAT(PFX, problem_t)*>(vp);                                            \
    std::string all;                                                                                 \
    if (p->ipm) for (auto& l : p->ipm->iterlog.lines) { all += l; all += '\n'; }                               \
    if (buf && cap) { size_t n = all.size() < cap - 1 ? all.size() : cap - 1; std::memcpy(buf, all.data(), n); buf[n] = 0; } \
    return all.size() + 1;                                                                           \
  }                                                                                                  \
  }
