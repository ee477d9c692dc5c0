// Reduced-space L-BFGS for canonical problems whose constraints only define auxiliary variables
// (BASELINE config C2: unconstrained NLP, "tape f / grad f evaluation + line search only").
//
// The reference solves such problems through IPOPT on the canonical form (aux variables +
// equalities, dnlp2smooth.py:42-111).  Here the user's variables x_f are the only unknowns:
//   forward :  t <- t - g(x_f, t)   repeated `depth` times  (every row of g is  t_k - expr_k,
//              so the fixed point is the exact forward substitution; depth = nesting depth)
//   adjoint :  lambda <- lambda + (grad_t f - (J^T lambda)_t)   repeated `depth` times
//   reduced gradient = grad_f f - (J^T lambda)_f
// built from the same tape kernels as the interior-point path (sweep, G spmv, Mg/MJ maps, COO
// J^T product).  The optimiser is limited-memory BFGS (two-loop recursion, history 10) with an
// Armijo backtracking line search; no linear system is ever formed.
#pragma once
#include <cmath>
#include <string>
#include <vector>

#include "fused_obj.h"
#include "ipm_core.h"

namespace dnlp {

template <class E>
class ReducedLbfgs {
 public:
  ReducedLbfgs(E* ex, Model<E>* md) : ex_(ex), md_(md) {}

  // optional fused native-form evaluator (fused_obj.h); when present and enabled, f / grad f of
  // the user's variables come from ONE kernel instead of the canonical tape's fixed-point passes
  FusedObjective<E>* fused = nullptr;
  bool use_fused = true;
  bool fused_used = false;
  bool allow_device_loop = true;
  bool device_loop_used = false;      // the device-resident loop (lbfgs_codegen.h) ran the solve
  double device_seconds = 0.0;
  int device_slots = 0;
  bool device_persistent = false;   // the single-launch persistent kernel ran (lbfgs_codegen.h: dnlp_lb_persist)

  double tol = 1e-7;
  int max_iter = 20000;
  int history = 10;
  int print_level = 0;
  int iterations = 0, evaluations = 0;
  double f_final = 0.0, gnorm_final = 0.0, wall = 0.0;
  std::vector<std::string> log_lines;

  // canonical full vector x (N); g (m); grad (N); jv (nnzJ); lam (m); jt (N)
  double *x = nullptr, *g = nullptr, *grad = nullptr, *jv = nullptr, *lam = nullptr, *jt = nullptr;
  double *xf = nullptr, *gf = nullptr, *xt = nullptr, *gt = nullptr, *dir = nullptr, *BV = nullptr;

  template <class T> T* A(i64 n) { return ex_->template alloc<T>(static_cast<size_t>(n > 0 ? n : 1)); }

  // f and reduced gradient at the free variables xfree (exec space, nfree); returns false on NaN
  bool eval(const double* xfree, double& fval, double* gred) {
    if (fused && fused->present && use_fused) {
      fval = fused->eval(xfree, gred);
      ++evaluations;
      fused_used = true;
      const double chk = ex_->sum(fused->nfree, [=] DNLP_HD(i64 k) { return gred[k] - gred[k]; });
      return std::isfinite(fval) && chk == 0.0;
    }
    const TapeView& t = md_->t;
    const i64 N = t.N, m = t.m, nf = t.nfree;
    const i32 *fi = t.free_idx, *dv = t.def_var;
    double* xx = x;
    ex_->map(nf, [=] DNLP_HD(i64 k) { xx[fi[k]] = xfree[k]; });
    double* gg = g;
    for (i64 pass = 0; pass < t.red_depth; ++pass) {
      md_->sweep(x, false);
      md_->eval_g_after_sweep(g);
      ex_->map(m, [=] DNLP_HD(i64 i) { xx[dv[i]] -= gg[i]; });
    }
    md_->sweep(x, false);
    fval = md_->eval_f_after_sweep();
    md_->eval_grad_after_sweep(grad);
    ++evaluations;
    if (m > 0) {
      md_->eval_jac_after_sweep(jv);
      ex_->zero(lam, sizeof(double) * static_cast<size_t>(m));
      ex_->zero(jt, sizeof(double) * static_cast<size_t>(N));
      double* ll = lam;
      const double *gr = grad, *jtt = jt;
      for (i64 pass = 0; pass <= t.red_depth; ++pass) {
        ex_->map(m, [=] DNLP_HD(i64 i) { ll[i] += gr[dv[i]] - jtt[dv[i]]; });
        md_->jac_tmult(jv, lam, jt);
      }
      ex_->map(nf, [=] DNLP_HD(i64 k) { gred[k] = gr[fi[k]] - jtt[fi[k]]; });
    } else {
      const double* gr = grad;
      ex_->map(nf, [=] DNLP_HD(i64 k) { gred[k] = gr[fi[k]]; });
    }
    const double chk = ex_->sum(nf, [=] DNLP_HD(i64 k) { return gred[k] - gred[k]; });
    return std::isfinite(fval) && chk == 0.0;
  }

  // returns 0 converged, -1 iteration limit, -13 invalid number at the start, 3 line search stuck
  int solve(const double* x0_host) {
    const double t0 = now_sec();
    const TapeView& t = md_->t;
    if (!t.reducible) return -11;
    const i64 N = t.N, m = t.m, nf = t.nfree;
    if (!x) {
      x = A<double>(N); g = A<double>(m); grad = A<double>(N); jv = A<double>(t.nnzJ); lam = A<double>(m);
      jt = A<double>(N); xf = A<double>(nf); gf = A<double>(nf); xt = A<double>(nf); gt = A<double>(nf);
      dir = A<double>(nf);
    }
    ex_->h2d(x, x0_host, sizeof(double) * static_cast<size_t>(N));
    {
      const i32* fi = t.free_idx;
      double* a = xf;
      const double* xx = x;
      ex_->map(nf, [=] DNLP_HD(i64 k) { a[k] = xx[fi[k]]; });
    }
    // Vector-free form of the two-loop recursion (Chen, Wang, Zhou: "Large-scale L-BFGS using
    // MapReduce", 2014): the basis B = [s_0..s_{M-1}, y_0..y_{M-1}, g] lives in one contiguous
    // buffer, its Gram matrix on the host.  Per iteration THREE fused multi-dot kernels (new s, new
    // y, new g against the whole basis: one scalar read-back each) replace the ~4M dependent
    // reductions of the textbook loop, the recursion runs on 2M+1 coefficients on the host, and the
    // direction is one fused linear combination.  Same mathematics, a handful of host round trips.
    const int M = history < 1 ? 1 : (history > 15 ? 15 : history);
    // Device-resident form (lbfgs_codegen.h): the same algorithm with every decision taken on the device
    // over the generated objective kernel; the host only enqueues.  Taken when the objective has a
    // generated form and no iteration log is asked for.
    device_loop_used = false;
    if (allow_device_loop && fused && fused->present && use_fused && print_level < 5) {
      typename E::LbfgsResult r;
      if (ex_->lbfgs_generated_solve(fused->progs, fused->consts, fused->c0, nf, xf, M, tol, max_iter, r)) {
        device_loop_used = true;
        fused_used = true;
        iterations = r.iterations;
        evaluations = r.evaluations;
        gnorm_final = r.gnorm;
        device_seconds = r.seconds;
        device_slots = r.slots;
        device_persistent = r.persistent;
        double fl;
        const bool keep = use_fused;
        use_fused = false;
        eval(xf, fl, gf);            // canonical vector (auxiliary variables) consistent with the solution
        use_fused = keep;
        --evaluations;
        // a tape without segments carries the user's variables and a zero objective (the direct path of
        // algorithm='lbfgs', dnlp_amd/problem.py): the objective lives in the fused program only
        f_final = (md_->t.nseg == 0 && md_->t.m == 0) ? r.f : fl;
        wall = now_sec() - t0;
        return r.status;
      }
    }
    const int nb = 2 * M + 1, GR = 2 * M;              // GR: row of the current gradient
    if (!BV) BV = A<double>(static_cast<i64>(nb) * nf);
    ex_->zero(BV, sizeof(double) * static_cast<size_t>(nb) * static_cast<size_t>(nf));
    auto row = [&](int r) { return BV + static_cast<i64>(r) * nf; };
    std::vector<double> G(static_cast<size_t>(nb) * nb, 0.0), coef(static_cast<size_t>(nb)), tmp(static_cast<size_t>(nb));
    std::vector<double> rho(static_cast<size_t>(M), 0.0), alpha(static_cast<size_t>(M), 0.0);
    auto gram_row = [&](int r) {
      ex_->vt_dot(nb, BV, nf, row(r), tmp.data());
      for (int j = 0; j < nb; ++j) G[static_cast<size_t>(r) * nb + j] = G[static_cast<size_t>(j) * nb + r] = tmp[static_cast<size_t>(j)];
    };
    double f = 0.0;
    if (!eval(xf, f, gf)) return -13;
    ex_->d2d(row(GR), gf, sizeof(double) * static_cast<size_t>(nf));
    gram_row(GR);
    int stored = 0, head = 0;
    iterations = 0;
    int status = -1;
    char buf[160];
    for (int it = 0; it < max_iter; ++it) {
      const double* gp = row(GR);
      const double gn = ex_->max(nf, [=] DNLP_HD(i64 k) { return fabs(gp[k]); });
      gnorm_final = gn;
      f_final = f;
      if (print_level >= 5 && (it % 10 == 0)) {
        std::snprintf(buf, sizeof buf, "%6d  f=%.10e  |g|_inf=%.3e  evals=%d", it, f, gn, evaluations);
        log_lines.emplace_back(buf);
      }
      if (gn <= tol * std::fmax(1.0, std::fabs(f))) { status = 0; break; }
      // two-loop recursion on the coefficients of q in the basis
      auto Gd = [&](int r, const std::vector<double>& c) { double v = 0.0; for (int j = 0; j < nb; ++j) v += c[static_cast<size_t>(j)] * G[static_cast<size_t>(r) * nb + j]; return v; };
      std::fill(coef.begin(), coef.end(), 0.0);
      coef[static_cast<size_t>(GR)] = 1.0;
      for (int j = 0; j < stored; ++j) {
        const int idx = (head - 1 - j + 2 * M) % M;
        const double a = rho[static_cast<size_t>(idx)] * Gd(idx, coef);
        alpha[static_cast<size_t>(idx)] = a;
        coef[static_cast<size_t>(M + idx)] -= a;
      }
      if (stored > 0) {
        const int idx = (head - 1 + M) % M;
        const double gam = G[static_cast<size_t>(idx) * nb + (M + idx)] / G[static_cast<size_t>(M + idx) * nb + (M + idx)];
        for (double& c : coef) c *= gam;
      }
      for (int j = stored - 1; j >= 0; --j) {
        const int idx = (head - 1 - j + 2 * M) % M;
        const double bta = rho[static_cast<size_t>(idx)] * Gd(M + idx, coef);
        coef[static_cast<size_t>(idx)] += alpha[static_cast<size_t>(idx)] - bta;
      }
      for (double& c : coef) c = -c;
      double gd = Gd(GR, coef);
      if (!(gd < 0.0)) {   // not a descent direction: restart from steepest descent
        stored = 0;
        std::fill(coef.begin(), coef.end(), 0.0);
        coef[static_cast<size_t>(GR)] = -1.0;
        gd = -G[static_cast<size_t>(GR) * nb + GR];
      }
      double* d = dir;
      ex_->v_comb(nb, BV, nf, coef.data(), dir);
      // Armijo backtracking (first iteration starts at 1/|g|)
      double step = (it == 0 && stored == 0) ? std::fmin(1.0, 1.0 / std::fmax(gn, 1e-300)) : 1.0;
      double fn = 0.0;
      bool ok = false;
      for (int ls = 0; ls < 60; ++ls) {
        double* xn = xt;
        const double* xc = xf;
        const double st = step;
        ex_->map(nf, [=] DNLP_HD(i64 k) { xn[k] = xc[k] + st * d[k]; });
        if (eval(xt, fn, gt) && fn <= f + 1e-4 * step * gd) { ok = true; break; }
        step *= 0.5;
      }
      if (!ok) { status = 3; break; }
      // history update: s, y into slot `head`, the new gradient into the g row, three Gram rows
      double* sn = row(head);
      double* yn = row(M + head);
      double* gr = row(GR);
      const double *xn = xt, *xc = xf, *gnew = gt;
      ex_->map(nf, [=] DNLP_HD(i64 k) { sn[k] = xn[k] - xc[k]; yn[k] = gnew[k] - gr[k]; });
      ex_->d2d(gr, gt, sizeof(double) * static_cast<size_t>(nf));
      gram_row(head);
      gram_row(M + head);
      gram_row(GR);
      const double sy = G[static_cast<size_t>(head) * nb + (M + head)];
      const double ss = G[static_cast<size_t>(head) * nb + head], yy = G[static_cast<size_t>(M + head) * nb + (M + head)];
      if (sy > 1e-10 * std::sqrt(ss) * std::sqrt(yy)) {
        rho[static_cast<size_t>(head)] = 1.0 / sy;
        head = (head + 1) % M;
        if (stored < M) ++stored;
      }
      ex_->d2d(xf, xt, sizeof(double) * static_cast<size_t>(nf));
      f = fn;
      iterations = it + 1;
    }
    ex_->d2d(gf, row(GR), sizeof(double) * static_cast<size_t>(nf));
    // leave the canonical vector consistent with the final free variables (tape pass: it also
    // fills the auxiliary variables the fused evaluator never forms)
    double fl;
    const bool keep = use_fused;
    use_fused = false;
    eval(xf, fl, gf);
    use_fused = keep;
    if (!(md_->t.nseg == 0 && md_->t.m == 0 && fused && fused->present && use_fused)) f_final = fl;
    wall = now_sec() - t0;
    return status;
  }

  void extract(double* x_host, double* obj) {
    if (x_host) ex_->d2h(x_host, x, sizeof(double) * static_cast<size_t>(md_->t.N));
    if (obj) *obj = f_final;
  }

 private:
  E* ex_;
  Model<E>* md_;
};

}  // namespace dnlp
