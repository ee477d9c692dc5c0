/* consumer.c — a SECOND consumer of the oracle-level C ABI, in plain C99.
 *
 * The reference proves that its callback protocol is solver-agnostic by registering the same
 * `Oracles` object with two solvers: cyipopt (nlp_solvers/ipopt_nlpif.py:140-170) and Knitro
 * (nlp_solvers/knitro_nlpif.py:211-309: EVALFC / EVALGA / EVALH callbacks, Hessian indices swapped
 * to Knitro's upper-triangular convention at :272-274).  This program plays that second role for
 * libdnlp_hip.so: it includes include/dnlp_hip.h, links the library, and drives ONLY
 *     dnlp_create / dnlp_dims / dnlp_bounds / dnlp_eval_f / _grad_f / _g / _jac_g / _h / dnlp_destroy
 * through callbacks of exactly IPOPT's C-interface shape (IpStdCInterface.h: Eval_F_CB ... Eval_H_CB,
 * `Bool eval_*(Index n, Number* x, Bool new_x, ..., UserDataPtr)`, structure requested with
 * values == NULL) from its OWN small interior-point solver below — it never calls dnlp_solve.
 *
 * The solver is deliberately independent of csrc/ipm_core.h: slacks as explicit variables, dense
 * KKT system solved by LU with partial pivoting, curvature-test ("inertia-free", Chiang & Zavala
 * 2016) regularisation instead of an inertia count, l1-merit backtracking instead of a filter,
 * monotone barrier.  Knitro-style, it asks for the Hessian in UPPER-triangular index order.
 *
 *   consumer <tape.blob> [device]      prints one line:  status iters objective x[0..n)
 *
 * With -DCONSUMER_ORACLE the same text links the CPU test oracle (orc_* symbols) so that the
 * consumer's own logic is exercised without a GPU (tests only).
 */
#ifdef CONSUMER_ORACLE
#define dnlp_create orc_create
#define dnlp_destroy orc_destroy
#define dnlp_dims orc_dims
#define dnlp_bounds orc_bounds
#define dnlp_eval_f orc_eval_f
#define dnlp_eval_grad_f orc_eval_grad_f
#define dnlp_eval_g orc_eval_g
#define dnlp_eval_jac_g orc_eval_jac_g
#define dnlp_eval_h orc_eval_h
#define dnlp_last_error orc_last_error
#endif
#include "dnlp_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- IPOPT C-interface types (IpStdCInterface.h) -------------------------------------------- */
typedef double Number;
typedef int Index;
typedef int Bool;
typedef void* UserDataPtr;
typedef Bool (*Eval_F_CB)(Index n, Number* x, Bool new_x, Number* obj_value, UserDataPtr user_data);
typedef Bool (*Eval_Grad_F_CB)(Index n, Number* x, Bool new_x, Number* grad_f, UserDataPtr user_data);
typedef Bool (*Eval_G_CB)(Index n, Number* x, Bool new_x, Index m, Number* g, UserDataPtr user_data);
typedef Bool (*Eval_Jac_G_CB)(Index n, Number* x, Bool new_x, Index m, Index nele_jac, Index* iRow, Index* jCol,
                              Number* values, UserDataPtr user_data);
typedef Bool (*Eval_H_CB)(Index n, Number* x, Bool new_x, Number obj_factor, Index m, Number* lambda, Bool new_lambda,
                          Index nele_hess, Index* iRow, Index* jCol, Number* values, UserDataPtr user_data);

/* ---- trampolines: IPOPT-shaped callbacks -> the C ABI (what INTEGRATION.md describes in prose) -- */
static Bool cb_eval_f(Index n, Number* x, Bool new_x, Number* obj, UserDataPtr ud) {
  (void)n;
  return dnlp_eval_f((dnlp_problem*)ud, x, new_x, obj) == 0;
}
static Bool cb_eval_grad_f(Index n, Number* x, Bool new_x, Number* grad, UserDataPtr ud) {
  (void)n;
  return dnlp_eval_grad_f((dnlp_problem*)ud, x, new_x, grad) == 0;
}
static Bool cb_eval_g(Index n, Number* x, Bool new_x, Index m, Number* g, UserDataPtr ud) {
  (void)n; (void)m;
  return dnlp_eval_g((dnlp_problem*)ud, x, new_x, g) == 0;
}
static Bool cb_eval_jac_g(Index n, Number* x, Bool new_x, Index m, Index nele, Index* iRow, Index* jCol, Number* vals,
                          UserDataPtr ud) {
  (void)n; (void)m; (void)nele;
  return dnlp_eval_jac_g((dnlp_problem*)ud, x, new_x, (int32_t*)iRow, (int32_t*)jCol, vals) == 0;
}
static Bool cb_eval_h(Index n, Number* x, Bool new_x, Number sigma, Index m, Number* lambda, Bool new_lambda, Index nele,
                      Index* iRow, Index* jCol, Number* vals, UserDataPtr ud) {
  (void)n; (void)m; (void)nele;
  return dnlp_eval_h((dnlp_problem*)ud, x, new_x, sigma, lambda, new_lambda, (int32_t*)iRow, (int32_t*)jCol, vals) == 0;
}

/* ---- the consumer's own solver ------------------------------------------------------------------ */
typedef struct {
  Index n, m, nele_jac, nele_hess;
  const Number *xL, *xU, *gL, *gU;
  Eval_F_CB eval_f; Eval_Grad_F_CB eval_grad_f; Eval_G_CB eval_g; Eval_Jac_G_CB eval_jac_g; Eval_H_CB eval_h;
  UserDataPtr ud;
} MiniProblem;

#define NLP_INF 1e19
static double dmax(double a, double b) { return a > b ? a : b; }
static double dmin(double a, double b) { return a < b ? a : b; }

/* LU with partial pivoting, in place; returns 0 when a pivot vanishes */
static int lu_solve(int n, double* A, double* b) {
  for (int k = 0; k < n; ++k) {
    int p = k;
    double best = fabs(A[k * n + k]);
    for (int i = k + 1; i < n; ++i) if (fabs(A[i * n + k]) > best) { best = fabs(A[i * n + k]); p = i; }
    if (!(best > 1e-300)) return 0;
    if (p != k) {
      for (int j = 0; j < n; ++j) { double t = A[k * n + j]; A[k * n + j] = A[p * n + j]; A[p * n + j] = t; }
      double t = b[k]; b[k] = b[p]; b[p] = t;
    }
    for (int i = k + 1; i < n; ++i) {
      const double f = A[i * n + k] / A[k * n + k];
      if (f == 0.0) continue;
      for (int j = k + 1; j < n; ++j) A[i * n + j] -= f * A[k * n + j];
      b[i] -= f * b[k];
    }
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int j = i + 1; j < n; ++j) s -= A[i * n + j] * b[j];
    b[i] = s / A[i * n + i];
  }
  return 1;
}

static double push_in(double v, double l, double u) {
  const int hl = l > -NLP_INF, hu = u < NLP_INF;
  if (hl && hu) {
    const double pl = dmin(1e-2 * dmax(1.0, fabs(l)), 1e-2 * (u - l)), pu = dmin(1e-2 * dmax(1.0, fabs(u)), 1e-2 * (u - l));
    return dmin(dmax(v, l + pl), u - pu);
  }
  if (hl) return dmax(v, l + 1e-2 * dmax(1.0, fabs(l)));
  if (hu) return dmin(v, u - 1e-2 * dmax(1.0, fabs(u)));
  return v;
}

/* returns 0 on success (KKT error <= tol), 1 iteration limit, 2 evaluation / linear algebra failure */
static int mini_solve(const MiniProblem* P, Number* x, Number* obj_out, Number* lambda_out, int* iters_out, double tol,
                      int max_iter) {
  const int n = P->n, m = P->m;
  /* Knitro-style evaluation groups and conventions: structures first (values == NULL); the Hessian is
   * held in UPPER-triangular index order (knitro_nlpif.py:272-274 swaps the lower-triangular pair) */
  Index* jr = (Index*)malloc(sizeof(Index) * (size_t)(P->nele_jac + 1));
  Index* jc = (Index*)malloc(sizeof(Index) * (size_t)(P->nele_jac + 1));
  Index* h1 = (Index*)malloc(sizeof(Index) * (size_t)(P->nele_hess + 1));
  Index* h2 = (Index*)malloc(sizeof(Index) * (size_t)(P->nele_hess + 1));
  if (m > 0 && !P->eval_jac_g(n, NULL, 0, m, P->nele_jac, jr, jc, NULL, P->ud)) return 2;
  if (!P->eval_h(n, NULL, 0, 1.0, m, NULL, 0, P->nele_hess, h2, h1, NULL, P->ud)) return 2;   /* (rows, cols) -> (h2, h1): h1 <= h2 */
  for (int k = 0; k < P->nele_hess; ++k) if (h1[k] > h2[k]) return 2;                         /* upper triangle after the swap */
  /* inequality rows get a slack variable */
  int* slack_of = (int*)malloc(sizeof(int) * (size_t)(m + 1));
  int mi = 0;
  for (int i = 0; i < m; ++i) slack_of[i] = (P->gL[i] == P->gU[i]) ? -1 : mi++;
  const int nv = n + mi, nk = nv + m;
  double* v = (double*)calloc((size_t)nv, 8); double* l = (double*)calloc((size_t)nv, 8); double* u = (double*)calloc((size_t)nv, 8);
  double* zl = (double*)calloc((size_t)nv, 8); double* zu = (double*)calloc((size_t)nv, 8);
  double* lam = (double*)calloc((size_t)(m + 1), 8);
  double* g = (double*)calloc((size_t)(m + 1), 8); double* c = (double*)calloc((size_t)(m + 1), 8);
  double* grad = (double*)calloc((size_t)nv, 8);
  double* jv = (double*)calloc((size_t)(P->nele_jac + 1), 8); double* hv = (double*)calloc((size_t)(P->nele_hess + 1), 8);
  double* K = (double*)malloc(sizeof(double) * (size_t)nk * (size_t)nk);
  double* W = (double*)malloc(sizeof(double) * (size_t)nv * (size_t)nv);
  double* rhs = (double*)calloc((size_t)nk, 8); double* rv = (double*)calloc((size_t)nv, 8);
  double* dzl = (double*)calloc((size_t)nv, 8); double* dzu = (double*)calloc((size_t)nv, 8);
  double* vt = (double*)calloc((size_t)nv, 8); double* gt = (double*)calloc((size_t)(m + 1), 8);
  double* sig = (double*)calloc((size_t)nv, 8);
  for (int j = 0; j < n; ++j) { l[j] = P->xL[j]; u[j] = P->xU[j]; v[j] = push_in(x[j], l[j], u[j]); }
  if (m > 0 && !P->eval_g(n, v, 1, m, g, P->ud)) return 2;
  for (int i = 0; i < m; ++i) if (slack_of[i] >= 0) {
    const int k = n + slack_of[i];
    l[k] = P->gL[i]; u[k] = P->gU[i]; v[k] = push_in(g[i], l[k], u[k]);
  }
  for (int k = 0; k < nv; ++k) { zl[k] = l[k] > -NLP_INF ? 1.0 : 0.0; zu[k] = u[k] < NLP_INF ? 1.0 : 0.0; }
  double mu = 0.1, nu = 1.0, delta_last = 0.0;
  int it = 0, status = 1;
  double f = 0.0;
  for (; it <= max_iter; ++it) {
    /* EVALFC + EVALGA + EVALH at the current point (new_x only on the first call of the group) */
    if (!P->eval_f(n, v, 1, &f, P->ud)) { status = 2; break; }
    if (m > 0 && !P->eval_g(n, v, 0, m, g, P->ud)) { status = 2; break; }
    if (!P->eval_grad_f(n, v, 0, grad, P->ud)) { status = 2; break; }
    for (int k = n; k < nv; ++k) grad[k] = 0.0;
    if (m > 0 && !P->eval_jac_g(n, v, 0, m, P->nele_jac, NULL, NULL, jv, P->ud)) { status = 2; break; }
    if (!P->eval_h(n, v, 0, 1.0, m, lam, 1, P->nele_hess, NULL, NULL, hv, P->ud)) { status = 2; break; }
    for (int i = 0; i < m; ++i) c[i] = slack_of[i] < 0 ? g[i] - P->gL[i] : g[i] - v[n + slack_of[i]];
    /* r_v(mu') = grad + A^T lam - mu'/(v-l) + mu'/(u-v);  errors for mu' = 0 and mu' = mu */
    for (int k = 0; k < nv; ++k) rv[k] = grad[k];
    for (int p = 0; p < P->nele_jac; ++p) rv[jc[p]] += jv[p] * lam[jr[p]];
    for (int i = 0; i < m; ++i) if (slack_of[i] >= 0) rv[n + slack_of[i]] -= lam[i];
    double e_du = 0.0, e_pr = 0.0, e_c0 = 0.0, e_cmu = 0.0;
    for (int k = 0; k < nv; ++k) {
      e_du = dmax(e_du, fabs(rv[k] - zl[k] + zu[k]));
      if (l[k] > -NLP_INF) { e_c0 = dmax(e_c0, fabs((v[k] - l[k]) * zl[k])); e_cmu = dmax(e_cmu, fabs((v[k] - l[k]) * zl[k] - mu)); }
      if (u[k] < NLP_INF) { e_c0 = dmax(e_c0, fabs((u[k] - v[k]) * zu[k])); e_cmu = dmax(e_cmu, fabs((u[k] - v[k]) * zu[k] - mu)); }
    }
    for (int i = 0; i < m; ++i) e_pr = dmax(e_pr, fabs(c[i]));
    if (dmax(dmax(e_du, e_pr), e_c0) <= tol) { status = 0; break; }
    if (it == max_iter) break;
    while (dmax(dmax(e_du, e_pr), e_cmu) <= 10.0 * mu && mu > tol / 10.0) {
      mu = dmax(tol / 10.0, dmin(0.2 * mu, pow(mu, 1.5)));
      e_cmu = 0.0;
      for (int k = 0; k < nv; ++k) {
        if (l[k] > -NLP_INF) e_cmu = dmax(e_cmu, fabs((v[k] - l[k]) * zl[k] - mu));
        if (u[k] < NLP_INF) e_cmu = dmax(e_cmu, fabs((u[k] - v[k]) * zu[k] - mu));
      }
    }
    /* W = Hessian of the Lagrangian (upper-triangular entries mirrored) + Sigma */
    memset(W, 0, sizeof(double) * (size_t)nv * (size_t)nv);
    for (int p = 0; p < P->nele_hess; ++p) {
      W[h1[p] * nv + h2[p]] += hv[p];
      if (h1[p] != h2[p]) W[h2[p] * nv + h1[p]] += hv[p];
    }
    for (int k = 0; k < nv; ++k) {
      sig[k] = 0.0;
      if (l[k] > -NLP_INF) { sig[k] += zl[k] / (v[k] - l[k]); rv[k] -= mu / (v[k] - l[k]); }
      if (u[k] < NLP_INF) { sig[k] += zu[k] / (u[k] - v[k]); rv[k] += mu / (u[k] - v[k]); }
    }
    /* Newton step with curvature-test regularisation */
    double delta = 0.0, delta_c = 0.0, curv = 0.0, dn2 = 0.0;
    int ok = 0;
    for (int attempt = 0; attempt < 60 && !ok; ++attempt) {
      memset(K, 0, sizeof(double) * (size_t)nk * (size_t)nk);
      for (int a = 0; a < nv; ++a) {
        for (int b = 0; b < nv; ++b) K[a * nk + b] = W[a * nv + b];
        K[a * nk + a] += sig[a] + delta;
      }
      for (int p = 0; p < P->nele_jac; ++p) { K[(nv + jr[p]) * nk + jc[p]] = jv[p]; K[jc[p] * nk + (nv + jr[p])] = jv[p]; }
      for (int i = 0; i < m; ++i) {
        if (slack_of[i] >= 0) { const int k = n + slack_of[i]; K[(nv + i) * nk + k] = -1.0; K[k * nk + (nv + i)] = -1.0; }
        K[(nv + i) * nk + (nv + i)] = -delta_c;
      }
      for (int k = 0; k < nv; ++k) rhs[k] = -rv[k];
      for (int i = 0; i < m; ++i) rhs[nv + i] = -c[i];
      int solved = lu_solve(nk, K, rhs);
      if (solved) for (int k = 0; k < nk; ++k) if (!(rhs[k] == rhs[k]) || fabs(rhs[k]) > 1e30) solved = 0;
      if (solved) {
        curv = 0.0; dn2 = 0.0;
        for (int a = 0; a < nv; ++a) {
          double s = (sig[a] + delta) * rhs[a];
          for (int b = 0; b < nv; ++b) s += W[a * nv + b] * rhs[b];
          curv += rhs[a] * s;
          dn2 += rhs[a] * rhs[a];
        }
        double lc = 0.0;
        for (int i = 0; i < m; ++i) lc += (lam[i] + rhs[nv + i]) * c[i];
        if (curv + dmax(-lc, 0.0) >= 1e-8 * dn2) ok = 1;
      } else if (delta_c == 0.0) {
        delta_c = 1e-8 * pow(mu, 0.25);
      }
      if (!ok) delta = (delta == 0.0) ? (delta_last == 0.0 ? 1e-4 : dmax(1e-20, delta_last / 3.0)) : 8.0 * delta;
      if (delta > 1e40) break;
    }
    if (!ok) { status = 2; break; }
    if (delta > 0.0) delta_last = delta;
    double* dv = rhs;
    double* dlam = rhs + nv;
    for (int k = 0; k < nv; ++k) {
      dzl[k] = l[k] > -NLP_INF ? (mu - zl[k] * dv[k]) / (v[k] - l[k]) - zl[k] : 0.0;
      dzu[k] = u[k] < NLP_INF ? (mu + zu[k] * dv[k]) / (u[k] - v[k]) - zu[k] : 0.0;
    }
    const double tau = dmax(0.99, 1.0 - mu);
    double a_max = 1.0, a_z = 1.0;
    for (int k = 0; k < nv; ++k) {
      if (l[k] > -NLP_INF && dv[k] < 0.0) a_max = dmin(a_max, -tau * (v[k] - l[k]) / dv[k]);
      if (u[k] < NLP_INF && dv[k] > 0.0) a_max = dmin(a_max, tau * (u[k] - v[k]) / dv[k]);
      if (dzl[k] < 0.0) a_z = dmin(a_z, -tau * zl[k] / dzl[k]);
      if (dzu[k] < 0.0) a_z = dmin(a_z, -tau * zu[k] / dzu[k]);
    }
    /* l1 merit  M = f - mu sum log + nu |c|_1 */
    double phi = f, dphi = 0.0, c1 = 0.0;
    for (int k = 0; k < nv; ++k) {
      double gk = grad[k];
      if (l[k] > -NLP_INF) { phi -= mu * log(v[k] - l[k]); gk -= mu / (v[k] - l[k]); }
      if (u[k] < NLP_INF) { phi -= mu * log(u[k] - v[k]); gk += mu / (u[k] - v[k]); }
      dphi += gk * dv[k];
    }
    for (int i = 0; i < m; ++i) c1 += fabs(c[i]);
    if (c1 > 0.0) nu = dmax(nu, (dphi + 0.5 * dmax(curv, 0.0)) / (0.9 * c1) + 1e-8);
    const double D = dphi - nu * c1;
    double alpha = a_max;
    int accepted = 0;
    for (int ls = 0; ls < 50; ++ls) {
      for (int k = 0; k < nv; ++k) vt[k] = v[k] + alpha * dv[k];
      double ft = 0.0;
      int fin = P->eval_f(n, vt, 1, &ft, P->ud) && (m == 0 || P->eval_g(n, vt, 0, m, gt, P->ud));
      if (fin && ft == ft) {
        double pt = ft, ct = 0.0;
        for (int k = 0; k < nv; ++k) {
          if (l[k] > -NLP_INF) pt -= mu * log(vt[k] - l[k]);
          if (u[k] < NLP_INF) pt -= mu * log(u[k] - vt[k]);
        }
        for (int i = 0; i < m; ++i) ct += fabs(slack_of[i] < 0 ? gt[i] - P->gL[i] : gt[i] - vt[n + slack_of[i]]);
        if (pt == pt && pt + nu * ct <= phi + nu * c1 + 1e-4 * alpha * D + 1e-13 * fabs(phi)) { accepted = 1; break; }
      }
      alpha *= 0.5;
    }
    if (!accepted) { status = 2; break; }
    for (int k = 0; k < nv; ++k) {
      v[k] += alpha * dv[k];
      zl[k] += a_z * dzl[k];
      zu[k] += a_z * dzu[k];
      if (l[k] > -NLP_INF) { const double t = v[k] - l[k]; zl[k] = dmax(dmin(zl[k], 1e10 * mu / t), mu / (1e10 * t)); }
      if (u[k] < NLP_INF) { const double t = u[k] - v[k]; zu[k] = dmax(dmin(zu[k], 1e10 * mu / t), mu / (1e10 * t)); }
    }
    for (int i = 0; i < m; ++i) lam[i] += alpha * dlam[i];
  }
  for (int j = 0; j < n; ++j) x[j] = v[j];
  *obj_out = f;
  if (lambda_out) for (int i = 0; i < m; ++i) lambda_out[i] = lam[i];
  *iters_out = it;
  free(jr); free(jc); free(h1); free(h2); free(slack_of); free(v); free(l); free(u); free(zl); free(zu); free(lam);
  free(g); free(c); free(grad); free(jv); free(hv); free(K); free(W); free(rhs); free(rv); free(dzl); free(dzu);
  free(vt); free(gt); free(sig);
  return status;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s tape.blob [device]\n", argv[0]); return 64; }
  FILE* fp = fopen(argv[1], "rb");
  if (!fp) { perror(argv[1]); return 66; }
  fseek(fp, 0, SEEK_END);
  const long len = ftell(fp);
  fseek(fp, 0, SEEK_SET);
  void* blob = malloc((size_t)len);
  if (fread(blob, 1, (size_t)len, fp) != (size_t)len) { fprintf(stderr, "short read\n"); return 66; }
  fclose(fp);
  const int device = argc > 2 ? atoi(argv[2]) : 0;
  dnlp_problem* p = dnlp_create(blob, (size_t)len, device);
  if (!p) { fprintf(stderr, "dnlp_create: %s\n", dnlp_last_error()); return 70; }
  int64_t n64 = 0, m64 = 0, nj = 0, nh = 0;
  if (dnlp_dims(p, &n64, &m64, &nj, &nh) != 0 || nh < 0) { fprintf(stderr, "dnlp_dims: %s\n", dnlp_last_error()); return 70; }
  const int n = (int)n64, m = (int)m64;
  double* xL = (double*)malloc(8 * (size_t)(n + 1)); double* xU = (double*)malloc(8 * (size_t)(n + 1));
  double* gL = (double*)malloc(8 * (size_t)(m + 1)); double* gU = (double*)malloc(8 * (size_t)(m + 1));
  double* x = (double*)malloc(8 * (size_t)(n + 1)); double* lam = (double*)malloc(8 * (size_t)(m + 1));
  if (dnlp_bounds(p, xL, xU, gL, gU, x) != 0) { fprintf(stderr, "dnlp_bounds: %s\n", dnlp_last_error()); return 70; }
  MiniProblem P;
  P.n = n; P.m = m; P.nele_jac = (Index)nj; P.nele_hess = (Index)nh;
  P.xL = xL; P.xU = xU; P.gL = gL; P.gU = gU;
  P.eval_f = cb_eval_f; P.eval_grad_f = cb_eval_grad_f; P.eval_g = cb_eval_g; P.eval_jac_g = cb_eval_jac_g; P.eval_h = cb_eval_h;
  P.ud = p;
  double obj = 0.0;
  int iters = 0;
  const int st = mini_solve(&P, x, &obj, lam, &iters, 1e-8, 500);
  printf("%d %d %.17g", st, iters, obj);
  for (int j = 0; j < n; ++j) printf(" %.17g", x[j]);
  printf("\n");
  dnlp_destroy(p);
  free(blob); free(xL); free(xU); free(gL); free(gU); free(x); free(lam);
  return st == 0 ? 0 : 1;
}
