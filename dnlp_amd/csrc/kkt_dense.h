// Dense reduced KKT system  K = [ W + Sigma_x + delta_w I   J^T ;  J   -D ]  in column-major
// lower storage, assembled straight from the tape's Hessian / Jacobian value arrays (the
// reference hands IPOPT COO triplets that MUMPS re-assembles on every callback).
//
// Factorisation is delegated to the execution space:
//   * order <= pivot_max_n : Bunch-Kaufman LDL^T (1x1 / 2x2 pivots, inertia from D) — the role
//     MUMPS plays for IPOPT; needed whenever W has structurally zero pivots (variables that
//     enter only linearly, common after dnlp2smooth).
//   * larger orders        : blocked unpivoted LDL^T whose Schur-complement (trailing) update
//     runs on FP64 MFMA; zero / wrong-sign pivots are reported through the inertia so the
//     caller's delta_w / delta_c regularisation (WB Algorithm IC) repairs them.
#pragma once
#include "model.h"
#include "sparse_ldl.h"

namespace dnlp {

template <class E>
struct DenseKkt {
  E* ex = nullptr;
  i64 N = 0, m = 0, n = 0, ld = 0;
  double* K = nullptr;
  i32* ipiv = nullptr;
  double* work = nullptr;      // exec-space scratch for solves
  i64 pivot_max_n = 2048;
  i64 n_fixed = 0;             // number of fixed variables (set by the interior-point driver)
  bool pivoted = true;
  // Orders between optimistic_min_n and pivot_max_n on a host-driven space start WITHOUT pivoting
  // (blocked LDL^T, ~50 launches per 512 columns instead of two per column: 17 -> 2 ms at order
  // 1472) and switch to Bunch-Kaufman for good at the first zero pivot met with delta_w = 0
  // (variables without curvature, as dnlp2smooth leaves them).
  bool optimistic = false;
  int optimistic_zero_streak = 0;   // consecutive delta_w = 0 attempts that met a zero pivot
  i64 optimistic_min_n = static_cast<i64>(1) << 40;   // off unless the option kkt_optimistic_min_n lowers it:
                                                      // nonconvex problems can reach another local optimum
  typename E::LdltWork lw;
  // sparse mode (sparse_plan.h / sparse_ldl.h): static-pattern LDL^T instead of the dense matrix
  bool sparse = false;
  bool skip_hessian = false;   // limited-memory quasi-Newton mode (ipm_core.h): the Hessian block is the diagonal the caller passes
  SparsePlan sp;
  double* svals = nullptr;     // plan-layout values: assembled matrix, then (D, L)
  double* swork = nullptr;     // scratch of the numeric phase (sparse_ldl_work_doubles)
  i64 fallback_max_n = 0;      // orders up to which a structurally singular static pivot sequence
                               // makes the instance switch to the dense Bunch-Kaufman path
  DNLP_HD bool can_fallback() const { return sparse && n <= fallback_max_n; }
  DNLP_HD void fallback_to_dense() {
    sparse = false;
    const i64 keep = pivot_max_n;
    if (pivot_max_n < n) pivot_max_n = n;      // the fallback is the pivoted factorisation
    init(ex, N, m);
    pivot_max_n = keep;
    pivoted = true;
  }

  DNLP_HD void init_sparse(E* e, i64 N_, i64 m_, const SparsePlan& plan) {
    ex = e; N = N_; m = m_; n = N + m; ld = 0;
    sparse = true;
    pivoted = true;            // (static 2x2 blocks) keeps the large-dense-only code paths of the IP loop off
    sp = plan;
    svals = ex->template alloc<double>(static_cast<size_t>(sp.nvals > 0 ? sp.nvals : 1));
    swork = ex->template alloc<double>(static_cast<size_t>(sparse_ldl_work_doubles(sp)));
  }

  DNLP_HD void init(E* e, i64 N_, i64 m_) {
    ex = e; N = N_; m = m_; n = N + m;
    ld = (n + 7) / 8 * 8;                       // 64-byte aligned columns
    K = ex->template alloc<double>(static_cast<size_t>(ld) * static_cast<size_t>(n) + 256);   // +slack: tile reads past the last row
    ipiv = ex->template alloc<i32>(static_cast<size_t>(n));
    work = ex->template alloc<double>(static_cast<size_t>(n));
    pivoted = n <= pivot_max_n;
    optimistic = false;
    optimistic_zero_streak = 0;
    if constexpr (E::has_host_control) {
      if (pivoted && n > optimistic_min_n) { optimistic = true; pivoted = false; }
    }
    lw.padded = true;
    ex->ldlt_prepare(lw, n, ld, pivoted);
  }

  // Assemble from the model's current Hessian (Hs + dense blocks) and factor.
  DNLP_HD bool assemble_factor(Model<E>& md, const double* jv, const double* Sx, const double* D,
                       const double* fixmask, double dw, int* nneg, int* nzero) {
    const TapeView& t = md.t;
    if (sparse) {
      double* V = svals;
      const i32 *hp = sp.hpos, *jp = sp.jpos, *dp = sp.dpos;
      const i32 *hr = t.hess_rows, *hc = t.hess_cols, *jc = t.jac_cols;
      const double* hs = md.Hs;
      const i64 NN = N;
      ex->zero(svals, sizeof(double) * static_cast<size_t>(sp.nvals));
      if (!skip_hessian) ex->map(t.nnzH, [=] DNLP_HD(i64 p) {
        if (fixmask[hr[p]] != 0.0 || fixmask[hc[p]] != 0.0 || hp[p] < 0) return;
        V[hp[p]] += hs[p];
      });
      ex->map(t.nnzJ, [=] DNLP_HD(i64 p) {
        if (fixmask[jc[p]] != 0.0 || jp[p] < 0) return;
        V[jp[p]] = jv[p];
      });
      ex->map(N, [=] DNLP_HD(i64 j) {
        if (fixmask[j] != 0.0) V[dp[j]] = 1.0;
        else V[dp[j]] += Sx[j] + dw;
      });
      ex->map(m, [=] DNLP_HD(i64 i) { V[dp[NN + i]] = -D[i]; });
      return ex->sparse_factor(sp, svals, swork, nneg, nzero);
    }
    double* Kp = K;
    const i64 ldk = ld, NN = N;
    // one dense block covering all of W lets us skip the memset of the n x n part
    bool full_block = t.nblk == 1 && t.blocks[0].n == N && N > 4096 && !skip_hessian;
    if (!full_block) {
      ex->zero(K, sizeof(double) * static_cast<size_t>(ld) * static_cast<size_t>(n));
    } else {
      const i64 mm = m, nn = n;
      ex->map(mm * nn, [=] DNLP_HD(i64 q) { Kp[(NN + q % mm) + (q / mm) * ldk] = 0.0; });
    }
    for (i64 k = 0; k < t.nblk && !skip_hessian; ++k) {
      const DenseBlock& B = t.blocks[k];
      const double* P = t.dense_ptr[B.cid];
      const i64 ldp = t.dense_ld[B.cid], nb = B.n, x0 = B.x0;
      const double wk = md.dense_w[k];
      ex->dense_block_add(Kp, ldk, x0, P, ldp, nb, wk, full_block);
    }
    const i32 *hr = t.hess_rows, *hc = t.hess_cols, *jr = t.jac_rows, *jc = t.jac_cols;
    const double* hs = md.Hs;
    if (!skip_hessian) ex->map(t.nnzH, [=] DNLP_HD(i64 p) {
      const i64 r = hr[p], c = hc[p];
      if (fixmask[r] != 0.0 || fixmask[c] != 0.0) return;
      Kp[r + c * ldk] += hs[p];
    });
    ex->map(t.nnzJ, [=] DNLP_HD(i64 p) {
      if (fixmask[jc[p]] != 0.0) return;
      Kp[(NN + jr[p]) + static_cast<i64>(jc[p]) * ldk] = jv[p];
    });
    ex->map(N, [=] DNLP_HD(i64 j) {
      if (fixmask[j] != 0.0) Kp[j + j * ldk] = 1.0;
      else Kp[j + j * ldk] += Sx[j] + dw;
    });
    ex->map(m, [=] DNLP_HD(i64 i) { Kp[(NN + i) + (NN + i) * ldk] = -D[i]; });
    if (n_fixed > 0 && t.nblk > 0) {
      // fixed variables (lb == ub) inside a dense block: pin them (unit row / column)
      for (i64 kb = 0; kb < t.nblk; ++kb) {
        const DenseBlock& B = t.blocks[kb];
        const i64 nb = B.n, x0 = B.x0;
        ex->map(nb * nb, [=] DNLP_HD(i64 q) {
          const i64 r = x0 + q % nb, c = x0 + q / nb;
          if (r <= c) return;
          if (fixmask[r] != 0.0 || fixmask[c] != 0.0) Kp[r + c * ldk] = 0.0;
        });
      }
    }
    lw.expect_neg = static_cast<int>(m);
    const bool ok = ex->ldlt_factor(lw, K, n, ld, ipiv, pivoted, nneg, nzero);
    if constexpr (E::has_host_control) {
      // one such attempt is tolerated (at the start the multipliers are zero and variables that only
      // occur in nonlinear constraints have no curvature yet: delta_w handles that iteration)
      if (optimistic && dw == 0.0) optimistic_zero_streak = (!ok || *nzero > 0) ? optimistic_zero_streak + 1 : 0;
      if (optimistic && dw == 0.0 && optimistic_zero_streak >= 2) {
        optimistic = false;
        pivoted = true;
        ex->ldlt_prepare(lw, n, ld, true);
        return assemble_factor(md, jv, Sx, D, fixmask, dw, nneg, nzero);
      }
    }
    return ok;
  }

  DNLP_HD void solve(const double* rhs, double* sol) {
    if (sol != rhs) ex->d2d(sol, rhs, sizeof(double) * static_cast<size_t>(n));
    if (sparse) { ex->sparse_solve(sp, svals, sol); return; }
    ex->ldlt_solve(lw, K, n, ld, ipiv, pivoted, sol);
  }
};

}  // namespace dnlp
