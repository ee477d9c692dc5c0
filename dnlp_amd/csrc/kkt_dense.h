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
  // Paired mode (host-driven spaces, dense patterns whose static pairing exists; set up by init_paired): the matrix is
  // assembled in the elimination order of the static analysis (sparse_plan.h: every equality row next to the variable
  // it was matched with, blocks in an order whose pivots are structurally non-zero), every matched pair is rotated by
  // Q = [[1, 1], [s, -s]] / sqrt(2) — the block [[w, a], [a, -d]] (zero diagonal for a variable without curvature and an
  // unregularised row) becomes [[(w-d)/2 + s a, (w+d)/2], [(w+d)/2, (w-d)/2 - s a]], s chosen so that the first
  // pivot is |w-d|/2 + |a| — and Q^T P K P^T Q is factorised WITHOUT pivoting by the blocked MFMA LDL^T: the same
  // pivot blocks as the static 2x2 sequence of the sparse path, as two 1x1 steps each (same inertia: Q is orthogonal).
  // 1.1 ms instead of 8.6 ms for Bunch-Kaufman at order 1472.  Two zero-pivot attempts with delta_w = 0 in a row
  // switch the handle to Bunch-Kaufman for good, as in optimistic mode.
  bool paired = false;
  i64 npairs = 0;
  int paired_factorizations = 0;
  double paired_growth = 0.0, paired_growth_max = 1e8;   // largest |L| of the last paired factorisation / the demotion threshold
  i32* pperm = nullptr;        // KKT index -> position in the permuted matrix
  i32* ppos = nullptr;         // first position of every pair
  double* psign = nullptr;     // s of the last factorisation's rotation, per pair
  double* pvec = nullptr;      // permuted right-hand side / solution
  typename E::LdltWork lw;
  // dense tail of a sparse plan (init_sparse)
  double* Kt = nullptr;
  double* tail_x = nullptr;
  i32* tail_ipiv = nullptr;
  i64 tail_ld = 0;
  bool tail_pivoted = true;
  typename E::LdltWork tail_lw;
  // sparse mode (sparse_plan.h / sparse_ldl.h): static-pattern LDL^T instead of the dense matrix
  bool sparse = false;
  bool skip_hessian = false;   // limited-memory quasi-Newton mode (ipm_core.h): the Hessian block is the diagonal the caller passes
  SparsePlan sp;
  VecP<E> svals;               // plan-layout values: assembled matrix, then (D, L)
  VecP<E> swork;               // scratch of the numeric phase (sparse_ldl_work_doubles)
  i64 fallback_max_n = 0;      // orders up to which a structurally singular static pivot sequence
                               // makes the instance switch to the dense Bunch-Kaufman path
  DNLP_HD bool can_fallback() const { return sparse && n <= fallback_max_n; }
  DNLP_HD void fallback_to_dense() {
    DNLP_THIS_IN_LDS(E); DNLP_PTR_IN_LDS(E, ex);
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
    if constexpr (E::has_host_control) {
      if (sp.tail_n > 0) {
        // the dense tail of the plan (sparse_plan.h): its own matrix, factorised by the dense path — Bunch-Kaufman
        // up to pivot_max_n (more robust than the static sequence it replaces), the blocked LDL^T above
        tail_ld = (sp.tail_n + 7) / 8 * 8;
        Kt = ex->template alloc<double>(static_cast<size_t>(tail_ld) * static_cast<size_t>(sp.tail_n) + 256);
        tail_ipiv = ex->template alloc<i32>(static_cast<size_t>(sp.tail_n));
        tail_x = ex->template alloc<double>(static_cast<size_t>(sp.tail_n));
        tail_pivoted = sp.tail_n <= pivot_max_n && !E::is_device;   // (device: unpivoted first, see assemble_factor)
        tail_lw.padded = true;
        ex->ldlt_prepare(tail_lw, sp.tail_n, tail_ld, tail_pivoted);
      }
    }
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

  // perm[k] = position of KKT index k; pair_pos[q] = first position of pair q (host arrays)
  void init_paired(E* e, i64 N_, i64 m_, const std::vector<i32>& perm, const std::vector<i32>& pair_pos) {
    const i64 keep = pivot_max_n;
    pivot_max_n = 0;                            // unpivoted factorisation of the rotated matrix
    init(e, N_, m_);
    pivot_max_n = keep;
    paired = true;
    optimistic = true;                          // (the zero-pivot streak rule ends the mode)
    npairs = static_cast<i64>(pair_pos.size());
    pperm = ex->template alloc<i32>(static_cast<size_t>(n));
    ppos = ex->template alloc<i32>(static_cast<size_t>(npairs > 0 ? npairs : 1));
    psign = ex->template alloc<double>(static_cast<size_t>(npairs > 0 ? npairs : 1));
    pvec = ex->template alloc<double>(static_cast<size_t>(n));
    ex->h2d(pperm, perm.data(), sizeof(i32) * static_cast<size_t>(n));
    if (npairs) ex->h2d(ppos, pair_pos.data(), sizeof(i32) * static_cast<size_t>(npairs));
  }

  // Q^T (P K P^T) Q in place on the lower triangle: the diagonal blocks (and the sign choice), then the rows of every
  // pair left of its block, then the columns below it — pairs own disjoint rows / columns, so a pass has no conflicts
  DNLP_HD void rotate_pairs() {
    DNLP_THIS_IN_LDS(E); DNLP_PTR_IN_LDS(E, ex);
    double* Kp = K;
    const i64 ldk = ld, nn = n, np = npairs;
    const i32* pos = ppos;
    double* sg = psign;
    const double r2 = 0.70710678118654752440;
    ex->map(np, [=] DNLP_HD(i64 q) {
      const i64 p = pos[q];
      const double w = Kp[p + p * ldk], a = Kp[(p + 1) + p * ldk], e = Kp[(p + 1) + (p + 1) * ldk];
      // a diagonal that dominates the coupling is a good 1x1 pivot as it stands (Bunch-Kaufman's test with the
      // coupling as the only off-diagonal candidate): no rotation — with a large delta_w the rotated block would
      // compute its second pivot -2|a| as a difference of two numbers of size delta_w / 2
      if (fabs(w) >= 0.6403882032022076 * fabs(a)) { sg[q] = 0.0; return; }
      const double sa = a < 0.0 ? -1.0 : 1.0;
      const double s = (w + e >= 0.0) ? sa : -sa;
      sg[q] = s;
      Kp[p + p * ldk] = 0.5 * (w + e) + s * a;
      Kp[(p + 1) + (p + 1) * ldk] = 0.5 * (w + e) - s * a;
      Kp[(p + 1) + p * ldk] = 0.5 * (w - e);
    });
    // rows p, p + 1 over the columns c < p   (one lane per (pair, column); pairs sit early in the order, p is small on average)
    ex->map(np * nn, [=] DNLP_HD(i64 t) {
      const i64 q = t / nn, c = t % nn, p = pos[q];
      if (c >= p) return;
      const double s = sg[q];
      if (s == 0.0) return;
      const double v1 = Kp[p + c * ldk], v2 = Kp[(p + 1) + c * ldk];
      Kp[p + c * ldk] = r2 * (v1 + s * v2);
      Kp[(p + 1) + c * ldk] = r2 * (v1 - s * v2);
    });
    // columns p, p + 1 over the rows r > p + 1
    ex->map(np * nn, [=] DNLP_HD(i64 t) {
      const i64 q = t / nn, r = t % nn, p = pos[q];
      if (r <= p + 1) return;
      const double s = sg[q];
      if (s == 0.0) return;
      const double u1 = Kp[r + p * ldk], u2 = Kp[r + (p + 1) * ldk];
      Kp[r + p * ldk] = r2 * (u1 + s * u2);
      Kp[r + (p + 1) * ldk] = r2 * (u1 - s * u2);
    });
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
      bool oks = ex->sparse_factor(sp, svals, swork, nneg, nzero);
      if constexpr (E::has_host_control) {
        if (sp.tail_n > 0 && oks) {
          double* T = Kt;
          const double* sv = svals;
          const i32 *src = sp.tg_src, *dst = sp.tg_dst;
          const i64 r = sp.tail_n, ldt = tail_ld;
          int nn2 = 0, nz2 = 0;
          // the static sequence this replaces was unpivoted: the blocked LDL^T goes first (0.2 ms at order 300 on the
          // MI355X against 1 ms for Bunch-Kaufman), and the first zero / non-finite pivot hands the tail to
          // Bunch-Kaufman for good (the gathered entries are still in the plan's storage: the attempt is repeated)
          for (int attempt = 0; attempt < 2; ++attempt) {
            ex->zero(Kt, sizeof(double) * static_cast<size_t>(ldt) * static_cast<size_t>(r));
            ex->map(sp.tg_count, [=] DNLP_HD(i64 q) { const i64 d = dst[q]; T[d % r + (d / r) * ldt] = sv[src[q]]; });
            if (sp.pg_maxcols > 0) {
              // + what the panel blocks of the levels before the tail contributed (sparse_ldl.h: T -= Pl Pw^T)
              const double* acc = swork + sparse_ldl_tail_acc_offset(sp);
              ex->map(r * r, [=] DNLP_HD(i64 e) { const i64 u = e % r, v = e / r; if (u >= v) T[u + v * ldt] += acc[u + v * ldt]; });
            }
            tail_lw.expect_neg = -1;
            nn2 = nz2 = 0;
            oks = ex->ldlt_factor(tail_lw, Kt, r, ldt, tail_ipiv, tail_pivoted, &nn2, &nz2);
            if (tail_pivoted || (oks && nz2 == 0) || r > pivot_max_n) break;
#if !DNLP_DEVICE_PASS
            if (std::getenv("DNLP_PAIRED_DEBUG")) std::fprintf(stderr, "[tail] unpivoted attempt: ok %d, %d zero pivots, delta_w %.2e -> Bunch-Kaufman from here on\n", oks ? 1 : 0, nz2, dw);
#endif
            tail_pivoted = true;
            ex->ldlt_prepare(tail_lw, r, ldt, true);
          }
          *nneg += nn2;
          *nzero += nz2;
        }
      }
      return oks;
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
    bool placed = false;
    if constexpr (E::has_host_control) if (paired) {
      placed = true;
      // the same entries at their permuted positions (lower triangle of P K P^T)
      const i32* pm = pperm;
      if (!skip_hessian) ex->map(t.nnzH, [=] DNLP_HD(i64 p) {
        const i64 r = hr[p], c = hc[p];
        if (fixmask[r] != 0.0 || fixmask[c] != 0.0) return;
        const i64 a = pm[r], b = pm[c];
        Kp[(a > b ? a : b) + (a > b ? b : a) * ldk] += hs[p];
      });
      ex->map(t.nnzJ, [=] DNLP_HD(i64 p) {
        if (fixmask[jc[p]] != 0.0) return;
        const i64 a = pm[NN + jr[p]], b = pm[jc[p]];
        Kp[(a > b ? a : b) + (a > b ? b : a) * ldk] = jv[p];
      });
      ex->map(N, [=] DNLP_HD(i64 j) {
        const i64 a = pm[j];
        if (fixmask[j] != 0.0) Kp[a + a * ldk] = 1.0;
        else Kp[a + a * ldk] += Sx[j] + dw;
      });
      ex->map(m, [=] DNLP_HD(i64 i) { const i64 a = pm[NN + i]; Kp[a + a * ldk] = -D[i]; });
      rotate_pairs();
    }
    if (!placed) {
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
    }
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
      // paired mode is stricter: its pivots are static, so a zero pivot later than the first two factorisations (where
      // multipliers are still zero) means a coupling of a pair has vanished at this iterate — Bunch-Kaufman picks
      // another partner there, a static sequence cannot.  The handle is demoted at once and the attempt repeated.
      bool demote = false;
      if (paired) {
        ++paired_factorizations;
        if (dw == 0.0 && (!ok || *nzero > 0) && paired_factorizations > 2) demote = true;
        // element growth: Bunch-Kaufman keeps |L| below 1 / alpha + 1 by choosing its pivots; a static sequence whose
        // multipliers reach 1e8 has lost half the digits of the Schur complement and its inertia count with them
        if (ok && !demote) {
          const double* Kp2 = K;
          const i64 ldk2 = ld, nn2 = n;
          const double g = ex->max(nn2 * nn2, [=] DNLP_HD(i64 q) {
            const i64 r = q % nn2, c = q / nn2;
            return r > c ? fabs(Kp2[r + c * ldk2]) : 0.0; });
          paired_growth = g;
          if (!(g <= paired_growth_max)) demote = true;
#if !DNLP_DEVICE_PASS
          if (std::getenv("DNLP_PAIRED_DEBUG")) std::fprintf(stderr, "[paired] factorisation %d delta_w %.2e: nneg %d nzero %d max|L| %.3e%s\n", paired_factorizations, dw, *nneg, *nzero, g, demote ? " -> Bunch-Kaufman from here on" : "");
#endif
        }
      }
      if (demote || (optimistic && dw == 0.0 && optimistic_zero_streak >= 2)) {
        optimistic = false;
        paired = false;
        pivoted = n <= pivot_max_n;
        ex->ldlt_prepare(lw, n, ld, pivoted);
        return assemble_factor(md, jv, Sx, D, fixmask, dw, nneg, nzero);
      }
    }
    return ok;
  }

  DNLP_HD void solve(const double* rhs, double* sol) {
    DNLP_THIS_IN_LDS(E); DNLP_PTR_IN_LDS(E, ex);
    if (sol != rhs) ex->d2d(sol, rhs, sizeof(double) * static_cast<size_t>(n));
    if (sparse) {
      bool tail_done = false;
      if constexpr (E::has_host_control) {
        if (sp.tail_n > 0) {
          // levels before the tail forward (+ D^-1), the tail by its dense factor, then the same levels backward
          SparsePlan ph = sp;
          ph.solve_phase = 1;
          ex->sparse_solve(ph, svals, sol);
          double* tx = tail_x;
          const i32* tn = sp.tnode;
          ex->map(sp.tail_n, [=] DNLP_HD(i64 j) { tx[j] = sol[tn[j]]; });
          ex->ldlt_solve(tail_lw, Kt, sp.tail_n, tail_ld, tail_ipiv, tail_pivoted, tail_x);
          ex->map(sp.tail_n, [=] DNLP_HD(i64 j) { sol[tn[j]] = tx[j]; });
          ph.solve_phase = 2;
          ex->sparse_solve(ph, svals, sol);
          tail_done = true;
        }
      }
      if (!tail_done) ex->sparse_solve(sp, svals, sol);
      return;
    }
    if constexpr (E::has_host_control) if (paired) {
      // b~ = Q^T P b;  K~ y = b~;  x = P^T Q y
      double* v = pvec;
      const i32 *pm = pperm, *pos = ppos;
      const double* sg = psign;
      const double r2 = 0.70710678118654752440;
      ex->map(n, [=] DNLP_HD(i64 k) { v[pm[k]] = sol[k]; });
      ex->map(npairs, [=] DNLP_HD(i64 q) {
        const i64 p = pos[q];
        const double s = sg[q], a = v[p], b = v[p + 1];
        if (s == 0.0) return;
        v[p] = r2 * (a + s * b);
        v[p + 1] = r2 * (a - s * b);
      });
      ex->ldlt_solve(lw, K, n, ld, ipiv, false, v);
      ex->map(npairs, [=] DNLP_HD(i64 q) {
        const i64 p = pos[q];
        const double s = sg[q], a = v[p], b = v[p + 1];
        if (s == 0.0) return;
        v[p] = r2 * (a + b);
        v[p + 1] = r2 * s * (a - b);
      });
      ex->map(n, [=] DNLP_HD(i64 k) { sol[k] = v[pm[k]]; });
      return;
    }
    ex->ldlt_solve(lw, K, n, ld, ipiv, pivoted, sol);
  }
};

}  // namespace dnlp
