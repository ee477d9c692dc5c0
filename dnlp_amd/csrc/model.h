// Tape evaluator: f, grad f, g, Jacobian values, Lagrangian-Hessian values from the lowered
// normal form (dnlp_amd/lowering.py).  Device counterpart of the reference's `Oracles`
// callbacks (cvxpy/reductions/solvers/nlp_solvers/nlp_solver.py:212-421):
//
//   objective(x)      -> sweep + dot            (nlp_solver.py:212-216)
//   gradient(x)       -> sweep + Mg spmv        (:218-235)
//   constraints(x)    -> sweep + G spmv         (:237-244)
//   jacobian(x)       -> sweep + MJ spmv        (:278-307; affine part Jc is the cached
//                                                 "stored affine jacobian" of :282-295)
//   hessian(x,lam,s)  -> Mw spmv, sweep, MH spmv (:394-421; sum_coo/insert_missing_zeros are
//                                                 folded into the constant map MH)
//
// Single source: every per-element formula is a DNLP_HD lambda run through E::map / E::sum,
// so hipcc emits gfx950 kernels (HipExec) and g++ emits host loops for the test oracle.
// The two bandwidth-critical pieces have hand-written HIP kernels behind E::gemv_sym /
// E::spmv when E is the device space (exec_hip.h).
#pragma once
#include "atom_math.h"
#include "tape.h"

namespace dnlp {

template <class E>
struct Model {
  E* ex = nullptr;
  TapeView t;                 // what the evaluators read (tape.h)
  Tape<E>* owner = nullptr;   // the loaded tape behind `t` (host / HIP spaces); null inside the batch kernel
  // work arrays (exec space)
  VecP<E> xz;               // [x; z]
  VecP<E> dvals;
  VecP<E> hvals;
  VecP<E> w;                // Z
  VecP<E> sl;               // [sigma; lambda]
  VecP<E> Hs;               // nnzH sparse-part Hessian values
  VecP<E> tmpN;             // N scratch (dense quad_form products)
  double* dense_w = nullptr;   // control space: current weight 2*w_z of every dense block

  DNLP_HD i64 N() const { return t.N; }
  DNLP_HD i64 m() const { return t.m; }

  void init(E* e, const TapeBlob& tb) {
    owner = new Tape<E>();
    owner->load(e, tb);
    init_view(e, *owner);
  }
  void destroy() { delete owner; owner = nullptr; }

  DNLP_HD void init_view(E* e, const TapeView& v) {
    ex = e;
    t = v;
    xz = ex->template alloc<double>(static_cast<size_t>(t.N + t.Z));
    dvals = ex->template alloc<double>(static_cast<size_t>(t.nd));
    hvals = ex->template alloc<double>(static_cast<size_t>(t.nh));
    w = ex->template alloc<double>(static_cast<size_t>(t.Z));
    sl = ex->template alloc<double>(static_cast<size_t>(1 + t.m));
    Hs = ex->template alloc<double>(static_cast<size_t>(t.nnzH));
    tmpN = ex->template alloc<double>(static_cast<size_t>(t.N));
    dense_w = ex->template ctl_alloc<double>(static_cast<size_t>(t.nblk));
  }
  DNLP_HD void clear_dense_w() { for (i64 k = 0; k < t.nblk; ++k) dense_w[k] = 0.0; }

  // y = base + M v   (base may be null)
  DNLP_HD void spmv(const Csr& M, const double* v, const double* base, double* y) {
    DNLP_THIS_IN_LDS(E); DNLP_PTR_IN_LDS(E, ex);
    if constexpr (E::is_device && E::has_host_control) {
      // long rows (dense constraint blocks: C3's A is 1e3 rows of 1e4 entries): a lane per row would walk
      // 1e4 entries alone; one wavefront per row reads them coalesced
      if (M.rows > 0 && M.nnz >= 32 * M.rows) { ex->spmv_long_rows(M, v, base, y); return; }
    }
    const i64* ptr = M.ptr;
    const i32* idx = M.idx;
    const double* val = M.val;
    DNLP_VEC_IN_LDS(E, v); DNLP_VEC_IN_LDS(E, y);      // (operand and result are solver vectors; `base` and the map's values are tape data)
    ex->map(M.rows, [=] DNLP_HD(i64 r) {
      double s = base ? base[r] : 0.0;
      for (i64 k = ptr[r]; k < ptr[r + 1]; ++k) s += val[k] * v[idx[k]];
      y[r] = s;
    });
  }

  // flat (elementwise-class) sweep: one work unit per output element
  DNLP_HD void sweep_flat(const double* x, bool with_h) {
    DNLP_THIS_IN_LDS(E); DNLP_PTR_IN_LDS(E, ex);
    if (t.flat_units == 0) return;
    if constexpr (E::is_device) {
      // hand-written gfx950 kernel (exec_hip.h); the lambda below is the same arithmetic
      typename E::FlatTableT ft{t.nflat, t.flat_units, t.flat_start, t.flat_op, t.flat_a0b, t.flat_a0o, t.flat_a1b,
                                t.flat_a1o, t.flat_zoff, t.flat_doff, t.flat_hoff, t.flat_n, t.flat_d0, t.flat_d1,
                                t.flat_d2, t.flat_p, t.flat_p2, t.gidx};
      ex->sweep_flat(ft, x, xz + t.N, dvals, hvals, w, with_h);
      return;
    }
    const i64 nflat = t.nflat;
    const i64* fstart = t.flat_start;
    const i32* fop = t.flat_op;
    const i64 *a0b = t.flat_a0b, *a0o = t.flat_a0o, *a1b = t.flat_a1b, *a1o = t.flat_a1o;
    const i64 *zoff = t.flat_zoff, *doff = t.flat_doff, *hoff = t.flat_hoff, *fn = t.flat_n;
    const i64 *d0 = t.flat_d0, *d1 = t.flat_d1, *d2 = t.flat_d2;
    const double *fp = t.flat_p, *fp2 = t.flat_p2;
    const i32* gidx = t.gidx;
    double* z = xz + t.N;
    double* dv = dvals;
    double* hv = hvals;
    const double* ww = w;
    ex->map(t.flat_units, [=] DNLP_HD(i64 e) {
      // binary search the segment of unit e
      i64 lo = 0, hi = nflat;
      while (hi - lo > 1) {
        i64 mid = (lo + hi) >> 1;
        if (fstart[mid] <= e) lo = mid; else hi = mid;
      }
      const i64 s = lo;
      const i64 i = e - fstart[s];
      const int op = fop[s];
      const i64 n = fn[s];
      if (op < OP_MUL) {
        const i64 xi = a0b[s] >= 0 ? a0b[s] + i : gidx[a0o[s] + i];
        double val, g1, g2;
        unary_rules(op, x[xi], fp[s], fp2[s], val, g1, g2);
        z[zoff[s] + i] = val;
        dv[doff[s] + i] = g1;
        if (with_h) hv[hoff[s] + i] = ww[zoff[s] + i] * g2;
      } else if (op == OP_MUL) {
        // bilinear u*v: binary_operators.py:586-591 (Jacobian), :543-546 (cross Hessian)
        const i64 xi = a0b[s] >= 0 ? a0b[s] + i : gidx[a0o[s] + i];
        const i64 yi = a1b[s] >= 0 ? a1b[s] + i : gidx[a1o[s] + i];
        const double u = x[xi], v = x[yi];
        z[zoff[s] + i] = u * v;
        dv[doff[s] + i] = v;
        dv[doff[s] + n + i] = u;
        if (with_h) hv[hoff[s] + i] = ww[zoff[s] + i];
      } else if (op == OP_REL_ENTR) {
        // rel_entr.py:37-40, :129-148, :150-179
        const i64 xi = a0b[s] >= 0 ? a0b[s] + i : gidx[a0o[s] + i];
        const i64 yi = a1b[s] >= 0 ? a1b[s] + i : gidx[a1o[s] + i];
        const double u = x[xi], v = x[yi];
        const double lr = log(u / v);
        z[zoff[s] + i] = u * lr;
        dv[doff[s] + i] = lr + 1.0;
        dv[doff[s] + n + i] = -u / v;
        if (with_h) {
          const double wi = ww[zoff[s] + i];
          hv[hoff[s] + i] = wi / u;
          hv[hoff[s] + n + i] = wi * u / (v * v);
          hv[hoff[s] + 2 * n + i] = -wi / v;
        }
      } else {  // OP_MATMUL: unit = output entry (r, c) of U(mm x kk) @ V(kk x pp), F-order
        const i64 mm = d0[s], kk = d1[s];
        const i64 r = i % mm, cidx = i / mm;
        double acc = 0.0;
        const i64 dbase = doff[s] + i * kk, cnt = mm * d2[s] * kk;
        for (i64 l = 0; l < kk; ++l) {
          const i64 ui = gidx[a0o[s] + r + l * mm];
          const i64 vi = gidx[a1o[s] + l + cidx * kk];
          const double u = x[ui], v = x[vi];
          acc += u * v;
          dv[dbase + l] = v;            // d z_rc / d U_rl
          dv[dbase + cnt + l] = u;      // d z_rc / d V_lc
          if (with_h) hv[hoff[s] + i * kk + l] = ww[zoff[s] + i];
        }
        z[zoff[s] + i] = acc;
      }
    });
  }

  // reduction-class segments: quad_form (dense / sparse), quad_over_lin
  DNLP_HD void sweep_reductions(const double* x, bool with_h) {
    DNLP_THIS_IN_LDS(E); DNLP_PTR_IN_LDS(E, ex);
    for (i64 rk = 0; rk < t.nred; ++rk) {
      const SegHost& g = t.segs[t.red_segs[rk]];
      double* z = xz + t.N;
      const i32* gidx = t.gidx;
      const i64 n = g.n, a0b = g.a0_base, a0o = g.a0_off;
      double* dv = dvals + g.doff;
      double* hv = hvals + g.hoff;
      const double* ww = w;
      const i64 zo = g.zoff;
      if (g.op == OP_QUAD_FORM_DENSE) {
        // quad_form.py:41-47 (x'Px), :154-160 (2Px); Hessian block 2wP stays dense
        const double* P = t.dense_ptr[g.aux];
        if (!P) DNLP_FAIL("dense quad_form matrix not bound (dnlp_bind_dense)");
        const i64 ld = t.dense_ld[g.aux];
        const double* u = x + a0b;
        ex->gemv_sym(n, P, ld, u, tmpN);            // tmpN = P u
        const double* pu = tmpN;
        double val = ex->sum(n, [=] DNLP_HD(i64 i) { return u[i] * pu[i]; });
        ex->map(n, [=] DNLP_HD(i64 i) { dv[i] = 2.0 * pu[i]; if (i == 0) z[zo] = val; });
      } else if (g.op == OP_QUAD_FORM_SPARSE) {
        const SparseConst& sc = t.sparse[g.aux];
        const Csr P = sc.P, PT = sc.PT;
        const double* hvc = sc.hv;
        const i64 nhv = sc.nh;
        double* pu = tmpN;
        ex->map(n, [=] DNLP_HD(i64 i) {
          double s1 = 0.0, s2 = 0.0;
          for (i64 k = P.ptr[i]; k < P.ptr[i + 1]; ++k) {
            const i64 j = P.idx[k];
            s1 += P.val[k] * x[a0b >= 0 ? a0b + j : gidx[a0o + j]];
          }
          for (i64 k = PT.ptr[i]; k < PT.ptr[i + 1]; ++k) {
            const i64 j = PT.idx[k];
            s2 += PT.val[k] * x[a0b >= 0 ? a0b + j : gidx[a0o + j]];
          }
          pu[i] = s1;
          dv[i] = s1 + s2;
        });
        double val = ex->sum(n, [=] DNLP_HD(i64 i) { return x[a0b >= 0 ? a0b + i : gidx[a0o + i]] * pu[i]; });
        ex->map(with_h ? (nhv > 1 ? nhv : 1) : 1, [=] DNLP_HD(i64 i) {
          if (i == 0) z[zo] = val;
          if (with_h && i < nhv) hv[i] = ww[zo] * hvc[i];
        });
      } else if (g.op == OP_QUAD_OVER_LIN) {
        // quad_over_lin.py:38-45, :178-185, :162-173
        const i64 a1b = g.a1_base, a1o = g.a1_off;
        double ss = ex->sum(n, [=] DNLP_HD(i64 i) { double u = x[a0b >= 0 ? a0b + i : gidx[a0o + i]]; return u * u; });
        ex->map(n, [=] DNLP_HD(i64 i) {
          const double y = x[a1b >= 0 ? a1b : gidx[a1o]];
          const double u = x[a0b >= 0 ? a0b + i : gidx[a0o + i]];
          dv[i] = 2.0 * u / y;
          if (with_h) {
            const double wz = ww[zo];
            hv[i] = 2.0 * wz / y;
            hv[n + 1 + i] = -2.0 * wz * u / (y * y);
          }
          if (i == 0) {
            z[zo] = ss / y;
            dv[n] = -ss / (y * y);
            if (with_h) hv[n] = 2.0 * ww[zo] * ss / (y * y * y);
          }
        });
      }
    }
  }

  DNLP_HD void set_x(const double* x) { ex->d2d(xz, x, static_cast<size_t>(t.N) * sizeof(double)); }

  // values + first-derivative element arrays at x (x: exec space, N)
  DNLP_HD void sweep(const double* x, bool with_h) {
    DNLP_THIS_IN_LDS(E); DNLP_PTR_IN_LDS(E, ex);
    set_x(x);
    sweep_flat(xz, with_h);
    sweep_reductions(xz, with_h);
  }

  // ---- reference callback set (all pointers exec space) ------------------------
  DNLP_HD double eval_f_after_sweep() {
    DNLP_THIS_IN_LDS(E); DNLP_PTR_IN_LDS(E, ex);
    const double* cc = t.c;
    const double* v = xz;
    return t.c0 + ex->sum(t.N + t.Z, [=] DNLP_HD(i64 i) { return cc[i] * v[i]; });
  }
  DNLP_HD void eval_g_after_sweep(double* g) { spmv(t.G, xz, t.b, g); }
  DNLP_HD void eval_grad_after_sweep(double* grad) { spmv(t.Mg, dvals, t.c, grad); }
  DNLP_HD void eval_jac_after_sweep(double* jv) { spmv(t.MJ, dvals, t.Jc, jv); }

  // Hessian of sigma f + lambda.g at x; sparse part -> Hs, dense block weights -> dense_w
  DNLP_HD void eval_hess(const double* x, double sigma, const double* lambda) {
    DNLP_THIS_IN_LDS(E); DNLP_PTR_IN_LDS(E, ex);
    double* s = sl;
    const i64 mm = t.m;
    ex->map(1 + mm, [=] DNLP_HD(i64 i) { s[i] = (i == 0) ? sigma : lambda[i - 1]; });
    spmv(t.Mw, sl, nullptr, w);
    sweep(x, true);
    spmv(t.MH, hvals, nullptr, Hs);
    {
      // only the block weights are needed on the host (one scalar per dense block)
      for (i64 k = 0; k < t.nblk; ++k) {
        double wz;
        ex->d2h(&wz, w + t.blocks[k].z, sizeof(double));
        dense_w[k] = 2.0 * wz;
      }
    }
  }

  // COO Hessian values (lower triangle, fixed pattern) into `out` (exec space, nnzH)
  DNLP_HD void hess_coo(double* out) {
    DNLP_THIS_IN_LDS(E); DNLP_PTR_IN_LDS(E, ex);
    if (!t.coo_complete) DNLP_FAIL("dense quad_form block too large for a COO Hessian; use the solver-level entry points");
    ex->d2d(out, Hs, static_cast<size_t>(t.nnzH) * sizeof(double));
    for (i64 k = 0; k < t.nblk; ++k) {
      const DenseBlock& B = t.blocks[k];
      const double* P = t.dense_ptr[B.cid];
      const i64 ld = t.dense_ld[B.cid];
      const i64* pos = B.coo_pos;
      const i64 base = B.coo_base;
      const double wk = dense_w[k];
      const i64 n = B.n;
      // tril_indices order: row-major over the lower triangle
      ex->map(n * (n + 1) / 2, [=] DNLP_HD(i64 q) {
        i64 r = static_cast<i64>((sqrt(8.0 * static_cast<double>(q) + 1.0) - 1.0) * 0.5);
        while (r * (r + 1) / 2 > q) --r;
        while ((r + 1) * (r + 2) / 2 <= q) ++r;
        const i64 cidx = q - r * (r + 1) / 2;
        out[pos ? pos[q] : base + q] += wk * P[r + cidx * ld];
      });
    }
  }

  // out = W v  (W = current Lagrangian Hessian, symmetric) ; v, out: exec space N
  DNLP_HD void hess_mult(const double* v, double* out) {
    DNLP_THIS_IN_LDS(E); DNLP_PTR_IN_LDS(E, ex);
    const i64 NN = t.N;
    // sparse part: lower entries contribute to both (r,c) and (c,r); the tape's index by output (tape.h CooIdx)
    // makes it an order-fixed gather that ASSIGNS every output (no zeroing pass before it); scatter with atomics
    // into a zeroed vector only for patterns too large to index
    const i32 *hr = t.hess_rows, *hc = t.hess_cols;
    const double* hs = Hs;
    if (t.hess_sym.ptr) ex->coo_gather(t.hess_sym, hs, v, out);
    else { ex->zero(out, static_cast<size_t>(NN) * sizeof(double)); ex->coo_sym_mult(t.nnzH, hr, hc, hs, v, out); }
    for (i64 k = 0; k < t.nblk; ++k) {
      const DenseBlock& B = t.blocks[k];
      const double* P = t.dense_ptr[B.cid];
      const i64 ld = t.dense_ld[B.cid];
      ex->gemv_sym(B.n, P, ld, v + B.x0, tmpN);
      const double wk = dense_w[k];
      const double* pv = tmpN;
      double* o = out + B.x0;
      ex->map(B.n, [=] DNLP_HD(i64 i) { o[i] += wk * pv[i]; });
    }
  }

  // out(m) = J v ; out(N) = J^T v with COO values jv on the fixed pattern
  DNLP_HD void jac_mult(const double* jv, const double* v, double* out) {
    DNLP_THIS_IN_LDS(E); DNLP_PTR_IN_LDS(E, ex);
    if (t.jac_by_row.ptr) { ex->coo_gather(t.jac_by_row, jv, v, out); return; }
    ex->zero(out, static_cast<size_t>(t.m) * sizeof(double));
    if constexpr (E::is_device && E::has_host_control) if (t.jac_rect_cols > 0) { ex->rect_mult(t.m, t.jac_rect_cols, t.jac_cols, jv, v, out); return; }
    ex->coo_mult(t.nnzJ, t.jac_rows, t.jac_cols, jv, v, out, false);
  }
  DNLP_HD void jac_tmult(const double* jv, const double* v, double* out) {
    DNLP_THIS_IN_LDS(E); DNLP_PTR_IN_LDS(E, ex);
    if (t.jac_by_col.ptr) { ex->coo_gather(t.jac_by_col, jv, v, out); return; }
    ex->zero(out, static_cast<size_t>(t.N) * sizeof(double));
    if constexpr (E::is_device && E::has_host_control) if (t.jac_rect_cols > 0) { ex->rect_tmult(t.m, t.jac_rect_cols, t.jac_cols, jv, v, out); return; }
    ex->coo_mult(t.nnzJ, t.jac_rows, t.jac_cols, jv, v, out, true);
  }
};

}  // namespace dnlp
