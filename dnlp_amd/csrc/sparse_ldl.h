// Numeric phase of the static-pattern sparse LDL^T (plan: sparse_plan.h).  Single source over a
// small parallel policy P:
//   par.lanes(), par.lane()   cooperating lanes and this lane's index
//   par.sync()                all lanes have finished the previous loop
//   par.sum(v)                sum of v over the lanes, returned to every lane
// Host (test oracle): one lane, no-ops.  Batch kernel: one wavefront or one 256-lane workgroup.
// HIP space: a single-workgroup kernel (exec_hip.h).
#pragma once
#include <cstdio>
#include <cstdlib>

#include "sparse_plan.h"

namespace dnlp {

struct SeqPar {
  DNLP_HD int lanes() const { return 1; }
  DNLP_HD int lane() const { return 0; }
  DNLP_HD void sync() const {}
  DNLP_HD double sum(double v) const { return v; }
};

// vals: assembled matrix in plan layout, overwritten by (D, L).  w: 2 * maxs scratch.
template <class P>
DNLP_HD inline bool sparse_ldl_factor(const SparsePlan& pl, double* vals, double* w, int* nneg_out, int* nzero_out, P par) {
  const int L = par.lanes(), me = par.lane();
  int nneg = 0, nzero = 0;
  bool ok = true;
  for (i64 k = 0; k < pl.nblk; ++k) {
    const i64 s = pl.soff[k + 1] - pl.soff[k];
    const bool two = pl.bnode[2 * k + 1] >= 0;
    double* Lk = vals + pl.loff[k];
    double* Dk = vals + pl.doff[k];
    const i64 t0 = pl.toff[k], t1 = pl.toff[k + 1];
    if (!two) {
      double d = Dk[0];
      if (!(d == d)) ok = false;
      if (fabs(d) < 1e-300) {
#if !DNLP_DEVICE_PASS
        if (std::getenv("DNLP_SPARSE_DEBUG")) std::fprintf(stderr, "[sparse] zero 1x1 pivot: block %lld node %d (N=%lld) struct %lld\n", (long long)k, pl.bnode[2 * k], (long long)pl.N, (long long)s);
#endif
        ++nzero; d = 1e-20; }
      if (d < 0.0) ++nneg;
      const double inv = 1.0 / d;
      for (i64 i = me; i < s; i += L) w[i] = Lk[i] * inv;
      par.sync();
      if (me == 0) Dk[0] = d;
      for (i64 q = t0 + me; q < t1; q += L) vals[pl.tdst[q]] -= Lk[pl.tiu[q]] * w[pl.tiv[q]];
      par.sync();
      for (i64 i = me; i < s; i += L) Lk[i] = w[i];
    } else {
      const double a = Dk[0], c = Dk[1], e = Dk[2];
      double det = a * e - c * c;
      if (!(det == det)) ok = false;
      if (fabs(det) < 1e-300) {
#if !DNLP_DEVICE_PASS
        if (std::getenv("DNLP_SPARSE_DEBUG")) std::fprintf(stderr, "[sparse] singular 2x2 pivot: block %lld nodes %d %d (N=%lld) a=%g c=%g e=%g\n", (long long)k, pl.bnode[2 * k], pl.bnode[2 * k + 1], (long long)pl.N, a, c, e);
#endif
        ++nzero; det = -1e-20; }
      if (det < 0.0) nneg += 1;
      else if (a < 0.0 || (a == 0.0 && e < 0.0)) nneg += 2;
      const double i11 = e / det, i21 = -c / det, i22 = a / det;
      for (i64 i = me; i < s; i += L) {
        const double l1 = Lk[2 * i], l2 = Lk[2 * i + 1];
        w[2 * i] = i11 * l1 + i21 * l2;
        w[2 * i + 1] = i21 * l1 + i22 * l2;
      }
      par.sync();
      for (i64 q = t0 + me; q < t1; q += L) {
        const i64 iu = pl.tiu[q], iv = pl.tiv[q];
        vals[pl.tdst[q]] -= Lk[2 * iu] * w[2 * iv] + Lk[2 * iu + 1] * w[2 * iv + 1];
      }
      par.sync();
      for (i64 i = me; i < 2 * s; i += L) Lk[i] = w[i];
    }
    par.sync();
  }
  *nneg_out = nneg;
  *nzero_out = nzero;
  return ok;
}

// x (n entries, node numbering of the KKT system: variables then constraint rows) <- K^-1 x
template <class P>
DNLP_HD inline void sparse_ldl_solve(const SparsePlan& pl, const double* vals, double* x, P par) {
  const int L = par.lanes(), me = par.lane();
  for (i64 k = 0; k < pl.nblk; ++k) {
    const i64 s0 = pl.soff[k], s = pl.soff[k + 1] - s0;
    const i32 u0 = pl.bnode[2 * k], u1 = pl.bnode[2 * k + 1];
    const double* Lk = vals + pl.loff[k];
    if (u1 < 0) {
      const double xp = x[u0];
      if (xp != 0.0)
        for (i64 i = me; i < s; i += L) x[pl.sidx[s0 + i]] -= Lk[i] * xp;
    } else {
      const double x0 = x[u0], x1 = x[u1];
      for (i64 i = me; i < s; i += L) x[pl.sidx[s0 + i]] -= Lk[2 * i] * x0 + Lk[2 * i + 1] * x1;
    }
    par.sync();
  }
  for (i64 k = me; k < pl.nblk; k += L) {
    const i32 u0 = pl.bnode[2 * k], u1 = pl.bnode[2 * k + 1];
    const double* Dk = vals + pl.doff[k];
    if (u1 < 0) {
      x[u0] /= Dk[0];
    } else {
      const double a = Dk[0], c = Dk[1], e = Dk[2];
      double det = a * e - c * c;
      if (fabs(det) < 1e-300) det = -1e-20;
      const double x0 = x[u0], x1 = x[u1];
      x[u0] = (e * x0 - c * x1) / det;
      x[u1] = (a * x1 - c * x0) / det;
    }
  }
  par.sync();
  for (i64 k = pl.nblk - 1; k >= 0; --k) {
    const i64 s0 = pl.soff[k], s = pl.soff[k + 1] - s0;
    if (s == 0) continue;
    const i32 u0 = pl.bnode[2 * k], u1 = pl.bnode[2 * k + 1];
    const double* Lk = vals + pl.loff[k];
    double a0 = 0.0, a1 = 0.0;
    if (u1 < 0) {
      for (i64 i = me; i < s; i += L) a0 += Lk[i] * x[pl.sidx[s0 + i]];
      a0 = par.sum(a0);
      if (me == 0) x[u0] -= a0;
    } else {
      for (i64 i = me; i < s; i += L) { const double xi = x[pl.sidx[s0 + i]]; a0 += Lk[2 * i] * xi; a1 += Lk[2 * i + 1] * xi; }
      a0 = par.sum(a0);
      a1 = par.sum(a1);
      if (me == 0) { x[u0] -= a0; x[u1] -= a1; }
    }
    par.sync();
  }
}

}  // namespace dnlp
