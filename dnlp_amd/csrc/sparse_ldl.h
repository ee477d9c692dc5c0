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
  template <class T> DNLP_HD T* vec(T* p) const { return p; }     // (a policy may know in which address space the vectors live)
};

// work: nvals (w, laid out like the L values) + 3 * nblk (inverse pivot blocks)
//       (+ with a dense tail: the accumulator T, tail_ld x tail_n, and the two panels, tail_ld x pg_maxcols each)
DNLP_HD inline i64 sparse_ldl_tail_acc_offset(const SparsePlan& pl) { return (pl.nvals + 3 * pl.nblk + 8 + 7) / 8 * 8; }
DNLP_HD inline i64 sparse_ldl_work_doubles(const SparsePlan& pl) {
  if (pl.tail_n <= 0) return pl.nvals + 3 * pl.nblk + 8;
  return sparse_ldl_tail_acc_offset(pl) + pl.tail_ld * pl.tail_n + 2 * pl.tail_ld * (pl.pg_maxcols > 0 ? pl.pg_maxcols : 1);
}

// ---- per-item bodies of the level phases (shared by the cooperative routine below and by the
// grid-wide level kernels of the HIP space) ------------------------------------------------------
// A: invert the pivot block k; returns inertia contribution through the three counters
DNLP_HD inline void sp_pivot(const SparsePlan& pl, double* vals, double* dinv, i64 k, double& nneg, double& nzero, double& bad) {
  double* Dk = vals + pl.doff[k];
  double* di = dinv + 3 * k;
  if (pl.bnode[2 * k + 1] < 0) {
    double d = Dk[0];
    if (!(d == d)) bad += 1.0;
    if (fabs(d) < 1e-300) {
#if !DNLP_DEVICE_PASS
      if (std::getenv("DNLP_SPARSE_DEBUG")) std::fprintf(stderr, "[sparse] zero 1x1 pivot: block %lld node %d (N=%lld)\n", (long long)k, pl.bnode[2 * k], (long long)pl.N);
#endif
      nzero += 1.0; d = 1e-20; Dk[0] = d;
    }
    if (d < 0.0) nneg += 1.0;
    di[0] = 1.0 / d;
  } else {
    const double a = Dk[0], c = Dk[1], e = Dk[2];
    double det = a * e - c * c;
    if (!(det == det)) bad += 1.0;
    if (fabs(det) < 1e-300) {
#if !DNLP_DEVICE_PASS
      if (std::getenv("DNLP_SPARSE_DEBUG")) std::fprintf(stderr, "[sparse] singular 2x2 pivot: block %lld nodes %d %d (N=%lld) a=%g c=%g e=%g\n", (long long)k, pl.bnode[2 * k], pl.bnode[2 * k + 1], (long long)pl.N, a, c, e);
#endif
      nzero += 1.0; det = -1e-20;
    }
    if (det < 0.0) nneg += 1.0;
    else if (a < 0.0 || (a == 0.0 && e < 0.0)) nneg += 2.0;
    di[0] = e / det; di[1] = -c / det; di[2] = a / det;
  }
}
// B: struct row r: L = l D^-1 goes into vals (its final place), the unscaled l into w (the update phase multiplies the
// two: no separate "keep L" pass over the level's values afterwards)
DNLP_HD inline void sp_scale(const SparsePlan& pl, double* vals, double* w, const double* dinv, i64 r) {
  const i64 k = pl.sblk[r], i = r - pl.soff[k];
  const double* di = dinv + 3 * k;
  if (pl.bnode[2 * k + 1] < 0) {
    const i64 a = pl.loff[k] + i;
    const double l1 = vals[a];
    w[a] = l1;
    vals[a] = l1 * di[0];
  } else {
    const i64 a = pl.loff[k] + 2 * i;
    const double l1 = vals[a], l2 = vals[a + 1];
    w[a] = l1; w[a + 1] = l2;
    vals[a] = di[0] * l1 + di[1] * l2;
    vals[a + 1] = di[1] * l1 + di[2] * l2;
  }
}
// C: value of update triple q: l_u (unscaled, in w) times (l D^-1)_v (in vals), direct addresses (sparse_plan.h tau / tav)
DNLP_HD inline double sp_update(const SparsePlan& pl, const double* vals, const double* w, i64 q) {
  const i32 au = pl.tau[q], av = pl.tav[q];
  if (av >= 0) return w[au] * vals[av];
  const i32 bv = ~av;
  return w[au] * vals[bv] + w[au + 1] * vals[bv + 1];
}
// forward substitution: contribution of gathered row q (sparse_plan.h fa / fu0 / fu1)
DNLP_HD inline double sp_fwd(const SparsePlan& pl, const double* vals, const double* x, i64 q) {
  const i32 a = pl.fa[q];
  if (a >= 0) return vals[a] * x[pl.fu0[q]];
  const i32 b = ~a;
  return vals[b] * x[pl.fu0[q]] + vals[b + 1] * x[pl.fu1[q]];
}
// D^-1 on block k
DNLP_HD inline void sp_dsolve(const SparsePlan& pl, const double* vals, double* x, i64 k) {
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

// vals: assembled matrix in plan layout, overwritten by (D, L).  Level by level: the blocks of an
// elimination-tree level are independent, so their pivots are inverted together, their rows scaled
// together and their update triples applied together.  The triples of a level are stored sorted by
// destination (sparse_plan.h: gdst / goff / lev_g): a destination is summed by ONE lane in storage order —
// or, on a narrow level with long groups, by all lanes and par.sum's fixed tree — and subtracted once.
// No floating-point atomics: the same bits on every run.
template <class P>
DNLP_HD inline bool sparse_ldl_factor(const SparsePlan& pl, double* vals, double* work, int* nneg_out, int* nzero_out, P par) {
  const int L = par.lanes(), me = par.lane();
  vals = par.vec(vals); work = par.vec(work);
  double* w = work;
  double* dinv = work + pl.nvals;
  double nneg = 0.0, nzero = 0.0, bad = 0.0;
  double* Tacc = work + sparse_ldl_tail_acc_offset(pl);
  double* Pl = Tacc + pl.tail_ld * pl.tail_n;
  double* Pw = Pl + pl.tail_ld * pl.pg_maxcols;
  if (pl.tail_n > 0) {
    for (i64 a = me; a < pl.tail_ld * pl.tail_n; a += L) Tacc[a] = 0.0;
    par.sync();
  }
  for (i64 lev = 0; lev < pl.nlev_run; ++lev) {
    const i64 b0 = pl.lev_off[lev], b1 = pl.lev_off[lev + 1];
    const i64 r0 = pl.soff[b0], r1 = pl.soff[b1];
    if (r1 - r0 <= 8 * (b1 - b0)) {
      // short structs: a lane inverts its block's pivot and scales the block's rows in one phase
      for (i64 k = b0 + me; k < b1; k += L) {
        sp_pivot(pl, vals, dinv, k, nneg, nzero, bad);
        for (i64 r = pl.soff[k]; r < pl.soff[k + 1]; ++r) sp_scale(pl, vals, w, dinv, r);
      }
      par.sync();
    } else {
      // A: pivot blocks
      for (i64 k = b0 + me; k < b1; k += L) sp_pivot(pl, vals, dinv, k, nneg, nzero, bad);
      par.sync();
      // B: L = l D^-1 for every struct row of the level
      for (i64 r = r0 + me; r < r1; r += L) sp_scale(pl, vals, w, dinv, r);
      par.sync();
    }
    // C': the level's panel blocks update the dense tail with one product (see SparsePlan)
    if (pl.tail_n > 0 && pl.pg_cols[lev] > 0) {
      const i64 cols = pl.pg_cols[lev], ldt = pl.tail_ld, r = pl.tail_n;
      for (i64 a = me; a < ldt * cols; a += L) { Pl[a] = 0.0; Pw[a] = 0.0; }
      par.sync();
      for (i64 q = pl.pg_off[lev] + me; q < pl.pg_off[lev + 1]; q += L) { Pl[pl.pg_dst[q]] = w[pl.pg_src[q]]; Pw[pl.pg_dst[q]] = vals[pl.pg_src[q]]; }
      par.sync();
      for (i64 e = me; e < r * r; e += L) {
        const i64 u = e % r, v = e / r;
        if (u < v) continue;
        double acc = 0.0;
        for (i64 c = 0; c < cols; ++c) acc += Pl[u + c * ldt] * Pw[v + c * ldt];
        Tacc[u + v * ldt] -= acc;
      }
      par.sync();
    }
    // C: Schur-complement updates of the level, one destination at a time
    {
      const i64 g0 = pl.lev_g[lev], g1 = pl.lev_g[lev + 1];
      const i64 ngr = g1 - g0, ntr = pl.goff[g1] - pl.goff[g0];
      if (ngr * 8 <= L && ntr >= 16 * ngr) {
        // few destinations, long groups (a separator's diagonal block under many children): all lanes per group
        for (i64 g = g0; g < g1; ++g) {
          double acc = 0.0;
          for (i64 q = pl.goff[g] + me; q < pl.goff[g + 1]; q += L) acc += sp_update(pl, vals, w, q);
          acc = par.sum(acc);
          if (me == 0) vals[pl.gdst[g]] -= acc;
        }
      } else {
        for (i64 g = g0 + me; g < g1; g += L) {
          double acc = 0.0;
          for (i64 q = pl.goff[g]; q < pl.goff[g + 1]; ++q) acc += sp_update(pl, vals, w, q);
          vals[pl.gdst[g]] -= acc;
        }
      }
    }
    par.sync();
  }
  nneg = par.sum(nneg);
  nzero = par.sum(nzero);
  bad = par.sum(bad);
  *nneg_out = static_cast<int>(nneg);
  *nzero_out = static_cast<int>(nzero);
  return bad == 0.0;
}

// x (n entries, node numbering of the KKT system: variables then constraint rows) <- K^-1 x
template <class P>
DNLP_HD inline void sparse_ldl_solve(const SparsePlan& pl, const double* vals, double* x, P par) {
  const int L = par.lanes(), me = par.lane();
  vals = par.vec(vals); x = par.vec(x);
  // forward, level by level, in GATHER form: a node of level lev collects the struct rows that point at it (all from
  // blocks of lower levels, so their x entries are final), in ascending row order, and is final itself afterwards
  // (a plan with a dense tail runs the levels before it in two calls: solve_phase 1, the tail's dense solve, then 2;
  //  the tail's nodes then gather only the rows of the blocks before the tail, in one last phase)
  if (pl.solve_phase != 2) {
  for (i64 lev = 1; lev <= pl.nlev_run; ++lev) {
    const bool last = lev == pl.nlev_run;
    if (last && pl.nlev_run == pl.nlev) break;
    const i64 h0 = pl.lev_f[lev], h1 = last ? pl.lev_f[pl.nlev] : pl.lev_f[lev + 1];
    if (h1 == h0) continue;
    const i64 nh = h1 - h0, nrw = pl.foff[h1] - pl.foff[h0];
    if (nh * 8 <= L && nrw > 2 * nh) {
      // few targets (the one-block levels of a dense separator chain: circle packing's last 21 levels have one node
      // each, under up to 20 rows): all lanes gather a target's rows and par.sum's fixed tree adds them — one lane
      // walking the rows is a chain of dependent loads per row
      for (i64 h = h0; h < h1; ++h) {
        const i64 q1 = pl.fend ? pl.fend[h] : pl.foff[h + 1];
        double acc = 0.0;
        for (i64 q = pl.foff[h] + me; q < q1; q += L) acc += sp_fwd(pl, vals, x, q);
        acc = par.sum(acc);
        if (me == 0) x[pl.fnode[h]] -= acc;
      }
    } else {
      for (i64 h = h0 + me; h < h1; h += L) {
        const i64 q1 = pl.fend ? pl.fend[h] : pl.foff[h + 1];
        double acc = 0.0;
        for (i64 q = pl.foff[h]; q < q1; ++q) acc += sp_fwd(pl, vals, x, q);
        x[pl.fnode[h]] -= acc;
      }
    }
    par.sync();
  }
  for (i64 k = me; k < pl.nblk_run; k += L) sp_dsolve(pl, vals, x, k);
  par.sync();
  }
  if (pl.solve_phase == 1) return;
  // backward, levels descending: a block gathers from its (already final) ancestors.  Wide levels:
  // one lane per block; narrow levels near the root (few blocks, long structs): all lanes per block.
  for (i64 lev = pl.nlev_run - 1; lev >= 0; --lev) {
    const i64 b0 = pl.lev_off[lev], b1 = pl.lev_off[lev + 1];
    // (a narrow level only pays for the all-lanes-per-block form when its structs are long compared
    // with the number of blocks: a lane-wide reduction per block costs about as much as two serial
    // struct entries, and the blocks of the level are then handled one after the other)
    const i64 nbl = b1 - b0, nrw = pl.soff[b1] - pl.soff[b0];
    if (nbl * 4 >= L || nrw <= 3 * nbl * nbl) {
      for (i64 k = b0 + me; k < b1; k += L) {
        const i64 s0 = pl.soff[k], s = pl.soff[k + 1] - s0;
        const i32 u0 = pl.bnode[2 * k], u1 = pl.bnode[2 * k + 1];
        const double* Lk = vals + pl.loff[k];
        double a0 = 0.0, a1 = 0.0;
        if (u1 < 0) { for (i64 i = 0; i < s; ++i) a0 += Lk[i] * x[pl.sidx[s0 + i]]; x[u0] -= a0; }
        else {
          for (i64 i = 0; i < s; ++i) { const double xi = x[pl.sidx[s0 + i]]; a0 += Lk[2 * i] * xi; a1 += Lk[2 * i + 1] * xi; }
          x[u0] -= a0; x[u1] -= a1;
        }
      }
    } else {
      for (i64 k = b0; k < b1; ++k) {
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
      }
    }
    par.sync();
  }
}

}  // namespace dnlp
