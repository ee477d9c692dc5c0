// Template-specialised batch solver, part 1: the PLAN of a parametrised template.
//
// The generic in-kernel solver (batch.h: Ipm<BlockExecT> over exec_block.h) walks the tape through the same 64-bit CSR
// / segment tables as the host-driven spaces, keeps its solver objects behind pointers and compiles to ~28 k instructions
// per interior-point iteration of a 102-variable KKT system.  The wavefront solver (wave_ipm.h) is a second, fixed-shape
// restatement of the same algorithm for SMALL SPARSE templates (the C5 members of BASELINE.json: what the serial best_of
// / re-solve loop of cvxpy/problems/problem.py:1256-1269 runs one by one).  Everything it reads that does not depend on
// the instance is flattened ONCE per template, on the host, into one block of 32-bit integers — this file:
//   * the tape's flat sweep as one row per work unit (op, argument, outputs, parameter row): no segment search
//   * the constant CSR maps G / Mg / MJ / Mw / MH with 32-bit row pointers
//   * the Jacobian / Hessian patterns and the three products by output (tape.h CooIdx)
//   * the static-pattern LDL^T program (sparse_plan.h)
//   * the layout of an instance's data row (batch.h BatchLayout) and of its solver state
// A workgroup copies the block into LDS once; every wavefront then solves instance after instance out of it.
// The block is plain data: the same bytes drive the host instantiation that the CPU tests compare with the oracle.
#pragma once
#include <stdexcept>
#include <vector>

#include "atom_math.h"
#include "sparse_plan.h"
#include "wave_hdr.h"

namespace dnlp {

struct WaveLayoutIn { i64 c0, c, b, Jc, G, Mg, Mw, MJ, MH, fp, fp2, x0, lb, ub, cl, cu, total; };

// the data row of an instance as the batch kernels read it (batch.h: per-segment parameters expanded to per-flat-row ones)
template <class E>
inline WaveLayoutIn wave_layout_of(const Tape<E>& t) {
  WaveLayoutIn l;
  i64 o = 0;
  auto take = [&](i64& f, i64 n) { f = o; o += n; };
  take(l.c0, 1); take(l.c, t.N + t.Z); take(l.b, t.m); take(l.Jc, t.nnzJ);
  take(l.G, t.G.nnz); take(l.Mg, t.Mg.nnz); take(l.Mw, t.Mw.nnz); take(l.MJ, t.MJ.nnz); take(l.MH, t.MH.nnz);
  take(l.fp, t.nflat); take(l.fp2, t.nflat);
  take(l.x0, t.N); take(l.lb, t.N); take(l.ub, t.N); take(l.cl, t.m); take(l.cu, t.m);
  l.total = o;
  return l;
}

// doubles of per-instance solver state (the order of wave_ipm.h w_layout)
inline i64 wave_state_doubles(i64 N, i64 m, i64 Z, i64 nd, i64 nh, i64 nnzJ, i64 nnzH, i64 nvals, i64 nblk, i64 scr, i64 nunits) {
  auto ev = [](i64 n) { return (n + 1) & ~static_cast<i64>(1); };      // 16-byte granules
  // (wave_ipm.h layout: the factorisation's work arrays share the region of what is dead while it runs, when they fit)
  const i64 dead = 4 * ev(N) + 7 * ev(m) + 6 * ev(N + m), work = ev(nvals + 3 * nblk + 8) + ev(scr);
  (void)nunits;
  const i64 hess = (nh <= N + m && Z <= N + m) ? 0 : ev(nh) + ev(Z) + ev(1 + m);      // (eval_hessian's transients borrow rhs / sol / res)
  return 10 * ev(N) + 13 * ev(m) + ev(nnzJ) + ev(Z) + ev(nd) + hess + ev(nnzH) + ev(nvals) + dead +
         (work <= dead ? 0 : work);
}

// doubles of global memory a wavefront parks an iterate in (wave_ipm.h polish)
inline i64 wave_park_doubles(i64 N, i64 m) { auto ev = [](i64 n) { return (n + 1) & ~static_cast<i64>(1); }; return 3 * ev(N) + 4 * ev(m); }

// Why a template cannot take the wavefront solver ("" = it can).  The generic kernel stays the solver of everything else.
template <class E>
inline const char* wave_plan_refusal(const Tape<E>& t, const SparsePlanHost* sp) {
  if (!sp) return "no static-pattern sparse plan";
  if (t.nblk > 0 || t.ndense > 0) return "dense quad_form blocks";
  if (t.nred > 0) return "reduction-class segments (quad_form / quad_over_lin)";
  for (const SegHost& g : t.h_segs)
    if (!(g.op < OP_MUL || g.op == OP_MUL || g.op == OP_REL_ENTR || g.op == OP_MATMUL)) return "an atom outside the elementwise / matmul set";
  if (sp->tail_n > 0 || sp->panels_dropped) return "plan with a dense tail";
  if (t.N + t.m > 4096 || t.nnzJ > 30000 || t.nnzH > 30000) return "too large for one wavefront";
  if (!t.jac_by_row.ptr || !t.jac_by_col.ptr || !t.hess_sym.ptr) return "no product index";
  return "";
}

// The block.  E is the execution space the tape was loaded into (its CSR index arrays live there: copied back once).
template <class E>
// (allow_tail = false: no dense tail in registers — the workgroup-per-instance kernel runs a short chain of one-block levels
//  as narrow generated phases instead: the register code was written for one wavefront)
inline std::vector<i32> build_wave_plan(E* ex, const Tape<E>& t, const SparsePlanHost& sp, const WaveLayoutIn& lay, bool allow_tail = true) {
  WaveHdr h;
  std::memset(&h, 0, sizeof h);
  std::vector<i32> out(sizeof(WaveHdr) / 4, 0);
  auto put = [&](const std::vector<i32>& v) { const i32 at = static_cast<i32>(out.size()); out.insert(out.end(), v.begin(), v.end()); if (out.size() & 1) out.push_back(0); return at; };
  auto narrow = [](i64 v) { if (v < -2147483647LL || v > 2147483647LL) throw std::runtime_error("wave plan: index beyond 32 bits"); return static_cast<i32>(v); };
  auto down32 = [&](const i32* d, i64 n) { std::vector<i32> v(static_cast<size_t>(n)); if (n) ex->d2h(v.data(), d, sizeof(i32) * static_cast<size_t>(n)); return v; };
  auto down64 = [&](const i64* d, i64 n) { std::vector<i64> w(static_cast<size_t>(n)); if (n) ex->d2h(w.data(), d, sizeof(i64) * static_cast<size_t>(n)); std::vector<i32> v(w.size()); for (size_t k = 0; k < w.size(); ++k) v[k] = narrow(w[k]); return v; };
  ex->sync();
  h.N = narrow(t.N); h.m = narrow(t.m); h.Z = narrow(t.Z); h.nd = narrow(t.nd); h.nh = narrow(t.nh); h.nnzJ = narrow(t.nnzJ); h.nnzH = narrow(t.nnzH);
  // ---- sweep units: one row per output element of every flat segment (model.h sweep_flat without its segment search)
  {
    i64 gcount = 0;                       // entries of the tape's gather table that the flat segments reach
    for (size_t f = 0; f < t.h_flat_seg.size(); ++f) {
      const SegHost& g = t.h_segs[static_cast<size_t>(t.h_flat_seg[f])];
      if (g.op == OP_MATMUL) { gcount = std::max(gcount, std::max(g.a0_off + g.d0 * g.d1, g.a1_off + g.d1 * g.d2)); continue; }
      if (g.a0_base < 0) gcount = std::max(gcount, g.a0_off + g.n);
      if ((g.op == OP_MUL || g.op == OP_REL_ENTR) && g.a1_base < 0) gcount = std::max(gcount, g.a1_off + g.n);
    }
    const std::vector<i32> gidx = down32(t.gidx, gcount);
    std::vector<i32> op, a0, a1, z, d0, d1, hh, pp, mm;
    for (size_t f = 0; f < t.h_flat_seg.size(); ++f) {
      const SegHost& g = t.h_segs[static_cast<size_t>(t.h_flat_seg[f])];
      const bool two = g.op == OP_MUL || g.op == OP_REL_ENTR;
      if (g.op == OP_MATMUL) {
        // unit = output entry (r, c) of U (mm x kk) @ V (kk x pp), F-order (model.h sweep_flat): a0 = start of its kk index
        // pairs in mm_idx, a1 = kk, d0 / d1 = its dz/dU and dz/dV runs, h = its Hessian run
        const i64 mmr = g.d0, kk = g.d1, cnt = mmr * g.d2 * kk;
        for (i64 i = 0; i < mmr * g.d2; ++i) {
          const i64 r = i % mmr, cidx = i / mmr;
          op.push_back(g.op);
          a0.push_back(narrow(static_cast<i64>(mm.size())));
          a1.push_back(narrow(kk));
          for (i64 l = 0; l < kk; ++l) {
            mm.push_back(gidx[static_cast<size_t>(g.a0_off + r + l * mmr)]);
            mm.push_back(gidx[static_cast<size_t>(g.a1_off + l + cidx * kk)]);
          }
          z.push_back(narrow(g.zoff + i));
          d0.push_back(narrow(g.doff + i * kk));
          d1.push_back(narrow(g.doff + i * kk + cnt));
          hh.push_back(narrow(g.hoff + i * kk));
          pp.push_back(narrow(static_cast<i64>(f)));
        }
        continue;
      }
      for (i64 i = 0; i < g.n; ++i) {
        op.push_back(g.op);
        a0.push_back(narrow(g.a0_base >= 0 ? g.a0_base + i : gidx[static_cast<size_t>(g.a0_off + i)]));
        a1.push_back(two ? narrow(g.a1_base >= 0 ? g.a1_base + i : gidx[static_cast<size_t>(g.a1_off + i)]) : -1);
        z.push_back(narrow(g.zoff + i));
        d0.push_back(narrow(g.doff + i));
        d1.push_back(two ? narrow(g.doff + g.n + i) : -1);
        hh.push_back(narrow(g.hoff + i));
        pp.push_back(g.op == OP_REL_ENTR ? narrow(g.n) : narrow(static_cast<i64>(f)));     // (rel_entr: the stride of its three Hessian runs)
      }
    }
    h.mm_idx = put(mm);
    h.nunits = narrow(static_cast<i64>(op.size()));
    h.u_op = put(op); h.u_a0 = put(a0); h.u_a1 = put(a1); h.u_z = put(z); h.u_d0 = put(d0); h.u_d1 = put(d1); h.u_h = put(hh); h.u_p = put(pp);
  }
  h.jac_rows = put(t.h_jac_rows); h.jac_cols = put(t.h_jac_cols); h.hess_rows = put(t.h_hess_rows); h.hess_cols = put(t.h_hess_cols);
  h.jac_rowptr = put(down64(t.jac_rowptr, t.m + 1));
  h.sp_nblk = narrow(sp.nblk()); h.sp_nvals = narrow(sp.nvals); h.sp_nlev = narrow(static_cast<i64>(sp.lev_off.size()) - 1);
  h.sp_ngrp = narrow(static_cast<i64>(sp.gdst.size())); h.sp_nfwd = narrow(static_cast<i64>(sp.fnode.size())); h.sp_ntrip = narrow(sp.ntrip);
  h.sp_rows = narrow(static_cast<i64>(sp.sidx.size()));
  {
    // dense tail (see WaveHdr): the longest chain of one-block 1x1 levels at the end whose trailing matrix is dense
    const i32 nlev = static_cast<i32>(sp.lev_off.size()) - 1;
    i32 Ls = nlev;
    auto one_by_one = [&](i32 lev) { const i32 b = sp.lev_off[static_cast<size_t>(lev)]; return sp.lev_off[static_cast<size_t>(lev) + 1] - b == 1 && sp.bnode[static_cast<size_t>(2 * b + 1)] < 0; };
    while (Ls > 0 && one_by_one(Ls - 1)) --Ls;
    i32 T = nlev - Ls;
    const i32 kTailMax = 32;
    if (T > kTailMax) { Ls = nlev - kTailMax; T = kTailMax; }
    std::vector<i32> tnode, td, tl, tfq, tfp;
    bool ok = allow_tail && T >= 3 && !std::getenv("DNLP_WAVE_NO_TAIL");
    if (ok) {
      std::vector<i32> pos(static_cast<size_t>(sp.n), -1);
      for (i32 t = 0; t < T; ++t) {
        const i32 b = sp.lev_off[static_cast<size_t>(Ls + t)];
        tnode.push_back(sp.bnode[static_cast<size_t>(2 * b)]);
        td.push_back(sp.doff[static_cast<size_t>(b)]);
        pos[static_cast<size_t>(tnode.back())] = t;
      }
      tl.assign(static_cast<size_t>(T) * T, -1);
      for (i32 t = 0; t < T && ok; ++t) {
        const i32 b = sp.lev_off[static_cast<size_t>(Ls + t)];
        i32 prev = t;
        for (i32 r = sp.soff[static_cast<size_t>(b)]; r < sp.soff[static_cast<size_t>(b) + 1]; ++r) {
          const i32 i = pos[static_cast<size_t>(sp.sidx[static_cast<size_t>(r)])];
          if (i <= prev) { ok = false; break; }              // struct rows: later tail nodes, ascending
          prev = i;
          tl[static_cast<size_t>(i) * T + t] = sp.loff[static_cast<size_t>(b)] + (r - sp.soff[static_cast<size_t>(b)]);
        }
        for (i32 i = t + 1; i < T && ok; ++i) if (tl[static_cast<size_t>(i) * T + t] < 0) ok = false;     // dense
      }
      // the forward gathers of the tail targets: one target per tail level; its entries from blocks before the tail come
      // first (ascending by source block), then the tail's own rows k = 0 .. t - 1 in order
      tfp.push_back(0);
      for (i32 t = 0; t < T && ok; ++t) {
        const i32 h0 = sp.lev_f[static_cast<size_t>(Ls + t)], h1 = sp.lev_f[static_cast<size_t>(Ls + t) + 1];
        if (Ls + t == 0) { if (h1 != h0) ok = false; tfp.push_back(static_cast<i32>(tfq.size())); continue; }
        if (h1 - h0 != 1 || sp.fnode[static_cast<size_t>(h0)] != tnode[static_cast<size_t>(t)]) { ok = false; break; }
        i32 q = sp.foff[static_cast<size_t>(h0)];
        const i32 q1 = sp.foff[static_cast<size_t>(h0) + 1];
        for (; q < q1 && pos[static_cast<size_t>(sp.fu0[static_cast<size_t>(q)])] < 0; ++q) tfq.push_back(q);
        for (i32 k = 0; k < t && ok; ++k, ++q)
          if (q >= q1 || sp.fa[static_cast<size_t>(q)] != tl[static_cast<size_t>(t) * T + k] || pos[static_cast<size_t>(sp.fu0[static_cast<size_t>(q)])] != k) ok = false;
        if (q != q1) ok = false;
        tfp.push_back(static_cast<i32>(tfq.size()));
      }
    }
    // (the tail's forward products use the five consecutive row-sized arrays dy dvL dvU st gt — dead during every solve — as scratch)
    if (ok && static_cast<i64>(tfq.size()) > 5 * ((static_cast<i64>(t.m) + 1) & ~static_cast<i64>(1))) ok = false;
    if (!ok) { T = 0; Ls = nlev; tnode.clear(); td.clear(); tl.clear(); tfq.clear(); tfp.assign(1, 0); }
    // Order of the static-pattern LDL^T's tables in the block: what every kernel reads first (assembly positions, the dense
    // tail's tables, the forward rows the tail gathers through), then what a kernel with generated phases (wave_gen.h) no
    // longer reads — the constant CSR maps, the products by output, the level machinery: such a kernel stages only the first
    // keep_gen ints in LDS.
    h.tail_L = Ls; h.tail_T = T;
    h.hpos = put(sp.hpos); h.jpos = put(sp.jpos); h.dpos = put(sp.dpos);
    h.t_node = put(tnode); h.t_d = put(td); h.t_l = put(tl); h.t_fq = put(tfq); h.t_fp = put(tfp);
    h.t_nf = narrow(static_cast<i64>(tfq.size()));
    if (T == 0) h.keep_gen = narrow(static_cast<i64>(out.size()));
    h.fa = put(sp.fa); h.fu0 = put(sp.fu0); h.fu1 = put(sp.fu1);
    if (T > 0) h.keep_gen = narrow(static_cast<i64>(out.size()));
  }
  // (... and the constant CSR maps and the products by output, which such a kernel also runs as generated phases)
  auto csr = [&](const Csr& M, i32& optr, i32& oidx) { optr = put(down64(M.ptr, M.rows + 1)); oidx = put(down32(M.idx, M.nnz)); };
  csr(t.G, h.G_ptr, h.G_idx); csr(t.Mg, h.Mg_ptr, h.Mg_idx); csr(t.MJ, h.MJ_ptr, h.MJ_idx); csr(t.Mw, h.Mw_ptr, h.Mw_idx); csr(t.MH, h.MH_ptr, h.MH_idx);
  auto coo = [&](const CooIdx& c, i32& p, i32& e, i32& s, i32& hv, i32& nh) {
    p = put(down32(c.ptr, c.nout + 1)); e = put(down32(c.ent, c.total)); s = put(down32(c.src, c.total)); hv = put(down32(c.heavy, c.nheavy)); nh = narrow(c.nheavy);
  };
  coo(t.jac_by_row, h.jr_ptr, h.jr_ent, h.jr_src, h.jr_heavy, h.jr_nheavy);
  coo(t.jac_by_col, h.jc_ptr, h.jc_ent, h.jc_src, h.jc_heavy, h.jc_nheavy);
  coo(t.hess_sym, h.hs_ptr, h.hs_ent, h.hs_src, h.hs_heavy, h.hs_nheavy);
  h.bnode = put(sp.bnode); h.soff = put(sp.soff); h.loff = put(sp.loff); h.doff = put(sp.doff); h.lev_off = put(sp.lev_off);
  h.sblk = put(sp.sblk); h.sidx = put(sp.sidx); h.lev_f = put(sp.lev_f); h.fnode = put(sp.fnode); h.foff = put(sp.foff);
  h.lev_g = put(sp.lev_g); h.gdst = put(sp.gdst); h.goff = put(sp.goff);
  {
    const size_t nl = sp.lev_off.size();
    std::vector<i32> lr(nl), lt(nl), lfe(nl);
    for (size_t l = 0; l < nl; ++l) {
      lr[l] = sp.soff[static_cast<size_t>(sp.lev_off[l])];
      lt[l] = sp.goff[static_cast<size_t>(sp.lev_g[l])];
      lfe[l] = sp.foff[static_cast<size_t>(sp.lev_f[l])];
    }
    h.lev_r = put(lr); h.lev_t = put(lt); h.lev_fe = put(lfe);
  }
  h.tau = put(sp.tau); h.tav = put(sp.tav);
  h.l_c0 = narrow(lay.c0); h.l_c = narrow(lay.c); h.l_b = narrow(lay.b); h.l_Jc = narrow(lay.Jc); h.l_G = narrow(lay.G); h.l_Mg = narrow(lay.Mg);
  h.l_Mw = narrow(lay.Mw); h.l_MJ = narrow(lay.MJ); h.l_MH = narrow(lay.MH); h.l_fp = narrow(lay.fp); h.l_fp2 = narrow(lay.fp2);
  h.l_x0 = narrow(lay.x0); h.l_lb = narrow(lay.lb); h.l_ub = narrow(lay.ub); h.l_cl = narrow(lay.cl); h.l_cu = narrow(lay.cu); h.l_total = narrow(lay.total);
  {
    // the largest phase of products: a level's update triples (wave_ipm.h ldl_factor)
    i64 scr = 1;
    const size_t nlev = sp.lev_off.size() - 1;
    for (size_t lev = 0; lev < nlev; ++lev)
      scr = std::max<i64>(scr, sp.goff[static_cast<size_t>(sp.lev_g[lev + 1])] - sp.goff[static_cast<size_t>(sp.lev_g[lev])]);
    h.scr_doubles = narrow(scr);
  }
  h.state_doubles = narrow(wave_state_doubles(t.N, t.m, t.Z, t.nd, t.nh, t.nnzJ, t.nnzH, sp.nvals, sp.nblk(), h.scr_doubles, h.nunits));
  h.total = narrow(static_cast<i64>(out.size()));
  std::memcpy(out.data(), &h, sizeof h);
  return out;
}

}  // namespace dnlp
