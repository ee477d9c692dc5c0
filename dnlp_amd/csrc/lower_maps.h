// Host side of the lowering (the role of cvxcore's build_matrix, cvxpy/cvxcore/src/cvxcore.cpp:161-215, and of
// the reference's per-callback COO bookkeeping, nlp_solvers/nlp_solver.py:246-276, 337-372): from the linear forms
// and the derivative triplets the Python DAG walk emits, build ONCE what every later oracle call multiplies with —
//
//   G   canonical CSR of the constraint rows over [x; z]          (rows sorted, duplicates summed)
//   Mg  N x nd     grad f    = c_x + Mg dvals
//   Mw  Z x (1+m)  w         = Mw [sigma; lambda]
//   MJ  nnzJ x nd, Jc        J vals = Jc + MJ dvals,   pattern (jac_rows, jac_cols) row-major sorted unique
//   MH  nnzH x nh            H vals = MH hvals (+ dense quad_form blocks), pattern lower oriented, sorted unique,
//                            position table (or base) of every listed dense block
//
// in plain C++ with a few host threads.  dnlp_amd/lowering.py keeps the same construction in numpy / scipy as
// the fallback and as the checker (tests/test_lower_maps.py compares every array); the results are identical.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <stdexcept>
#include <thread>
#include <vector>

namespace dnlp {

using lm_i64 = long long;
using lm_i32 = int32_t;

struct LmCsr {
  std::vector<lm_i64> ptr;
  std::vector<lm_i32> idx;
  std::vector<double> val;
};

struct LowerMapsIn {
  lm_i64 N = 0, Z = 0, m = 0, nd = 0, nh = 0;
  const lm_i64* Gp = nullptr;      // m + 1
  const lm_i32* Gi = nullptr;      // columns in [0, N + Z), any order, duplicates allowed
  const double* Gv = nullptr;
  const double* c = nullptr;       // N + Z objective coefficients
  const lm_i64 *drow = nullptr, *dcol = nullptr;      // nd: d z_drow / d x_dcol
  const lm_i64 *hrow = nullptr, *hcol = nullptr;      // nh: lower oriented second-derivative positions
  int nblk = 0;                    // dense quad_form blocks listed in the COO Hessian
  const lm_i64 *blk_x0 = nullptr, *blk_n = nullptr;
};

struct LowerMapsOut {
  int G_changed = 0;               // 0: the caller's arrays already are the canonical form
  int jac_is_G = 0;                // 1: every constraint row is affine in x alone and G is canonical — the Jacobian pattern
                                   //    and Jc ARE G's arrays (not copied: BASELINE C3 has 1e7 of them); nnzJ = nnz(G)
  LmCsr G, Mg, Mw, MJ, MH;
  std::vector<double> Jc;
  std::vector<lm_i32> jr, jc, hr, hc;
  std::vector<int> blk_mode;       // 2: contiguous run (pos holds the base), 1: table
  std::vector<std::vector<lm_i64>> blk_pos;
};

template <class F>
inline void lm_par_for(lm_i64 n, lm_i64 grain, F f) {
  unsigned T = std::thread::hardware_concurrency();
  if (T > 16) T = 16;
  if (T < 2 || n < 2 * grain) { f(static_cast<lm_i64>(0), n); return; }
  const lm_i64 chunks = std::min<lm_i64>(T, (n + grain - 1) / grain);
  const lm_i64 per = (n + chunks - 1) / chunks;
  std::vector<std::thread> th;
  for (lm_i64 k = 1; k < chunks; ++k) {
    const lm_i64 lo = k * per, hi = std::min(n, lo + per);
    if (lo < hi) th.emplace_back([=] { f(lo, hi); });
  }
  f(static_cast<lm_i64>(0), std::min(n, per));
  for (auto& t : th) t.join();
}

// exact symmetry of a dense n x n matrix (leading dimension ld): 64 x 64 tiles against their mirrors, tile rows dealt
// round-robin to the threads (the triangle's rows are not equally long); stops at the first difference
inline bool lm_is_symmetric(const double* P, lm_i64 n, lm_i64 ld) {
  const lm_i64 B = 64, nb = (n + B - 1) / B;
  unsigned T = std::thread::hardware_concurrency();
  if (T > 32) T = 32;
  if (T < 1 || n < 512) T = 1;
  std::atomic<int> bad{0};
  auto work = [&](unsigned t) {
    for (lm_i64 bi = t; bi < nb && !bad.load(std::memory_order_relaxed); bi += T) {
      const lm_i64 i0 = bi * B, i1 = std::min(n, i0 + B);
      for (lm_i64 bj = 0; bj <= bi; ++bj) {
        const lm_i64 j0 = bj * B, j1 = std::min(n, j0 + B);
        for (lm_i64 i = i0; i < i1; ++i) {
          const double* row = P + i * ld;
          for (lm_i64 j = j0; j < j1; ++j) if (row[j] != P[j * ld + i]) { bad.store(1, std::memory_order_relaxed); return; }
        }
      }
    }
  };
  std::vector<std::thread> th;
  for (unsigned t = 1; t < T; ++t) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
  return bad.load() == 0;
}

// sorted unique keys of `keys` (int64) and, for every input key, its position among them
inline void lm_unique(const std::vector<lm_i64>& keys, std::vector<lm_i64>& uniq, std::vector<lm_i64>& inv) {
  const lm_i64 n = static_cast<lm_i64>(keys.size());
  bool increasing = true;
  for (lm_i64 i = 1; i < n && increasing; ++i) increasing = keys[static_cast<size_t>(i)] > keys[static_cast<size_t>(i - 1)];
  if (increasing) {
    uniq = keys;
    inv.resize(static_cast<size_t>(n));
    std::iota(inv.begin(), inv.end(), static_cast<lm_i64>(0));
    return;
  }
  uniq = keys;
  // chunks sorted by threads, then merged pairwise
  unsigned T = std::thread::hardware_concurrency();
  if (T > 16) T = 16;
  lm_i64 parts = (n >= (1 << 18) && T >= 2) ? static_cast<lm_i64>(T) : 1;
  std::vector<lm_i64> cut(static_cast<size_t>(parts) + 1);
  for (lm_i64 k = 0; k <= parts; ++k) cut[static_cast<size_t>(k)] = n * k / parts;
  lm_par_for(parts, 1, [&](lm_i64 lo, lm_i64 hi) {
    for (lm_i64 k = lo; k < hi; ++k) std::sort(uniq.begin() + cut[static_cast<size_t>(k)], uniq.begin() + cut[static_cast<size_t>(k) + 1]);
  });
  for (lm_i64 width = 1; width < parts; width *= 2)
    for (lm_i64 k = 0; k + width < parts; k += 2 * width)
      std::inplace_merge(uniq.begin() + cut[static_cast<size_t>(k)], uniq.begin() + cut[static_cast<size_t>(k + width)],
                         uniq.begin() + cut[static_cast<size_t>(std::min(parts, k + 2 * width))]);
  uniq.erase(std::unique(uniq.begin(), uniq.end()), uniq.end());
  inv.resize(static_cast<size_t>(n));
  lm_par_for(n, 1 << 16, [&](lm_i64 lo, lm_i64 hi) {
    for (lm_i64 i = lo; i < hi; ++i)
      inv[static_cast<size_t>(i)] = std::lower_bound(uniq.begin(), uniq.end(), keys[static_cast<size_t>(i)]) - uniq.begin();
  });
}

inline void lower_maps_build(const LowerMapsIn& in, LowerMapsOut& out) {
  const lm_i64 N = in.N, Z = in.Z, m = in.m, nd = in.nd, nh = in.nh, ncol = N + Z;
  if (N + Z >= (static_cast<lm_i64>(1) << 31) || nd >= (static_cast<lm_i64>(1) << 31) || nh >= (static_cast<lm_i64>(1) << 31))
    throw std::runtime_error("lower_maps: index range exceeds int32");
  // ---- G: canonical form (rows sorted by column, duplicates summed) ----------------------------------------------
  bool canonical = true;
  {
    std::vector<char> flags(16, 0);
    lm_par_for(m, 4096, [&](lm_i64 lo, lm_i64 hi) {
      bool ok = true;
      for (lm_i64 i = lo; i < hi && ok; ++i)
        for (lm_i64 e = in.Gp[i] + 1; e < in.Gp[i + 1]; ++e)
          if (in.Gi[e] <= in.Gi[e - 1]) { ok = false; break; }
      if (!ok) flags[0] = 1;     // (benign race: every writer stores the same value)
    });
    canonical = flags[0] == 0;
  }
  const lm_i64* Gp = in.Gp;
  const lm_i32* Gi = in.Gi;
  const double* Gv = in.Gv;
  if (!canonical) {
    out.G_changed = 1;
    std::vector<lm_i64> cnt(static_cast<size_t>(m) + 1, 0);
    // pass 1: distinct columns per row
    std::vector<std::vector<std::pair<lm_i32, double>>> rows(static_cast<size_t>(m));
    lm_par_for(m, 1024, [&](lm_i64 lo, lm_i64 hi) {
      for (lm_i64 i = lo; i < hi; ++i) {
        auto& r = rows[static_cast<size_t>(i)];
        r.reserve(static_cast<size_t>(in.Gp[i + 1] - in.Gp[i]));
        for (lm_i64 e = in.Gp[i]; e < in.Gp[i + 1]; ++e) r.emplace_back(in.Gi[e], in.Gv[e]);
        std::stable_sort(r.begin(), r.end(), [](const auto& a, const auto& b) { return a.first < b.first; });
        size_t w = 0;
        for (size_t k = 0; k < r.size(); ++k) {
          if (w && r[w - 1].first == r[k].first) r[w - 1].second += r[k].second;
          else r[w++] = r[k];
        }
        r.resize(w);
        cnt[static_cast<size_t>(i) + 1] = static_cast<lm_i64>(w);
      }
    });
    for (lm_i64 i = 0; i < m; ++i) cnt[static_cast<size_t>(i) + 1] += cnt[static_cast<size_t>(i)];
    out.G.ptr = cnt;
    out.G.idx.resize(static_cast<size_t>(cnt[static_cast<size_t>(m)]));
    out.G.val.resize(out.G.idx.size());
    lm_par_for(m, 1024, [&](lm_i64 lo, lm_i64 hi) {
      for (lm_i64 i = lo; i < hi; ++i) {
        lm_i64 o = out.G.ptr[static_cast<size_t>(i)];
        for (const auto& pr : rows[static_cast<size_t>(i)]) { out.G.idx[static_cast<size_t>(o)] = pr.first; out.G.val[static_cast<size_t>(o)] = pr.second; ++o; }
      }
    });
    Gp = out.G.ptr.data(); Gi = out.G.idx.data(); Gv = out.G.val.data();
  }
  // first z entry of every row (rows are sorted: entries of x come first)
  std::vector<lm_i64> zbeg(static_cast<size_t>(m));
  lm_par_for(m, 4096, [&](lm_i64 lo, lm_i64 hi) {
    for (lm_i64 i = lo; i < hi; ++i)
      zbeg[static_cast<size_t>(i)] = std::lower_bound(Gi + Gp[i], Gi + Gp[i + 1], static_cast<lm_i32>(N)) - Gi;
  });
  const double* cz = in.c + N;
  // ---- Mg: rows = dcol, columns = derivative slots k, values cz[drow[k]] (non-zero ones) -----------------------
  {
    std::vector<lm_i64>& p = out.Mg.ptr;
    p.assign(static_cast<size_t>(N) + 1, 0);
    for (lm_i64 k = 0; k < nd; ++k) if (cz[in.drow[k]] != 0.0) ++p[static_cast<size_t>(in.dcol[k]) + 1];
    for (lm_i64 j = 0; j < N; ++j) p[static_cast<size_t>(j) + 1] += p[static_cast<size_t>(j)];
    out.Mg.idx.resize(static_cast<size_t>(p[static_cast<size_t>(N)]));
    out.Mg.val.resize(out.Mg.idx.size());
    std::vector<lm_i64> w(p.begin(), p.end() - 1);
    for (lm_i64 k = 0; k < nd; ++k) {
      const double v = cz[in.drow[k]];
      if (v == 0.0) continue;
      const lm_i64 o = w[static_cast<size_t>(in.dcol[k])]++;
      out.Mg.idx[static_cast<size_t>(o)] = static_cast<lm_i32>(k);
      out.Mg.val[static_cast<size_t>(o)] = v;
    }
  }
  // ---- Mw = [cz | Gz^T]: Z x (1 + m) ------------------------------------------------------------------------------
  {
    std::vector<lm_i64>& p = out.Mw.ptr;
    p.assign(static_cast<size_t>(Z) + 1, 0);
    if (Z) {
      for (lm_i64 z = 0; z < Z; ++z) if (cz[z] != 0.0) ++p[static_cast<size_t>(z) + 1];
      for (lm_i64 i = 0; i < m; ++i)
        for (lm_i64 e = zbeg[static_cast<size_t>(i)]; e < Gp[i + 1]; ++e) ++p[static_cast<size_t>(Gi[e] - N) + 1];
      for (lm_i64 z = 0; z < Z; ++z) p[static_cast<size_t>(z) + 1] += p[static_cast<size_t>(z)];
      out.Mw.idx.resize(static_cast<size_t>(p[static_cast<size_t>(Z)]));
      out.Mw.val.resize(out.Mw.idx.size());
      std::vector<lm_i64> w(p.begin(), p.end() - 1);
      for (lm_i64 z = 0; z < Z; ++z)
        if (cz[z] != 0.0) { const lm_i64 o = w[static_cast<size_t>(z)]++; out.Mw.idx[static_cast<size_t>(o)] = 0; out.Mw.val[static_cast<size_t>(o)] = cz[z]; }
      for (lm_i64 i = 0; i < m; ++i)
        for (lm_i64 e = zbeg[static_cast<size_t>(i)]; e < Gp[i + 1]; ++e) {
          const lm_i64 o = w[static_cast<size_t>(Gi[e] - N)]++;
          out.Mw.idx[static_cast<size_t>(o)] = static_cast<lm_i32>(1 + i);
          out.Mw.val[static_cast<size_t>(o)] = Gv[e];
        }
    }
  }
  // ---- Jacobian: J = Gx + Gz D -------------------------------------------------------------------------------------
  // derivative slots of every z: E[z] = {k : drow[k] == z}, k ascending
  std::vector<lm_i64> Ep(static_cast<size_t>(Z) + 1, 0), Ek(static_cast<size_t>(nd));
  for (lm_i64 k = 0; k < nd; ++k) ++Ep[static_cast<size_t>(in.drow[k]) + 1];
  for (lm_i64 z = 0; z < Z; ++z) Ep[static_cast<size_t>(z) + 1] += Ep[static_cast<size_t>(z)];
  {
    std::vector<lm_i64> w(Ep.begin(), Ep.end() - 1);
    for (lm_i64 k = 0; k < nd; ++k) Ek[static_cast<size_t>(w[static_cast<size_t>(in.drow[k])]++)] = k;
  }
  struct Cand { lm_i32 col; lm_i32 k; double v; };          // k < 0: an affine (Gx) entry
  // candidates of every row with derivative entries in ONE flat array (a vector per row is 3e5 allocations for
  // BASELINE C2's canonical form); affine rows have none: their sorted G entries are the pattern
  std::vector<lm_i64> jcount(static_cast<size_t>(m) + 1, 0), mjcount(static_cast<size_t>(m) + 1, 0), coff(static_cast<size_t>(m) + 1, 0);
  lm_par_for(m, 2048, [&](lm_i64 lo, lm_i64 hi) {
    for (lm_i64 i = lo; i < hi; ++i) {
      lm_i64 extra = 0;
      for (lm_i64 e = zbeg[static_cast<size_t>(i)]; e < Gp[i + 1]; ++e) extra += Ep[static_cast<size_t>(Gi[e] - N) + 1] - Ep[static_cast<size_t>(Gi[e] - N)];
      mjcount[static_cast<size_t>(i) + 1] = extra;
      coff[static_cast<size_t>(i) + 1] = extra ? (zbeg[static_cast<size_t>(i)] - Gp[i]) + extra : 0;
    }
  });
  for (lm_i64 i = 0; i < m; ++i) coff[static_cast<size_t>(i) + 1] += coff[static_cast<size_t>(i)];
  std::vector<Cand> cand(static_cast<size_t>(coff[static_cast<size_t>(m)]));
  lm_par_for(m, 1024, [&](lm_i64 lo, lm_i64 hi) {
    for (lm_i64 i = lo; i < hi; ++i) {
      const lm_i64 c0 = coff[static_cast<size_t>(i)], c1 = coff[static_cast<size_t>(i) + 1];
      if (c1 == c0) {                                        // affine row
        jcount[static_cast<size_t>(i) + 1] = zbeg[static_cast<size_t>(i)] - Gp[i];
        continue;
      }
      Cand* r = cand.data() + c0;
      lm_i64 w = 0;
      for (lm_i64 e = Gp[i]; e < zbeg[static_cast<size_t>(i)]; ++e) r[w++] = {Gi[e], -1, Gv[e]};
      for (lm_i64 e = zbeg[static_cast<size_t>(i)]; e < Gp[i + 1]; ++e) {
        const lm_i64 z = Gi[e] - N;
        for (lm_i64 q = Ep[static_cast<size_t>(z)]; q < Ep[static_cast<size_t>(z) + 1]; ++q) {
          const lm_i64 k = Ek[static_cast<size_t>(q)];
          r[w++] = {static_cast<lm_i32>(in.dcol[k]), static_cast<lm_i32>(k), Gv[e]};
        }
      }
      std::stable_sort(r, r + w, [](const Cand& a, const Cand& b) { return a.col < b.col || (a.col == b.col && a.k < b.k); });
      lm_i64 u = 0;
      for (lm_i64 q = 0; q < w; ++q) if (q == 0 || r[q].col != r[q - 1].col) ++u;
      jcount[static_cast<size_t>(i) + 1] = u;
    }
  });
  for (lm_i64 i = 0; i < m; ++i) { jcount[static_cast<size_t>(i) + 1] += jcount[static_cast<size_t>(i)]; mjcount[static_cast<size_t>(i) + 1] += mjcount[static_cast<size_t>(i)]; }
  const lm_i64 nnzJ = jcount[static_cast<size_t>(m)];
  if (canonical && mjcount[static_cast<size_t>(m)] == 0 && nnzJ == Gp[m]) {
    out.jac_is_G = 1;
    out.MJ.ptr.assign(static_cast<size_t>(nnzJ) + 1, 0);
  } else {
  out.jr.resize(static_cast<size_t>(nnzJ)); out.jc.resize(static_cast<size_t>(nnzJ)); out.Jc.assign(static_cast<size_t>(nnzJ), 0.0);
  out.MJ.ptr.assign(static_cast<size_t>(nnzJ) + 1, 0);
  out.MJ.idx.resize(static_cast<size_t>(mjcount[static_cast<size_t>(m)]));
  out.MJ.val.resize(out.MJ.idx.size());
  lm_par_for(m, 1024, [&](lm_i64 lo, lm_i64 hi) {
    for (lm_i64 i = lo; i < hi; ++i) {
      lm_i64 pos = jcount[static_cast<size_t>(i)];
      const lm_i64 c0 = coff[static_cast<size_t>(i)], c1 = coff[static_cast<size_t>(i) + 1];
      if (c1 == c0) {
        for (lm_i64 e = Gp[i]; e < zbeg[static_cast<size_t>(i)]; ++e, ++pos) {
          out.jr[static_cast<size_t>(pos)] = static_cast<lm_i32>(i); out.jc[static_cast<size_t>(pos)] = Gi[e]; out.Jc[static_cast<size_t>(pos)] = Gv[e];
          out.MJ.ptr[static_cast<size_t>(pos) + 1] = 0;
        }
        continue;
      }
      const Cand* r = cand.data() + c0;
      lm_i64 mo = mjcount[static_cast<size_t>(i)];
      --pos;
      for (lm_i64 q = 0; q < c1 - c0; ++q) {
        if (q == 0 || r[q].col != r[q - 1].col) {
          ++pos;
          out.jr[static_cast<size_t>(pos)] = static_cast<lm_i32>(i); out.jc[static_cast<size_t>(pos)] = r[q].col;
          out.MJ.ptr[static_cast<size_t>(pos) + 1] = 0;
        }
        if (r[q].k < 0) out.Jc[static_cast<size_t>(pos)] += r[q].v;
        else { out.MJ.idx[static_cast<size_t>(mo)] = r[q].k; out.MJ.val[static_cast<size_t>(mo)] = r[q].v; ++mo; ++out.MJ.ptr[static_cast<size_t>(pos) + 1]; }
      }
    }
  });
  for (lm_i64 p = 0; p < nnzJ; ++p) out.MJ.ptr[static_cast<size_t>(p) + 1] += out.MJ.ptr[static_cast<size_t>(p)];
  }
  // ---- Hessian pattern and MH --------------------------------------------------------------------------------------
  lm_i64 nkeys = nh;
  for (int b = 0; b < in.nblk; ++b) nkeys += in.blk_n[b] * (in.blk_n[b] + 1) / 2;
  std::vector<lm_i64> keys(static_cast<size_t>(nkeys));
  for (lm_i64 k = 0; k < nh; ++k) keys[static_cast<size_t>(k)] = in.hrow[k] * N + in.hcol[k];
  {
    lm_i64 o = nh;
    for (int b = 0; b < in.nblk; ++b) {
      const lm_i64 x0 = in.blk_x0[b], nb = in.blk_n[b];
      for (lm_i64 i = 0; i < nb; ++i)
        for (lm_i64 j = 0; j <= i; ++j) keys[static_cast<size_t>(o++)] = (x0 + i) * N + (x0 + j);   // np.tril_indices order
    }
  }
  std::vector<lm_i64> huniq, hinv;
  lm_unique(keys, huniq, hinv);
  const lm_i64 nnzH = static_cast<lm_i64>(huniq.size());
  out.hr.resize(static_cast<size_t>(nnzH)); out.hc.resize(static_cast<size_t>(nnzH));
  lm_par_for(nnzH, 1 << 16, [&](lm_i64 lo, lm_i64 hi) {
    for (lm_i64 p = lo; p < hi; ++p) { out.hr[static_cast<size_t>(p)] = static_cast<lm_i32>(huniq[static_cast<size_t>(p)] / N); out.hc[static_cast<size_t>(p)] = static_cast<lm_i32>(huniq[static_cast<size_t>(p)] % N); }
  });
  {
    std::vector<lm_i64>& p = out.MH.ptr;
    p.assign(static_cast<size_t>(nnzH) + 1, 0);
    for (lm_i64 k = 0; k < nh; ++k) ++p[static_cast<size_t>(hinv[static_cast<size_t>(k)]) + 1];
    for (lm_i64 q = 0; q < nnzH; ++q) p[static_cast<size_t>(q) + 1] += p[static_cast<size_t>(q)];
    out.MH.idx.resize(static_cast<size_t>(nh));
    out.MH.val.assign(static_cast<size_t>(nh), 1.0);
    std::vector<lm_i64> w(p.begin(), p.end() - 1);
    for (lm_i64 k = 0; k < nh; ++k) out.MH.idx[static_cast<size_t>(w[static_cast<size_t>(hinv[static_cast<size_t>(k)])]++)] = static_cast<lm_i32>(k);
  }
  out.blk_mode.assign(static_cast<size_t>(in.nblk), 1);
  out.blk_pos.resize(static_cast<size_t>(in.nblk));
  {
    lm_i64 o = nh;
    for (int b = 0; b < in.nblk; ++b) {
      const lm_i64 cnt = in.blk_n[b] * (in.blk_n[b] + 1) / 2;
      bool run = cnt > 0 && hinv[static_cast<size_t>(o + cnt - 1)] - hinv[static_cast<size_t>(o)] == cnt - 1;
      for (lm_i64 q = 1; q < cnt && run; ++q) run = hinv[static_cast<size_t>(o + q)] > hinv[static_cast<size_t>(o + q - 1)];
      if (run) { out.blk_mode[static_cast<size_t>(b)] = 2; out.blk_pos[static_cast<size_t>(b)] = {hinv[static_cast<size_t>(o)]}; }
      else out.blk_pos[static_cast<size_t>(b)].assign(hinv.begin() + o, hinv.begin() + o + cnt);
      o += cnt;
    }
  }
}

}  // namespace dnlp
