// Static-pattern sparse LDL^T of the reduced KKT matrix
//
//     K = [ W + Sigma_x + delta_w I    J^T ]      (order n = N + m)
//         [ J                          -D  ]
//
// for problems whose KKT matrix is sparse (every example of the paper: nnz(L) is 0.4 % of the
// dense triangle for path planning / power flow, 5 % for localization).  The role
// MUMPS plays for IPOPT in the reference (ipopt_nlpif.py:140-170).
//
// Everything that depends only on the sparsity pattern is computed ONCE per tape on the host
// (this file):
//   * static pivot blocks.  dnlp2smooth introduces auxiliary variables that enter linearly and
//     are defined by equality rows, so both their diagonal and the row's diagonal are
//     structurally zero: no 1x1 pivot order exists without regularisation.  Every equality row
//     is therefore paired, when possible, with one adjacent variable into a 2x2 pivot block
//     [[w, a], [a, -d]] (det = -w d - a^2 < 0 for a != 0), preferring structurally-zero-diagonal
//     variables and constant Jacobian coefficients (the defining rows `t - expr == 0` have a = 1);
//   * a minimum-degree elimination order of the blocks, the fill pattern, the address of every
//     entry of the reduced matrices, and the UPDATE PROGRAM: for every pivot block the list of
//     (target address, row i, row j) triples of its Schur-complement update.
// The numeric phase (sparse_ldl.h) is then branch-free index arithmetic over flat arrays: the
// same single-source routine runs as host loops (test oracle), as one wavefront / workgroup per
// instance inside the batch kernel, and as a kernel of the HIP space.
//
// Pivoting is static: a zero or wrong-sign pivot is reported through the inertia (nzero /
// nneg), and the interior-point loop's delta_w / delta_c regularisation (WB Algorithm IC)
// repairs it, exactly as for the unpivoted blocked dense path.
#pragma once
#include <limits>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <queue>
#include <vector>
#include <atomic>
#include <exception>
#include <thread>

#include "tape.h"

namespace dnlp {

// exec-space view of the plan (plain pointers)
struct SparsePlan {
  i64 n = 0, N = 0, m = 0, nblk = 0, nvals = 0, ntrip = 0, maxs = 0, nlev = 0;
  // Dense tail (host-driven spaces, kkt_dense.h): the last levels of the elimination order of a pattern with a dense
  // separator are a chain of one-block levels over a dense Schur complement (NMF at notebook size: 301 of 305 levels,
  // 1 200 dependent launches per factorisation and 600 per solve).  When tail_n > 0 the level loops stop at nlev_run /
  // nblk_run, the tail's entries are gathered from the plan's storage into a dense tail_n x tail_n matrix
  // (tg_src -> tg_dst) and that matrix is factorised and solved by the dense path.
  i64 nlev_run = 0, nblk_run = 0;     // levels / blocks the sparse phases process (= nlev / nblk without a tail)
  int solve_phase = 0;                // sparse_solve: 0 everything, 1 forward + D^-1, 2 backward
  i64 tail_n = 0, tg_count = 0;
  i32* tnode = nullptr;               // tail_n: KKT node of every tail position
  i32* tg_src = nullptr;              // tg_count: value index in the plan's storage ...
  i32* tg_dst = nullptr;              // ... and its place r + c * tail_n (r >= c) in the dense tail
  // Panels: a block before the tail whose struct ends in >= 12 tail nodes (NMF: 1 200 blocks with 300 each, 5.4e7 of
  // the 5.5e7 update triples) carries NO triples for its tail x tail pairs.  After the level's rows are scaled its tail
  // rows are gathered into two dense panels (l and w = l D^-1, tail_ld x pg_cols[level]) and the level's whole
  // contribution to the tail is ONE product  T -= Pl Pw^T  into the accumulator T (tail_ld x tail_n, lower triangle),
  // which joins the gathered tail before its dense factorisation.
  i64 tail_ld = 0, pg_maxcols = 0;
  i64* pg_off = nullptr;              // nlev_run + 1: offsets into pg_src / pg_dst per level
  i32* pg_src = nullptr;              // value index of a tail row of a panel block's column
  i32* pg_dst = nullptr;              // its place row + col * tail_ld in the level's panel
  const i64 *h_pg_off = nullptr, *h_pg_cols = nullptr;   // host copies (level loops of the host-driven space)
  i64* pg_cols = nullptr;             // nlev_run: panel columns per level (exec space; the in-kernel routine reads it)
  i32* bnode = nullptr;   // 2 per block: pivot nodes (second = -1 for a 1x1 block)
  i32* soff = nullptr;    // nblk + 1: offsets into sidx (every offset array of the plan is 32-bit: the in-kernel solver keeps them in LDS)
  i32* sidx = nullptr;    // struct of every block: node ids (0..N-1 variables, N.. constraint rows)
  i32* doff = nullptr;    // per block: D values (1x1: d ; 2x2: d11, d21, d22)
  i32* loff = nullptr;    // per block: L values, s_k rows x b_k, row-major
  i32* toff = nullptr;    // nblk + 1: offsets into the triples
  // update program in DIRECT ADDRESSES: triple q multiplies the unscaled l at value index tau[q] with L = l D^-1 at
  // tav[q] (a 2x2 pivot block: the pairs tau, tau + 1 and ~tav, ~tav + 1 — the complement marks the block size), into
  // the destination of q's group (gdst).  No block / row lookups in the numeric phase.
  i32* tau = nullptr;
  i32* tav = nullptr;
  i32* hpos = nullptr;    // value index of every Hessian COO entry (lower triangle)
  i32* jpos = nullptr;    // value index of every Jacobian COO entry
  i32* dpos = nullptr;    // value index of the diagonal of every node
  // level schedule: blocks are stored level-major (elimination-tree levels); blocks of one level
  // are independent and are processed together
  i32* lev_off = nullptr; // nlev + 1 offsets into the block order
  i32* sblk = nullptr;    // block of every struct row (soff-indexed)
  // ORDER-FIXED ACCUMULATION (no floating-point atomics anywhere in the numeric phase).  The update triples of a level
  // are stored sorted by destination: group g = one destination value gdst[g] and the triples goff[g] .. goff[g+1];
  // lev_g[lev] .. lev_g[lev+1] are the groups of level lev.  One lane (or a fixed reduction tree over lanes) sums a
  // group and subtracts it from its destination once.  Forward substitution likewise gathers: target node fnode[h]
  // (a node of a block of level lev for h in lev_f[lev] .. lev_f[lev+1]) collects the struct rows frow[foff[h] ..
  // foff[h+1]) that point at it, ascending (= by source block), so a node is final when its level is reached.
  i64 ngrp = 0, nfwd = 0;
  i32* gdst = nullptr;
  i32* goff = nullptr;
  i32* lev_g = nullptr;
  i32* fnode = nullptr;
  i32* foff = nullptr;
  i32* lev_f = nullptr;
  // a gathered row in direct addresses: value index fa (complemented for a row of a 2x2 block: two values), source
  // nodes fu0 / fu1 (fu1 unused for 1x1)
  i32* fa = nullptr;
  i32* fu0 = nullptr;
  i32* fu1 = nullptr;
  i32* fend = nullptr;    // dense-tail plans only: end of target h's rows that come from blocks before the tail (else null: foff[h + 1])
  const i32 *h_lev_g = nullptr, *h_lev_f = nullptr, *h_fwd_rows = nullptr;   // (h_fwd_rows = host copy of foff)
  // host copies of the level boundaries (blocks / struct rows / triples / values), for the space
  // that launches one kernel per level phase; unused inside kernels
  const i64 *h_lev_blk = nullptr, *h_lev_row = nullptr, *h_lev_trip = nullptr, *h_lev_val = nullptr;
};

struct SparsePlanHost {
  i64 n = 0, N = 0, m = 0, nvals = 0, maxs = 0;
  std::vector<i32> bnode, sidx, tdst, tiu, tiv, tau, tav, hpos, jpos, dpos;   // (tdst / tiu / tiv: scratch of layout(), empty afterwards)
  std::vector<i32> soff, doff, loff, toff, lev_off;
  std::vector<i64> lev_blk, lev_row, lev_trip, lev_val;     // per-level boundaries in blocks / rows / triples / values
  std::vector<i32> sblk, tblk;
  std::vector<i32> gdst, fnode, fa, fu0, fu1, fend;  // order-fixed accumulation (see SparsePlan)
  i64 ntrip = 0;
  std::vector<i32> goff, lev_g, foff, lev_f;
  // dense tail (see SparsePlan): chosen by layout(), handed to the exec space only by upload(ex, true)
  i64 tail_lev = -1, tail_n = 0, tail_ld = 0, pg_maxcols = 0;
  std::vector<i32> tnode, tg_src, tg_dst, pg_src, pg_dst;
  std::vector<i64> pg_off, pg_cols;
  bool allow_tail = false;            // build_sparse_plan(..., allow_tail): host-driven spaces only
  bool panels_dropped = false;        // tail x tail triples of the panel blocks are not in the update program
  i64 nnzL = 0, n_delayed = 0, n_pairs = 0;
  double fill_ratio = 0.0;    // factor values / dense lower triangle
  // analyse() results: what the numeric phase will cost (known before the layout and the update program,
  // which are the expensive part of the analysis, are generated), and the state layout() continues from
  i64 pred_triples = 0, pred_levels = 0;
  std::vector<std::array<i32, 2>> bn_;
  std::vector<i32> order_;
  std::vector<std::vector<i32>> bstruct_;

  i64 nblk() const { return static_cast<i64>(doff.size()); }

  // zero_diag_var[j]: no finite bound; eq_row[i]: cl == cu;
  // jac_const[p]: |constant coefficient| of Jacobian entry p, or 0 when the entry is not constant;
  // jac_abs0[p] (optional): |value| of entry p at the start point — a coupling that is numerically
  // zero there makes a singular 2x2 block, so such entries are the last resort of the matching
  void build(i64 N_, i64 m_, const std::vector<i32>& hr, const std::vector<i32>& hc, const std::vector<i32>& jr,
             const std::vector<i32>& jc, const std::vector<char>& zero_diag_var, const std::vector<char>& eq_row,
             const std::vector<double>& jac_const, const std::vector<char>& fixed_var, int relax = 0,
             const std::vector<double>* jac_abs0 = nullptr) {
    analyse(N_, m_, hr, hc, jr, jc, zero_diag_var, eq_row, jac_const, fixed_var, relax, jac_abs0);
    layout(hr, hc, jr, jc, fixed_var);
  }

  // pairing, elimination order, elimination-tree levels
  void analyse(i64 N_, i64 m_, const std::vector<i32>& hr, const std::vector<i32>& hc, const std::vector<i32>& jr,
               const std::vector<i32>& jc, const std::vector<char>& zero_diag_var, const std::vector<char>& eq_row,
               const std::vector<double>& jac_const, const std::vector<char>& fixed_var, int relax = 0,
               const std::vector<double>* jac_abs0 = nullptr) {
    *this = SparsePlanHost();
    N = N_; m = m_; n = N + m;
    const i64 nn = n;
    // fixed variables (lb == ub) are pinned by the interior-point loop: unit diagonal, all their
    // couplings masked.  They are isolated nodes of the pattern and never pivot partners.
    auto is_fixed = [&](i32 u) { return u < N && fixed_var[static_cast<size_t>(u)] != 0; };
    const bool tplan = std::getenv("DNLP_TIME_PLAN") != nullptr;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tp0 = tnow();
    auto tick = [&](const char* what) { if (tplan) { const double t1 = tnow(); std::fprintf(stderr, "[dnlp plan] %-28s %.3f s\n", what, t1 - tp0); tp0 = t1; } };
    // ---- symmetric adjacency (no diagonal) ----
    // (flat: two counting passes instead of 700 000 little vectors for the canonical Rosenbrock chain — the lists are
    //  only read afterwards; node u's sorted, duplicate-free neighbours are aidx[aptr[u] .. aptr[u] + alen[u]))
    std::vector<i64> aptr(static_cast<size_t>(nn) + 1, 0);
    std::vector<i32> aidx, alen(static_cast<size_t>(nn), 0);
    {
      auto keep = [&](i32 a, i32 b) { return a != b && !is_fixed(a) && !is_fixed(b); };
      for (size_t p = 0; p < hr.size(); ++p) if (keep(hr[p], hc[p])) { ++aptr[static_cast<size_t>(hr[p]) + 1]; ++aptr[static_cast<size_t>(hc[p]) + 1]; }
      for (size_t p = 0; p < jr.size(); ++p) if (keep(static_cast<i32>(N + jr[p]), jc[p])) { ++aptr[static_cast<size_t>(N + jr[p]) + 1]; ++aptr[static_cast<size_t>(jc[p]) + 1]; }
      for (i64 u = 0; u < nn; ++u) aptr[static_cast<size_t>(u) + 1] += aptr[static_cast<size_t>(u)];
      aidx.resize(static_cast<size_t>(aptr[static_cast<size_t>(nn)]));
      std::vector<i64> fill(aptr.begin(), aptr.end() - 1);
      auto put = [&](i32 a, i32 b) { aidx[static_cast<size_t>(fill[static_cast<size_t>(a)]++)] = b; aidx[static_cast<size_t>(fill[static_cast<size_t>(b)]++)] = a; };
      for (size_t p = 0; p < hr.size(); ++p) if (keep(hr[p], hc[p])) put(hr[p], hc[p]);
      for (size_t p = 0; p < jr.size(); ++p) if (keep(static_cast<i32>(N + jr[p]), jc[p])) put(static_cast<i32>(N + jr[p]), jc[p]);
      for (i64 u = 0; u < nn; ++u) {
        i32* b0 = aidx.data() + aptr[static_cast<size_t>(u)];
        i32* b1 = aidx.data() + aptr[static_cast<size_t>(u) + 1];
        std::sort(b0, b1);
        alen[static_cast<size_t>(u)] = static_cast<i32>(std::unique(b0, b1) - b0);
      }
    }
    // ---- static 2x2 blocks: equality row <-> one adjacent variable ----
    std::vector<i32> partner(static_cast<size_t>(nn), -1);
    {
      // best coefficient per (row, var): from the Jacobian COO
      // candidate weight: > 1 constant coefficient, (0, 1] varying coefficient that is non-zero at the
      // start point, 0 varying coefficient that vanishes there
      // (flat as well: row i's candidates in the order of the Jacobian's entries)
      struct RowVars {
        std::vector<i64> ptr;
        std::vector<std::pair<i32, double>> ent;
        struct Range { const std::pair<i32, double>*b, *e; const std::pair<i32, double>* begin() const { return b; } const std::pair<i32, double>* end() const { return e; } size_t size() const { return static_cast<size_t>(e - b); } };
        Range operator[](size_t i) const { return Range{ent.data() + ptr[i], ent.data() + ptr[i + 1]}; }
      } rowvars;
      rowvars.ptr.assign(static_cast<size_t>(m) + 1, 0);
      for (size_t p = 0; p < jr.size(); ++p) if (!is_fixed(jc[p])) ++rowvars.ptr[static_cast<size_t>(jr[p]) + 1];
      for (i64 i = 0; i < m; ++i) rowvars.ptr[static_cast<size_t>(i) + 1] += rowvars.ptr[static_cast<size_t>(i)];
      rowvars.ent.resize(static_cast<size_t>(rowvars.ptr[static_cast<size_t>(m)]));
      {
        std::vector<i64> fill(rowvars.ptr.begin(), rowvars.ptr.end() - 1);
        for (size_t p = 0; p < jr.size(); ++p) {
          if (is_fixed(jc[p])) continue;
          double wgt;
          if (jac_const[p] > 0.0) wgt = 1.0 + std::min(jac_const[p], 1.0);
          else if (jac_abs0) wgt = std::min((*jac_abs0)[p], 1.0) * 0.999;
          else wgt = 0.5;
          rowvars.ent[static_cast<size_t>(fill[static_cast<size_t>(jr[p])]++)] = {jc[p], wgt};
        }
      }
      // rows with the fewest candidates first (they have the least choice)
      std::vector<i32> rows;
      for (i64 i = 0; i < m; ++i) if (eq_row[static_cast<size_t>(i)]) rows.push_back(static_cast<i32>(i));
      std::stable_sort(rows.begin(), rows.end(), [&](i32 a, i32 b) { return rowvars[static_cast<size_t>(a)].size() < rowvars[static_cast<size_t>(b)].size(); });
      for (i32 i : rows) {
        i32 best = -1;
        double best_score = -1.0;
        for (const auto& vc : rowvars[static_cast<size_t>(i)]) {
          const i32 v = vc.first;
          if (partner[static_cast<size_t>(v)] >= 0 || !(vc.second > 0.0)) continue;
          // constant coefficient > large coefficient > structurally zero diagonal > low degree
          double score = 4.0 * vc.second + (zero_diag_var[static_cast<size_t>(v)] ? 2.0 : 0.0) +
                         1.0 / (1.0 + static_cast<double>(alen[static_cast<size_t>(v)]));
          if (score > best_score) { best_score = score; best = v; }
        }
        if (best >= 0) { partner[static_cast<size_t>(best)] = static_cast<i32>(N + i); partner[static_cast<size_t>(N + i)] = best; }
      }
      // Rows the greedy pass left without a partner: augmenting paths (Kuhn's algorithm, iterative
      // DFS).  An unpaired equality row is a structurally zero 1x1 pivot whose fill from a
      // neighbouring [[w, a], [a, 0]] block is -l^T D^-1 l = 0 when w = 0, so a maximum matching is
      // what makes the static pivot sequence non-singular whenever J has full structural row rank.
      // Constant coefficients are tried first.
      std::vector<i64> visit_mark(static_cast<size_t>(N), -1);
      std::vector<i32> from_row(static_cast<size_t>(N), -1);
      for (int pass = 0; pass < 3; ++pass) {
        for (i32 i : rows) {
          if (partner[static_cast<size_t>(N + i)] >= 0) continue;
          // BFS over alternating paths: row -> candidate variable -> the row it is matched to -> ...
          const i64 stamp = static_cast<i64>(pass) * (m + 1) + i;     // unique per (pass, root)
          std::vector<i32> queue{i};
          i32 end_var = -1;
          for (size_t qh = 0; qh < queue.size() && end_var < 0; ++qh) {
            const i32 r = queue[qh];
            for (const auto& vc : rowvars[static_cast<size_t>(r)]) {
              const i32 v = vc.first;
              if (pass == 0 && !(vc.second > 1.0)) continue;       // constant coefficients only
              if (pass == 1 && !(vc.second > 0.0)) continue;       // non-zero at the start point
              if (visit_mark[static_cast<size_t>(v)] == stamp) continue;
              visit_mark[static_cast<size_t>(v)] = stamp;
              from_row[static_cast<size_t>(v)] = r;
              const i32 pr = partner[static_cast<size_t>(v)];
              if (pr < 0) { end_var = v; break; }
              queue.push_back(static_cast<i32>(pr - N));
            }
          }
          // flip the path
          i32 v = end_var;
          while (v >= 0) {
            const i32 r = from_row[static_cast<size_t>(v)];
            const i32 prev = partner[static_cast<size_t>(N + r)];     // variable r was matched to (-1 for the root)
            partner[static_cast<size_t>(v)] = static_cast<i32>(N + r);
            partner[static_cast<size_t>(N + r)] = v;
            v = prev;
            if (r == i) break;
          }
        }
      }
    }
    if (std::getenv("DNLP_SPARSE_DEBUG")) {
      i64 un = 0, tot = 0;
      for (i64 i = 0; i < m; ++i) if (eq_row[static_cast<size_t>(i)]) { ++tot; if (partner[static_cast<size_t>(N + i)] < 0) { ++un; if (un <= 5) std::fprintf(stderr, "[plan] unmatched eq row %lld\n", (long long)i); } }
      std::fprintf(stderr, "[plan] equality rows %lld, unmatched %lld\n", (long long)tot, (long long)un);
    }
    tick("adjacency + matching");
    // ---- blocks ----
    std::vector<i32> blk_of(static_cast<size_t>(nn), -1);
    std::vector<std::array<i32, 2>> bn;
    for (i64 u = 0; u < nn; ++u) {
      if (blk_of[static_cast<size_t>(u)] >= 0) continue;
      const i32 p = partner[static_cast<size_t>(u)];
      const i32 id = static_cast<i32>(bn.size());
      if (p >= 0) { bn.push_back({static_cast<i32>(std::min<i64>(u, p)), static_cast<i32>(std::max<i64>(u, p))}); blk_of[static_cast<size_t>(p)] = id; }
      else bn.push_back({static_cast<i32>(u), -1});
      blk_of[static_cast<size_t>(u)] = id;
    }
    const i64 nb = static_cast<i64>(bn.size());
    n_pairs = nn - nb;
    auto bsize = [&](i32 b) { return bn[static_cast<size_t>(b)][1] >= 0 ? 2 : 1; };
    // block adjacency
    std::vector<std::vector<i32>> badj(static_cast<size_t>(nb));
    for (i64 u = 0; u < nn; ++u)
      for (i64 q = aptr[static_cast<size_t>(u)], q1 = q + alen[static_cast<size_t>(u)]; q < q1; ++q) {
        const i32 v = aidx[static_cast<size_t>(q)];
        if (blk_of[static_cast<size_t>(u)] != blk_of[static_cast<size_t>(v)]) badj[static_cast<size_t>(blk_of[static_cast<size_t>(u)])].push_back(blk_of[static_cast<size_t>(v)]);
      }
    for (auto& a : badj) { std::sort(a.begin(), a.end()); a.erase(std::unique(a.begin(), a.end()), a.end()); }
    // ---- minimum (external) degree elimination of the blocks, explicit fill ----
    // A 1x1 block whose diagonal is structurally zero (unpaired equality row, unpaired linear
    // free variable) is not eligible until one of its neighbours has been eliminated: the
    // neighbour's update puts -l^2/d on its diagonal, so the static pivot is non-zero.
    std::vector<char> gone(static_cast<size_t>(nb), 0), ready(static_cast<size_t>(nb), 1);
    for (i64 b = 0; b < nb; ++b) {
      if (bsize(static_cast<i32>(b)) == 2) continue;
      const i32 u = bn[static_cast<size_t>(b)][0];
      const bool zd = (u < N) ? zero_diag_var[static_cast<size_t>(u)] != 0 : eq_row[static_cast<size_t>(u - N)] != 0;
      if (zd && !badj[static_cast<size_t>(b)].empty()) ready[static_cast<size_t>(b)] = 0;
    }
    n_delayed = 0;
    for (char r : ready) n_delayed += r ? 0 : 1;
    std::vector<i64> deg(static_cast<size_t>(nb), 0);
    auto degree = [&](i32 b) { i64 d = 0; for (i32 c : badj[static_cast<size_t>(b)]) d += bsize(c); return d; };
    // The queue of (degree, block) pairs, smallest first, ties by block number — as buckets by degree (a binary heap of
    // 900 000 pushes was 0.04 s of the canonical Rosenbrock chain's plan): an entry is stale when its block is gone or its
    // degree has changed since; a round takes the whole buckets mind .. mind + relax, drops the stale entries and sorts
    // what is left by block number — the order in which the heap handed them out.
    std::vector<std::vector<i32>> bucket;
    i64 low = 0;                                     // no valid entry sits in a bucket below this one
    auto push = [&](i64 d, i32 b) {
      if (d >= static_cast<i64>(bucket.size())) bucket.resize(static_cast<size_t>(d) + 1 + bucket.size() / 2);
      bucket[static_cast<size_t>(d)].push_back(b);
      if (d < low) low = d;
    };
    // smallest degree with a valid entry (stale entries of the buckets passed are dropped), or -1
    auto min_valid = [&]() -> i64 {
      for (; low < static_cast<i64>(bucket.size()); ++low) {
        std::vector<i32>& bk = bucket[static_cast<size_t>(low)];
        size_t keep = 0;
        for (i32 b : bk) if (!gone[static_cast<size_t>(b)] && deg[static_cast<size_t>(b)] == low) bk[keep++] = b;
        bk.resize(keep);
        if (keep) return low;
      }
      return -1;
    };
    for (i64 b = 0; b < nb; ++b) {
      deg[static_cast<size_t>(b)] = degree(static_cast<i32>(b));
      if (ready[static_cast<size_t>(b)]) push(deg[static_cast<size_t>(b)], static_cast<i32>(b));
    }
    std::vector<i32> order;
    std::vector<std::vector<i32>> bstruct(static_cast<size_t>(nb));   // neighbour blocks at elimination time
    order.reserve(static_cast<size_t>(nb));
    std::vector<i32> merged;
    i64 remaining = nb;
    // Multiple elimination with a relaxed threshold (`relax`): in one round, a maximal independent
    // set of the blocks whose degree is within `relax` of the minimum is eliminated; neighbours of
    // an eliminated block wait for the next round.  relax = 0 is plain minimum degree.  On
    // chain-like patterns (Rosenbrock chain, path planning) relax = 1 removes every other interior
    // block per round, so the elimination tree is logarithmically shallow instead of n/2 deep —
    // the depth is the number of sequential steps of the level-parallel numeric phase.
    std::vector<i64> mark(static_cast<size_t>(nb), -1);
    std::vector<i32> cand;
    i64 round = 0;
    // One round eliminates an INDEPENDENT set: no eliminated block is a neighbour of another, so the blocks'
    // structs are their adjacency lists as they stand, and the fill can be applied per surviving neighbour
    // ONCE per round -- the union of its own list and the lists of all its eliminated neighbours -- instead
    // of one merge per (eliminated block, neighbour) pair: a variable coupled to 400 paired rows that all
    // go in the same round (the NMF example) used to be merged 400 times.
    std::vector<i32> round_elim, touched;
    std::vector<std::vector<i32>> contrib(static_cast<size_t>(nb));     // eliminated neighbours of a block, this round
    std::vector<i64> seen(static_cast<size_t>(nb), -1);
    i64 stamp = 0;
    double t_merge = 0.0; i64 n_touched = 0;
    while (remaining > 0) {
      const i64 mind = min_valid();
      if (mind < 0) {
        // only not-ready blocks are left (a component made of zero-diagonal nodes): release them
        low = 0;
        for (i64 b = 0; b < nb; ++b)
          if (!gone[static_cast<size_t>(b)] && !ready[static_cast<size_t>(b)]) { ready[static_cast<size_t>(b)] = 1; push(deg[static_cast<size_t>(b)], static_cast<i32>(b)); }
        continue;
      }
      ++round;
      const i64 thresh = mind + relax;
      cand.clear();
      for (i64 d = mind; d <= thresh && d < static_cast<i64>(bucket.size()); ++d) {
        std::vector<i32>& bk = bucket[static_cast<size_t>(d)];
        const size_t c0 = cand.size();
        for (i32 b : bk) if (!gone[static_cast<size_t>(b)] && deg[static_cast<size_t>(b)] == d) cand.push_back(b);
        bk.clear();
        std::sort(cand.begin() + static_cast<std::ptrdiff_t>(c0), cand.end());
      }
      round_elim.clear();
      touched.clear();
      for (i32 b : cand) {
        if (gone[static_cast<size_t>(b)]) continue;
        if (mark[static_cast<size_t>(b)] == round) continue;     // a neighbour goes this round
        gone[static_cast<size_t>(b)] = 1;
        --remaining;
        order.push_back(b);
        round_elim.push_back(b);
        for (i32 c : badj[static_cast<size_t>(b)]) {
          if (contrib[static_cast<size_t>(c)].empty()) touched.push_back(c);
          contrib[static_cast<size_t>(c)].push_back(b);
          mark[static_cast<size_t>(c)] = round;
        }
      }
      const double tm0 = tplan ? tnow() : 0.0;
      for (i32 c : touched) {
        ++stamp;
        merged.clear();
        seen[static_cast<size_t>(c)] = stamp;
        for (i32 x : badj[static_cast<size_t>(c)])
          if (!gone[static_cast<size_t>(x)] && seen[static_cast<size_t>(x)] != stamp) { seen[static_cast<size_t>(x)] = stamp; merged.push_back(x); }
        for (i32 b : contrib[static_cast<size_t>(c)])
          for (i32 x : badj[static_cast<size_t>(b)])
            if (seen[static_cast<size_t>(x)] != stamp) { seen[static_cast<size_t>(x)] = stamp; merged.push_back(x); }
        // (unsorted: the list is a set until its block is eliminated — nothing below looks at the order of a LIVE list;
        //  it is sorted once, when it becomes the block's struct.  Sorting every merge was 0.1 s of the NMF plan's 0.35)
        badj[static_cast<size_t>(c)].assign(merged.begin(), merged.end());
        contrib[static_cast<size_t>(c)].clear();
        deg[static_cast<size_t>(c)] = degree(c);
        ready[static_cast<size_t>(c)] = 1;
        push(deg[static_cast<size_t>(c)], c);
      }
      if (tplan) { t_merge += tnow() - tm0; n_touched += static_cast<i64>(touched.size()); }
      for (i32 b : round_elim) {
        std::sort(badj[static_cast<size_t>(b)].begin(), badj[static_cast<size_t>(b)].end());
        bstruct[static_cast<size_t>(b)] = std::move(badj[static_cast<size_t>(b)]);
        badj[static_cast<size_t>(b)] = std::vector<i32>();
      }
    }
    if (tplan) std::fprintf(stderr, "[dnlp plan]   rounds %lld, touched %lld, merge phase %.3f s\n", (long long)round, (long long)n_touched, t_merge);
    tick("elimination");
    // ---- elimination-tree levels: level-major order is a topological order of the same tree ----
    {
      std::vector<i64> pos0(static_cast<size_t>(nb));
      for (i64 k = 0; k < nb; ++k) pos0[static_cast<size_t>(order[static_cast<size_t>(k)])] = k;
      std::vector<i64> level(static_cast<size_t>(nb), 0);
      for (i64 k = 0; k < nb; ++k) {
        const i32 b = order[static_cast<size_t>(k)];
        i64 par = -1;
        for (i32 c : bstruct[static_cast<size_t>(b)]) if (par < 0 || pos0[static_cast<size_t>(c)] < par) par = pos0[static_cast<size_t>(c)];
        if (par >= 0) {
          const i32 pb = order[static_cast<size_t>(par)];
          level[static_cast<size_t>(pb)] = std::max(level[static_cast<size_t>(pb)], level[static_cast<size_t>(b)] + 1);
        }
      }
      {
        // stable by level: a counting sort (levels are small integers)
        i64 maxl = 0;
        for (i64 k = 0; k < nb; ++k) maxl = std::max(maxl, level[static_cast<size_t>(k)]);
        std::vector<i64> at(static_cast<size_t>(maxl) + 2, 0);
        for (i64 k = 0; k < nb; ++k) ++at[static_cast<size_t>(level[static_cast<size_t>(k)]) + 1];
        for (i64 l = 0; l <= maxl; ++l) at[static_cast<size_t>(l) + 1] += at[static_cast<size_t>(l)];
        std::vector<i32> sorted(order.size());
        for (i32 b : order) sorted[static_cast<size_t>(at[static_cast<size_t>(level[static_cast<size_t>(b)])]++)] = b;
        order.swap(sorted);
      }
      lev_off.assign(1, 0);
      for (i64 k = 1; k <= nb; ++k)
        if (k == nb || level[static_cast<size_t>(order[static_cast<size_t>(k)])] != level[static_cast<size_t>(order[static_cast<size_t>(k - 1)])]) lev_off.push_back(static_cast<i32>(k));
      if (nb == 0) lev_off.assign(1, 0);
    }
    pred_levels = static_cast<i64>(lev_off.size()) - 1;
    pred_triples = 0;
    for (i64 b = 0; b < nb; ++b) {
      i64 sz = 0;
      for (i32 c : bstruct[static_cast<size_t>(b)]) sz += bsize(c);
      pred_triples += sz * (sz + 1) / 2;
    }
    bn_ = std::move(bn);
    order_ = std::move(order);
    bstruct_ = std::move(bstruct);
    tick("levels");
  }

  // value layout, assembly maps and the update program of the analysed order
  void layout(const std::vector<i32>& hr, const std::vector<i32>& hc, const std::vector<i32>& jr, const std::vector<i32>& jc,
              const std::vector<char>& fixed_var) {
    const bool tplan = std::getenv("DNLP_TIME_PLAN") != nullptr;
    auto tnow = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double tp0 = tnow();
    auto tick = [&](const char* what) { if (tplan) { const double t1 = tnow(); std::fprintf(stderr, "[dnlp plan] %-28s %.3f s\n", what, t1 - tp0); tp0 = t1; } };
    const i64 nn = n;
    auto is_fixed = [&](i32 u) { return u < N && fixed_var[static_cast<size_t>(u)] != 0; };
    const std::vector<std::array<i32, 2>>& bn = bn_;
    const std::vector<i32>& order = order_;
    const std::vector<std::vector<i32>>& bstruct = bstruct_;
    const i64 nb = static_cast<i64>(bn.size());
    auto bsize = [&](i32 b) { return bn[static_cast<size_t>(b)][1] >= 0 ? 2 : 1; };
    // ---- layout: values = [D blocks | L blocks], in elimination order ----
    std::vector<i64> pos(static_cast<size_t>(nb));          // elimination position of every block
    for (i64 k = 0; k < nb; ++k) pos[static_cast<size_t>(order[static_cast<size_t>(k)])] = k;
    bnode.assign(static_cast<size_t>(2 * nb), -1);
    soff.assign(static_cast<size_t>(nb + 1), 0);
    doff.assign(static_cast<size_t>(nb), 0);
    loff.assign(static_cast<size_t>(nb), 0);
    i64 v = 0;
    maxs = 0;
    for (i64 k = 0; k < nb; ++k) {
      const i32 b = order[static_cast<size_t>(k)];
      bnode[static_cast<size_t>(2 * k)] = bn[static_cast<size_t>(b)][0];
      bnode[static_cast<size_t>(2 * k + 1)] = bn[static_cast<size_t>(b)][1];
      doff[static_cast<size_t>(k)] = v;
      v += bsize(b) == 2 ? 3 : 1;
    }
    for (i64 k = 0; k < nb; ++k) {
      const i32 b = order[static_cast<size_t>(k)];
      // struct nodes sorted by (elimination position of their block, node)
      std::vector<std::pair<i64, i32>> sn;
      for (i32 c : bstruct[static_cast<size_t>(b)]) {
        sn.push_back({pos[static_cast<size_t>(c)], bn[static_cast<size_t>(c)][0]});
        if (bn[static_cast<size_t>(c)][1] >= 0) sn.push_back({pos[static_cast<size_t>(c)], bn[static_cast<size_t>(c)][1]});
      }
      std::sort(sn.begin(), sn.end());
      soff[static_cast<size_t>(k + 1)] = soff[static_cast<size_t>(k)] + static_cast<i64>(sn.size());
      for (auto& e : sn) { sidx.push_back(e.second); sblk.push_back(static_cast<i32>(k)); }
      loff[static_cast<size_t>(k)] = v;
      v += static_cast<i64>(sn.size()) * bsize(b);
      maxs = std::max<i64>(maxs, static_cast<i64>(sn.size()));
    }
    nvals = v;
    nnzL = v;
    fill_ratio = static_cast<double>(v) / (0.5 * static_cast<double>(nn) * static_cast<double>(nn + 1));
    // position of a node inside its block, elimination position of its block
    std::vector<i64> npos(static_cast<size_t>(nn));
    std::vector<i32> ncol(static_cast<size_t>(nn));
    for (i64 k = 0; k < nb; ++k)
      for (int c = 0; c < 2; ++c) {
        const i32 u = bnode[static_cast<size_t>(2 * k + c)];
        if (u >= 0) { npos[static_cast<size_t>(u)] = k; ncol[static_cast<size_t>(u)] = c; }
      }
    // address of the entry (u, v) of the (reduced) matrix
    auto addr = [&](i32 u, i32 w) -> i64 {
      i64 ku = npos[static_cast<size_t>(u)], kw = npos[static_cast<size_t>(w)];
      if (ku == kw) {
        if (u == w) return doff[static_cast<size_t>(ku)] + (ncol[static_cast<size_t>(u)] == 0 ? 0 : 2);
        return doff[static_cast<size_t>(ku)] + 1;
      }
      if (ku > kw) { std::swap(ku, kw); std::swap(u, w); }
      // u's block is eliminated first: row of w in its struct
      const i64 s0 = soff[static_cast<size_t>(ku)], s1 = soff[static_cast<size_t>(ku + 1)];
      const int bk = bnode[static_cast<size_t>(2 * ku + 1)] >= 0 ? 2 : 1;
      // struct is sorted by (block position, node): linear search is fine for the small structs,
      // binary search on the pair key for the large ones
      i64 lo = s0, hi = s1;
      while (lo < hi) {
        const i64 mid = (lo + hi) / 2;
        const i32 x = sidx[static_cast<size_t>(mid)];
        const bool less = npos[static_cast<size_t>(x)] < kw || (npos[static_cast<size_t>(x)] == kw && x < w);
        if (less) lo = mid + 1; else hi = mid;
      }
      if (lo >= s1 || sidx[static_cast<size_t>(lo)] != w) return -1;
      return loff[static_cast<size_t>(ku)] + (lo - s0) * bk + ncol[static_cast<size_t>(u)];
    };
    tick("layout");
    // ---- assembly maps ----
    hpos.resize(hr.size());
    for (size_t p = 0; p < hr.size(); ++p) hpos[p] = static_cast<i32>(addr(hr[p], hc[p]));
    jpos.resize(jr.size());
    for (size_t p = 0; p < jr.size(); ++p) jpos[p] = static_cast<i32>(addr(static_cast<i32>(N + jr[p]), jc[p]));
    dpos.resize(static_cast<size_t>(nn));
    for (i64 u = 0; u < nn; ++u) dpos[static_cast<size_t>(u)] = static_cast<i32>(addr(static_cast<i32>(u), static_cast<i32>(u)));
    // entries that touch a fixed variable have no address (-1): the assembly masks them anyway
    for (size_t p = 0; p < hr.size(); ++p)
      if (hpos[p] < 0 && !is_fixed(hr[p]) && !is_fixed(hc[p])) throw std::runtime_error("sparse KKT plan: Hessian entry outside the pattern");
    for (size_t p = 0; p < jr.size(); ++p)
      if (jpos[p] < 0 && !is_fixed(jc[p])) throw std::runtime_error("sparse KKT plan: Jacobian entry outside the pattern");
    tick("assembly maps");
    // ---- update program ----
    // Every pair (iu >= iv) of a block's struct updates the entry (S[iu], S[iv]).  For a fixed iv the
    // destinations live in the struct of S[iv]'s block, which is sorted by the same (block position, node)
    // key as S: one merge scan per iv instead of a binary search per pair (6e7 pairs in the NMF example).
    // Blocks are independent and their triple ranges are known up front (sz (sz + 1) / 2 each), so the program is
    // written in place by a few host threads, blocks handed out in chunks from a shared counter.
    // (dense tail: chosen here, before the program is written, so that the tail x tail pairs of the panel blocks are
    //  never generated — they are most of the program where a tail exists)
    std::vector<i64> tcut(static_cast<size_t>(nb));
    for (i64 k = 0; k < nb; ++k) tcut[static_cast<size_t>(k)] = soff[static_cast<size_t>(k + 1)] - soff[static_cast<size_t>(k)];
    panels_dropped = false;
    pg_src.clear(); pg_dst.clear(); pg_off.clear(); pg_cols.clear(); pg_maxcols = 0;
    tail_lev = -1; tail_n = 0; tail_ld = 0;
    if (allow_tail) choose_tail();
    if (tail_n > 0) {
      tail_ld = (tail_n + 7) / 8 * 8;
      const i64 tb0 = lev_off[static_cast<size_t>(tail_lev)];
      std::vector<i32> tpos(static_cast<size_t>(nn), -1);
      for (i64 j = 0; j < tail_n; ++j) tpos[static_cast<size_t>(tnode[static_cast<size_t>(j)])] = static_cast<i32>(j);
      pg_off.assign(1, 0);
      for (i64 lev = 0; lev < tail_lev; ++lev) {
        i64 cols = 0;
        for (i64 k = lev_off[static_cast<size_t>(lev)]; k < lev_off[static_cast<size_t>(lev) + 1]; ++k) {
          const i64 s0 = soff[static_cast<size_t>(k)], sz = soff[static_cast<size_t>(k + 1)] - s0;
          i64 h = sz;                                   // the struct is sorted by elimination position: tail nodes last
          while (h > 0 && npos[static_cast<size_t>(sidx[static_cast<size_t>(s0 + h - 1)])] >= tb0) --h;
          if (sz - h < 12) continue;
          tcut[static_cast<size_t>(k)] = h;
          panels_dropped = true;
          const int bk = bnode[static_cast<size_t>(2 * k + 1)] >= 0 ? 2 : 1;
          for (int c = 0; c < bk; ++c, ++cols)
            for (i64 i = h; i < sz; ++i) {
              pg_src.push_back(static_cast<i32>(loff[static_cast<size_t>(k)] + i * bk + c));
              pg_dst.push_back(static_cast<i32>(tpos[static_cast<size_t>(sidx[static_cast<size_t>(s0 + i)])] + cols * tail_ld));
            }
        }
        pg_cols.push_back(cols);
        pg_off.push_back(static_cast<i64>(pg_src.size()));
        pg_maxcols = std::max(pg_maxcols, cols);
      }
      // the tail's own blocks are never walked by the level loops: no triples for them either (r^3 / 6 of them)
      for (i64 k = tb0; k < nb; ++k) if (tcut[static_cast<size_t>(k)] > 0) { tcut[static_cast<size_t>(k)] = 0; panels_dropped = true; }
    }
    toff.assign(static_cast<size_t>(nb + 1), 0);
    for (i64 k = 0; k < nb; ++k) {
      const i64 sz = soff[static_cast<size_t>(k + 1)] - soff[static_cast<size_t>(k)], h = tcut[static_cast<size_t>(k)];
      const i64 nxt = static_cast<i64>(toff[static_cast<size_t>(k)]) + h * sz - h * (h - 1) / 2;      // pairs (iu >= iv) with iv < h
      if (nxt >= (static_cast<i64>(1) << 31)) throw std::runtime_error("sparse KKT plan: update program too long for 32-bit offsets");
      toff[static_cast<size_t>(k + 1)] = static_cast<i32>(nxt);
    }
    {
      const size_t total = static_cast<size_t>(toff[static_cast<size_t>(nb)]);
      tdst.resize(total); tiu.resize(total); tiv.resize(total); tblk.resize(total);
      std::atomic<i64> next{0};
      std::atomic<bool> failed{false};
      auto work = [&]() {
        const i64 chunk = 64;
        for (;;) {
          const i64 kb = next.fetch_add(chunk);
          if (kb >= nb || failed.load()) return;
          for (i64 k = kb; k < std::min(nb, kb + chunk); ++k) {
            const i64 s0 = soff[static_cast<size_t>(k)], sz = soff[static_cast<size_t>(k + 1)] - s0, hcut = tcut[static_cast<size_t>(k)];
            size_t q = static_cast<size_t>(toff[static_cast<size_t>(k)]);
            for (i64 iv = 0; iv < hcut; ++iv) {
              const i32 w = sidx[static_cast<size_t>(s0 + iv)];
              const i64 kw = npos[static_cast<size_t>(w)];
              const i64 w0 = soff[static_cast<size_t>(kw)], w1 = soff[static_cast<size_t>(kw + 1)];
              const int bk = bnode[static_cast<size_t>(2 * kw + 1)] >= 0 ? 2 : 1;
              i64 pcur = w0;
              for (i64 iu = iv; iu < sz; ++iu, ++q) {
                const i32 u = sidx[static_cast<size_t>(s0 + iu)];
                i64 a;
                if (npos[static_cast<size_t>(u)] == kw) {
                  a = (u == w) ? doff[static_cast<size_t>(kw)] + (ncol[static_cast<size_t>(u)] == 0 ? 0 : 2) : doff[static_cast<size_t>(kw)] + 1;
                } else {
                  while (pcur < w1 && sidx[static_cast<size_t>(pcur)] != u) ++pcur;
                  if (pcur >= w1) { failed.store(true); return; }
                  a = loff[static_cast<size_t>(kw)] + (pcur - w0) * bk + ncol[static_cast<size_t>(w)];
                }
                tdst[q] = static_cast<i32>(a);
                tiu[q] = static_cast<i32>(iu);
                tiv[q] = static_cast<i32>(iv);
                tblk[q] = static_cast<i32>(k);
              }
            }
          }
        }
      };
      unsigned nthreads = total > (1u << 20) ? std::min(4u, std::max(1u, std::thread::hardware_concurrency())) : 1u;
      if (const char* ev = std::getenv("DNLP_PLAN_THREADS")) nthreads = static_cast<unsigned>(std::max(1, std::atoi(ev)));
      std::vector<std::thread> pool;
      for (unsigned t = 1; t < nthreads; ++t) pool.emplace_back(work);
      work();
      for (std::thread& th : pool) th.join();
      if (failed.load()) throw std::runtime_error("sparse KKT plan: fill entry outside the pattern");
    }
    tick("update program");
    lev_blk.clear(); lev_row.clear(); lev_trip.clear(); lev_val.clear();
    for (i64 b : lev_off) {
      lev_blk.push_back(b);
      lev_row.push_back(soff[static_cast<size_t>(b)]);
      lev_trip.push_back(toff[static_cast<size_t>(b)]);
      lev_val.push_back(b < nb ? loff[static_cast<size_t>(b)] : nvals);
    }
    // ---- order-fixed accumulation: the triples of a level sorted by destination (stable counting sort), one group per
    // destination; struct rows grouped by the node they point at ----
    {
      const i64 nl = static_cast<i64>(lev_off.size()) - 1;
      gdst.clear(); goff.assign(1, 0); lev_g.assign(1, 0);
      std::vector<i32> cnt(static_cast<size_t>(nvals), 0);      // (32-bit: the random walk over this array is the phase's cost)
      std::vector<i32> touched, t_dst, t_iu, t_iv, t_blk;
      for (i64 lev = 0; lev < nl; ++lev) {
        const size_t t0 = static_cast<size_t>(lev_trip[static_cast<size_t>(lev)]), t1 = static_cast<size_t>(lev_trip[static_cast<size_t>(lev) + 1]);
        touched.clear();
        i32 dmin = std::numeric_limits<i32>::max(), dmax = -1;
        for (size_t q = t0; q < t1; ++q) {
          const i32 d = tdst[q];
          if (cnt[static_cast<size_t>(d)]++ == 0) { touched.push_back(d); dmin = std::min(dmin, d); dmax = std::max(dmax, d); }
        }
        // the destinations in ascending order: a dense level (the first levels of a chain: 400 000 destinations spread over
        // all of the factor) is read off the counters in one sweep of its range instead of being sorted
        if (!touched.empty() && static_cast<size_t>(dmax - dmin) < 16 * touched.size()) {
          size_t k = 0;
          for (i32 d = dmin; d <= dmax; ++d) if (cnt[static_cast<size_t>(d)] != 0) touched[k++] = d;
        } else {
          std::sort(touched.begin(), touched.end());
        }
        i64 at = static_cast<i64>(t0);
        for (i32 d : touched) {
          const i64 c = cnt[static_cast<size_t>(d)];
          cnt[static_cast<size_t>(d)] = static_cast<i32>(at);              // from here on: next free position of this destination
          at += c;
          gdst.push_back(d);
          goff.push_back(static_cast<i32>(at));
        }
        t_dst.assign(tdst.begin() + static_cast<std::ptrdiff_t>(t0), tdst.begin() + static_cast<std::ptrdiff_t>(t1));
        t_iu.assign(tiu.begin() + static_cast<std::ptrdiff_t>(t0), tiu.begin() + static_cast<std::ptrdiff_t>(t1));
        t_iv.assign(tiv.begin() + static_cast<std::ptrdiff_t>(t0), tiv.begin() + static_cast<std::ptrdiff_t>(t1));
        t_blk.assign(tblk.begin() + static_cast<std::ptrdiff_t>(t0), tblk.begin() + static_cast<std::ptrdiff_t>(t1));
        for (size_t q = 0; q < t1 - t0; ++q) {
          const size_t to = static_cast<size_t>(cnt[static_cast<size_t>(t_dst[q])]++);
          tdst[to] = t_dst[q]; tiu[to] = t_iu[q]; tiv[to] = t_iv[q]; tblk[to] = t_blk[q];
        }
        for (i32 d : touched) cnt[static_cast<size_t>(d)] = 0;
        lev_g.push_back(static_cast<i32>(gdst.size()));
      }
      // (toff keeps the level boundaries toff[lev_off[l]]; inside a level the triples are no longer grouped by block)
      ntrip = static_cast<i64>(tdst.size());
      tau.resize(tdst.size()); tav.resize(tdst.size());
      for (size_t q = 0; q < tdst.size(); ++q) {
        const i64 k = tblk[q];
        const bool two = bnode[static_cast<size_t>(2 * k + 1)] >= 0;
        const i64 au = loff[static_cast<size_t>(k)] + (two ? 2 : 1) * static_cast<i64>(tiu[q]);
        const i64 av = loff[static_cast<size_t>(k)] + (two ? 2 : 1) * static_cast<i64>(tiv[q]);
        tau[q] = static_cast<i32>(au);
        tav[q] = two ? ~static_cast<i32>(av) : static_cast<i32>(av);
      }
      std::vector<i32>().swap(tdst); std::vector<i32>().swap(tiu); std::vector<i32>().swap(tiv); std::vector<i32>().swap(tblk);
      std::vector<i32> frow;
      fnode.clear(); foff.assign(1, 0); lev_f.assign(1, 0);
      std::vector<i64> fcnt(static_cast<size_t>(nn) + 1, 0);
      for (i32 u : sidx) ++fcnt[static_cast<size_t>(u) + 1];
      for (i64 u = 0; u < nn; ++u) fcnt[static_cast<size_t>(u) + 1] += fcnt[static_cast<size_t>(u)];
      std::vector<i32> rows_by_node(sidx.size());
      {
        std::vector<i64> fill(fcnt.begin(), fcnt.end() - 1);
        for (size_t r = 0; r < sidx.size(); ++r) rows_by_node[static_cast<size_t>(fill[static_cast<size_t>(sidx[r])]++)] = static_cast<i32>(r);
      }
      for (i64 lev = 0; lev < nl; ++lev) {
        for (i64 k = lev_off[static_cast<size_t>(lev)]; k < lev_off[static_cast<size_t>(lev) + 1]; ++k)
          for (int c = 0; c < 2; ++c) {
            const i32 u = bnode[static_cast<size_t>(2 * k + c)];
            if (u < 0) continue;
            const i64 a = fcnt[static_cast<size_t>(u)], b = fcnt[static_cast<size_t>(u) + 1];
            if (b == a) continue;
            fnode.push_back(u);
            frow.insert(frow.end(), rows_by_node.begin() + static_cast<std::ptrdiff_t>(a), rows_by_node.begin() + static_cast<std::ptrdiff_t>(b));
            foff.push_back(static_cast<i32>(frow.size()));
          }
        lev_f.push_back(static_cast<i32>(fnode.size()));
      }
      fa.resize(frow.size()); fu0.resize(frow.size()); fu1.resize(frow.size());
      for (size_t q = 0; q < frow.size(); ++q) {
        const i64 r = frow[q], k = sblk[static_cast<size_t>(r)], i = r - soff[static_cast<size_t>(k)];
        const i32 u0 = bnode[static_cast<size_t>(2 * k)], u1 = bnode[static_cast<size_t>(2 * k + 1)];
        const i64 a = loff[static_cast<size_t>(k)] + (u1 >= 0 ? 2 : 1) * i;
        fa[q] = u1 >= 0 ? ~static_cast<i32>(a) : static_cast<i32>(a);
        fu0[q] = u0;
        fu1[q] = u1 >= 0 ? u1 : u0;
      }
      fend.clear();
      if (tail_n > 0) {
        // rows of a target that come from blocks before the tail (its rows are ascending): the tail's own rows belong to
        // the dense solve
        const i64 rmax = soff[static_cast<size_t>(lev_off[static_cast<size_t>(tail_lev)])];
        fend.resize(fnode.size());
        for (size_t h = 0; h < fnode.size(); ++h) {
          i64 e = foff[h];
          while (e < foff[h + 1] && frow[static_cast<size_t>(e)] < rmax) ++e;
          fend[h] = static_cast<i32>(e);
        }
      }
    }
    tick("order-fixed groups");
    if (nvals >= (static_cast<i64>(1) << 31)) throw std::runtime_error("sparse KKT plan: factor too large for 32-bit addresses");
    bn_.clear(); bn_.shrink_to_fit();
    order_.clear(); order_.shrink_to_fit();
    bstruct_.clear(); bstruct_.shrink_to_fit();
  }

  // The longest suffix of levels that is a dense chain: 1x1 blocks only, at least 12 levels and 24 nodes (the 36-node
  // tail of the 12-image NMF: 0.31 -> 0.26 s per solve on the MI355X with the unpivoted dense factor; with Bunch-Kaufman
  // on that tail it was 2.6x SLOWER than the chain), at most
  // 8192 nodes, at least 60 % of the tail's lower triangle structurally present, and narrow (two blocks per level on
  // average at most: a wide level is parallel work the level kernels handle well).
  void choose_tail() {
    tail_lev = -1; tail_n = 0;
    tnode.clear(); tg_src.clear(); tg_dst.clear();
    const i64 nl = static_cast<i64>(lev_off.size()) - 1, nb = nblk();
    i64 best = -1, rows = 0;
    for (i64 lev = nl - 1; lev >= 0; --lev) {
      const i64 b0 = lev_off[static_cast<size_t>(lev)], b1 = lev_off[static_cast<size_t>(lev) + 1];
      bool single = true;
      for (i64 k = b0; k < b1 && single; ++k) single = bnode[static_cast<size_t>(2 * k + 1)] < 0;
      if (!single) break;
      rows += soff[static_cast<size_t>(b1)] - soff[static_cast<size_t>(b0)];
      const i64 r = nb - b0, levels = nl - lev;
      if (r > 8192 || r > 2 * levels) break;
      if (levels >= 12 && r >= 24 && static_cast<double>(rows) >= 0.6 * 0.5 * static_cast<double>(r) * static_cast<double>(r - 1)) best = lev;
    }
    if (best < 0) return;
    tail_lev = best;
    const i64 b0 = lev_off[static_cast<size_t>(best)];
    tail_n = nb - b0;
    std::vector<i32> tpos(static_cast<size_t>(n), -1);
    tnode.resize(static_cast<size_t>(tail_n));
    for (i64 j = 0; j < tail_n; ++j) { tnode[static_cast<size_t>(j)] = bnode[static_cast<size_t>(2 * (b0 + j))]; tpos[static_cast<size_t>(tnode[static_cast<size_t>(j)])] = static_cast<i32>(j); }
    for (i64 j = 0; j < tail_n; ++j) {
      const i64 k = b0 + j;
      tg_src.push_back(static_cast<i32>(doff[static_cast<size_t>(k)]));
      tg_dst.push_back(static_cast<i32>(j + j * tail_n));
      for (i64 i = 0; i < soff[static_cast<size_t>(k) + 1] - soff[static_cast<size_t>(k)]; ++i) {
        const i32 row = tpos[static_cast<size_t>(sidx[static_cast<size_t>(soff[static_cast<size_t>(k)] + i)])];
        if (row < 0 || row <= j) { tail_lev = -1; tail_n = 0; tnode.clear(); tg_src.clear(); tg_dst.clear(); return; }   // (cannot happen: structs point forward)
        tg_src.push_back(static_cast<i32>(loff[static_cast<size_t>(k)] + i));
        tg_dst.push_back(static_cast<i32>(row + j * tail_n));
      }
    }
  }

  template <class E> SparsePlan upload(E* ex, bool with_tail = false) const {
    SparsePlan p;
    p.n = n; p.N = N; p.m = m; p.nblk = nblk(); p.nvals = nvals; p.ntrip = ntrip; p.maxs = maxs;
    auto up = [&](auto*& dst, const auto& src) {
      using T = std::remove_pointer_t<std::remove_reference_t<decltype(dst)>>;
      dst = ex->template alloc<T>(src.size());
      if (!src.empty()) ex->h2d(dst, src.data(), src.size() * sizeof(T));
    };
    up(p.bnode, bnode); up(p.soff, soff); up(p.sidx, sidx); up(p.doff, doff); up(p.loff, loff); up(p.toff, toff);
    up(p.tau, tau); up(p.tav, tav); up(p.hpos, hpos); up(p.jpos, jpos); up(p.dpos, dpos);
    up(p.lev_off, lev_off); up(p.sblk, sblk);
    up(p.gdst, gdst); up(p.goff, goff); up(p.lev_g, lev_g); up(p.fnode, fnode); up(p.foff, foff); up(p.lev_f, lev_f);
    up(p.fa, fa); up(p.fu0, fu0); up(p.fu1, fu1);
    if (with_tail && tail_n > 0) up(p.fend, fend);
    p.ngrp = static_cast<i64>(gdst.size()); p.nfwd = static_cast<i64>(fnode.size());
    p.h_lev_g = lev_g.data(); p.h_lev_f = lev_f.data(); p.h_fwd_rows = foff.data();
    p.nlev = static_cast<i64>(lev_off.size()) - 1;
    p.nlev_run = p.nlev; p.nblk_run = p.nblk;
    if (panels_dropped && !(with_tail && tail_n > 0))
      throw std::runtime_error("sparse plan: built for the dense tail (its panel blocks carry no tail x tail triples); this space needs a full plan");
    if (with_tail && tail_n > 0) {
      p.tail_n = tail_n; p.tg_count = static_cast<i64>(tg_src.size());
      p.nlev_run = tail_lev; p.nblk_run = lev_off[static_cast<size_t>(tail_lev)];
      up(p.tnode, tnode); up(p.tg_src, tg_src); up(p.tg_dst, tg_dst);
      p.tail_ld = tail_ld; p.pg_maxcols = pg_maxcols;
      up(p.pg_off, pg_off); up(p.pg_src, pg_src); up(p.pg_dst, pg_dst); up(p.pg_cols, pg_cols);
      p.h_pg_off = pg_off.data(); p.h_pg_cols = pg_cols.data();
    }
    p.h_lev_blk = lev_blk.data(); p.h_lev_row = lev_row.data(); p.h_lev_trip = lev_trip.data(); p.h_lev_val = lev_val.data();
    return p;
  }
};

// Plan of a loaded tape: structural information from the host copies.
template <class E>
inline void build_sparse_plan(const Tape<E>& t, SparsePlanHost& plan, bool bounds_relaxed, const std::vector<double>* jac_abs0 = nullptr,
                              bool allow_tail = false) {
  std::vector<char> zero_diag(static_cast<size_t>(t.N), 1), eq(static_cast<size_t>(t.m), 0), fixed(static_cast<size_t>(t.N), 0);
  if (!bounds_relaxed)
    for (i64 j = 0; j < t.N; ++j) fixed[static_cast<size_t>(j)] = t.h_lb[static_cast<size_t>(j)] == t.h_ub[static_cast<size_t>(j)];
  // a Hessian diagonal entry is not a safe pivot: entries weighted by multipliers vanish when
  // the multiplier does (iteration 0 with y = 0).  Only bound terms (Sigma > 0) are relied upon.
  for (i64 j = 0; j < t.N; ++j)
    if (t.h_lb[static_cast<size_t>(j)] > -1e19 || t.h_ub[static_cast<size_t>(j)] < 1e19) zero_diag[static_cast<size_t>(j)] = 0;
  for (i64 i = 0; i < t.m; ++i) eq[static_cast<size_t>(i)] = t.h_cl[static_cast<size_t>(i)] == t.h_cu[static_cast<size_t>(i)];
  // two candidate orders: plain minimum degree, and the relaxed multiple elimination that gives
  // shallow trees on chain-like patterns; the relaxed one is kept unless it costs noticeably more
  // (both are analysed -- pairing, elimination, levels -- and compared on what the numeric phase will
  // cost; the value layout and the update program, 80 % of the analysis time, are generated once)
  SparsePlanHost strict;
  if (t.N + t.m >= 20000) {
    // the two analyses are independent: side by side (NMF at notebook size: 0.2 s each, most of the symbolic phase)
    std::exception_ptr err;
    std::thread other([&] {
      try { strict.analyse(t.N, t.m, t.h_hess_rows, t.h_hess_cols, t.h_jac_rows, t.h_jac_cols, zero_diag, eq, t.h_jac_const, fixed, 0, jac_abs0); }
      catch (...) { err = std::current_exception(); }
    });
    try { plan.analyse(t.N, t.m, t.h_hess_rows, t.h_hess_cols, t.h_jac_rows, t.h_jac_cols, zero_diag, eq, t.h_jac_const, fixed, 1, jac_abs0); }
    catch (...) { other.join(); throw; }
    other.join();
    if (err) std::rethrow_exception(err);
  } else {
    strict.analyse(t.N, t.m, t.h_hess_rows, t.h_hess_cols, t.h_jac_rows, t.h_jac_cols, zero_diag, eq, t.h_jac_const, fixed, 0, jac_abs0);
    plan.analyse(t.N, t.m, t.h_hess_rows, t.h_hess_cols, t.h_jac_rows, t.h_jac_cols, zero_diag, eq, t.h_jac_const, fixed, 1, jac_abs0);
  }
  // cost model of the level-parallel numeric phase: a level costs a few barriers, a triple a
  // fraction of that per lane — halving the depth is worth up to 3x the update work
  const double tr = static_cast<double>(plan.pred_triples), ts = static_cast<double>(strict.pred_triples);
  const double lr = static_cast<double>(plan.pred_levels + 1), ls = static_cast<double>(strict.pred_levels + 1);
  const bool relaxed_ok = (lr <= 0.5 * ls && tr <= 3.0 * ts + 64.0) || (lr <= ls && tr <= 1.2 * ts + 64.0);
  if (!relaxed_ok) plan = std::move(strict);
  plan.allow_tail = allow_tail;
  plan.layout(t.h_hess_rows, t.h_hess_cols, t.h_jac_rows, t.h_jac_cols, fixed);
  if (std::getenv("DNLP_PLAN_LEVELS")) {
    std::fprintf(stderr, "[plan] dense tail: %lld nodes from level %lld on, panel columns per level up to %lld, %zu triples\n", (long long)plan.tail_n,
                 (long long)plan.tail_lev, (long long)plan.pg_maxcols, static_cast<size_t>(plan.ntrip));
    const i64 nl = static_cast<i64>(plan.lev_off.size()) - 1;
    for (i64 l = 0; l < nl; ++l)
      std::fprintf(stderr, "[plan] level %lld: blocks %lld struct rows %lld triples %lld\n", (long long)l,
                   (long long)(plan.lev_off[static_cast<size_t>(l) + 1] - plan.lev_off[static_cast<size_t>(l)]),
                   (long long)(plan.lev_row[static_cast<size_t>(l) + 1] - plan.lev_row[static_cast<size_t>(l)]),
                   (long long)(plan.lev_trip[static_cast<size_t>(l) + 1] - plan.lev_trip[static_cast<size_t>(l)]));
  }
}

// The pivot blocks of a built plan in elimination order (kkt_dense.h paired mode: a dense matrix assembled in this order
// has every matched pair adjacent and no structurally zero pivot).  Empty when no plan was built (a pattern that is
// dense to begin with: its minimum-degree analysis alone would cost more than the factorisations it saves).
inline void static_pivot_order(const SparsePlanHost& plan, i64 n, std::vector<i32>& perm, std::vector<i32>& pair_pos) {
  perm.clear();
  pair_pos.clear();
  if (plan.nblk() <= 0 || static_cast<i64>(plan.bnode.size()) != 2 * plan.nblk() || plan.n != n) return;
  perm.assign(static_cast<size_t>(n), -1);
  i32 pos = 0;
  for (i64 k = 0; k < plan.nblk(); ++k) {
    const i32 u0 = plan.bnode[static_cast<size_t>(2 * k)], u1 = plan.bnode[static_cast<size_t>(2 * k + 1)];
    if (u1 >= 0) pair_pos.push_back(pos);
    perm[static_cast<size_t>(u0)] = pos++;
    if (u1 >= 0) perm[static_cast<size_t>(u1)] = pos++;
  }
  bool ok = pos == n;
  for (i64 k = 0; k < n && ok; ++k) ok = perm[static_cast<size_t>(k)] >= 0;
  if (!ok) { perm.clear(); pair_pos.clear(); }
}

}  // namespace dnlp
