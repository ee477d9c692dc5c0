// Template-specialised batch solver, part 7: the generator of the per-template LDL^T phases.
//
// From a template's plan block (wave_plan.h) this writes (a) the 32-bit work tables G — one task per lane and phase, see
// wave_gen_rt.h for the word formats — and (b) the TEXT of
//     wgen::ldl_factor<P>(S)          = wave_ipm.h ldl_factor_impl
//     wgen::ldl_solve<P, TWO>(S, x, y) = wave_ipm.h ldl_solve
// as a sequence of wgrt:: helper calls whose template arguments are the literals of this template: table bases, active
// lanes, entry counts, block kinds.  wave_ipm.h calls them when it is compiled with -DDNLP_WAVE_GEN (wave_codegen.h does
// that for the per-template kernel; tests do it for the host lane and compare bits with the interpreted text).
// Role in the reference: what MUMPS does behind ipopt_nlpif.py:170 for one small KKT system, once per iteration and solve.
//
// Plain host C++ (no HIP).
#pragma once
#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "wave_hdr.h"

namespace dnlp {

struct WaveGen {
  std::vector<uint32_t> G;        // work tables of all phases
  std::string code;               // namespace dnlp::wgen { ... } (function templates)
  int phases_factor = 0, phases_solve = 0;
};

namespace wgen_detail {

struct Emit {
  std::vector<uint32_t>& G;
  std::string& s;
  Emit(std::vector<uint32_t>& g, std::string& t) : G(g), s(t) {}
  int reserve(size_t words) { const int at = static_cast<int>(G.size()); G.resize(G.size() + words, 0u); return at; }
  void line(const char* fmt, ...) __attribute__((format(printf, 2, 3))) {
    char b[512];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(b, sizeof b, fmt, ap);
    va_end(ap);
    s += b;
  }
};

inline uint32_t lo16(int v) {
  if (v < 0 || v > 0xffff) throw std::runtime_error("wave gen: an index does not fit 16 bits");
  return static_cast<uint32_t>(v);
}

}  // namespace wgen_detail

// can the phases of this plan be generated?  ("" = yes)
inline const char* wave_gen_refusal(const WaveHdr& h) {
  if (h.sp_nvals >= 32768 || h.N + h.m >= 65535 || 3 * h.sp_nblk + 8 >= (1 << 30)) return "plan beyond the generated tables' 16-bit fields";
  return "";
}

// a level of at most kWideMaxTasks tasks with at least kWideMinEntries entries each is run task by task with the entries
// across the lanes (wave_gen_rt.h fwdw / bwdw) instead of a task per lane
constexpr int kWideMaxTasks = 4, kWideMinEntries = 8;

inline WaveGen wave_generate(const std::vector<i32>& blk) {
  using namespace wgen_detail;
  const WaveHdr& h = *reinterpret_cast<const WaveHdr*>(blk.data());
  if (wave_gen_refusal(h)[0]) throw std::runtime_error(wave_gen_refusal(h));
  auto T = [&](i32 off) { return blk.data() + off; };
  const i32 *bnode = T(h.bnode), *soff = T(h.soff), *loff = T(h.loff), *doff = T(h.doff), *lev_off = T(h.lev_off), *sblk = T(h.sblk), *sidx = T(h.sidx),
            *lev_f = T(h.lev_f), *fnode = T(h.fnode), *foff = T(h.foff), *fa = T(h.fa), *fu0 = T(h.fu0), *fu1 = T(h.fu1), *lev_g = T(h.lev_g),
            *gdst = T(h.gdst), *goff = T(h.goff), *tau = T(h.tau), *tav = T(h.tav), *lev_r = T(h.lev_r), *lev_t = T(h.lev_t);
  const int nlev = h.tail_L;                // (the levels before the dense tail; = sp_nlev without one)
  const int nblk = h.sp_nblk;
  WaveGen out;
  Emit E(out.G, out.code);
  E.line("namespace dnlp {\nnamespace wgen {\n");
  // ================================================================ factorisation
  E.line("// wave_ipm.h ldl_factor_impl for THIS template: %d levels before the dense tail (order %d), %d blocks, %d values, %d update triples\n",
         nlev, h.tail_T, nblk, h.sp_nvals, h.sp_ntrip);
  E.line("template <class P, class WS> DNLP_HD bool ldl_factor(WS* S) {\n"
         "  typedef typename P::D WD;\n  typedef WaveIpm<P> W;\n"
         "  WD* vals = WV(svals);\n  WD* w = WV(swork);\n  WD* dinv = WV(swork) + wspec::k_nvals;\n  WD* scr = WV(scr);\n"
         "  typename P::G G = P::gtab();\n"
         "  double nneg = 0.0, nzero = 0.0, bad = 0.0;\n");
  for (int lev = 0; lev < nlev; ++lev) {
    const int b0 = lev_off[lev], b1 = lev_off[lev + 1], r0 = lev_r[lev], r1 = lev_r[lev + 1];
    E.line("  // level %d: %d blocks, %d struct rows\n", lev, b1 - b0, r1 - r0);
    // pivots
    for (int s0 = b0; s0 < b1; s0 += 64) {
      const int nact = std::min(64, b1 - s0);
      const int at = E.reserve(2 * static_cast<size_t>(nact));
      int kinds = 0;
      for (int j = 0; j < nact; ++j) {
        const int k = s0 + j, kind = bnode[2 * k + 1] < 0 ? 1 : 2;
        kinds |= kind;
        out.G[static_cast<size_t>(at + j)] = lo16(doff[k]) | (static_cast<uint32_t>(kind) << 16);
        out.G[static_cast<size_t>(at + nact + j)] = static_cast<uint32_t>(3 * k);
      }
      E.line("  WG_BEGIN wgrt::piv<%d, %d, %d>(lane, G, vals, dinv, nneg, nzero, bad); WG_END\n", at, nact, kinds);
      ++out.phases_factor;
    }
    if (r1 == r0) continue;
    // row scaling
    for (int s0 = r0; s0 < r1; s0 += 64) {
      const int nact = std::min(64, r1 - s0);
      const int at = E.reserve(2 * static_cast<size_t>(nact));
      int kinds = 0;
      for (int j = 0; j < nact; ++j) {
        const int r = s0 + j, k = sblk[r], i = r - soff[k];
        const bool one = bnode[2 * k + 1] < 0;
        kinds |= one ? 1 : 2;
        const int a = one ? loff[k] + i : loff[k] + 2 * i;
        out.G[static_cast<size_t>(at + j)] = lo16(a) | (static_cast<uint32_t>(one ? 1 : 2) << 16);
        out.G[static_cast<size_t>(at + nact + j)] = static_cast<uint32_t>(3 * k);
      }
      E.line("  WG_BEGIN wgrt::scl<%d, %d, %d>(lane, G, vals, w, dinv); WG_END\n", at, nact, kinds);
      ++out.phases_factor;
    }
    const int g0 = lev_g[lev], g1 = lev_g[lev + 1], t0 = lev_t[lev], ntr = lev_t[lev + 1] - t0;
    if (ntr == 0) continue;
    // products of the update triples, side by side
    for (int q0 = 0; q0 < ntr; q0 += 64) {
      const int nact = std::min(64, ntr - q0);
      const int at = E.reserve(static_cast<size_t>(nact));
      int kinds = 0;
      for (int j = 0; j < nact; ++j) {
        const int q = t0 + q0 + j, au = tau[q], av = tav[q];
        const bool two = av < 0;
        kinds |= two ? 2 : 1;
        const int bv = two ? ~av : av;
        if (bv > 0x7fff) throw std::runtime_error("wave gen: a value index does not fit 15 bits");
        out.G[static_cast<size_t>(at + j)] = lo16(au) | (static_cast<uint32_t>(bv) << 16) | (two ? 0x80000000u : 0u);
      }
      E.line("  WG_BEGIN wgrt::upd<%d, %d, %d, %d>(lane, G, vals, w, scr); WG_END\n", at, nact, q0, kinds);
      ++out.phases_factor;
    }
    // every destination's run, added in storage order
    for (int s0 = g0; s0 < g1; s0 += 64) {
      const int nact = std::min(64, g1 - s0);
      const int at = E.reserve(2 * static_cast<size_t>(nact));
      int maxc = 0, minc = 1 << 30;
      for (int j = 0; j < nact; ++j) {
        const int gq = s0 + j, cnt = goff[gq + 1] - goff[gq];
        maxc = std::max(maxc, cnt); minc = std::min(minc, cnt);
        out.G[static_cast<size_t>(at + j)] = lo16(gdst[gq]) | (lo16(cnt) << 16);
        out.G[static_cast<size_t>(at + nact + j)] = static_cast<uint32_t>(goff[gq] - t0);
      }
      E.line("  WG_BEGIN wgrt::gsum<%d, %d, %d, %s>(lane, G, vals, scr); WG_END\n", at, nact, maxc, maxc != minc ? "true" : "false");
      ++out.phases_factor;
    }
  }
  E.line("  if (wspec::k_tail_T > 0) W::tail_factor(S, nneg, nzero, bad);\n"
         "  nneg = P::sum(nneg); nzero = P::sum(nzero); bad = P::sum(bad);\n"
         "  S->o_i[1] = static_cast<int>(nneg);\n  S->o_i[2] = static_cast<int>(nzero);\n  return bad == 0.0;\n}\n\n");
  // ================================================================ solve
  E.line("// wave_ipm.h ldl_solve for THIS template (TWO: a second right-hand side through the same phases)\n");
  E.line("template <class P, bool TWO, class WS, class WD> DNLP_HD void ldl_solve(WS* S, WD* x, WD* y) {\n"
         "  typedef WaveIpm<P> W;\n  WD* vals = WV(svals);\n  typename P::G G = P::gtab();\n");
  // forward: level by level, a target per lane
  for (int lev = 1; lev < nlev; ++lev) {
    const int h0 = lev_f[lev], h1 = lev_f[lev + 1];
    if (h1 == h0) continue;
    E.line("  // forward, level %d: %d targets\n", lev, h1 - h0);
    // a narrow level of long rows: every target by all lanes (wave_gen_rt.h fwdw), one after the other
    bool wide = h1 - h0 <= kWideMaxTasks;
    for (int hq = h0; hq < h1 && wide; ++hq) wide = foff[hq + 1] - foff[hq] >= kWideMinEntries;
    if (wide) {
      for (int hq = h0; hq < h1; ++hq) {
        const int c = foff[hq + 1] - foff[hq];
        E.line("  { double acc = 0.0, acc2 = 0.0;\n");
        for (int e0 = 0; e0 < c; e0 += 64) {
          const int cnt = std::min(64, c - e0);
          const int ea = E.reserve(2 * static_cast<size_t>(cnt));
          int kinds = 0;
          for (int j = 0; j < cnt; ++j) {
            const int q = foff[hq] + e0 + j, a = fa[q];
            const bool two = a < 0;
            kinds |= two ? 2 : 1;
            out.G[static_cast<size_t>(ea + j)] = lo16(two ? ~a : a) | (lo16(fu0[q]) << 16);
            out.G[static_cast<size_t>(ea + cnt + j)] = lo16(two ? fu1[q] : fu0[q]) | (static_cast<uint32_t>(two ? 2 : 1) << 16);
          }
          E.line("    wgrt::fwdw<P, TWO, %d, %d, %d>(G, vals, x, y, acc, acc2);\n", ea, cnt, kinds);
        }
        E.line("    wgrt::fwdw_fin<P, TWO, %d>(x, y, acc, acc2); }\n", fnode[hq]);
        ++out.phases_solve;
      }
      continue;
    }
    for (int s0 = h0; s0 < h1; s0 += 64) {
      const int nact = std::min(64, h1 - s0);
      int maxc = 0, minc = 1 << 30, kinds = 0;
      for (int j = 0; j < nact; ++j) { const int c = foff[s0 + j + 1] - foff[s0 + j]; maxc = std::max(maxc, c); minc = std::min(minc, c); }
      const int at = E.reserve(static_cast<size_t>(nact)), ea = E.reserve(2 * static_cast<size_t>(maxc) * static_cast<size_t>(nact));
      for (int j = 0; j < nact; ++j) {
        const int hq = s0 + j, c = foff[hq + 1] - foff[hq];
        out.G[static_cast<size_t>(at + j)] = lo16(fnode[hq]) | (lo16(c) << 16);
        for (int e = 0; e < c; ++e) {
          const int q = foff[hq] + e, a = fa[q];
          const bool two = a < 0;
          kinds |= two ? 2 : 1;
          out.G[static_cast<size_t>(ea + (2 * e) * nact + j)] = lo16(two ? ~a : a) | (lo16(fu0[q]) << 16);
          out.G[static_cast<size_t>(ea + (2 * e + 1) * nact + j)] = lo16(two ? fu1[q] : fu0[q]) | (static_cast<uint32_t>(two ? 2 : 1) << 16);
        }
      }
      E.line("  WG_BEGIN wgrt::fwd<TWO, %d, %d, %d, %d, %d, %s>(lane, G, vals, x, y); WG_END\n", at, ea, nact, maxc, kinds, maxc != minc ? "true" : "false");
      ++out.phases_solve;
    }
  }
  E.line("  if (wspec::k_tail_T > 0) W::tail_forward(S, x, y);\n");
  // D^-1: every block (the tail's included), side by side
  E.line("  // D^-1, %d blocks\n", nblk);
  for (int s0 = 0; s0 < nblk; s0 += 64) {
    const int nact = std::min(64, nblk - s0);
    const int at = E.reserve(2 * static_cast<size_t>(nact));
    int kinds = 0;
    for (int j = 0; j < nact; ++j) {
      const int k = s0 + j, u1 = bnode[2 * k + 1];
      kinds |= u1 < 0 ? 1 : 2;
      out.G[static_cast<size_t>(at + j)] = lo16(bnode[2 * k]) | ((u1 < 0 ? 0xffffu : lo16(u1)) << 16);
      out.G[static_cast<size_t>(at + nact + j)] = static_cast<uint32_t>(doff[k]);
    }
    E.line("  WG_BEGIN wgrt::dsol<TWO, %d, %d, %d>(lane, G, vals, x, y); WG_END\n", at, nact, kinds);
    ++out.phases_solve;
  }
  E.line("  if (wspec::k_tail_T > 0) W::tail_backward(S, x, y);\n");
  // backward: levels descending, a block per lane (blocks without struct rows subtract nothing)
  for (int lev = nlev - 1; lev >= 0; --lev) {
    const int b0 = lev_off[lev], b1 = lev_off[lev + 1];
    if (lev_r[lev + 1] == lev_r[lev]) continue;
    std::vector<int> blocks;
    for (int k = b0; k < b1; ++k) if (soff[k + 1] > soff[k]) blocks.push_back(k);
    E.line("  // backward, level %d: %d blocks with struct rows\n", lev, static_cast<int>(blocks.size()));
    bool wide = static_cast<int>(blocks.size()) <= kWideMaxTasks;
    for (size_t j = 0; j < blocks.size() && wide; ++j) wide = soff[blocks[j] + 1] - soff[blocks[j]] >= kWideMinEntries;
    if (wide) {
      for (size_t j = 0; j < blocks.size(); ++j) {
        const int k = blocks[j], sn = soff[k + 1] - soff[k], u1 = bnode[2 * k + 1];
        E.line("  { double a0 = 0.0, a1 = 0.0, c0 = 0.0, c1 = 0.0;\n");
        for (int i0 = 0; i0 < sn; i0 += 64) {
          const int cnt = std::min(64, sn - i0);
          const int ea = E.reserve(static_cast<size_t>(cnt));
          for (int i = 0; i < cnt; ++i) out.G[static_cast<size_t>(ea + i)] = lo16(sidx[soff[k] + i0 + i]);
          E.line("    wgrt::bwdw<P, TWO, %s, %d, %d, %d, %d>(G, vals, x, y, a0, a1, c0, c1);\n", u1 < 0 ? "true" : "false", ea, cnt, loff[k], i0);
        }
        E.line("    wgrt::bwdw_fin<P, TWO, %s, %d, %d>(x, y, a0, a1, c0, c1); }\n", u1 < 0 ? "true" : "false", bnode[2 * k], u1 < 0 ? 0 : u1);
        ++out.phases_solve;
      }
      continue;
    }
    for (size_t s0 = 0; s0 < blocks.size(); s0 += 64) {
      const int nact = static_cast<int>(std::min<size_t>(64, blocks.size() - s0));
      int maxc = 0, minc = 1 << 30, kinds = 0;
      for (int j = 0; j < nact; ++j) { const int k = blocks[s0 + static_cast<size_t>(j)], c = soff[k + 1] - soff[k]; maxc = std::max(maxc, c); minc = std::min(minc, c); }
      const int ew = (maxc + 1) / 2;
      const int at = E.reserve(2 * static_cast<size_t>(nact)), ea = E.reserve(static_cast<size_t>(ew) * static_cast<size_t>(nact));
      for (int j = 0; j < nact; ++j) {
        const int k = blocks[s0 + static_cast<size_t>(j)], sn = soff[k + 1] - soff[k], u1 = bnode[2 * k + 1];
        kinds |= u1 < 0 ? 1 : 2;
        out.G[static_cast<size_t>(at + j)] = lo16(bnode[2 * k]) | ((u1 < 0 ? 0xffffu : lo16(u1)) << 16);
        out.G[static_cast<size_t>(at + nact + j)] = lo16(loff[k]) | (lo16(sn) << 16);
        for (int i = 0; i < sn; ++i) {
          uint32_t& wd = out.G[static_cast<size_t>(ea + (i >> 1) * nact + j)];
          wd |= lo16(sidx[soff[k] + i]) << ((i & 1) ? 16 : 0);
        }
      }
      E.line("  WG_BEGIN wgrt::bwd<TWO, %d, %d, %d, %d, %d, %s>(lane, G, vals, x, y); WG_END\n", at, ea, nact, maxc, kinds, maxc != minc ? "true" : "false");
      ++out.phases_solve;
    }
  }
  E.line("}\n\n}  // namespace wgen\n}  // namespace dnlp\n");
  return out;
}

}  // namespace dnlp
