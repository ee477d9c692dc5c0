// Template-specialised batch solver, part 7: the generator of the per-template phases.
//
// From a template's plan block (wave_plan.h) this writes (a) the 32-bit work tables G — one task per lane and phase, see
// wave_gen_rt.h for the word formats — and (b) the TEXT of
//     wgen::ldl_factor<P>(S)                 = wave_ipm.h ldl_factor_impl   (pivots + row scaling in ONE phase per level)
//     wgen::ldl_solve<P, TWO>(S, x, y)        = wave_ipm.h ldl_solve
//     wgen::kkt_residual<P, TWO>(S, ...)      = wave_ipm.h kkt_residual_impl / kkt_residual2 (one or two systems)
//     wgen::jac_tmult<P>(S, v, out)           = wave_ipm.h jac_tmult (out of the residual's tables)
//     wgen::spmv<P, SPLIT>(S, id, ...)        = wave_ipm.h spmv for the five constant CSR maps
// as sequences of wgrt:: helper calls whose template arguments are the literals of this template: table bases, active
// lanes, entry counts, block kinds.  wave_ipm.h calls them when it is compiled with -DDNLP_WAVE_GEN (wave_codegen.h does
// that for the per-template kernels; tests do it for the host lane and compare bits with the interpreted text).
// LW = lanes that share a phase: 64 (one wavefront per instance) or 64 x the wavefronts of a workgroup per instance — then a
// phase of more than 64 tasks spreads over the wavefronts between two workgroup barriers, a NARROW one (and the wide forms)
// runs on the first wavefront alone, and what is independent inside a level (long forward targets, long outputs of the
// residual) is dealt out to the wavefronts.
// Role in the reference: what MUMPS does behind ipopt_nlpif.py:170 for one small KKT system, once per iteration and solve.
//
// Plain host C++ (no HIP).
#pragma once
#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "wave_hdr.h"

namespace dnlp {

struct WaveGen {
  std::vector<uint32_t> G;        // work tables of all phases
  std::string code;               // namespace dnlp::wgen { ... } (function templates)
  int phases_factor = 0, phases_solve = 0;
  int stage_words = 0;            // LDS words of the staging buffer the narrow phases read their tables from (0: none)
  int wwin_doubles = 0, swin_doubles = 0;      // LDS windows of the factorisation's unscaled rows / products (0: none)
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
  if (h.sp_nvals >= 32768 || h.N + h.m >= 65535 || h.sp_nblk >= 16384) return "plan beyond the generated tables' 16-bit fields";
  return "";
}

// a level of at most kWideMaxTasks tasks with at least kWideMinEntries entries each is run task by task with the entries
// across the lanes (wave_gen_rt.h fwdw / bwdw) instead of a task per lane
constexpr int kWideMaxTasks = 4, kWideMinEntries = 8;
// the largest table a narrow phase reads from the LDS staging buffer (work tables in global memory: see wave_generate)
constexpr int kStageMaxWords = 2048;

// LW: lanes that share a phase (64: one wavefront per instance; 64 x wavefronts of a workgroup per instance)
// (ww_cap / sw_cap: doubles of LDS the caller can give the factorisation's two windows — see the level loop)
inline WaveGen wave_generate(const std::vector<i32>& blk, int LW = 64, int ww_cap = 0, int sw_cap = 0) {
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
  // NARROW phases (at most 64 tasks: one slot of one wavefront) are run by the FIRST wavefront alone, with a wavefront-level
  // barrier behind them; the other wavefronts of a workgroup-per-instance kernel run ahead and wait at the next phase that
  // needs them (WG_NCLOSE: path planning's 49 two-block levels are 196 dependent narrow phases per factorisation — a
  // workgroup barrier each was most of their time).  With one wavefront per instance every phase of up to 64 tasks is
  // narrow and the two kinds of barrier are the same instruction.
  bool region_open = false;
  // STAGING (work tables in global memory — a workgroup-per-instance kernel, LW > 64).  A narrow phase is a table word from
  // L2 (300-500 cycles), then a little LDS work: path planning's solve is ~120 such phases in a row, ~670 cycles each.  The
  // phases of a region are run by ONE wavefront in program order, so each one first issues the loads of the NEXT narrow
  // phase's table into registers (WG_STAGE_NEXT), does its own work, and puts the registers into an LDS buffer at its end
  // (WG_STAGE_PUT: by then they have arrived); the next phase reads its table from that buffer (sg0_ = where its table
  // starts in G: WG_SO rebases the offsets, WG_SG is the buffer).  The first narrow phase of a region and a phase whose
  // table exceeds kStageMaxWords read G as before.  One buffer: a wavefront's LDS operations keep their order.
  // ($DNLP_WAVE_STAGE=0 turns it off.  Tried before and dropped: touching the next table's lines at a phase's start so
  //  that L1 holds them — loads return in order, the phase's own table word then waits behind the touch: 14.1 -> 13.3 k
  //  problems/s on path planning.)
  const bool stage = LW > 64 && !(std::getenv("DNLP_WAVE_STAGE") && std::atoi(std::getenv("DNLP_WAVE_STAGE")) == 0);
  int tok = 0, prev_tok = -1, cur_tok = -1, cur_g0 = 0;
  auto put = [&](const std::string& token, const std::string& val) {
    const size_t at = out.code.rfind(token);
    if (at != std::string::npos) out.code.replace(at, token.size(), val);
  };
  auto drop_pending = [&] { if (prev_tok >= 0) { put("@@N" + std::to_string(prev_tok) + "@@", "0, 0"); prev_tok = -1; } };
  auto narrow_open = [&] {          // (behind WG_NBEGIN / WG_WBEGIN)
    if (!stage) { E.line("    constexpr int sg0_ = -1;\n"); return; }
    cur_tok = tok++;
    cur_g0 = static_cast<int>(out.G.size());
    E.line("    WG_STAGE_NEXT(@@N%d@@)\n    constexpr int sg0_ = @@S%d@@;\n", cur_tok, cur_tok);
  };
  auto narrow_close = [&] {         // (in front of WG_NEND / WG_WEND)
    if (!stage) return;
    E.line("    WG_STAGE_PUT\n");
    const int words = static_cast<int>(out.G.size()) - cur_g0;
    const bool staged = prev_tok >= 0 && words > 0 && words <= kStageMaxWords;
    put("@@S" + std::to_string(cur_tok) + "@@", staged ? std::to_string(cur_g0) : std::string("-1"));
    if (prev_tok >= 0) put("@@N" + std::to_string(prev_tok) + "@@", staged ? std::to_string(cur_g0) + ", " + std::to_string(words) : std::string("0, 0"));
    if (staged) out.stage_words = std::max(out.stage_words, (words + 63) & ~63);
    prev_tok = cur_tok;
  };
  auto begin_phase = [&](int tasks) {
    const bool narrow = tasks <= 64;
    if (!narrow && region_open) { drop_pending(); E.line("  WG_NCLOSE\n"); region_open = false; }
    E.line(narrow ? "  WG_NBEGIN\n" : "  WG_BEGIN\n");
    if (narrow) narrow_open(); else E.line("    constexpr int sg0_ = -1;\n");
    return narrow;
  };
  auto end_phase = [&](bool narrow) {
    if (narrow) narrow_close();
    E.line(narrow ? "  WG_NEND\n" : "  WG_END\n");
    if (narrow) region_open = true;
  };
  auto close_region = [&] { drop_pending(); if (region_open) { E.line("  WG_NCLOSE\n"); region_open = false; } };
  E.line("#define WG_LANES %d\nnamespace dnlp {\nnamespace wgen {\n", LW);
  // ================================================================ factorisation
  E.line("// wave_ipm.h ldl_factor_impl for THIS template: %d levels before the dense tail (order %d), %d blocks, %d values, %d update triples\n",
         nlev, h.tail_T, nblk, h.sp_nvals, h.sp_ntrip);
  E.line("template <class P, class WS> DNLP_HD bool ldl_factor(WS* S) {\n"
         "  typedef typename P::D WD;\n  typedef WaveIpm<P> W;\n"
         "  WD* vals = WV(svals);\n  WD* w = WV(swork);\n  WD* scr = WV(scr);\n"
         "  typename P::G G = P::gtab();\n"
         "  double nneg = 0.0, nzero = 0.0, bad = 0.0;\n");
  for (int lev = 0; lev < nlev; ++lev) {
    const int b0 = lev_off[lev], b1 = lev_off[lev + 1], r0 = lev_r[lev], r1 = lev_r[lev + 1];
    // WINDOWS.  A level's struct rows are copied unscaled into w (scl2) for the products of its update triples (upd), which go
    // into scr for the destinations' sums (gsum): w is as long as the factor and scr as the largest level's triples — they live
    // in the slab when the vectors do (workgroup kernel), two dependent global round trips per level.  But a level touches only
    // ITS rows' span of w and its own triples: when those fit the LDS windows the caller has room for, the level's phases get
    // window pointers instead (WG_WW(lo): the window rebased to the span's first value; WG_SW) — path planning's 49 chain
    // levels are 76 values and a few dozen triples each.
    int wlo = 1 << 30, whi = 0;
    for (int r = r0; r < r1; ++r) {
      const int k = sblk[r], i = r - soff[k];
      const bool one = bnode[2 * k + 1] < 0;
      const int aa = one ? loff[k] + i : loff[k] + 2 * i;
      wlo = std::min(wlo, aa); whi = std::max(whi, aa + (one ? 1 : 2));
    }
    const int lev_ntr = lev_t[lev + 1] - lev_t[lev];
    const bool use_ww = r1 > r0 && whi - wlo <= ww_cap, use_sw = lev_ntr > 0 && lev_ntr <= sw_cap;
    if (use_ww) out.wwin_doubles = std::max(out.wwin_doubles, (whi - wlo + 1) & ~1);
    if (use_sw) out.swin_doubles = std::max(out.swin_doubles, (lev_ntr + 1) & ~1);
    char wexp[48], sexp[16];
    if (use_ww) std::snprintf(wexp, sizeof wexp, "WG_WW(%d)", wlo); else std::snprintf(wexp, sizeof wexp, "w");
    std::snprintf(sexp, sizeof sexp, use_sw ? "WG_SW" : "scr");
    E.line("  // level %d: %d blocks, %d struct rows\n", lev, b1 - b0, r1 - r0);
    // pivots (inertia, tiny 1x1 pivots fixed in place) and — in the SAME phase — the struct rows scaled by the inverse pivot that
    // every row's lane recomputes from its block's D entries (wave_gen_rt.h scl2): no inverse is stored, the rows do not wait
    const bool nb_ps = begin_phase(std::max(b1 - b0, r1 - r0));
    for (int s0 = b0; s0 < b1; s0 += LW) {
      const int nact = std::min(LW, b1 - s0);
      const int at = E.reserve(static_cast<size_t>(nact));
      int kinds = 0;
      for (int j = 0; j < nact; ++j) {
        const int k = s0 + j, kind = bnode[2 * k + 1] < 0 ? 1 : 2;
        kinds |= kind;
        out.G[static_cast<size_t>(at + j)] = lo16(doff[k]) | (static_cast<uint32_t>(kind) << 30);
      }
      E.line("    wgrt::piv2<WG_SO(%d), %d, %d>(lane, WG_SG, vals, nneg, nzero, bad);\n", at, nact, kinds);
      ++out.phases_factor;
    }
    for (int s0 = r0; s0 < r1; s0 += LW) {
      const int nact = std::min(LW, r1 - s0);
      const int at = E.reserve(static_cast<size_t>(nact));
      int kinds = 0;
      for (int j = 0; j < nact; ++j) {
        const int r = s0 + j, k = sblk[r], i = r - soff[k];
        const bool one = bnode[2 * k + 1] < 0;
        kinds |= one ? 1 : 2;
        const int aa = one ? loff[k] + i : loff[k] + 2 * i;
        if (aa > 0x7fff || doff[k] > 0x7fff) throw std::runtime_error("wave gen: a value index does not fit 15 bits");
        out.G[static_cast<size_t>(at + j)] = static_cast<uint32_t>(aa) | (static_cast<uint32_t>(doff[k]) << 15) | (static_cast<uint32_t>(one ? 1 : 2) << 30);
      }
      E.line("    wgrt::scl2<WG_SO(%d), %d, %d>(lane, WG_SG, vals, %s);\n", at, nact, kinds, wexp);
    }
    end_phase(nb_ps);
    if (r1 == r0) continue;
    const int g0 = lev_g[lev], g1 = lev_g[lev + 1], t0 = lev_t[lev], ntr = lev_t[lev + 1] - t0;
    if (ntr == 0) continue;
    // products of the update triples, side by side
    const bool nb_upd = begin_phase(ntr);
    for (int q0 = 0; q0 < ntr; q0 += LW) {
      const int nact = std::min(LW, ntr - q0);
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
      E.line("    wgrt::upd<WG_SO(%d), %d, %d, %d>(lane, WG_SG, vals, %s, %s);\n", at, nact, q0, kinds, wexp, sexp);
      ++out.phases_factor;
    }
    end_phase(nb_upd);
    // every destination's run, added in storage order
    const bool nb_gs = begin_phase(g1 - g0);
    for (int s0 = g0; s0 < g1; s0 += LW) {
      const int nact = std::min(LW, g1 - s0);
      const int at = E.reserve(2 * static_cast<size_t>(nact));
      int maxc = 0, minc = 1 << 30;
      for (int j = 0; j < nact; ++j) {
        const int gq = s0 + j, cnt = goff[gq + 1] - goff[gq];
        maxc = std::max(maxc, cnt); minc = std::min(minc, cnt);
        out.G[static_cast<size_t>(at + j)] = lo16(gdst[gq]) | (lo16(cnt) << 16);
        out.G[static_cast<size_t>(at + nact + j)] = static_cast<uint32_t>(goff[gq] - t0);
      }
      E.line("    wgrt::gsum<WG_SO(%d), %d, %d, %s>(lane, WG_SG, vals, %s);\n", at, nact, maxc, maxc != minc ? "true" : "false", sexp);
      ++out.phases_factor;
    }
    end_phase(nb_gs);
  }
  close_region();
  E.line("  if (wspec::k_tail_T > 0) W::tail_factor(S, nneg, nzero, bad);\n"
         "  { double r3[3] = {nneg, nzero, bad}; P::sum_n(r3); nneg = r3[0]; nzero = r3[1]; bad = r3[2]; }\n"
         "  S->o_i[1] = static_cast<int>(nneg);\n  S->o_i[2] = static_cast<int>(nzero);\n  return bad == 0.0;\n}\n\n");
  // ================================================================ solve
  E.line("// wave_ipm.h ldl_solve for THIS template (TWO: a second right-hand side through the same phases)\n");
  // (XP / YP: the right-hand sides' pointer types — the workgroup kernel hands them over typed by where they live, LDS or the
  //  slab: wave_ipm.h ldl_solve)
  E.line("template <class P, bool TWO, class WS, class XP, class YP> DNLP_HD void ldl_solve(WS* S, XP x, YP y) {\n"
         "  typedef WaveIpm<P> W;\n  typedef typename P::D WD;\n  const auto vals = wgrt::tvec<P, wspec::v_svals>(S);\n  typename P::G G = P::gtab();\n");
  // forward: level by level, a target per lane
  for (int lev = 1; lev < nlev; ++lev) {
    const int h0 = lev_f[lev], h1 = lev_f[lev + 1];
    if (h1 == h0) continue;
    E.line("  // forward, level %d: %d targets\n", lev, h1 - h0);
    // a narrow level of long rows: every target by all lanes (wave_gen_rt.h fwdw), one after the other
    bool wide = h1 - h0 <= kWideMaxTasks;
    for (int hq = h0; hq < h1 && wide; ++hq) wide = foff[hq + 1] - foff[hq] >= kWideMinEntries;
    if (wide) {
      // several targets too long to share a wavefront (power flow: two of ~40 rows per chain level), work tables in global
      // memory: a wavefront each, side by side, between two workgroup barriers (a phase there is a table word from L2, then
      // little: two in a row cost more than two barriers)
      bool deal = LW > 64 && h1 - h0 >= 2 && h1 - h0 <= LW / 64;
      for (int hq = h0; hq < h1 && deal; ++hq) deal = foff[hq + 1] - foff[hq] > 32 && foff[hq + 1] - foff[hq] <= 64;
      if (deal) {
        close_region();
        for (int hq = h0; hq < h1; ++hq) {
          const int c = foff[hq + 1] - foff[hq], wv = hq - h0;
          const int ea = E.reserve(2 * static_cast<size_t>(c));
          int kinds = 0;
          for (int j = 0; j < c; ++j) {
            const int q = foff[hq] + j, a = fa[q];
            const bool two = a < 0;
            kinds |= two ? 2 : 1;
            out.G[static_cast<size_t>(ea + j)] = lo16(two ? ~a : a) | (lo16(fu0[q]) << 16);
            out.G[static_cast<size_t>(ea + c + j)] = lo16(two ? fu1[q] : fu0[q]) | (static_cast<uint32_t>(two ? 2 : 1) << 16);
          }
          E.line("  WG_WAVE(%d) { double acc = 0.0, acc2 = 0.0; wgrt::fwdw<P, TWO, %d, %d, %d, %d>(G, vals, x, y, acc, acc2); wgrt::fwdw_fin<P, TWO, %d, %d>(x, y, acc, acc2); }\n",
                 wv, ea, c, kinds, wv, fnode[hq], wv);
          ++out.phases_solve;
        }
        E.line("  P::sync();\n");
        continue;
      }
      for (int hq = h0; hq < h1; ++hq) {
        const int c = foff[hq + 1] - foff[hq];
        // two targets of at most 32 rows each share ONE phase, a half of the wavefront each (path planning's 49 chain levels
        // have two targets of 19 rows: a phase there is a table word from global memory, then little — one phase less per
        // level and solve)
        if (hq + 1 < h1 && c <= 32 && foff[hq + 2] - foff[hq + 1] <= 32) {
          const int cb = foff[hq + 2] - foff[hq + 1];
          E.line("  WG_WBEGIN\n");
          narrow_open();
          const int ea = E.reserve(128);
          int kinds = 0;
          for (int half = 0; half < 2; ++half)
            for (int j = 0; j < (half ? cb : c); ++j) {
              const int q = foff[hq + half] + j, a = fa[q];
              const bool two = a < 0;
              kinds |= two ? 2 : 1;
              out.G[static_cast<size_t>(ea + 32 * half + j)] = lo16(two ? ~a : a) | (lo16(fu0[q]) << 16);
              out.G[static_cast<size_t>(ea + 64 + 32 * half + j)] = lo16(two ? fu1[q] : fu0[q]) | (static_cast<uint32_t>(two ? 2 : 1) << 16);
            }
          E.line("    wgrt::fwdw2<P, TWO, WG_SO(%d), %d, %d, %d, %d, %d>(WG_SG, vals, x, y);\n", ea, c, cb, kinds, fnode[hq], fnode[hq + 1]);
          narrow_close();
          E.line("  WG_WEND\n");
          region_open = true;
          ++out.phases_solve;
          ++hq;
          continue;
        }
        E.line("  WG_WBEGIN\n");
        narrow_open();
        E.line("    { double acc = 0.0, acc2 = 0.0;\n");
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
          E.line("    wgrt::fwdw<P, TWO, WG_SO(%d), %d, %d>(WG_SG, vals, x, y, acc, acc2);\n", ea, cnt, kinds);
        }
        E.line("    wgrt::fwdw_fin<P, TWO, %d>(x, y, acc, acc2); }\n", fnode[hq]);
        narrow_close();
        E.line("  WG_WEND\n");
        region_open = true;
        ++out.phases_solve;
      }
      continue;
    }
    const bool nb_fwd = begin_phase(h1 - h0);
    for (int s0 = h0; s0 < h1; s0 += LW) {
      const int nact = std::min(LW, h1 - s0);
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
      E.line("    wgrt::fwd<TWO, WG_SO(%d), WG_SO(%d), %d, %d, %d, %s>(lane, WG_SG, vals, x, y);\n", at, ea, nact, maxc, kinds, maxc != minc ? "true" : "false");
      ++out.phases_solve;
    }
    end_phase(nb_fwd);
  }
  close_region();
  E.line("  if (wspec::k_tail_T > 0) W::tail_forward(S, (WD*)x, (WD*)y);\n");
  // D^-1: every block (the tail's included), side by side
  E.line("  // D^-1, %d blocks\n", nblk);
  const bool nb_ds = begin_phase(nblk);
  for (int s0 = 0; s0 < nblk; s0 += LW) {
    const int nact = std::min(LW, nblk - s0);
    const int at = E.reserve(2 * static_cast<size_t>(nact));
    int kinds = 0;
    for (int j = 0; j < nact; ++j) {
      const int k = s0 + j, u1 = bnode[2 * k + 1];
      kinds |= u1 < 0 ? 1 : 2;
      out.G[static_cast<size_t>(at + j)] = lo16(bnode[2 * k]) | ((u1 < 0 ? 0xffffu : lo16(u1)) << 16);
      out.G[static_cast<size_t>(at + nact + j)] = static_cast<uint32_t>(doff[k]);
    }
    E.line("    wgrt::dsol<TWO, WG_SO(%d), %d, %d>(lane, WG_SG, vals, x, y);\n", at, nact, kinds);
    ++out.phases_solve;
  }
  end_phase(nb_ds);
  close_region();
  E.line("  if (wspec::k_tail_T > 0) W::tail_backward(S, (WD*)x, (WD*)y);\n");
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
        E.line("  WG_WBEGIN\n");
        narrow_open();
        E.line("    { double a0 = 0.0, a1 = 0.0, c0 = 0.0, c1 = 0.0;\n");
        for (int i0 = 0; i0 < sn; i0 += 64) {
          const int cnt = std::min(64, sn - i0);
          const int ea = E.reserve(static_cast<size_t>(cnt));
          for (int i = 0; i < cnt; ++i) out.G[static_cast<size_t>(ea + i)] = lo16(sidx[soff[k] + i0 + i]);
          E.line("    wgrt::bwdw<P, TWO, %s, WG_SO(%d), %d, %d, %d>(WG_SG, vals, x, y, a0, a1, c0, c1);\n", u1 < 0 ? "true" : "false", ea, cnt, loff[k], i0);
        }
        E.line("    wgrt::bwdw_fin<P, TWO, %s, %d, %d>(x, y, a0, a1, c0, c1); }\n", u1 < 0 ? "true" : "false", bnode[2 * k], u1 < 0 ? 0 : u1);
        narrow_close();
        E.line("  WG_WEND\n");
        region_open = true;
        ++out.phases_solve;
      }
      continue;
    }
    const bool nb_bwd = begin_phase(static_cast<int>(blocks.size()));
    for (size_t s0 = 0; s0 < blocks.size(); s0 += LW) {
      const int nact = static_cast<int>(std::min<size_t>(static_cast<size_t>(LW), blocks.size() - s0));
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
      E.line("    wgrt::bwd<TWO, WG_SO(%d), WG_SO(%d), %d, %d, %d, %s>(lane, WG_SG, vals, x, y);\n", at, ea, nact, maxc, kinds, maxc != minc ? "true" : "false");
      ++out.phases_solve;
    }
    end_phase(nb_bwd);
  }
  close_region();
  E.line("}\n\n");
  // ================================================================ KKT residual (products by output fused with the combination)
  {
    const i32 *hsp = T(h.hs_ptr), *hse = T(h.hs_ent), *hss = T(h.hs_src), *jcp = T(h.jc_ptr), *jce = T(h.jc_ent), *jcs = T(h.jc_src),
              *jrp = T(h.jr_ptr), *jre = T(h.jr_ent), *jrs = T(h.jr_src);
    const int N = h.N, m = h.m;
    E.line("// wave_ipm.h kkt_residual_impl / kkt_residual2 for THIS template: out = rhs - K v (and the same for a second system)\n");
    E.line("template <class P, bool TWO, class WS, class WD> DNLP_WINL DNLP_HD void kkt_residual(WS* S, double dw, const WD* v, const WD* rhsv, WD* out,\n"
           "    const WD* v2, const WD* rhsv2, WD* out2, double& en, double& sn, double& en2, double& sn2) {\n"
           "  const WD *Hs = WV(Hs), *jv = WV(jv), *sx = WV(Sx), *dd = WV(Dd), *fm = WV(fixm);\n"
           "  WD *preH = out, *preJt = WV(xt), *preJ = WV(tM), *preH2 = out2, *preJt2 = WV(dx), *preJ2 = WV(ds);\n"
           "  typename P::G G = P::gtab();\n");
    // long outputs first, by all lanes
    bool any_heavy = false;
    int n_heavy = 0;
    std::string jt_heavy, jt_light;          // the same tables drive J^T y alone (jac_tmult below)
    auto heavy = [&](const i32* ptr, const i32* ent, const i32* src, int nout, const char* a, const char* vv, const char* v2, const char* pre, const char* pre2, bool is_jc) {
      for (int g = 0; g < nout; ++g) {
        const int c = ptr[g + 1] - ptr[g];
        if (c <= kCooHeavy) continue;
        any_heavy = true;
        const int wv = n_heavy++ % (LW / 64);          // (dealt out to the wavefronts of the workgroup)
        E.line("  WG_WAVE(%d) { double acc = 0.0, acc2 = 0.0;\n", wv);
        char b[256];
        if (is_jc) { std::snprintf(b, sizeof b, "  WG_WAVE(%d) { double acc = 0.0, acc2 = 0.0;\n", wv); jt_heavy += b; }
        for (int e0 = 0; e0 < c; e0 += 64) {
          const int cnt = std::min(64, c - e0);
          const int ea = E.reserve(static_cast<size_t>(cnt));
          for (int j = 0; j < cnt; ++j) out.G[static_cast<size_t>(ea + j)] = lo16(ent[ptr[g] + e0 + j]) | (lo16(src[ptr[g] + e0 + j]) << 16);
          E.line("    wgrt::wdot<P, TWO, %d, %d, %d>(G, %s, %s, %s, acc, acc2);\n", wv, ea, cnt, a, vv, v2);
          if (is_jc) { std::snprintf(b, sizeof b, "    wgrt::wdot<P, false, %d, %d, %d>(G, jv, v, v, acc, acc2);\n", wv, ea, cnt); jt_heavy += b; }
        }
        E.line("    wgrt::wdot_fin<P, TWO, %d, %d>(%s, %s, acc, acc2); }\n", wv, g, pre, pre2);
        if (is_jc) { std::snprintf(b, sizeof b, "    wgrt::wdot_fin<P, false, %d, %d>(out, out, acc, acc2); }\n", wv, g); jt_heavy += b; }
      }
    };
    heavy(hsp, hse, hss, N, "Hs", "v", "v2", "preH", "preH2", false);
    {
      char vb[64], v2b[64];
      std::snprintf(vb, sizeof vb, "v + %d", N); std::snprintf(v2b, sizeof v2b, "v2 + %d", N);
      heavy(jcp, jce, jcs, N, "jv", vb, v2b, "preJt", "preJt2", true);
    }
    heavy(jrp, jre, jrs, m, "jv", "v", "v2", "preJ", "preJ2", false);
    if (any_heavy) E.line("  P::sync();\n");
    E.line("  double m0 = -kInf, m1 = -kInf, n0 = -kInf, n1 = -kInf;\n");
    E.line("  WG_BEGIN\n");
    for (int k0 = 0; k0 < N; k0 += LW) {
      const int nact = std::min(LW, N - k0);
      int maxh = 0, maxj = 0;
      for (int j = 0; j < nact; ++j) {
        const int k = k0 + j, ch = hsp[k + 1] - hsp[k], cj = jcp[k + 1] - jcp[k];
        if (ch <= kCooHeavy) maxh = std::max(maxh, ch);
        if (cj <= kCooHeavy) maxj = std::max(maxj, cj);
      }
      const int at = E.reserve(static_cast<size_t>(nact)), ea = E.reserve(static_cast<size_t>(maxh + maxj) * static_cast<size_t>(nact));
      for (int j = 0; j < nact; ++j) {
        const int k = k0 + j, ch = hsp[k + 1] - hsp[k], cj = jcp[k + 1] - jcp[k];
        const bool hh = ch > kCooHeavy, hj = cj > kCooHeavy;
        out.G[static_cast<size_t>(at + j)] = static_cast<uint32_t>(hh ? 0 : ch) | (static_cast<uint32_t>(hj ? 0 : cj) << 8) | (hh ? 0x10000u : 0u) | (hj ? 0x20000u : 0u);
        if (!hh) for (int e = 0; e < ch; ++e) out.G[static_cast<size_t>(ea + e * nact + j)] = lo16(hse[hsp[k] + e]) | (lo16(hss[hsp[k] + e]) << 16);
        if (!hj) for (int e = 0; e < cj; ++e) out.G[static_cast<size_t>(ea + (maxh + e) * nact + j)] = lo16(jce[jcp[k] + e]) | (lo16(N + jcs[jcp[k] + e]) << 16);
      }
      E.line("    wgrt::kres_var<TWO, %d, %d, %d, %d, %d, %d>(lane, G, Hs, jv, sx, fm, dw, v, rhsv, out, (const WD*)preH, (const WD*)preJt, v2, rhsv2, out2, (const WD*)preH2, (const WD*)preJt2, m0, m1, n0, n1);\n",
             at, ea, nact, k0, maxh, maxj);
      { char b[256]; std::snprintf(b, sizeof b, "    wgrt::cojt<%d, %d, %d, %d, %d, %d>(lane, G, jv, v - %d, out);\n", at, ea, nact, k0, maxh, maxj, N); jt_light += b; }
    }
    E.line("  WG_END\n");
    E.line("  WG_BEGIN\n");
    for (int i0 = 0; i0 < m; i0 += LW) {
      const int nact = std::min(LW, m - i0);
      int maxj = 0;
      for (int j = 0; j < nact; ++j) { const int c = jrp[i0 + j + 1] - jrp[i0 + j]; if (c <= kCooHeavy) maxj = std::max(maxj, c); }
      const int at = E.reserve(static_cast<size_t>(nact)), ea = E.reserve(static_cast<size_t>(maxj) * static_cast<size_t>(nact));
      for (int j = 0; j < nact; ++j) {
        const int i = i0 + j, c = jrp[i + 1] - jrp[i];
        const bool hv = c > kCooHeavy;
        out.G[static_cast<size_t>(at + j)] = static_cast<uint32_t>(hv ? 0 : c) | (hv ? 0x10000u : 0u);
        if (!hv) for (int e = 0; e < c; ++e) out.G[static_cast<size_t>(ea + e * nact + j)] = lo16(jre[jrp[i] + e]) | (lo16(jrs[jrp[i] + e]) << 16);
      }
      E.line("    wgrt::kres_row<TWO, %d, %d, %d, %d, %d, %d>(lane, G, jv, dd, v, rhsv, out, (const WD*)preJ, v2, rhsv2, out2, (const WD*)preJ2, m0, m1, n0, n1);\n",
             at, ea, nact, i0, N, maxj);
    }
    E.line("  WG_END\n");
    E.line("  if (TWO) { double r4[4] = {m0, m1, n0, n1}; P::vmax_n(r4); en = r4[0]; sn = r4[1]; en2 = r4[2]; sn2 = r4[3]; }\n"
           "  else { double r2[2] = {m0, m1}; P::vmax_n(r2); en = r2[0]; sn = r2[1]; }\n  P::sync();\n}\n\n");
    E.line("// wave_ipm.h jac_tmult (the product by output J^T v of the tape's index jc) out of the same tables\n");
    E.line("template <class P, class WS, class WD> DNLP_WINL DNLP_HD void jac_tmult(WS* S, const WD* v, WD* out) {\n"
           "  const WD* jv = WV(jv);\n  typename P::G G = P::gtab();\n");
    out.code += "  WG_BEGIN\n" + jt_light + "  WG_END\n";
    out.code += jt_heavy;
    E.line("  P::sync();\n}\n\n");
  }
  // ================================================================ the five constant CSR maps
  {
    struct M { const char* name; i32 ptr, idx, rows, val; };
    const M mats[5] = {{"G", h.G_ptr, h.G_idx, h.m, h.l_G}, {"Mg", h.Mg_ptr, h.Mg_idx, h.N, h.l_Mg}, {"MJ", h.MJ_ptr, h.MJ_idx, h.nnzJ, h.l_MJ},
                       {"Mw", h.Mw_ptr, h.Mw_idx, h.Z, h.l_Mw}, {"MH", h.MH_ptr, h.MH_idx, h.nnzH, h.l_MH}};
    E.line("// wave_ipm.h spmv for THIS template's constant maps (id: 0 G, 1 Mg, 2 MJ, 3 Mw, 4 MH)\n");
    E.line("template <class P, bool SPLIT, class WS, class WD> DNLP_WINL DNLP_HD void spmv(WS* S, int id, const WD* v, i32 base_off, WD* y, int scale_kind, double scalar,\n"
           "    const WD* vhi, i32 split) {\n"
           "  typename P::G G = P::gtab();\n  const WD* sg = WV(sg);\n  typename P::I* jr = WT(jac_rows);\n"
           "  WG* base = base_off >= 0 ? S->row + base_off : nullptr;\n");
    for (int q = 0; q < 5; ++q) {
      const M& mt = mats[q];
      const i32 *ptr = T(mt.ptr), *idx = T(mt.idx);
      E.line("  %sif (id == %d) {      // %s: %d rows\n    WG* val = S->row + %d;\n", q ? "else " : "", q, mt.name, mt.rows, mt.val);
      E.line("    WG_BEGIN\n");
      for (int r0 = 0; r0 < mt.rows; r0 += LW) {
        const int nact = std::min(LW, mt.rows - r0);
        int maxc = 0;
        for (int j = 0; j < nact; ++j) maxc = std::max(maxc, ptr[r0 + j + 1] - ptr[r0 + j]);
        const int at = E.reserve(static_cast<size_t>(nact)), ea = E.reserve(static_cast<size_t>((maxc + 1) / 2) * static_cast<size_t>(nact));
        for (int j = 0; j < nact; ++j) {
          const int r = r0 + j, c = ptr[r + 1] - ptr[r];
          out.G[static_cast<size_t>(at + j)] = lo16(ptr[r]) | (lo16(c) << 16);
          for (int e = 0; e < c; ++e) out.G[static_cast<size_t>(ea + (e >> 1) * nact + j)] |= lo16(idx[ptr[r] + e]) << ((e & 1) ? 16 : 0);
        }
        E.line("      wgrt::spmv<SPLIT, %d, %d, %d, %d, %d>(lane, G, val, base, v, vhi, split, y, scale_kind, scalar, sg, jr);\n", at, ea, nact, r0, maxc);
      }
      E.line("    WG_END\n");
      E.line("  }\n");
    }
    E.line("}\n\n");
  }
  E.line("}  // namespace wgen\n}  // namespace dnlp\n");
  if (stage) out.G.resize(out.G.size() + 64, 0u);      // (WG_STAGE_NEXT reads whole 64-word rows: up to 63 words past a table)
  return out;
}

}  // namespace dnlp
