// Template-specialised batch solver, part 6: the PHASES of the static-pattern LDL^T, of the KKT residual and of the constant
// CSR maps as straight-line code.
//
// wave_ipm.h's ldl_factor_impl / ldl_solve / kkt_residual / spmv walk the plan's index tables: per level a handful of uniform
// bound loads, per entry an index load, a dependent operand load and loop control — 9.7 k cycles for one solve of an
// order-102 factor.  In a kernel compiled per template (wave_codegen.h) the structure is known when the text is written:
// wave_gen.h lays the work of every phase out as ONE TASK PER LANE (descriptor words G[base + lane], entry words
// G[base + e * lanes_active + lane], 32-bit; staged in LDS next to the plan, or read from global memory by the kernels whose
// shares fill LDS) and emits, per phase slot, one call of a helper below with every base, count and block kind as a template
// argument.  After inlining that is straight-line code: loads with immediate offsets for the descriptors, unrolled entry
// loops, no level tables, no scalar loop control.  Every operand is LOADED unconditionally and SELECTED afterwards: a load
// under a condition is a branch, and a branch per padded entry a memory round trip.
//
// The ARITHMETIC is that of wave_ipm.h / sparse_ldl.h, entry by entry: every destination (a forward target, a block's
// backward sum, an update group, an output of a product) is summed in storage order by one lane — what the host lane of the
// interpreted text does — so the generated phases reproduce its bits on the CPU (tests/test_wave_gen_cpu.py builds this text
// for the host, its lanes played one after the other, against the interpreted host lane).  A padded entry multiplies two
// selected zeros: adding +0.0 to a sum that started at +0.0 never changes it.  The device runs the same sums, with ONE
// exception: the wide forms (fwdw / fwdw2 / bwdw / wdot: the rows of one long target across the lanes) add their products
// through the wavefront's fixed DPP tree instead of a chain of CNT dependent additions — the same bits on every run, another
// order than the host's, as every P::sum of the interpreted text.
//
// Free of standard-library includes (the text travels into hiprtc as wave_gen_rt_src.inc).
#pragma once

// a phase: every lane of the wavefront runs the body once; on the host one thread plays the 64 lanes one after the other
// (phases have no cross-lane dependence inside: that is what makes them phases)
// A NARROW phase (at most 64 tasks) and the wide forms below are run by the first wavefront alone (WG_NBEGIN / WG_WBEGIN),
// with a wavefront-level barrier behind them; the other wavefronts of a workgroup-per-instance kernel run ahead and wait at
// WG_NCLOSE, which the generator puts in front of the next thing that needs them.  With one wavefront per instance the first
// wavefront is the only one and the barriers coincide.
#if DNLP_DEVICE_PASS
#define WG_BEGIN { const int lane = P::lane();
#define WG_END } P::sync();
#define WG_NBEGIN if (P::lane() < 64) { const int lane = P::lane();
#define WG_NEND wave_sync(); }
#define WG_WBEGIN if (P::lane() < 64) {
#define WG_WEND wave_sync(); }
#define WG_NCLOSE P::sync();
#define WG_WAVE(w) if ((P::lane() >> 6) == (w))
#define WG_INLINE __attribute__((always_inline)) __device__ inline
#else
#define WG_BEGIN for (int lane = 0; lane < WG_LANES; ++lane) {      // (WG_LANES: the generated text says how many lanes share a phase)
#define WG_END }
#define WG_NBEGIN for (int lane = 0; lane < 64; ++lane) {
#define WG_NEND }
#define WG_WBEGIN {
#define WG_WEND }
#define WG_NCLOSE
#define WG_WAVE(w) if (true)
#define WG_INLINE inline
#endif

namespace dnlp {
namespace wgrt {

typedef unsigned int u32;

// STAGING of a narrow phase's table (wave_gen.h: work tables in global memory): WG_STAGE_NEXT issues the loads of the NEXT
// narrow phase's table (whole 64-word rows from G[g0n] on) into registers, WG_STAGE_PUT stores them to the policy's LDS
// buffer P::stage() at the phase's end; a phase whose table was staged (sg0_ >= 0: where the table starts in G) reads the
// buffer (WG_SG) at offsets from its start (WG_SO).  On the host the lanes are played one after the other from G itself.
// a vector of the instance by its LITERAL place, typed by where it lives when the policy's vectors are generic pointers (the
// workgroup kernel: LDS or the slab — wave_ipm.h ldl_solve): the factor's values are read in every solve phase
#if DNLP_DEVICE_PASS
template <class P, int OFF, class WS>
WG_INLINE auto tvec(WS* S) { if constexpr (P::lds_generic) return P::template vec_typed<OFF>(); else return P::vec(S, OFF); }
#else
template <class P, int OFF, class WS>
WG_INLINE auto tvec(WS* S) { return P::vec(S, OFF); }
#endif
#if DNLP_DEVICE_PASS
template <bool ST, class P, class GP>
WG_INLINE auto gsel(GP G) { if constexpr (ST) return P::stage(); else return G; }
#define WG_SO(a) ((sg0_) >= 0 ? (a) - (sg0_) : (a))
#define WG_SG wgrt::gsel<(sg0_ >= 0), P>(G)
#define WG_STAGE_NEXT_(g0n, nw) constexpr int sn_ = ((nw) + 63) / 64; unsigned st_[sn_ > 0 ? sn_ : 1]; \
  _Pragma("unroll") for (int i_ = 0; i_ < sn_; ++i_) st_[i_] = G[(g0n) + 64 * i_ + static_cast<int>(P::lane())];
#define WG_STAGE_NEXT(...) WG_STAGE_NEXT_(__VA_ARGS__)
#define WG_WW(lo) (P::wwin() - (lo))
#define WG_SW (P::swin())
#define WG_STAGE_PUT _Pragma("unroll") for (int i_ = 0; i_ < sn_; ++i_) P::stage()[64 * i_ + static_cast<int>(P::lane())] = st_[i_];
#else
#define WG_SO(a) (a)
#define WG_SG G
#define WG_WW(lo) w
#define WG_SW scr
#define WG_STAGE_NEXT(...)
#define WG_STAGE_PUT
#endif

// ---- forward substitution: target node `node` collects its rows (wave_ipm.h ldl_solve, first half) -------------------
// descriptor (1 word): node | rows << 16.  entry e (2 words, [2e][lane], [2e + 1][lane]): a | u0 << 16, u1 | kind << 16
// (kind 1: a row of a 1x1 block, value index a; 2: of a 2x2 block, values a, a + 1, sources u0, u1).
// KINDS: 1 / 2 = every real entry of the phase is of that kind, 3 = mixed.  RAGGED: lanes differ in their number of rows.
template <bool TWO, int D0, int E0, int NACT, int MAXC, int KINDS, bool RAGGED, class GP, class AP, class XP, class YP>
WG_INLINE void fwd(int lane, GP G, const AP vals, XP x, YP y) {
  if (lane < NACT) {
    const u32 d = G[D0 + lane];
    const int node = static_cast<int>(d & 0xffffu), cnt = static_cast<int>(d >> 16);
    double acc = 0.0, acc2 = 0.0;
#pragma unroll
    for (int e = 0; e < MAXC; ++e) {
      const u32 w0 = G[E0 + (2 * e) * NACT + lane], w1 = G[E0 + (2 * e + 1) * NACT + lane];
      const int a = static_cast<int>(w0 & 0xffffu), u0 = static_cast<int>(w0 >> 16), u1 = static_cast<int>(w1 & 0xffffu);
      const bool on = !RAGGED || e < cnt;
      const bool two = KINDS == 2 || (KINDS == 3 && (w1 >> 16) == 2u);
      // (every operand is LOADED unconditionally — a padded entry's indices are zeros, valid places — and SELECTED afterwards:
      //  a load under a condition is a branch, and a branch per entry is an LDS round trip per entry)
      if (KINDS == 1) {
        const double lr = vals[a], xr = x[u0];
        const double l = on ? lr : 0.0, xv = on ? xr : 0.0;
        acc += l * xv;
        if (TWO) { const double yr = y[u0]; const double yv = on ? yr : 0.0; acc2 += l * yv; }
      } else if (KINDS == 2) {
        const double lr0 = vals[a], lr1 = vals[a + 1], xr0 = x[u0], xr1 = x[u1];
        const double l0 = on ? lr0 : 0.0, l1 = on ? lr1 : 0.0, x0 = on ? xr0 : 0.0, x1 = on ? xr1 : 0.0;
        acc += l0 * x0 + l1 * x1;
        if (TWO) { const double yr0 = y[u0], yr1 = y[u1]; const double y0 = on ? yr0 : 0.0, y1 = on ? yr1 : 0.0; acc2 += l0 * y0 + l1 * y1; }
      } else {
        // mixed: the loads side by side, the two forms of the sum under the lane's own kind
        const double lr0 = vals[a], lr1 = vals[a + 1], xr0 = x[u0], xr1 = x[u1];
        const double l0 = on ? lr0 : 0.0, l1 = (on && two) ? lr1 : 0.0, x0 = on ? xr0 : 0.0, x1 = (on && two) ? xr1 : 0.0;
        double y0 = 0.0, y1 = 0.0;
        if (TWO) { const double yr0 = y[u0], yr1 = y[u1]; y0 = on ? yr0 : 0.0; y1 = (on && two) ? yr1 : 0.0; }
        if (two) { acc += l0 * x0 + l1 * x1; if (TWO) acc2 += l0 * y0 + l1 * y1; }
        else { acc += l0 * x0; if (TWO) acc2 += l0 * y0; }
      }
    }
    x[node] -= acc;
    if (TWO) y[node] -= acc2;
  }
}

// ---- forward substitution, WIDE form: ONE target whose many rows lie across the lanes ------------------------------------------
// (the last levels of a plan with a dense row — localization's two position variables meet all twenty range rows: one lane
//  walking twenty entries issues 160 LDS loads by itself.)  Lane e forms the product of row e; the products are then added
//  — on the host in row order (the lanes are played in that order: the serial sum of the interpreted text), on the device by
//  the wavefront's fixed reduction tree.  entry e (2 words): as in fwd.  The caller chains the slots of a
// target with more than 64 rows through acc / acc2 and subtracts once (fwdw_fin).
// (W: the wavefront of the workgroup that runs this target — 0 unless the generator deals a level's targets out: WG_WAVE(W))
template <class P, bool TWO, int E0, int CNT, int KINDS, int W = 0, class GP, class AP, class XP, class YP>
WG_INLINE void fwdw(GP G, const AP vals, const XP x, const YP y, double& acc, double& acc2) {
#if DNLP_DEVICE_PASS
  const int lane = P::lane() - 64 * W;
  double p = 0.0, p2 = 0.0;
  if (lane < CNT) {
#else
  for (int lane = 0; lane < CNT; ++lane) {
    double p = 0.0, p2 = 0.0;
#endif
    const u32 w0 = G[E0 + lane], w1 = G[E0 + CNT + lane];
    const int a = static_cast<int>(w0 & 0xffffu), u0 = static_cast<int>(w0 >> 16), u1 = static_cast<int>(w1 & 0xffffu);
    const bool two = KINDS == 2 || (KINDS == 3 && (w1 >> 16) == 2u);
    if (two) {
      const double l0 = vals[a], l1 = vals[a + 1];
      p = l0 * x[u0] + l1 * x[u1];
      if (TWO) p2 = l0 * y[u0] + l1 * y[u1];
    } else {
      const double l = vals[a];
      p = l * x[u0];
      if (TWO) p2 = l * y[u0];
    }
#if DNLP_DEVICE_PASS
  }
  // (the wavefront's fixed DPP tree — wave_ops.h — instead of CNT dependent additions through v_readlane: a 47-row target
  //  of power flow's dense end spent 1 000 cycles in that chain, twice per right-hand side and level.  The same bits on
  //  every run; the order differs from the host lane's serial sum, as every P::sum of the interpreted text does)
  acc += wave_all_sum(p);
  if (TWO) acc2 += wave_all_sum(p2);
#else
    acc += p;
    if (TWO) acc2 += p2;
  }
#endif
}
template <class P, bool TWO, int NODE, int W = 0, class XP, class YP>
WG_INLINE void fwdw_fin(XP x, YP y, double acc, double acc2) {
#if DNLP_DEVICE_PASS
  if (P::lane() == 64 * W) {
#else
  {
#endif
    x[NODE] -= acc;
    if (TWO) y[NODE] -= acc2;
  }
}

// ---- ... TWO targets of at most 32 rows in one phase: target A on lanes 0 .. 31, target B on lanes 32 .. 63 (table: 64 + 64 words,
// zero where a half has fewer rows).  Each target's products are added in row order on the host, by the wavefront's reduction
// tree on the device (the other half contributes zeros).
template <class P, bool TWO, int E0, int CNT_A, int CNT_B, int KINDS, int NODE_A, int NODE_B, class GP, class AP, class XP, class YP>
WG_INLINE void fwdw2(GP G, const AP vals, XP x, YP y) {
  double accA = 0.0, accB = 0.0, acc2A = 0.0, acc2B = 0.0;
#if DNLP_DEVICE_PASS
  const int lane = P::lane();
  double p = 0.0, p2 = 0.0;
  if (lane < CNT_A || (lane >= 32 && lane < 32 + CNT_B)) {
#else
  for (int lane = 0; lane < 64; ++lane) {
    if (!(lane < CNT_A || (lane >= 32 && lane < 32 + CNT_B))) continue;
    double p = 0.0, p2 = 0.0;
#endif
    const u32 w0 = G[E0 + lane], w1 = G[E0 + 64 + lane];
    const int a = static_cast<int>(w0 & 0xffffu), u0 = static_cast<int>(w0 >> 16), u1 = static_cast<int>(w1 & 0xffffu);
    const bool two = KINDS == 2 || (KINDS == 3 && (w1 >> 16) == 2u);
    if (two) {
      const double l0 = vals[a], l1 = vals[a + 1];
      p = l0 * x[u0] + l1 * x[u1];
      if (TWO) p2 = l0 * y[u0] + l1 * y[u1];
    } else {
      const double l = vals[a];
      p = l * x[u0];
      if (TWO) p2 = l * y[u0];
    }
#if DNLP_DEVICE_PASS
  }
  {
    double r[2] = {lane < 32 ? p : 0.0, lane < 32 ? 0.0 : p};
    wave_all_sum_n<2>(r);
    accA += r[0]; accB += r[1];
    if (TWO) {
      double r2[2] = {lane < 32 ? p2 : 0.0, lane < 32 ? 0.0 : p2};
      wave_all_sum_n<2>(r2);
      acc2A += r2[0]; acc2B += r2[1];
    }
  }
  if (lane == 0) {
#else
    if (lane < 32) { accA += p; if (TWO) acc2A += p2; }
    else { accB += p; if (TWO) acc2B += p2; }
  }
  {
#endif
    x[NODE_A] -= accA; x[NODE_B] -= accB;
    if (TWO) { y[NODE_A] -= acc2A; y[NODE_B] -= acc2B; }
  }
}

// ---- backward substitution, WIDE form: ONE block whose many struct rows lie across the lanes -------------------------------
// entry i (1 word): source node u.  ONE: a 1x1 block (values LOF + i), else a 2x2 block (LOF + 2 i, LOF + 2 i + 1).
template <class P, bool TWO, bool ONE, int E0, int CNT, int LOF, int I0, class GP, class AP, class XP, class YP>
WG_INLINE void bwdw(GP G, const AP vals, const XP x, const YP y, double& a0, double& a1, double& c0, double& c1) {
#if DNLP_DEVICE_PASS
  const int lane = P::lane();
  double p0 = 0.0, p1 = 0.0, q0 = 0.0, q1 = 0.0;
  if (lane < CNT) {
#else
  for (int lane = 0; lane < CNT; ++lane) {
    double p0 = 0.0, p1 = 0.0, q0 = 0.0, q1 = 0.0;
#endif
    const int u = static_cast<int>(G[E0 + lane]);
    const int i = I0 + lane;
    const double xi = x[u];
    double yi = 0.0;
    if (TWO) yi = y[u];
    if (ONE) {
      const double l = vals[LOF + i];
      p0 = l * xi;
      if (TWO) q0 = l * yi;
    } else {
      const double l0 = vals[LOF + 2 * i], l1 = vals[LOF + 2 * i + 1];
      p0 = l0 * xi; p1 = l1 * xi;
      if (TWO) { q0 = l0 * yi; q1 = l1 * yi; }
    }
#if DNLP_DEVICE_PASS
  }
  a0 += wave_all_sum(p0);
  if (!ONE) a1 += wave_all_sum(p1);
  if (TWO) { c0 += wave_all_sum(q0); if (!ONE) c1 += wave_all_sum(q1); }
#else
    a0 += p0;
    if (!ONE) a1 += p1;
    if (TWO) { c0 += q0; if (!ONE) c1 += q1; }
  }
#endif
}
template <class P, bool TWO, bool ONE, int U0, int U1, class XP, class YP>
WG_INLINE void bwdw_fin(XP x, YP y, double a0, double a1, double c0, double c1) {
#if DNLP_DEVICE_PASS
  if (P::lane() == 0) {
#else
  {
#endif
    x[U0] -= a0;
    if (TWO) y[U0] -= c0;
    if (!ONE) {
      x[U1] -= a1;
      if (TWO) y[U1] -= c1;
    }
  }
}

// ---- D^-1 on every block (wave_ipm.h dsolve) -------------------------------------------------------------------------
// descriptor (2 words): u0 | u1 << 16 (u1 = 0xffff: a 1x1 block), doff
template <bool TWO, int D0, int NACT, int KINDS, class GP, class AP, class XP, class YP>
WG_INLINE void dsol(int lane, GP G, const AP vals, XP x, YP y) {
  if (lane < NACT) {
    const u32 w0 = G[D0 + lane], w1 = G[D0 + NACT + lane];
    const int u0 = static_cast<int>(w0 & 0xffffu), u1r = static_cast<int>(w0 >> 16), dof = static_cast<int>(w1);
    const bool one = KINDS == 1 || (KINDS == 3 && u1r == 0xffff);
    if (one) {
      const double d = vals[dof];
      x[u0] /= d;
      if (TWO) y[u0] /= d;
    } else {
      const int u1 = u1r;
      const double a = vals[dof], c = vals[dof + 1], e = vals[dof + 2];
      double det = a * e - c * c;
      if (fabs(det) < 1e-300) det = -1e-20;
      const double x0 = x[u0], x1 = x[u1];
      x[u0] = (e * x0 - c * x1) / det;
      x[u1] = (a * x1 - c * x0) / det;
      if (TWO) {
        const double y0 = y[u0], y1 = y[u1];
        y[u0] = (e * y0 - c * y1) / det;
        y[u1] = (a * y1 - c * y0) / det;
      }
    }
  }
}

// ---- backward substitution: a block subtracts its struct rows' share (wave_ipm.h ldl_solve, second half) ---------------
// descriptor (2 words): u0 | u1 << 16 (0xffff: 1x1), loff | rows << 16.  entries: source node u, two per word
// ([e / 2][lane], low half first).
template <bool TWO, int D0, int E0, int NACT, int MAXC, int KINDS, bool RAGGED, class GP, class AP, class XP, class YP>
WG_INLINE void bwd(int lane, GP G, const AP vals, XP x, YP y) {
  if (lane < NACT) {
    const u32 w0 = G[D0 + lane], w1 = G[D0 + NACT + lane];
    const int u0 = static_cast<int>(w0 & 0xffffu), u1r = static_cast<int>(w0 >> 16);
    const int lof = static_cast<int>(w1 & 0xffffu), sn = static_cast<int>(w1 >> 16);
    const bool one = KINDS == 1 || (KINDS == 3 && u1r == 0xffff);
    double a0 = 0.0, a1 = 0.0, c0 = 0.0, c1 = 0.0;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const u32 w = G[E0 + (i >> 1) * NACT + lane];
      const int u = static_cast<int>((i & 1) ? (w >> 16) : (w & 0xffffu));
      const bool on = !RAGGED || i < sn;
      // (loads unconditional, selected afterwards: see fwd.  A padded row reads within the level's values: i < MAXC)
      if (KINDS == 1) {
        const double lr = vals[on ? lof + i : lof], xr = x[u];
        const double l = on ? lr : 0.0, xi = on ? xr : 0.0;
        a0 += l * xi;
        if (TWO) { const double yr = y[u]; const double yi = on ? yr : 0.0; c0 += l * yi; }
      } else if (KINDS == 2) {
        const int p0 = on ? lof + 2 * i : lof;
        const double lr0 = vals[p0], lr1 = vals[p0 + 1], xr = x[u];
        const double l0 = on ? lr0 : 0.0, l1 = on ? lr1 : 0.0, xi = on ? xr : 0.0;
        a0 += l0 * xi; a1 += l1 * xi;
        if (TWO) { const double yr = y[u]; const double yi = on ? yr : 0.0; c0 += l0 * yi; c1 += l1 * yi; }
      } else {
        const int p0 = on ? (one ? lof + i : lof + 2 * i) : lof;
        const double lr0 = vals[p0], lr1 = vals[p0 + 1], xr = x[u];
        const double l0 = on ? lr0 : 0.0, l1 = (on && !one) ? lr1 : 0.0, xi = on ? xr : 0.0;
        a0 += l0 * xi;
        if (!one) a1 += l1 * xi;
        if (TWO) { const double yr = y[u]; const double yi = on ? yr : 0.0; c0 += l0 * yi; if (!one) c1 += l1 * yi; }
      }
    }
    x[u0] -= a0;
    if (TWO) y[u0] -= c0;
    if (!one) {
      x[u1r] -= a1;
      if (TWO) y[u1r] -= c1;
    }
  }
}

// ---- factorisation: pivots of a level (sparse_ldl.h sp_pivot) -----------------------------------------------------------
// descriptor (1 word): doff | k << 16 | kind << 30 (kind 1: 1x1, 2: 2x2; the block's inverse goes to dinv[3 k ..])
template <int D0, int NACT, int KINDS, class GP, class VP>
WG_INLINE void piv(int lane, GP G, VP vals, VP dinv, double& nneg, double& nzero, double& bad) {
  if (lane < NACT) {
    const u32 w0 = G[D0 + lane];
    const int dof = static_cast<int>(w0 & 0xffffu), k3 = 3 * static_cast<int>((w0 >> 16) & 0x3fffu);
    const bool one = KINDS == 1 || (KINDS == 3 && (w0 >> 30) == 1u);
    if (one) {
      double d = vals[dof];
      if (!(d == d)) bad += 1.0;
      if (fabs(d) < 1e-300) { nzero += 1.0; d = 1e-20; vals[dof] = d; }
      if (d < 0.0) nneg += 1.0;
      dinv[k3] = 1.0 / d;
    } else {
      const double a = vals[dof], c = vals[dof + 1], e = vals[dof + 2];
      double det = a * e - c * c;
      if (!(det == det)) bad += 1.0;
      if (fabs(det) < 1e-300) { nzero += 1.0; det = -1e-20; }
      if (det < 0.0) nneg += 1.0;
      else if (a < 0.0 || (a == 0.0 && e < 0.0)) nneg += 2.0;
      dinv[k3] = e / det; dinv[k3 + 1] = -c / det; dinv[k3 + 2] = a / det;
    }
  }
}

// ---- factorisation: struct rows of a level scaled by their block's inverse pivot (sp_scale) ---------------------------------
// descriptor (1 word): a | k << 16 | kind << 30
template <int D0, int NACT, int KINDS, class GP, class VP>
WG_INLINE void scl(int lane, GP G, VP vals, VP w, const VP dinv) {
  if (lane < NACT) {
    const u32 w0 = G[D0 + lane];
    const int a = static_cast<int>(w0 & 0xffffu), k3 = 3 * static_cast<int>((w0 >> 16) & 0x3fffu);
    const bool one = KINDS == 1 || (KINDS == 3 && (w0 >> 30) == 1u);
    if (one) {
      const double l1 = vals[a];
      w[a] = l1;
      vals[a] = l1 * dinv[k3];
    } else {
      const double l1 = vals[a], l2 = vals[a + 1];
      const double d0 = dinv[k3], d1 = dinv[k3 + 1], d2 = dinv[k3 + 2];
      w[a] = l1; w[a + 1] = l2;
      vals[a] = d0 * l1 + d1 * l2;
      vals[a + 1] = d1 * l1 + d2 * l2;
    }
  }
}

// ---- ... the same, with the inverse pivot RECOMPUTED from the block's D entries by every row's lane (the expressions of
// sp_pivot, the tiny-pivot substitutions included): the rows no longer wait for the pivots' phase — pivots (which still count
// the inertia and fix a tiny 1x1 pivot in place) and row scaling are ONE phase, a dependent round trip less per level.
// descriptor (1 word): a | doff << 15 | kind << 30
template <int D0, int NACT, int KINDS, class GP, class VP, class WP>
WG_INLINE void scl2(int lane, GP G, VP vals, WP w) {
  if (lane < NACT) {
    const u32 w0 = G[D0 + lane];
    const int a = static_cast<int>(w0 & 0x7fffu), dof = static_cast<int>((w0 >> 15) & 0x7fffu);
    const bool one = KINDS == 1 || (KINDS == 3 && (w0 >> 30) == 1u);
    if (one) {
      double d = vals[dof];
      if (fabs(d) < 1e-300) d = 1e-20;
      const double l1 = vals[a];
      w[a] = l1;
      vals[a] = l1 * (1.0 / d);
    } else {
      const double pa = vals[dof], pc = vals[dof + 1], pe = vals[dof + 2];
      double det = pa * pe - pc * pc;
      if (fabs(det) < 1e-300) det = -1e-20;
      const double d0 = pe / det, d1 = -pc / det, d2 = pa / det;
      const double l1 = vals[a], l2 = vals[a + 1];
      w[a] = l1; w[a + 1] = l2;
      vals[a] = d0 * l1 + d1 * l2;
      vals[a + 1] = d1 * l1 + d2 * l2;
    }
  }
}
// the pivots' part of that phase: inertia counts and the in-place fix of a tiny 1x1 pivot (descriptor as piv; no inverse stored)
template <int D0, int NACT, int KINDS, class GP, class VP>
WG_INLINE void piv2(int lane, GP G, VP vals, double& nneg, double& nzero, double& bad) {
  if (lane < NACT) {
    const u32 w0 = G[D0 + lane];
    const int dof = static_cast<int>(w0 & 0xffffu);
    const bool one = KINDS == 1 || (KINDS == 3 && (w0 >> 30) == 1u);
    if (one) {
      double d = vals[dof];
      if (!(d == d)) bad += 1.0;
      if (fabs(d) < 1e-300) { nzero += 1.0; d = 1e-20; vals[dof] = d; }
      if (d < 0.0) nneg += 1.0;
    } else {
      const double a = vals[dof], c = vals[dof + 1], e = vals[dof + 2];
      double det = a * e - c * c;
      if (!(det == det)) bad += 1.0;
      if (fabs(det) < 1e-300) { nzero += 1.0; det = -1e-20; }
      if (det < 0.0) nneg += 1.0;
      else if (a < 0.0 || (a == 0.0 && e < 0.0)) nneg += 2.0;
    }
  }
}

// ---- factorisation: the products of a level's update triples, side by side into the scratch array (sp_update) -------------
// one word per triple: au | av << 16 | (2x2 pivot block) << 31; the triple's place in the scratch array is Q0 + lane
template <int D0, int NACT, int Q0, int KINDS, class GP, class VP, class WP, class SP>
WG_INLINE void upd(int lane, GP G, const VP vals, const WP w, SP scr) {
  if (lane < NACT) {
    const u32 t = G[D0 + lane];
    const int au = static_cast<int>(t & 0xffffu), av = static_cast<int>((t >> 16) & 0x7fffu);
    const bool two = KINDS == 2 || (KINDS == 3 && (t >> 31) != 0u);
    double p;
    if (two) p = w[au] * vals[av] + w[au + 1] * vals[av + 1];
    else p = w[au] * vals[av];
    scr[Q0 + lane] = p;
  }
}

// ---- factorisation: every destination subtracts its run of products, added in storage order (run_sum) --------------------
// descriptor (2 words): dst | count << 16, start of the run in the scratch array
template <int D0, int NACT, int MAXC, bool RAGGED, class GP, class VP, class SP>
WG_INLINE void gsum(int lane, GP G, VP vals, const SP scr) {
  if (lane < NACT) {
    const u32 w0 = G[D0 + lane], w1 = G[D0 + NACT + lane];
    const int dst = static_cast<int>(w0 & 0xffffu), cnt = static_cast<int>(w0 >> 16), q0 = static_cast<int>(w1);
    double acc = 0.0;
#pragma unroll
    for (int e = 0; e < MAXC; ++e) {
      const bool on = !RAGGED || e < cnt;
      const double t = scr[on ? q0 + e : q0];
      acc += on ? t : 0.0;
    }
    vals[dst] -= acc;
  }
}

// =====================================================================================================================
// products by output (the tape's indices hs / jc / jr: wave_ipm.h coo, kkt_residual) and the constant CSR maps (spmv)
// =====================================================================================================================
// NaN conventions of the reductions (wave_ipm.h mxin)
WG_INLINE double mxin(double acc, double v) { return fmax(acc, v != v ? __builtin_inf() : v); }

// ---- ONE long output by all lanes: sum over its entries of a[ent] v[src] (+ the same with v2), entries across the lanes,
// added in entry order on the host (the interpreted text's coo_heavy on the host lane), by the fixed reduction tree on the device.  entry (1 word): ent | src << 16.
// W: the wavefront of the workgroup that runs this output (the generator deals the long outputs out to the wavefronts: power
// flow has 46 of them per residual, and one wavefront took them one after the other; WG_WAVE(W) keeps the others out)
template <class P, bool TWO, int W, int E0, int CNT, class GP, class CP>
WG_INLINE void wdot(GP G, CP a, CP v, CP v2, double& acc, double& acc2) {
#if DNLP_DEVICE_PASS
  const int lane = P::lane() - 64 * W;
  double p = 0.0, p2 = 0.0;
  if (lane < CNT) {
#else
  for (int lane = 0; lane < CNT; ++lane) {
    double p = 0.0, p2 = 0.0;
#endif
    const u32 w = G[E0 + lane];
    const double c = a[w & 0xffffu];
    p = c * v[w >> 16];
    if (TWO) p2 = c * v2[w >> 16];
#if DNLP_DEVICE_PASS
  }
  acc += wave_all_sum(p);
  if (TWO) acc2 += wave_all_sum(p2);
#else
    acc += p;
    if (TWO) acc2 += p2;
  }
#endif
}
template <class P, bool TWO, int W, int OUT, class VP>
WG_INLINE void wdot_fin(VP pre, VP pre2, double acc, double acc2) {
#if DNLP_DEVICE_PASS
  if (P::lane() == 64 * W) {
#else
  {
#endif
    pre[OUT] = acc;
    if (TWO) pre2[OUT] = acc2;
  }
}

// ---- KKT residual, variable part: out[k] = rhs[k] - (sym(H) v + (Sx + dw) v + J^T v_y)[k], k = K0 + lane (kkt_residual) --------
// descriptor (1 word): nh | nj << 8 | (H output long) << 16 | (J^T output long) << 17.  entries (1 word each, [e][lane]):
// first MAXH of H (ent | src << 16), then MAXJ of J^T (ent | (N + src) << 16).  A long output was summed beforehand into pre*.
template <bool TWO, int D0, int E0, int NACT, int K0, int MAXH, int MAXJ, class GP, class CP, class VP>
WG_INLINE void kres_var(int lane, GP G, CP Hs, CP jv, CP sx, CP fm, double dw, CP v, CP rhsv, VP out,
                        CP preH, CP preJt, CP v2, CP rhsv2, VP out2, CP preH2, CP preJt2,
                        double& m0, double& m1, double& n0, double& n1) {
  if (lane < NACT) {
    const int k = K0 + lane;
    const u32 d = G[D0 + lane];
    const int nh = static_cast<int>(d & 0xffu), nj = static_cast<int>((d >> 8) & 0xffu);
    double hv = 0.0, jt = 0.0, hv2 = 0.0, jt2 = 0.0;
#pragma unroll
    for (int e = 0; e < MAXH; ++e) {
      const u32 w = G[E0 + e * NACT + lane];
      const bool on = e < nh;
      const double cr = Hs[w & 0xffffu], vr = v[w >> 16];      // (loads unconditional, selected afterwards: see fwd)
      const double c = on ? cr : 0.0;
      hv += c * (on ? vr : 0.0);
      if (TWO) { const double vr2 = v2[w >> 16]; hv2 += c * (on ? vr2 : 0.0); }
    }
#pragma unroll
    for (int e = 0; e < MAXJ; ++e) {
      const u32 w = G[E0 + (MAXH + e) * NACT + lane];
      const bool on = e < nj;
      const double cr = jv[w & 0xffffu], vr = v[w >> 16];
      const double c = on ? cr : 0.0;
      jt += c * (on ? vr : 0.0);
      if (TWO) { const double vr2 = v2[w >> 16]; jt2 += c * (on ? vr2 : 0.0); }
    }
    {
      const double ph = preH[k], pj = preJt[k];
      hv = (d & 0x10000u) ? ph : hv;
      jt = (d & 0x20000u) ? pj : jt;
      if (TWO) { const double ph2 = preH2[k], pj2 = preJt2[k]; hv2 = (d & 0x10000u) ? ph2 : hv2; jt2 = (d & 0x20000u) ? pj2 : jt2; }
    }
    const bool fx = fm[k] != 0.0;
    const double sd = sx[k] + dw, vk = v[k];
    const double kv = fx ? vk : hv + sd * vk + jt;
    const double r = rhsv[k] - kv;
    out[k] = r;
    m0 = mxin(m0, fabs(r)); m1 = mxin(m1, fabs(vk));
    if (TWO) {
      const double vk2 = v2[k];
      const double kv2 = fx ? vk2 : hv2 + sd * vk2 + jt2;
      const double r2 = rhsv2[k] - kv2;
      out2[k] = r2;
      n0 = mxin(n0, fabs(r2)); n1 = mxin(n1, fabs(vk2));
    }
  }
}
// ---- J^T y alone (wave_ipm.h jac_tmult) out of the variable part's tables: the J^T entries carry N + src, so the caller
// passes y - N.  A long output is written by the wdot pass around this call.
template <int D0, int E0, int NACT, int K0, int MAXH, int MAXJ, class GP, class CP, class VP>
WG_INLINE void cojt(int lane, GP G, CP jv, CP vsh, VP out) {
  if (lane < NACT) {
    const u32 d = G[D0 + lane];
    const int nj = static_cast<int>((d >> 8) & 0xffu);
    double jt = 0.0;
#pragma unroll
    for (int e = 0; e < MAXJ; ++e) {
      const u32 w = G[E0 + (MAXH + e) * NACT + lane];
      const bool on = e < nj;
      const double cr = jv[w & 0xffffu], vr = vsh[w >> 16];
      jt += (on ? cr : 0.0) * (on ? vr : 0.0);
    }
    if (!(d & 0x20000u)) out[K0 + lane] = jt;
  }
}

// ---- ... row part: out[N + i] = rhs[N + i] - (J v_x - D v_y)[i], i = I0 + lane.  descriptor: nj | (long) << 16 ------------------
template <bool TWO, int D0, int E0, int NACT, int I0, int NV, int MAXJ, class GP, class CP, class VP>
WG_INLINE void kres_row(int lane, GP G, CP jv, CP dd, CP v, CP rhsv, VP out, CP preJ,
                        CP v2, CP rhsv2, VP out2, CP preJ2, double& m0, double& m1, double& n0, double& n1) {
  if (lane < NACT) {
    const int i = I0 + lane, k = NV + i;
    const u32 d = G[D0 + lane];
    const int nj = static_cast<int>(d & 0xffu);
    double jx = 0.0, jx2 = 0.0;
#pragma unroll
    for (int e = 0; e < MAXJ; ++e) {
      const u32 w = G[E0 + e * NACT + lane];
      const bool on = e < nj;
      const double cr = jv[w & 0xffffu], vr = v[w >> 16];
      const double c = on ? cr : 0.0;
      jx += c * (on ? vr : 0.0);
      if (TWO) { const double vr2 = v2[w >> 16]; jx2 += c * (on ? vr2 : 0.0); }
    }
    {
      const double pj = preJ[i];
      jx = (d & 0x10000u) ? pj : jx;
      if (TWO) { const double pj2 = preJ2[i]; jx2 = (d & 0x10000u) ? pj2 : jx2; }
    }
    const double di = dd[i], vk = v[k];
    const double kv = jx - di * vk;
    const double r = rhsv[k] - kv;
    out[k] = r;
    m0 = mxin(m0, fabs(r)); m1 = mxin(m1, fabs(vk));
    if (TWO) {
      const double vk2 = v2[k];
      const double kv2 = jx2 - di * vk2;
      const double r2 = rhsv2[k] - kv2;
      out2[k] = r2;
      n0 = mxin(n0, fabs(r2)); n1 = mxin(n1, fabs(vk2));
    }
  }
}

// ---- a constant CSR map applied to a vector: y[r] = (base[r] + sum_k val[k] v[idx[k]]) [x scale], r = R0 + lane (spmv) ------
// descriptor (1 word): first value index | entries << 16.  entries: column, two per word ([e / 2][lane], low half first).
// SPLIT: column c >= split reads vhi[c] (the vector [x | z] in two places).  val / base: the instance's data row (global memory).
template <bool SPLIT, int D0, int E0, int NACT, int R0, int MAXC, class GP, class CP, class VP, class RP, class IP>
WG_INLINE void spmv(int lane, GP G, RP val, RP base, CP v, CP vhi, int split, VP y, int scale_kind, double scalar, CP sg, IP jr) {
  if (lane < NACT) {
    const int r = R0 + lane;
    const u32 w0 = G[D0 + lane];
    const int k0 = static_cast<int>(w0 & 0xffffu), cnt = static_cast<int>(w0 >> 16);
    double sacc = base ? base[r] : 0.0;
#pragma unroll
    for (int e = 0; e < MAXC; ++e) {
      const u32 w = G[E0 + (e >> 1) * NACT + lane];
      const int c = static_cast<int>((e & 1) ? (w >> 16) : (w & 0xffffu));
      const double a = val[k0 + e];
      const double vv = ((SPLIT && c >= split) ? vhi : v)[c];
      // (no padding by zeros here: the sum starts at base[r], which may be -0.0)
      sacc = e < cnt ? sacc + a * vv : sacc;
    }
    if (scale_kind == 1) sacc *= scalar;
    else if (scale_kind == 2) sacc *= sg[r];
    else if (scale_kind == 3) sacc *= sg[jr[r]];
    y[r] = sacc;
  }
}

}  // namespace wgrt
}  // namespace dnlp
