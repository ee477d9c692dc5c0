// Template-specialised batch solver, part 2: the interior-point loop of ONE small sparse instance by ONE wavefront.
//
// Role: the serial best_of / re-solve loop of the reference (cvxpy/problems/problem.py:1256-1269 -> IPOPT through
// ipopt_nlpif.py:140-170), for templates that wave_plan.h accepts.  The ALGORITHM is ipm_core.h's (Waechter & Biegler
// 2006 with IPOPT's defaults; every function below names the Ipm<E, K> member it restates) — same formulas, same
// decisions, same safeguards and retry ladder; what differs is how it is laid out for a 64-lane wavefront:
//   * every vector of the instance lives at a fixed place in the wavefront's share of LDS and is read / written through
//     LDS-typed pointers (ds_read / ds_write; the generic kernel reaches most of them through flat instructions, its
//     sparse factor through the global slab, and spills 3.3 KB per lane to scratch);
//   * variables and constraint rows are separate loops (the generic text runs both branch sides of every
//     [variables | rows] pass in each of its two trips);
//   * the tape is one table row per work unit and 32-bit CSR maps (wave_plan.h) staged once per workgroup;
//   * reductions are lane-strided partial sums + the fixed DPP tree of wave_ops.h: the same bits on every run.
// Single source over a lane policy P (lanes, lane(), sync(), sum / vmax, now()): WaveLanes (wave_batch.h) on the device, one
// host lane in the test oracle (oracle/oracle_lib.cpp) — where its serial sums reproduce the HostExec build of
// ipm_core.h, which is how the restatement is pinned on the CPU (tests/test_wave_ipm_cpu.py).
// What it does not have: the dense Bunch-Kaufman fallback of a structurally singular static pivot sequence — such an
// instance ends with status kWaveNeedsGeneric and the host hands it to the generic kernel (batch.h).
#pragma once
#ifndef DNLP_RTC
#include "ipm_core.h"
#include "wave_plan.h"
#endif

#if DNLP_DEVICE_PASS
#define DNLP_WLDS __attribute__((address_space(3)))
#define DNLP_WGLB __attribute__((address_space(1)))
#else
#define DNLP_WLDS
#define DNLP_WGLB
#endif

// Cycle profile of the phases (a -DDNLP_WAVE_PROF build; tools/wave_profile.sh): S->prof[k] accumulates s_memtime ticks per
// category, the kernel adds them up over the launch (wave_batch.h) and the host prints them (batch.h solve_wave).
#if defined(DNLP_WAVE_PROF) && DNLP_DEVICE_PASS
#define W_P0() const unsigned long long wp0_ = __builtin_readcyclecounter()
#define W_P1(k) S->prof[k] += __builtin_readcyclecounter() - wp0_
#else
#define W_P0() do { } while (0)
#define W_P1(k) do { } while (0)
#endif

// How the algorithm text reaches what never changes between the instances of a template — sizes, tables of the plan
// block, the places of an instance's vectors.  The library's own kernels and the host lane read them from the state
// record (set once per kernel by layout()).  A kernel compiled PER TEMPLATE at run time (wave_codegen.h, -DDNLP_WAVE_SPEC)
// knows them as constants: a size is a literal (a lane-strided loop over N <= 64 entries is one predicated trip), a vector
// is the wavefront's LDS share plus an immediate offset (no pointer load from the state record in front of every loop),
// a table is a fixed LDS address.
#ifdef DNLP_WAVE_SPEC
#define WK(f) (wspec::k_##f)
#define WV(f) (P::vec(S, wspec::v_##f))
#define WT(f) (P::tab(wspec::t_##f))
#define WDIR(set, k) (P::vec(S, wspec::v_dir[set][k]))
#define WCSR(f) (WCsr{P::tab(wspec::t_##f##_ptr), P::tab(wspec::t_##f##_idx), wspec::k_##f##_rows, wspec::k_##f##_val, wspec::k_##f##_id})
#define WCOO(f) (WCoo{P::tab(wspec::t_##f##_ptr), P::tab(wspec::t_##f##_ent), P::tab(wspec::t_##f##_src), P::tab(wspec::t_##f##_heavy), wspec::k_##f##_nout, wspec::k_##f##_nheavy})
#else
#define WK(f) (S->f)
#define WV(f) (S->f)
#define WT(f) (S->f)
#define WDIR(set, k) (S->dir[set][k])
#define WCSR(f) (S->f)
#define WCOO(f) (S->f)
#endif

namespace dnlp {

constexpr int kWaveProfSlots = 28;

// (device: the phases are real functions over one LDS pointer — the all-inlined generic kernel is a 28 k-instruction body)
#if DNLP_DEVICE_PASS
#define DNLP_WFN __attribute__((noinline))
#define DNLP_WINL __attribute__((always_inline))
#else
#define DNLP_WFN
#define DNLP_WINL
#endif

// Where a wavefront's vectors and the staged plan live is part of the lane policy P: P::D (a double of the instance's state)
// and P::I (a table entry) are LDS-qualified types when they fit the compute unit's LDS, plain (global memory) otherwise.
typedef DNLP_WLDS double WLdsD;
typedef DNLP_WLDS const int16_t WLdsI;       // (a plan staged in LDS is narrowed to 16 bits: wave_batch.h)
typedef DNLP_WGLB double WGlbD;
typedef DNLP_WGLB const i32 WGlbI;
typedef DNLP_WGLB const int16_t WGlbI16;
typedef DNLP_WGLB const double WG;    // a double of the instance's data row (global memory)

constexpr int kWaveNeedsGeneric = -197;
#ifndef DNLP_WAVE_FILTER_CAP
#define DNLP_WAVE_FILTER_CAP 32        // (BlockExecT::kFilterCap; the test oracle builds with HostExec's 1024)
#endif
constexpr int kWaveFilterCap = DNLP_WAVE_FILTER_CAP;

template <class WI> struct WCsrT { WI* ptr; WI* idx; i32 rows; i32 val; i32 id; };       // val: offset of the values in the instance's data row; id: 0 G, 1 Mg, 2 MJ, 3 Mw, 4 MH
template <class WI> struct WCooT { WI* ptr; WI* ent; WI* src; WI* heavy; i32 nout; i32 nheavy; };

// the fields of the state record that layout() sets once per kernel (X-macro lists: the record's declaration below, and
// wave_codegen.h, which prints them as the constants of a per-template kernel)
//   sizes (tail_L / tail_T: the dense tail of wave_plan.h — first tail level (= nlev without one), its order)
#define WAVE_SIZE_FIELDS(X) X(N) X(m) X(Z) X(nd) X(nh) X(nnzJ) X(nnzH) X(nunits) X(nblk) X(nvals) X(nlev) X(ngrp) X(nfwd) X(tail_L) X(tail_T) \
  X(l_c0) X(l_c) X(l_b) X(l_Jc) X(l_fp) X(l_fp2) X(l_x0) X(l_lb) X(l_ub) X(l_cl) X(l_cu)
//   tables of the plan block
#define WAVE_TAB_FIELDS(X) X(u_op) X(u_a0) X(u_a1) X(u_z) X(u_d0) X(u_d1) X(u_h) X(u_p) X(mm_idx) X(jac_rows) X(jac_cols) X(hess_rows) X(hess_cols) \
  X(jac_rowptr) X(bnode) X(soff) X(loff) X(doff) X(lev_off) X(sblk) X(sidx) X(lev_f) X(fnode) X(foff) X(fa) X(fu0) X(fu1) X(lev_g) X(gdst) X(goff) \
  X(upd_u) X(upd_v) X(hpos) X(jpos) X(dpos) X(lev_r) X(lev_t) X(lev_fe) X(t_node) X(t_d) X(t_l) X(t_fq) X(t_fp)
//   vectors of the instance
#define WAVE_VEC_FIELDS(X) X(x) X(zL) X(zU) X(xL) X(xU) X(grad) X(dx) X(dzL) X(dzU) X(xt) X(Sx) X(rx) X(tN) X(fixm) \
  X(s) X(y) X(vL) X(vU) X(sL) X(sU) X(eq) X(g) X(sg) X(ds) X(dy) X(dvL) X(dvU) X(st) X(gt) X(Dd) X(Ss) X(rs) X(rp) X(tM) X(csoc) \
  X(rhs) X(sol) X(res) X(cor) X(jv) X(xz) X(dvals) X(hvals) X(w) X(sl) X(Hs) X(svals) X(swork) X(scr)

// Everything one wavefront knows about the instance it is solving.  Lives in LDS (one per wavefront).
template <class WD, class WI>
struct WStateT {
  typedef WCsrT<WI> WCsr;
  typedef WCooT<WI> WCoo;
  // ---- sizes, tables, vectors (set once per kernel by layout(); a per-template kernel knows them as constants: WK / WT / WV) ----
#ifndef DNLP_WAVE_SPEC
#define W_DECL_K(f) i32 f;
#define W_DECL_T(f) WI* f;
#define W_DECL_V(f) WD* f;
  WAVE_SIZE_FIELDS(W_DECL_K)
  WAVE_TAB_FIELDS(W_DECL_T)
  WCsr G, Mg, MJ, Mw, MH;
  WCoo jr, jc, hs;
  WAVE_VEC_FIELDS(W_DECL_V)
  WD *dir[3][7];                     // [0] = dx ds dy dzL dzU dvL dvU, [1] = affine-scaling, [2] = centering direction
#undef W_DECL_K
#undef W_DECL_T
#undef W_DECL_V
#endif
  // ---- the instance (set per instance: w_bind) ----
  WG* row;                           // its data row (batch.h layout)
  double* park;                      // 3 N + 4 m doubles of global memory: where polish() keeps the iterate it may have to come back to
  const double *ws_g, *ws_l, *ws_u;  // warm-start multipliers or null
  i64 fallback_max_n;
  // ---- interior-point state (Ipm<E, K> members of the same names) ----
  IpmOptions opt;
  double sf, f, mu, tau, delta_w_last, theta_max, theta_min, last_obj, last_ratio, streak_theta0, streak_f0, resto_theta, t_begin;
  double filt_th[kWaveFilterCap], filt_ph[kWaveFilterCap], kkt_hist[4];
  double e_dual, e_primal, e_cmpl, e_sd, e_sc, e_total, e_primal_unscaled;      // e_cached_
  double inf_pr, inf_du, cmpl, nlp_error, wall;
  i32 nfilt, n_hist, acceptable_count, iter, status, factorizations, nb_cache, n_eq, n_fixed, last_nneg, ladder_rung,
      sparse_singular_streak, dc_fixed_count, tiny_streak;
  bool initialized, fixed_mode, e_cached_valid, jty_valid, delta_w_used_last_iter, dc_fixed_last, always_dc, in_solve,
       resto_stationary, bail,
       swept_xt;                    // the tape's z / dvals belong to the point in xt (the last sweep was a trial evaluation)
  // the quality function's constants and the golden section's two results: here, not on the stack (a struct passed by
  // reference and two reference outputs were private-memory traffic in every evaluation)
  // output slots of the phase functions: a function that is CALLED hands its scalar results back through these (the caller's
  // wrapper copies them out at once) — a reference parameter makes the caller's variable a private-memory variable, a
  // scratch store before and a scratch load after every call
  double o_d[10];
  i32 o_i[4];
  double qf_avg, qf_nd2, qf_np2, qf_n_dual, qf_n_pri, qf_fsel;
  i64 qf_nb;
  i32 qf_endpoint;
#ifdef DNLP_WAVE_PROF
  unsigned long long prof[kWaveProfSlots];
#endif
};

struct WErr { double dual, primal, cmpl, sd, sc, total, primal_unscaled; };
struct WMeasures { double theta, phi, chk; };

#ifdef DNLP_WAVE_GEN
// the phases of the static-pattern LDL^T as per-template straight-line code (wave_gen.h writes them, wave_gen_rt.h holds their
// building blocks); defined behind this header in the same translation unit
namespace wgen {
template <class P, class WS> DNLP_HD bool ldl_factor(WS* S);
template <class P, bool TWO, class WS, class XP, class YP> DNLP_HD void ldl_solve(WS* S, XP x, YP y);
template <class P, bool TWO, class WS, class WD> DNLP_HD void kkt_residual(WS* S, double dw, const WD* v, const WD* rhsv, WD* out, const WD* v2, const WD* rhsv2, WD* out2,
                                                                        double& en, double& sn, double& en2, double& sn2);
template <class P, class WS, class WD> DNLP_HD void jac_tmult(WS* S, const WD* v, WD* out);
template <class P, bool SPLIT, class WS, class WD> DNLP_HD void spmv(WS* S, int id, const WD* v, i32 base_off, WD* y, int scale_kind, double scalar, const WD* vhi, i32 split);
}  // namespace wgen
#endif

template <class P>
struct WaveIpm {
  typedef typename P::D WD;
  typedef typename P::I WI;
  typedef WStateT<WD, WI> WState;
  typedef DNLP_WLDS WState WS;          // (the state record itself is always in LDS)
  typedef WCsrT<WI> WCsr;
  typedef WCooT<WI> WCoo;
  // lane-strided loops
#define W_FOR(i, n) for (int i = P::lane(); i < (n); i += P::lanes)

#ifndef DNLP_WAVE_SPEC
  // ---- layout: the tables out of the staged block, the vectors out of the wavefront's share ---------------------------
  // (returns the doubles of state it laid out: wave_plan.h wave_state_doubles says the same number)
  DNLP_HD static i32 layout(WS* S, const WaveHdr* h, WI* blk, WD* base) {
    S->N = h->N; S->m = h->m; S->Z = h->Z; S->nd = h->nd; S->nh = h->nh; S->nnzJ = h->nnzJ; S->nnzH = h->nnzH; S->nunits = h->nunits;
    S->u_op = blk + h->u_op; S->u_a0 = blk + h->u_a0; S->u_a1 = blk + h->u_a1; S->u_z = blk + h->u_z; S->u_d0 = blk + h->u_d0;
    S->u_d1 = blk + h->u_d1; S->u_h = blk + h->u_h; S->u_p = blk + h->u_p; S->mm_idx = blk + h->mm_idx;
    S->G = WCsr{blk + h->G_ptr, blk + h->G_idx, h->m, h->l_G, 0};
    S->Mg = WCsr{blk + h->Mg_ptr, blk + h->Mg_idx, h->N, h->l_Mg, 1};
    S->MJ = WCsr{blk + h->MJ_ptr, blk + h->MJ_idx, h->nnzJ, h->l_MJ, 2};
    S->Mw = WCsr{blk + h->Mw_ptr, blk + h->Mw_idx, h->Z, h->l_Mw, 3};
    S->MH = WCsr{blk + h->MH_ptr, blk + h->MH_idx, h->nnzH, h->l_MH, 4};
    S->jac_rows = blk + h->jac_rows; S->jac_cols = blk + h->jac_cols; S->hess_rows = blk + h->hess_rows; S->hess_cols = blk + h->hess_cols;
    S->jac_rowptr = blk + h->jac_rowptr;
    S->jr = WCoo{blk + h->jr_ptr, blk + h->jr_ent, blk + h->jr_src, blk + h->jr_heavy, h->m, h->jr_nheavy};
    S->jc = WCoo{blk + h->jc_ptr, blk + h->jc_ent, blk + h->jc_src, blk + h->jc_heavy, h->N, h->jc_nheavy};
    S->hs = WCoo{blk + h->hs_ptr, blk + h->hs_ent, blk + h->hs_src, blk + h->hs_heavy, h->N, h->hs_nheavy};
    S->nblk = h->sp_nblk; S->nvals = h->sp_nvals; S->nlev = h->sp_nlev; S->ngrp = h->sp_ngrp; S->nfwd = h->sp_nfwd;
    S->bnode = blk + h->bnode; S->soff = blk + h->soff; S->loff = blk + h->loff; S->doff = blk + h->doff; S->lev_off = blk + h->lev_off;
    S->sblk = blk + h->sblk; S->sidx = blk + h->sidx; S->lev_f = blk + h->lev_f; S->fnode = blk + h->fnode; S->foff = blk + h->foff;
    S->fa = blk + h->fa; S->fu0 = blk + h->fu0; S->fu1 = blk + h->fu1; S->lev_g = blk + h->lev_g; S->gdst = blk + h->gdst;
    S->goff = blk + h->goff; S->upd_u = blk + h->tau; S->upd_v = blk + h->tav; S->hpos = blk + h->hpos; S->jpos = blk + h->jpos; S->dpos = blk + h->dpos;
    S->lev_r = blk + h->lev_r; S->lev_t = blk + h->lev_t; S->lev_fe = blk + h->lev_fe;
    S->t_node = blk + h->t_node; S->t_d = blk + h->t_d; S->t_l = blk + h->t_l; S->t_fq = blk + h->t_fq; S->t_fp = blk + h->t_fp;
    S->tail_L = h->tail_L; S->tail_T = h->tail_T;
    S->l_c0 = h->l_c0; S->l_c = h->l_c; S->l_b = h->l_b; S->l_Jc = h->l_Jc; S->l_fp = h->l_fp; S->l_fp2 = h->l_fp2;
    S->l_x0 = h->l_x0; S->l_lb = h->l_lb; S->l_ub = h->l_ub; S->l_cl = h->l_cl; S->l_cu = h->l_cu;
    // vectors, 16-byte granules (the count per class is wave_plan.h wave_state_doubles)
    const i32 N = h->N, m = h->m;
    WD* p = base;
    auto take = [&](i32 n) { WD* q = p; p += (n + 1) & ~1; return q; };
    S->x = take(N); S->zL = take(N); S->zU = take(N); S->xL = take(N); S->xU = take(N); S->grad = take(N);
    S->Sx = take(N); S->rx = take(N); S->tN = take(N); S->fixm = take(N);
    S->s = take(m); S->y = take(m); S->vL = take(m); S->vU = take(m); S->sL = take(m); S->sU = take(m); S->eq = take(m); S->g = take(m);
    S->sg = take(m); S->Dd = take(m); S->Ss = take(m); S->rs = take(m); S->rp = take(m);
    // (xz: the units' values z only — an argument index below N reads the point the sweep is at, x or the trial point, where it
    //  lies: no copy of it in front of z)
    S->jv = take(h->nnzJ); S->xz = take(h->Z); S->dvals = take(h->nd);
    // what only lives inside eval_hessian — the scaled multipliers sl, the unit weights w, the units' second derivatives hvals
    // (sl -> w -> hvals -> Hs) — borrows the three arrays of a linear solve, which are idle then, when it fits them
    const bool hess_borrows = h->nh <= N + m && h->Z <= N + m;
    if (!hess_borrows) { S->hvals = take(h->nh); S->w = take(h->Z); S->sl = take(1 + m); }
    S->Hs = take(h->nnzH); S->svals = take(h->sp_nvals);
    // One contiguous region of everything that is DEAD while the KKT matrix is being factorised — the step (dx .. dvU), the
    // trial point (xt, st, gt), the centering direction and the four arrays of a linear solve: the factorisation's work
    // arrays (the unscaled L, the inverse pivots, the scratch of a level's products) live there for its duration instead
    // of in 22 KB (circle packing n = 10) of LDS of their own, when they fit (wave_plan.h wave_state_doubles: same rule).
    WD* dead0 = p;
    S->dx = take(N); S->dzL = take(N); S->dzU = take(N); S->xt = take(N);
    S->ds = take(m); S->dy = take(m); S->dvL = take(m); S->dvU = take(m); S->st = take(m); S->gt = take(m);
    S->tM = S->gt;        // (the residual passes' scratch: the trial point's constraint values are dead inside a linear solve — the
                          //  dense end's scratch has always counted on that)
    WD* cy = take(m);
    // the centering direction as three [variables | rows] pairs: until its outputs are written they hold the second
    // right-hand side, solution and residual of the mu oracle's joint solve (quality_function_mu)
    WD* cx = take(N + m); WD* cs = cx + N; WD* czL = take(N + m); WD* cvL = czL + N; WD* czU = take(N + m); WD* cvU = czU + N;
    S->rhs = take(N + m); S->sol = take(N + m); S->res = take(N + m);
    if (hess_borrows) { S->hvals = S->rhs; S->w = S->sol; S->sl = S->res; }
    S->cor = S->res;      // (a refinement step's correction is solved IN PLACE in the residual's array: kkt_solve(res, cor) copies nothing)
    {
      const i32 nwork = (h->sp_nvals + 3 * h->sp_nblk + 8 + 1) & ~1, nscr = (h->scr_doubles + 1) & ~1;
      if (nwork + nscr <= static_cast<i32>(p - dead0)) { S->swork = dead0; S->scr = dead0 + nwork; }
      else { S->swork = take(nwork); S->scr = take(nscr); }
    }
    S->dir[0][0] = S->dx; S->dir[0][1] = S->ds; S->dir[0][2] = S->dy; S->dir[0][3] = S->dzL; S->dir[0][4] = S->dzU; S->dir[0][5] = S->dvL; S->dir[0][6] = S->dvU;
    // (the second-order correction's right-hand side lives in the CENTERING direction's slack part: that direction is dead
    //  once the barrier parameter is chosen, the correction runs inside the line search that follows and factors nothing —
    //  the affine-scaling arrays are not free for this: polish() parks the iterate there)
    S->csoc = cs;
    // The affine-scaling direction lives in the STEP's arrays: it is written after the mu oracle's joint solve (until then
    // those arrays are scratch, as they always were), read by the quality function, and the step itself — affine + mu x
    // centering — is formed in place.  3 N + 4 m doubles of state less: LDS decides how many wavefronts a compute unit holds.
    for (int k = 0; k < 7; ++k) S->dir[1][k] = S->dir[0][k];
    S->dir[2][0] = cx; S->dir[2][1] = cs; S->dir[2][2] = cy; S->dir[2][3] = czL; S->dir[2][4] = czU; S->dir[2][5] = cvL; S->dir[2][6] = cvU;
    return static_cast<i32>(p - base);
  }
#endif

  // ---- lane reductions (NaN conventions of BlockExecT::reduce: max NaN -> +inf, min NaN -> -inf) -------------------------
  DNLP_HD static double mxin(double acc, double v) { return fmax(acc, v != v ? kInf : v); }
  DNLP_HD static double mnin(double acc, double v) { v = v != v ? -kInf : v; return fmax(acc, -v); }      // (minimum as a maximum of negatives)

  // ====================================================================================================================
  // tape evaluation (model.h)
  // ====================================================================================================================
  // the general rule table (atom_math.h unary_rules: pow / exp / log / trigonometric ... a few thousand instructions inline) as a
  // function of its own: inside the sweep it cost the sweep 34 saved registers per call in the 256-register forms, whether
  // or not a unit ever took it
  struct U3 { double v, g1, g2; };
  DNLP_WFN DNLP_HD static U3 unary_slow(int op, double u, double p, double p2) {
    U3 r;
    unary_rules(op, u, p, p2, r.v, r.g1, r.g2);
    return r;
  }
  // Model::sweep: xz[0..N) <- src (unless it is xz already), then every flat unit: z, dvals (and hvals with the weights w)
  DNLP_WFN DNLP_HD static void sweep(WS* S, const WD* src, bool with_h) {
    W_P0();
    const int N = WK(N), nu = WK(nunits);
    WD *dv = WV(dvals), *hv = WV(hvals);
    WD* zz = WV(xz);
    const WD* ww = WV(w);
    const WD* zlo = WV(xz) - N;          // argument index u: u < N -> src[u], else zlo[u] = z[u - N]
    auto at = [&](i32 u) -> double { return (u < N ? src : zlo)[u]; };
    S->swept_xt = src == WV(xt);
    WI *uop = WT(u_op), *ua0 = WT(u_a0), *ua1 = WT(u_a1), *uz = WT(u_z), *ud0 = WT(u_d0), *ud1 = WT(u_d1), *uh = WT(u_h), *up = WT(u_p);
    // the per-segment parameters of the unary atoms come straight out of the instance's data row (global memory, read-only,
    // two loads per unit and sweep that travel beside the LDS loads): copies of them in LDS were 3 x nunits doubles of state,
    // and state is what decides how many wavefronts a compute unit holds (batch.h wave_form)
    WG *fp = S->row + WK(l_fp), *fp2 = S->row + WK(l_fp2);
    W_FOR(e, nu) {
      const int op = uop[e];
      const i32 zi = uz[e];
      if (op < OP_MUL) {
        double val, g1, g2;
        const i32 f = up[e];
        const double u = at(ua0[e]);
        // the two power atoms every canonical form is full of take their branch of pow_fast directly (atom_math.h: the
        // same expressions, so the same bits — the chain of comparisons in front of them is what a sweep was made of).
        // (the parameters are loaded where they are used: live across the general rule's code they cost the function 32
        //  more saved registers per call)
        int cls = 0;
        if (op == OP_POWER) {
          const double pd = fp[f], p2 = fp2[f];
          cls = (pd == 2.0 && p2 == 2.0) ? 1 : (pd == 0.5 && p2 == 0.5) ? 2 : 0;
        }
        if (cls == 1) {                // x^2: pow_fast(u, 2) = u u; 2 pow_fast(u, 1) = 2 u; 2 (2 - 1) pow_fast(u, 0) = 2
          const double pd = 2.0;
          val = u * u; g1 = pd * u; g2 = pd * (pd - 1.0) * 1.0;
        } else if (cls == 2) {         // sqrt: pow_fast(u, 0.5) = sqrt(u); 0.5 pow_fast(u, -0.5); 0.5 (-0.5) pow_fast(u, -1.5)
          const double pd = 0.5, sq = sqrt(u);
          val = sq; g1 = pd * (1.0 / sq); g2 = pd * (pd - 1.0) * (1.0 / (u * sq));
        } else {
          const U3 r = unary_slow(op, u, fp[f], fp2[f]);
          val = r.v; g1 = r.g1; g2 = r.g2;
        }
        zz[zi] = val;
        dv[ud0[e]] = g1;
        if (with_h) hv[uh[e]] = ww[zi] * g2;
      } else if (op == OP_MUL) {
        // bilinear u*v: binary_operators.py:586-591 (Jacobian), :543-546 (cross Hessian)
        const double u = at(ua0[e]), v = at(ua1[e]);
        zz[zi] = u * v;
        dv[ud0[e]] = v;
        dv[ud1[e]] = u;
        if (with_h) hv[uh[e]] = ww[zi];
      } else if (op == OP_REL_ENTR) {
        // rel_entr.py:37-40, :129-148, :150-179
        const double u = at(ua0[e]), v = at(ua1[e]);
        const double lr = log(u / v);
        zz[zi] = u * lr;
        dv[ud0[e]] = lr + 1.0;
        dv[ud1[e]] = -u / v;
        if (with_h) {
          const double wi = ww[zi];
          const i32 hb = uh[e], n = up[e];
          hv[hb] = wi / u;
          hv[hb + n] = wi * u / (v * v);
          hv[hb + 2 * n] = -wi / v;
        }
      } else {
        // OP_MATMUL: one output entry of U @ V (model.h sweep_flat)
        const i32 kk = ua1[e], db0 = ud0[e], db1 = ud1[e], hb = uh[e];
        WI* mi = WT(mm_idx) + ua0[e];
        double acc = 0.0;
        for (i32 q = 0; q < kk; ++q) {
          const double u = at(mi[2 * q]), v = at(mi[2 * q + 1]);
          acc += u * v;
          dv[db0 + q] = v;
          dv[db1 + q] = u;
          if (with_h) hv[hb + q] = ww[zi];
        }
        zz[zi] = acc;
      }
    }
    W_P1(15);
    P::sync();
  }
  // Sums of products, the pattern of every index-driven piece below (CSR maps, the products by output, the update
  // program and the substitutions of the LDL^T): a lane that walks "its" output's entries runs a chain of dependent LDS
  // trips per entry (index -> operands -> add).  Instead the products of ALL entries of the phase are formed side by
  // side into the scratch array (one entry per lane and trip: the index loads of different entries overlap), and the
  // owner of an output then adds its run of consecutive scratch entries — in the same order as before, so the sums keep
  // their bits.
  DNLP_HD static double run_sum(double acc, const WD* p, int n) {
    int q = 0;
    for (; q + 4 <= n; q += 4) {
      const double t0 = p[q], t1 = p[q + 1], t2 = p[q + 2], t3 = p[q + 3];
      acc += t0; acc += t1; acc += t2; acc += t3;
    }
    for (; q < n; ++q) acc += p[q];
    return acc;
  }
  // Model::spmv: y = (base + M v) [* scale];  scale_kind 0 none, 1 a scalar, 2 sg[r], 3 sg[jac_rows[r]]
  // (SPLIT: v is the vector [x | z] in two places — column c < split reads v[c], the others vhi[c]: the sweep's point and WV(xz) - N)
  template <bool SPLIT = false>
  DNLP_WFN DNLP_HD static void spmv(DNLP_WLDS WState* S, const WCsr M, const WD* v, i32 base_off, WD* y, int scale_kind, double scalar,
                                    const WD* vhi = nullptr, i32 split = 0) {
#ifdef DNLP_WAVE_GEN
    { W_P0(); wgen::spmv<P, SPLIT>(S, M.id, v, base_off, y, scale_kind, scalar, vhi, split); W_P1(16); return; }
#endif
    W_P0();
    WI *ptr = M.ptr, *idx = M.idx;
    WG* val = S->row + M.val;
    WG* base = base_off >= 0 ? S->row + base_off : nullptr;
    const WD* sg = WV(sg);
    WI* jr = WT(jac_rows);
    W_FOR(r, M.rows) {
      double sacc = base ? base[r] : 0.0;
      const i32 k1 = ptr[r + 1];
      for (i32 k = ptr[r]; k < k1; ++k) {
        const i32 c = idx[k];
        sacc += val[k] * ((SPLIT && c >= split) ? vhi : v)[c];
      }
      if (scale_kind == 1) sacc *= scalar;
      else if (scale_kind == 2) sacc *= sg[r];
      else if (scale_kind == 3) sacc *= sg[jr[r]];
      y[r] = sacc;
    }
    P::sync();
    W_P1(16);
  }
  // Ipm::eval_fg (check folded into the callers): f~ and g~ at xp; returns isfinite(f~)
  DNLP_HD static bool eval_fg(WS* S, const WD* xp, double& fval, WD* gout) {
    const bool ok = eval_fg_impl(S, xp, gout);
    fval = S->o_d[0];
    return ok;
  }
#if DNLP_DEVICE_PASS
  DNLP_HD static bool eval_fg(WS* S, const WD* xp, DNLP_WLDS double& fval, WD* gout) {      // (a field of the state record as the destination)
    const bool ok = eval_fg_impl(S, xp, gout);
    fval = S->o_d[0];
    return ok;
  }
#endif
  DNLP_WFN DNLP_HD static bool eval_fg_impl(WS* S, const WD* xp, WD* gout) {
    auto& fval = S->o_d[0];
    sweep(S, xp, false);
    const int NZ = WK(N) + WK(Z);
    WG* cc = S->row + WK(l_c);
    const int N = WK(N);
    const WD* zlo = WV(xz) - N;
    double acc = 0.0;
    W_FOR(i, NZ) acc += cc[i] * (i < N ? xp : zlo)[i];
    fval = S->sf * (S->row[WK(l_c0)] + P::sum(acc));
    spmv<true>(S, WCSR(G), xp, WK(l_b), gout, 2, 0.0, zlo, N);
    return std::isfinite(fval);
  }
  DNLP_HD static double nan_check(WS* S, const WD* gg) {
    double acc = 0.0;
    W_FOR(i, WK(m)) acc += gg[i] - gg[i];
    return P::sum(acc);
  }
  // Ipm::eval_derivs_after_sweep
  DNLP_HD static void eval_derivs(WS* S) {
    S->jty_valid = false;
    spmv(S, WCSR(Mg), WV(dvals), WK(l_c), WV(grad), 1, S->sf);
    spmv(S, WCSR(MJ), WV(dvals), WK(l_Jc), WV(jv), 3, 0.0);
  }
  // Ipm::eval_hessian + Model::eval_hess
  DNLP_WFN DNLP_HD static void eval_hessian(WS* S) {
    W_P0();
    const int m = WK(m);
    WD* sl = WV(sl);
    const WD *sg = WV(sg), *yy = WV(y);
    const double sff = S->sf;
    if (P::lane() == 0) sl[0] = sff;
    W_FOR(i, m) sl[1 + i] = sg[i] * yy[i];
    P::sync();
    spmv(S, WCSR(Mw), WV(sl), -1, WV(w), 0, 0.0);
    sweep(S, WV(x), true);
    spmv(S, WCSR(MH), WV(hvals), -1, WV(Hs), 0, 0.0);
    W_P1(3);
  }
  // BlockExecT::coo_gather through the tape's index by output: out = J v / J^T v / sym(H) v
  DNLP_WFN DNLP_WFN DNLP_HD static void coo(DNLP_WLDS WState* S, const WCoo ix, const WD* a, const WD* v, WD* out) {
    W_P0();
    WI *ptr = ix.ptr, *ent = ix.ent, *src = ix.src;
    W_FOR(gq, ix.nout) {
      const i32 p0 = ptr[gq], p1 = ptr[gq + 1];
      if (p1 - p0 > static_cast<i32>(kCooHeavy)) continue;
      double sacc = 0.0;
      for (i32 p = p0; p < p1; ++p) sacc += a[ent[p]] * v[src[p]];
      out[gq] = sacc;
    }
    for (i32 hq = 0; hq < ix.nheavy; ++hq) {
      const i32 gq = ix.heavy[hq];
      const i32 p0 = ptr[gq], cnt = ptr[gq + 1] - p0;
      double sacc = 0.0;
      W_FOR(q, cnt) sacc += a[ent[p0 + q]] * v[src[p0 + q]];
      sacc = P::sum(sacc);
      if (P::lane() == 0) out[gq] = sacc;
    }
    P::sync();
    W_P1(8);
  }
  DNLP_HD static void hess_mult(WS* S, const WD* v, WD* out) { coo(S, WCOO(hs), WV(Hs), v, out); }
  DNLP_HD static void jac_mult(WS* S, const WD* v, WD* out) { coo(S, WCOO(jr), WV(jv), v, out); }
#ifdef DNLP_WAVE_GEN
  DNLP_WFN DNLP_HD static void jac_tmult(WS* S, const WD* v, WD* out) { W_P0(); wgen::jac_tmult<P>(S, v, out); W_P1(8); }
#else
  DNLP_HD static void jac_tmult(WS* S, const WD* v, WD* out) { coo(S, WCOO(jc), WV(jv), v, out); }
#endif

  // ====================================================================================================================
  // KKT system: assembly (kkt_dense.h assemble_factor, sparse branch) and the static-pattern LDL^T (sparse_ldl.h)
  // ====================================================================================================================
  DNLP_HD static bool assemble_factor(WS* S, const WD* Sx, const WD* D, double dw, bool zero_h, int* nneg_out, int* nzero_out) {
    const bool ok = assemble_factor_impl(S, Sx, D, dw, zero_h);
    *nneg_out = S->o_i[1]; *nzero_out = S->o_i[2];
    return ok;
  }
  DNLP_WFN DNLP_HD static bool assemble_factor_impl(WS* S, const WD* Sx, const WD* D, double dw, bool zero_h) {
    const int N = WK(N), m = WK(m), nnzH = WK(nnzH), nnzJ = WK(nnzJ), nvals = WK(nvals);
    W_P0();
    WD* V = WV(svals);
    const WD *fixm = WV(fixm), *hs = WV(Hs), *jv = WV(jv);
    WI *hp = WT(hpos), *jp = WT(jpos), *dp = WT(dpos), *hr = WT(hess_rows), *hc = WT(hess_cols), *jc = WT(jac_cols);
    W_FOR(a, nvals) V[a] = 0.0;
    P::sync();
    if (!zero_h) {
      W_FOR(p, nnzH) {
        if (fixm[hr[p]] != 0.0 || fixm[hc[p]] != 0.0 || hp[p] < 0) continue;
        V[hp[p]] += hs[p];
      }
    }
    W_FOR(p, nnzJ) {
      if (fixm[jc[p]] != 0.0 || jp[p] < 0) continue;
      V[jp[p]] = jv[p];
    }
    P::sync();
    W_FOR(j, N) {
      if (fixm[j] != 0.0) V[dp[j]] = 1.0;
      else V[dp[j]] += Sx[j] + dw;
    }
    W_FOR(i, m) V[dp[N + i]] = -D[i];
    P::sync();
    S->factorizations++;
    W_P1(5);
    return ldl_factor_impl(S);
  }
  // sparse_ldl.h sp_pivot
  DNLP_HD static void sp_pivot(WS* S, WD* vals, WD* dinv, int k, double& nneg, double& nzero, double& bad) {
    WD* Dk = vals + WT(doff)[k];
    WD* di = dinv + 3 * k;
    if (WT(bnode)[2 * k + 1] < 0) {
      double d = Dk[0];
      if (!(d == d)) bad += 1.0;
      if (fabs(d) < 1e-300) { nzero += 1.0; d = 1e-20; Dk[0] = d; }
      if (d < 0.0) nneg += 1.0;
      di[0] = 1.0 / d;
    } else {
      const double a = Dk[0], c = Dk[1], e = Dk[2];
      double det = a * e - c * c;
      if (!(det == det)) bad += 1.0;
      if (fabs(det) < 1e-300) { nzero += 1.0; det = -1e-20; }
      if (det < 0.0) nneg += 1.0;
      else if (a < 0.0 || (a == 0.0 && e < 0.0)) nneg += 2.0;
      di[0] = e / det; di[1] = -c / det; di[2] = a / det;
    }
  }
  // sparse_ldl.h sp_scale
  DNLP_HD static void sp_scale(WS* S, WD* vals, WD* w, const WD* dinv, int r) {
    const int k = WT(sblk)[r], i = r - WT(soff)[k];
    const WD* di = dinv + 3 * k;
    if (WT(bnode)[2 * k + 1] < 0) {
      const int a = WT(loff)[k] + i;
      const double l1 = vals[a];
      w[a] = l1;
      vals[a] = l1 * di[0];
    } else {
      const int a = WT(loff)[k] + 2 * i;
      const double l1 = vals[a], l2 = vals[a + 1];
      w[a] = l1; w[a + 1] = l2;
      vals[a] = di[0] * l1 + di[1] * l2;
      vals[a + 1] = di[1] * l1 + di[2] * l2;
    }
  }
  DNLP_HD static double sp_update(WI* tau, WI* tav, const WD* vals, const WD* w, int q) {
    const i32 au = tau[q], av = tav[q];
    if (av >= 0) return w[au] * vals[av];
    const i32 bv = ~av;
    return w[au] * vals[bv] + w[au + 1] * vals[bv + 1];
  }
  // ---- dense tail (wave_plan.h WaveHdr): the trailing T x T matrix of a chain of one-block levels, ONE ROW PER LANE in
  // registers.  The operations on every entry are those of the level code (sp_pivot, sp_scale, one update product per
  // destination and level, subtracted in level order), so a host lane that owns all rows reproduces the generic text's bits;
  // what goes away is the level machinery — four phases of index walks per block — for what is T^3 / 6 multiply-adds.
  // (three widths of the unrolled loops — 12, 24, 32 rows — so that a tail of 9 or 21 rows does not pay for 32: the
  //  padding columns cost products, loads and skipped steps; circle packing n = 4 has T = 9, n = 10 T = 21)
  static constexpr int kTailMax = 32;
  struct T3 { double nneg, nzero, bad; };      // (lane-local counts — lane 0 counts — handed back in registers)
  template <int kTailMax>
  DNLP_WFN DNLP_HD static T3 tail_factor_n(WS* S) {
    constexpr int kTailSlots = (kTailMax + P::lanes - 1) / P::lanes;      // rows of the tail a lane owns (device 1, host: all)
    W_P0();
    const int T = P::uni(WK(tail_T)), me = P::lane();      // (uniform: a scalar register, so that the unrolled loops below test it on the scalar unit)
    double nneg = 0.0, nzero = 0.0, bad = 0.0;           // (lane 0 counts; handed back once)
    WD* vals = WV(svals);
    WI *td = WT(t_d), *tl = WT(t_l);
    double A[kTailSlots][kTailMax];
#pragma clang loop unroll(full)
    for (int sl = 0; sl < kTailSlots; ++sl) {
      const int i = me + P::lanes * sl;
#pragma clang loop unroll(full)
      for (int j = 0; j < kTailMax; ++j) {
        double v = 0.0;
        if (i < T && j <= i) v = vals[j == i ? td[i] : tl[i * T + j]];
        A[sl][j] = v;
      }
    }
#pragma clang loop unroll(full)
    for (int k = 0; k < kTailMax; ++k) if (k < T) {
      double col[kTailSlots];
#pragma clang loop unroll(full)
      for (int sl = 0; sl < kTailSlots; ++sl) col[sl] = A[sl][k];
      // sp_pivot of block k (a 1x1 block): every lane computes it from the same value, lane 0 counts
      double d = P::row_get(col, k);
      if (!(d == d) && me == 0) bad += 1.0;
      bool fixed = false;
      if (fabs(d) < 1e-300) { if (me == 0) nzero += 1.0; d = 1e-20; fixed = true; }
      if (d < 0.0 && me == 0) nneg += 1.0;
      const double dinv = 1.0 / d;
      // sp_scale: rows i > k keep the unscaled entry for the updates (w) and store L = l / d
      double wv[kTailSlots], sc[kTailSlots];
#pragma clang loop unroll(full)
      for (int sl = 0; sl < kTailSlots; ++sl) {
        const int i = me + P::lanes * sl;
        const bool below = i > k && i < T;
        wv[sl] = below ? col[sl] : 0.0;
        sc[sl] = below ? col[sl] * dinv : 0.0;
        if (below) A[sl][k] = sc[sl];
        if (i == k && fixed) A[sl][k] = d;
      }
      // the level's update triples: destination (i, j), i >= j > k, gets  - l_ik (l_jk / d).  No predicate on (i, j): rows
      // up to k and rows / columns from T on carry wv = 0 or sc = 0, and what the products do to the entries ABOVE the diagonal
      // is never read — three instructions per pair (two v_readlane, one fma) instead of a mask per pair
#pragma clang loop unroll(full)
      for (int j = k + 1; j < kTailMax; ++j) {
        const double sj = P::row_get(sc, j);
#pragma clang loop unroll(full)
        for (int sl = 0; sl < kTailSlots; ++sl) A[sl][j] -= wv[sl] * sj;
      }
    }
#pragma clang loop unroll(full)
    for (int sl = 0; sl < kTailSlots; ++sl) {
      const int i = me + P::lanes * sl;
#pragma clang loop unroll(full)
      for (int j = 0; j < kTailMax; ++j)
        if (i < T && j <= i) vals[j == i ? td[i] : tl[i * T + j]] = A[sl][j];
    }
    P::sync();
    W_P1(24);
    T3 r;
    r.nneg = nneg; r.nzero = nzero; r.bad = bad;
    return r;
  }
  // forward substitution through the tail: the gathers of the tail's targets from the blocks before it (products side by
  // side, runs added in storage order), then row t adds L_tk x_k for k < t as x_k becomes final — the order of the level code
  template <int kTailMax>
  DNLP_WFN DNLP_HD static void tail_forward_n(WS* S, WD* x, WD* y) {
    constexpr int kTailSlots = (kTailMax + P::lanes - 1) / P::lanes;      // rows of the tail a lane owns (device 1, host: all)
    W_P0();
    const int T = P::uni(WK(tail_T)), me = P::lane();      // (uniform: a scalar register, so that the unrolled loops below test it on the scalar unit)
    const WD* vals = WV(svals);
    WI *tn = WT(t_node), *tl = WT(t_l), *tfq = WT(t_fq), *tfp = WT(t_fp), *fa = WT(fa), *fu0 = WT(fu0), *fu1 = WT(fu1);
    const bool two = y != nullptr;
    WD* scr = WV(dy);                        // (dy dvL dvU st gt: consecutive and dead during a solve; wave_plan.h keeps t_nf within them)
    WD* scr2 = scr + tfp[T];
    const int nf = tfp[T];
    for (int n = me; n < nf; n += P::lanes) {
      const int q = tfq[n];
      const i32 a = fa[q], u0 = fu0[q];
      if (a >= 0) { const double l = vals[a]; scr[n] = l * x[u0]; if (two) scr2[n] = l * y[u0]; }
      else {
        const i32 b = ~a, u1 = fu1[q];
        const double l0 = vals[b], l1 = vals[b + 1];
        scr[n] = l0 * x[u0] + l1 * x[u1];
        if (two) scr2[n] = l0 * y[u0] + l1 * y[u1];
      }
    }
    P::sync();
    double acc[kTailSlots], acc2[kTailSlots], xr[kTailSlots], yr[kTailSlots], Lr[kTailSlots][kTailMax];
#pragma clang loop unroll(full)
    for (int sl = 0; sl < kTailSlots; ++sl) {
      const int t = me + P::lanes * sl;
      const bool in = t < T;
      const int f0 = in ? tfp[t] : 0, cnt = in ? tfp[t + 1] - f0 : 0;
      acc[sl] = run_sum(0.0, scr + f0, cnt);
      acc2[sl] = two ? run_sum(0.0, scr2 + f0, cnt) : 0.0;
      xr[sl] = in ? x[tn[t]] : 0.0;
      yr[sl] = (in && two) ? y[tn[t]] : 0.0;
#pragma clang loop unroll(full)
      for (int k = 0; k < kTailMax; ++k) Lr[sl][k] = (in && k < t) ? vals[tl[t * T + k]] : 0.0;
    }
#pragma clang loop unroll(full)
    for (int k = 0; k < kTailMax; ++k) if (k < T) {
      double xf[kTailSlots], yf[kTailSlots];
#pragma clang loop unroll(full)
      for (int sl = 0; sl < kTailSlots; ++sl) { xf[sl] = xr[sl] - acc[sl]; yf[sl] = yr[sl] - acc2[sl]; }
      const double xk = P::row_get(xf, k), yk = two ? P::row_get(yf, k) : 0.0;
#pragma clang loop unroll(full)
      for (int sl = 0; sl < kTailSlots; ++sl) {
        const int t = me + P::lanes * sl;
        if (t == k) { xr[sl] = xk; yr[sl] = yk; }
        // (no predicate: Lr[.][k] is zero for the rows up to k, whose sums are never looked at again)
        acc[sl] += Lr[sl][k] * xk;
        if (two) acc2[sl] += Lr[sl][k] * yk;
      }
    }
#pragma clang loop unroll(full)
    for (int sl = 0; sl < kTailSlots; ++sl) {
      const int t = me + P::lanes * sl;
      if (t < T) { x[tn[t]] = xr[sl]; if (two) y[tn[t]] = yr[sl]; }
    }
    P::sync();
    W_P1(25);
  }
  // backward substitution through the tail (after D^-1): x_t -= sum over i > t of L_it x_i, t descending
  template <int kTailMax>
  DNLP_WFN DNLP_HD static void tail_backward_n(WS* S, WD* x, WD* y) {
    constexpr int kTailSlots = (kTailMax + P::lanes - 1) / P::lanes;      // rows of the tail a lane owns (device 1, host: all)
    W_P0();
    const int T = P::uni(WK(tail_T)), me = P::lane();      // (uniform: a scalar register, so that the unrolled loops below test it on the scalar unit)
    const WD* vals = WV(svals);
    WI *tn = WT(t_node), *tl = WT(t_l);
    const bool two = y != nullptr;
    double xr[kTailSlots], yr[kTailSlots], Lr[kTailSlots][kTailMax];
#pragma clang loop unroll(full)
    for (int sl = 0; sl < kTailSlots; ++sl) {
      const int i = me + P::lanes * sl;
      const bool in = i < T;
      xr[sl] = in ? x[tn[i]] : 0.0;
      yr[sl] = (in && two) ? y[tn[i]] : 0.0;
#pragma clang loop unroll(full)
      for (int k = 0; k < kTailMax; ++k) Lr[sl][k] = (in && k < i) ? vals[tl[i * T + k]] : 0.0;
    }
#pragma clang loop unroll(full)
    for (int t = kTailMax - 1; t >= 0; --t) if (t < T) {
      double a0 = 0.0, c0 = 0.0;
#pragma clang loop unroll(full)
      for (int sl = 0; sl < kTailSlots; ++sl) {
        a0 += Lr[sl][t] * xr[sl];          // (Lr[.][t] is zero for the rows up to t and beyond the tail)
        if (two) c0 += Lr[sl][t] * yr[sl];
      }
      a0 = P::sum(a0);
      if (two) c0 = P::sum(c0);
#pragma clang loop unroll(full)
      for (int sl = 0; sl < kTailSlots; ++sl) {
        const int i = me + P::lanes * sl;
        if (i == t) { xr[sl] -= a0; if (two) yr[sl] -= c0; }
      }
    }
#pragma clang loop unroll(full)
    for (int sl = 0; sl < kTailSlots; ++sl) {
      const int i = me + P::lanes * sl;
      if (i < T) { x[tn[i]] = xr[sl]; if (two) y[tn[i]] = yr[sl]; }
    }
    P::sync();
    W_P1(26);
  }
  DNLP_HD static void tail_factor(WS* S, double& nneg_io, double& nzero_io, double& bad_io) {
    const int T = P::uni(WK(tail_T));
    const T3 r = T <= 12 ? tail_factor_n<12>(S) : T <= 24 ? tail_factor_n<24>(S) : tail_factor_n<32>(S);
    nneg_io += r.nneg; nzero_io += r.nzero; bad_io += r.bad;
  }
  DNLP_HD static void tail_forward(WS* S, WD* x, WD* y) {
    const int T = P::uni(WK(tail_T));
    if (T <= 12) tail_forward_n<12>(S, x, y);
    else if (T <= 24) tail_forward_n<24>(S, x, y);
    else tail_forward_n<32>(S, x, y);
  }
  DNLP_HD static void tail_backward(WS* S, WD* x, WD* y) {
    const int T = P::uni(WK(tail_T));
    if (T <= 12) tail_backward_n<12>(S, x, y);
    else if (T <= 24) tail_backward_n<24>(S, x, y);
    else tail_backward_n<32>(S, x, y);
  }
  // sparse_ldl.h sparse_ldl_factor (no dense tail).  Per level: pivots, row scaling, then the update triples — their
  // products side by side into the scratch array, each destination's run added in storage order (see run_sum).
  DNLP_WFN DNLP_HD static bool ldl_factor_impl(WS* S) {
#ifdef DNLP_WAVE_GEN
    { W_P0(); const bool okg = wgen::ldl_factor<P>(S); W_P1(6); return okg; }
#endif
    auto* nneg_out = &S->o_i[1];
    auto* nzero_out = &S->o_i[2];
    W_P0();
    const int L = P::lanes, me = P::lane();
    WD* vals = WV(svals);
    WD* w = WV(swork);
    WD* dinv = WV(swork) + WK(nvals);
    WD* scr = WV(scr);
    WI *lev_off = WT(lev_off), *soff = WT(soff), *lev_g = WT(lev_g), *goff = WT(goff), *gdst = WT(gdst), *tau = WT(upd_u), *tav = WT(upd_v);
    double nneg = 0.0, nzero = 0.0, bad = 0.0;
    const int nlev = WK(tail_L), nt = WK(nlev) + 1;          // (the levels before the dense tail; tail_L = nlev without one)
    // (the per-level bounds: one table entry per lane in a register, read back with v_readlane — two dependent uniform
    //  LDS trips less in front of every level phase)
    WI *lev_r = WT(lev_r), *lev_t = WT(lev_t);
    const int c_off = P::tab_load(lev_off, nt), c_r = P::tab_load(lev_r, nt), c_g = P::tab_load(lev_g, nt), c_t = P::tab_load(lev_t, nt);
    for (int lev = 0; lev < nlev; ++lev) {
      const int b0 = P::tab_at(lev_off, c_off, lev, nt), b1 = P::tab_at(lev_off, c_off, lev + 1, nt);
      const int r0 = P::tab_at(lev_r, c_r, lev, nt), r1 = P::tab_at(lev_r, c_r, lev + 1, nt);
      for (int k = b0 + me; k < b1; k += L) sp_pivot(S, vals, dinv, k, nneg, nzero, bad);
      if (r1 == r0) continue;          // (a level without struct rows — the last block: nothing to scale, nothing to update)
      P::sync();
      for (int r = r0 + me; r < r1; r += L) sp_scale(S, vals, w, dinv, r);
      P::sync();
      const int g0 = P::tab_at(lev_g, c_g, lev, nt), g1 = P::tab_at(lev_g, c_g, lev + 1, nt);
      const int t0 = P::tab_at(lev_t, c_t, lev, nt), ntr = P::tab_at(lev_t, c_t, lev + 1, nt) - t0, ngr = g1 - g0;
      if (ntr == 0) continue;
      for (int q = me; q < ntr; q += L) scr[q] = sp_update(tau, tav, vals, w, t0 + q);
      P::sync();
      if (ngr * 8 <= L && ntr >= 16 * ngr) {
        for (int gq = g0; gq < g1; ++gq) {
          double acc = 0.0;
          const int q0 = goff[gq] - t0, qe = goff[gq + 1] - t0;
          for (int q = q0 + me; q < qe; q += L) acc += scr[q];
          acc = P::sum(acc);
          if (me == 0) vals[gdst[gq]] -= acc;
        }
      } else {
        for (int gq = g0 + me; gq < g1; gq += L) {
          const int q0 = goff[gq] - t0;
          vals[gdst[gq]] -= run_sum(0.0, scr + q0, goff[gq + 1] - t0 - q0);
        }
      }
      P::sync();
    }
    if (WK(tail_T) > 0) tail_factor(S, nneg, nzero, bad);
    { double r3[3] = {nneg, nzero, bad}; P::sum_n(r3); nneg = r3[0]; nzero = r3[1]; bad = r3[2]; }
    *nneg_out = static_cast<int>(nneg);
    *nzero_out = static_cast<int>(nzero);
    W_P1(6);
    return bad == 0.0;
  }
  // (measured on MI355X and not kept: forming the terms of four entries of a lane's run side by side so that their index and
  //  operand loads overlap — the substitutions went from 43.6 to 49.8 k cycles per iteration, the residual pass from 14.3 to
  //  15.4: at one wavefront per SIMD these loops are bound by the NUMBER of instructions issued, not by the LDS round trips)
  // sparse_ldl.h sp_dsolve: D^-1 on block k of x (and y)
  DNLP_HD static void dsolve(const WD* vals, WI* doff, i32 u0, i32 u1, int k, WD* x, WD* y) {
    const WD* Dk = vals + doff[k];
    if (u1 < 0) {
      const double d = Dk[0];
      x[u0] /= d;
      if (y) y[u0] /= d;
    } else {
      const double a = Dk[0], c = Dk[1], e = Dk[2];
      double det = a * e - c * c;
      if (fabs(det) < 1e-300) det = -1e-20;
      const double x0 = x[u0], x1 = x[u1];
      x[u0] = (e * x0 - c * x1) / det;
      x[u1] = (a * x1 - c * x0) / det;
      if (y) {
        const double y0 = y[u0], y1 = y[u1];
        y[u0] = (e * y0 - c * y1) / det;
        y[u1] = (a * y1 - c * y0) / det;
      }
    }
  }
  // sparse_ldl.h sparse_ldl_solve: x <- K^-1 x — and y <- K^-1 y in the same level phases when a second right-hand side is
  // given (the mu oracle's affine-scaling and centering systems share the factor: one walk of the index arrays, one
  // chain of level barriers for both; each vector sees exactly the operations of a solve of its own)
  DNLP_WFN DNLP_HD static void ldl_solve(WS* S, WD* x, WD* y) {
#ifdef DNLP_WAVE_GEN
    {
      W_P0();
#if DNLP_DEVICE_PASS
      // a policy whose vectors are GENERIC pointers (the workgroup kernel: the solves' arrays in LDS, the rest in the slab) hands
      // the right-hand sides over typed by where they live: through a generic pointer every access is a FLAT instruction —
      // 1 576 of them in path planning's solve — whose wait is for global memory AND LDS, so an LDS operand also waited for the
      // table words the phase had just sent for.  (wspec::kSolve2Lds: the second system's arrays are in LDS too — then no
      // right-hand side is ever in the slab; else x may be the second system's residual, y always is.)
      if constexpr (P::lds_generic) {
        typedef DNLP_WLDS double* LP;
        typedef DNLP_WGLB double* QP;
        // (a right-hand side somewhere these cases do not name would be read through the wrong address space: stop the kernel
        //  instead — the launch then fails with a HIP error)
        const bool xs = __builtin_amdgcn_is_shared(x), ys = y && __builtin_amdgcn_is_shared(y);
        if constexpr (wspec::kSolve2Lds) {
          if (!xs || (y && !ys)) __builtin_trap();
          if (y) wgen::ldl_solve<P, true>(S, (LP)x, (LP)y); else wgen::ldl_solve<P, false>(S, (LP)x, (LP)nullptr);
        } else if (xs) {
          if (ys) __builtin_trap();
          if (y) wgen::ldl_solve<P, true>(S, (LP)x, (QP)y); else wgen::ldl_solve<P, false>(S, (LP)x, (LP)nullptr);
        } else {
          if (ys) __builtin_trap();
          if (y) wgen::ldl_solve<P, true>(S, (QP)x, (QP)y); else wgen::ldl_solve<P, false>(S, (QP)x, (QP)nullptr);
        }
      } else
#endif
      if (y) wgen::ldl_solve<P, true>(S, x, y); else wgen::ldl_solve<P, false>(S, x, y);
      W_P1(7);
      return;
    }
#endif
    W_P0();
    const int L = P::lanes, me = P::lane();
    const WD* vals = WV(svals);
    WI *lev_f = WT(lev_f), *foff = WT(foff), *fnode = WT(fnode), *lev_off = WT(lev_off), *soff = WT(soff), *bnode = WT(bnode), *loff = WT(loff),
       *sidx = WT(sidx), *doff = WT(doff), *fa = WT(fa), *fu0 = WT(fu0), *fu1 = WT(fu1);
    const int nlev = WK(tail_L), nblk = WK(nblk), nt = WK(nlev) + 1;      // (the levels before the dense tail)
    WI *lev_r = WT(lev_r), *lev_fe = WT(lev_fe);
    const bool two = y != nullptr;
    for (int lev = 1; lev < nlev; ++lev) {
      const int h0 = lev_f[lev], h1 = lev_f[lev + 1];
      if (h1 == h0) continue;
      const int nh = h1 - h0, nrw = lev_fe[lev + 1] - lev_fe[lev];
      if (nh * 8 <= L && nrw > 2 * nh) {
        for (int hq = h0; hq < h1; ++hq) {
          const int q1 = foff[hq + 1];
          double acc = 0.0, acc2 = 0.0;
          for (int q = foff[hq] + me; q < q1; q += L) {
            const i32 a = fa[q], u0 = fu0[q];
            if (a >= 0) { const double l = vals[a]; acc += l * x[u0]; if (two) acc2 += l * y[u0]; }
            else {
              const i32 b = ~a, u1 = fu1[q];
              const double l0 = vals[b], l1 = vals[b + 1];
              acc += l0 * x[u0] + l1 * x[u1];
              if (two) acc2 += l0 * y[u0] + l1 * y[u1];
            }
          }
          acc = P::sum(acc);
          if (two) acc2 = P::sum(acc2);
          if (me == 0) { x[fnode[hq]] -= acc; if (two) y[fnode[hq]] -= acc2; }
        }
      } else {
        for (int hq = h0 + me; hq < h1; hq += L) {
          const int q1 = foff[hq + 1];
          double acc = 0.0, acc2 = 0.0;
          for (int q = foff[hq]; q < q1; ++q) {
            const i32 a = fa[q], u0 = fu0[q];
            if (a >= 0) { const double l = vals[a]; acc += l * x[u0]; if (two) acc2 += l * y[u0]; }
            else {
              const i32 b = ~a, u1 = fu1[q];
              const double l0 = vals[b], l1 = vals[b + 1];
              acc += l0 * x[u0] + l1 * x[u1];
              if (two) acc2 += l0 * y[u0] + l1 * y[u1];
            }
          }
          const i32 u = fnode[hq];
          x[u] -= acc;
          if (two) y[u] -= acc2;
        }
      }
      P::sync();
    }
    if (WK(tail_T) > 0) tail_forward(S, x, y);
    // D^-1, all blocks side by side (measured: folded into the backward pass — one level barrier less — the division
    // joins each block's dependent chain and the solve gets 8 % slower)
    for (int k = me; k < nblk; k += L) dsolve(vals, doff, bnode[2 * k], bnode[2 * k + 1], k, x, y);
    P::sync();
    if (WK(tail_T) > 0) tail_backward(S, x, y);
    for (int lev = nlev - 1; lev >= 0; --lev) {
      const int b0 = lev_off[lev], b1 = lev_off[lev + 1];
      const int nbl = b1 - b0, nrw = lev_r[lev + 1] - lev_r[lev];
      if (nrw == 0) continue;            // (the last block: nothing to gather)
      if (nbl * 4 >= L || nrw <= 3 * nbl * nbl) {
        for (int k = b0 + me; k < b1; k += L) {
          const int s0 = soff[k], sn = soff[k + 1] - s0;
          const i32 u0 = bnode[2 * k], u1 = bnode[2 * k + 1];
          const WD* Lk = vals + loff[k];
          double a0 = 0.0, a1 = 0.0, c0 = 0.0, c1 = 0.0;
          if (u1 < 0) {
            for (int i = 0; i < sn; ++i) { const i32 u = sidx[s0 + i]; const double l = Lk[i]; a0 += l * x[u]; if (two) c0 += l * y[u]; }
            x[u0] -= a0;
            if (two) y[u0] -= c0;
          } else {
            for (int i = 0; i < sn; ++i) {
              const i32 u = sidx[s0 + i];
              const double l0 = Lk[2 * i], l1 = Lk[2 * i + 1], xi = x[u];
              a0 += l0 * xi; a1 += l1 * xi;
              if (two) { const double yi = y[u]; c0 += l0 * yi; c1 += l1 * yi; }
            }
            x[u0] -= a0; x[u1] -= a1;
            if (two) { y[u0] -= c0; y[u1] -= c1; }
          }
        }
      } else {
        for (int k = b0; k < b1; ++k) {
          const int s0 = soff[k], sn = soff[k + 1] - s0;
          const i32 u0 = bnode[2 * k], u1 = bnode[2 * k + 1];
          if (sn == 0) continue;
          const WD* Lk = vals + loff[k];
          double a0 = 0.0, a1 = 0.0, c0 = 0.0, c1 = 0.0;
          if (u1 < 0) {
            for (int i = me; i < sn; i += L) { const i32 u = sidx[s0 + i]; const double l = Lk[i]; a0 += l * x[u]; if (two) c0 += l * y[u]; }
            a0 = P::sum(a0);
            if (two) c0 = P::sum(c0);
            if (me == 0) { x[u0] -= a0; if (two) y[u0] -= c0; }
          } else {
            for (int i = me; i < sn; i += L) {
              const i32 u = sidx[s0 + i];
              const double l0 = Lk[2 * i], l1 = Lk[2 * i + 1], xi = x[u];
              a0 += l0 * xi; a1 += l1 * xi;
              if (two) { const double yi = y[u]; c0 += l0 * yi; c1 += l1 * yi; }
            }
            a0 = P::sum(a0);
            a1 = P::sum(a1);
            if (two) { c0 = P::sum(c0); c1 = P::sum(c1); }
            if (me == 0) { x[u0] -= a0; x[u1] -= a1; if (two) { y[u0] -= c0; y[u1] -= c1; } }
          }
        }
      }
      P::sync();
    }
    W_P1(7);
  }
  // DenseKkt::solve (sparse) == Ipm::kkt_solve without the quasi-Newton part
  DNLP_HD static void kkt_solve(WS* S, const WD* r, WD* out, const WD* r2 = nullptr, WD* out2 = nullptr) {
    const int n = WK(N) + WK(m);
    { W_P0();
      if (out != r) W_FOR(k, n) out[k] = r[k];
      if (out2 && out2 != r2) W_FOR(k, n) out2[k] = r2[k];
      P::sync();
      W_P1(22); }
    ldl_solve(S, out, out2);
  }

  // ====================================================================================================================
  // interior-point loop (ipm_core.h; the member restated is named at each function)
  // ====================================================================================================================
  DNLP_HD static void filter_add(WS* S, double th, double ph) {
    if (S->nfilt == kWaveFilterCap) {
      for (int k = 1; k < S->nfilt; ++k) { S->filt_th[k - 1] = S->filt_th[k]; S->filt_ph[k - 1] = S->filt_ph[k]; }
      --S->nfilt;
    }
    S->filt_th[S->nfilt] = th; S->filt_ph[S->nfilt] = ph; ++S->nfilt;
  }
  DNLP_HD static bool filter_ok(WS* S, double th, double ph) {
    const double gth = 1e-5, gph = 1e-8;
    for (int k = 0; k < S->nfilt; ++k)
      if (!(th <= (1.0 - gth) * S->filt_th[k] || ph <= S->filt_ph[k] - gph * S->filt_th[k])) return false;
    return true;
  }

  // Ipm::begin
  DNLP_WFN DNLP_HD static int begin(WS* S) {
    W_P0();
    const double t_start = P::now();
    const int N = WK(N), m = WK(m);
    const IpmOptions& opt = S->opt;
    const double inf = opt.nlp_inf, brf = opt.bound_relax_factor;
    const bool warm = opt.warm_start != 0 && S->ws_g != nullptr && S->ws_l != nullptr && S->ws_u != nullptr;
    const double k1 = warm ? opt.warm_start_bound_push : opt.bound_push, k2 = warm ? opt.warm_start_bound_frac : opt.bound_frac;
    WG *lb = S->row + WK(l_lb), *ub = S->row + WK(l_ub), *cl = S->row + WK(l_cl), *cu = S->row + WK(l_cu), *x0 = S->row + WK(l_x0);
    WD *l = WV(xL), *u = WV(xU), *sl = WV(sL), *su = WV(sU), *sgp = WV(sg), *xx = WV(x), *px = WV(xt);
    {
      double bad = -kInf;
      W_FOR(j, N) {
        double a = lb[j], b = ub[j];
        if (brf > 0) {
          if (a > -inf) a -= fmin(brf * fmax(1.0, fabs(a)), 1e-3);
          if (b < inf) b += fmin(brf * fmax(1.0, fabs(b)), 1e-3);
        }
        const double lj = a <= -inf ? -kInf : a, uj = b >= inf ? kInf : b;
        l[j] = lj; u[j] = uj;
        bad = mxin(bad, lj > uj ? 1.0 : 0.0);
        xx[j] = x0[j];
      }
      W_FOR(i, m) {
        const double a = cl[i] <= -inf ? -kInf : cl[i], b = cu[i] >= inf ? kInf : cu[i];
        sl[i] = a; su[i] = b;
        bad = mxin(bad, a > b ? 1.0 : 0.0);
        sgp[i] = 1.0;
      }
      bad = P::vmax(bad);
      P::sync();
      if (bad > 0.0) return S->status = Invalid_Option;
    }
    S->sf = 1.0;
    if (opt.nlp_scaling) {
      W_FOR(j, N) px[j] = push_into_bounds1(xx[j], l[j], u[j], k1, k2);
      P::sync();
      sweep(S, WV(xt), false);
      spmv(S, WCSR(Mg), WV(dvals), WK(l_c), WV(grad), 0, 0.0);
      spmv(S, WCSR(MJ), WV(dvals), WK(l_Jc), WV(jv), 0, 0.0);
      const WD* gr = WV(grad);
      double gmax = -kInf;
      W_FOR(j, N) gmax = mxin(gmax, fabs(gr[j]));
      gmax = P::vmax(gmax);
      const double smax = opt.nlp_scaling_max_gradient;
      if (std::isfinite(gmax) && gmax > smax) S->sf = std::max(smax / gmax, 1e-8);
      if (m > 0) {
        WI* rp_ = WT(jac_rowptr);
        const WD* jvv = WV(jv);
        W_FOR(i, m) {
          double rmax = 0.0;
          bool fin = true;
          const i32 pe = rp_[i + 1];
          for (i32 p = rp_[i]; p < pe; ++p) { const double a = fabs(jvv[p]); if (!(a <= kInf) || a == kInf) fin = false; if (a > rmax) rmax = a; }
          sgp[i] = (fin && rmax > smax) ? fmax(smax / rmax, 1e-8) : 1.0;
        }
        P::sync();
      }
    }
    // scaled constraint bounds, equality mask, counts; start point pushed into the bounds, fixed variables pinned
    {
      WD *eq = WV(eq), *fm = WV(fixm);
      double neq = 0.0, nfree = 0.0;
      W_FOR(i, m) {
        const double e = (sl[i] == su[i]) ? 1.0 : 0.0;
        eq[i] = e;
        sl[i] *= sgp[i];
        su[i] *= sgp[i];
        neq += e;
      }
      W_FOR(j, N) nfree += l[j] == u[j] ? 0.0 : 1.0;
      const i64 n_eq = m ? static_cast<i64>(P::sum(neq)) : 0;
      const i64 n_free = static_cast<i64>(P::sum(nfree));
      if (n_eq > n_free) return S->status = Not_Enough_Degrees_Of_Freedom;
      S->n_eq = static_cast<i32>(n_eq);
      S->n_fixed = static_cast<i32>(N - n_free);
      W_FOR(j, N) {
        const bool fx = l[j] == u[j];
        xx[j] = fx ? l[j] : push_into_bounds1(xx[j], l[j], u[j], k1, k2);
        fm[j] = fx ? 1.0 : 0.0;
        if (fx) { l[j] = -kInf; u[j] = kInf; }
      }
      P::sync();
    }
    S->nb_cache = -1;
    if (!eval_fg(S, WV(x), S->f, WV(g)) || nan_check(S, WV(g)) != 0.0) return S->status = Invalid_Number_Detected;
    eval_derivs(S);
    {
      WD *ss = WV(s), *a = WV(zL), *b = WV(zU), *c = WV(vL), *d = WV(vU), *yy = WV(y);
      const WD *gg = WV(g), *eq = WV(eq);
      const double zi = opt.bound_mult_init_val;
      W_FOR(i, m) {
        ss[i] = eq[i] != 0.0 ? sl[i] : push_into_bounds1(gg[i], sl[i], su[i], k1, k2);
        c[i] = (eq[i] == 0.0 && sl[i] > -kInf) ? zi : 0.0;
        d[i] = (eq[i] == 0.0 && su[i] < kInf) ? zi : 0.0;
        yy[i] = 0.0;
      }
      W_FOR(j, N) { a[j] = (l[j] > -kInf) ? zi : 0.0; b[j] = (u[j] < kInf) ? zi : 0.0; }
      P::sync();
    }
    S->mu = opt.mu_init;
    S->tau = std::max(0.99, 1.0 - S->mu);
    S->jty_valid = false;
    if (warm) {
      const double sff = S->sf, mp = opt.warm_start_mult_bound_push;
      const double *wy = S->ws_g, *wl = S->ws_l, *wu = S->ws_u;
      const WD* eq = WV(eq);
      WD *yy = WV(y), *a = WV(zL), *b = WV(zU), *c = WV(vL), *d = WV(vU);
      W_FOR(i, m) {
        const double yi = wy[i] * sff / sgp[i];
        yy[i] = yi;
        const bool in = eq[i] == 0.0;
        c[i] = (in && sl[i] > -kInf) ? fmax(-yi, mp) : 0.0;
        d[i] = (in && su[i] < kInf) ? fmax(yi, mp) : 0.0;
      }
      W_FOR(j, N) {
        a[j] = (l[j] > -kInf) ? fmax(wl[j] * sff, mp) : 0.0;
        b[j] = (u[j] < kInf) ? fmax(wu[j] * sff, mp) : 0.0;
      }
      P::sync();
    } else if (m > 0 && opt.least_square_init_duals >= 0) {
      init_multipliers_ls(S);
    }
    S->nfilt = 0;
    const double th0 = theta_at(S, WV(g), WV(s));
    S->theta_max = 1e4 * std::max(1.0, th0);
    S->theta_min = 1e-4 * std::max(1.0, th0);
    S->iter = 0;
    S->acceptable_count = 0;
    S->delta_w_last = 0.0;
    S->dc_fixed_count = 0; S->dc_fixed_last = false; S->always_dc = false;
    S->resto_stationary = false; S->resto_theta = 0.0;
    S->tiny_streak = 0;
    S->e_cached_valid = false;
    S->fixed_mode = false;
    S->n_hist = 0;
    S->initialized = true;
    S->status = Internal_Error;
    S->factorizations = 0;            // (stats = IpmStats())
    S->inf_pr = S->inf_du = S->cmpl = S->nlp_error = 0.0;
    S->t_begin = t_start;
    W_P1(1);
    return 0;
  }

  // Ipm::init_multipliers_ls (the general path: neither of the host-only condensed forms)
  DNLP_WFN DNLP_HD static void init_multipliers_ls(WS* S) {
    const int N = WK(N), m = WK(m);
    S->jty_valid = false;
    const WD* eq = WV(eq);
    WD *sx = WV(Sx), *dd = WV(Dd);
    W_FOR(j, N) sx[j] = 1.0;
    W_FOR(i, m) dd[i] = (eq[i] == 0.0) ? 1.0 : 0.0;
    {
      WD* hs = WV(Hs);                      // (ex_->zero(md_->Hs): the next eval_hessian refills it)
      W_FOR(p, WK(nnzH)) hs[p] = 0.0;
    }
    P::sync();
    int nneg = 0, nzero = 0;
    bool ok = assemble_factor(S, WV(Sx), WV(Dd), 0.0, true, &nneg, &nzero);
    if (!ok || nzero > 0 || nneg != m) {
      W_FOR(i, m) dd[i] += 1e-8;
      P::sync();
      ok = assemble_factor(S, WV(Sx), WV(Dd), 0.0, true, &nneg, &nzero);
      if (!ok) return;
    }
    WD* r = WV(rhs);
    const WD *gr = WV(grad), *a = WV(zL), *b = WV(zU), *c = WV(vL), *d = WV(vU);
    W_FOR(j, N) r[j] = -(gr[j] - a[j] + b[j]);
    W_FOR(i, m) r[N + i] = (eq[i] == 0.0) ? -(-c[i] + d[i]) : 0.0;
    P::sync();
    kkt_solve(S, WV(rhs), WV(sol));
    const WD* so = WV(sol);
    double ymax = -kInf;
    W_FOR(i, m) ymax = mxin(ymax, fabs(so[N + i]));
    ymax = P::vmax(ymax);
    if (std::isfinite(ymax) && ymax <= S->opt.constr_mult_init_max) {
      WD* yy = WV(y);
      W_FOR(i, m) yy[i] = so[N + i];
      P::sync();
    }
  }

  // Ipm::theta_at
  DNLP_HD static double theta_at(WS* S, const WD* gg, const WD* ss) {
    const WD *eq = WV(eq), *sl = WV(sL);
    double acc = 0.0;
    W_FOR(i, WK(m)) acc += fabs(eq[i] != 0.0 ? gg[i] - sl[i] : gg[i] - ss[i]);
    return P::sum(acc);
  }
  DNLP_HD static double bterm(double v, double lo, double hi, double kd) {
    double bt = 0.0;
    const bool hl = lo > -kInf, hu = hi < kInf;
    if (hl) bt -= log(v - lo);
    if (hu) bt -= log(hi - v);
    if (hl && !hu) bt += kd * (v - lo);
    if (hu && !hl) bt += kd * (hi - v);
    return bt;
  }
  // Ipm::barrier_at (two sums: variables, rows)
  DNLP_WFN DNLP_HD static double barrier_at(WS* S, double fv, const WD* xx, const WD* ss, double muv) {
    const WD *l = WV(xL), *u = WV(xU), *sl = WV(sL), *su = WV(sU), *eq = WV(eq);
    const double kd = S->opt.kappa_d;
    double ax = 0.0, as = 0.0;
    W_FOR(j, WK(N)) ax += bterm(xx[j], l[j], u[j], kd);
    W_FOR(i, WK(m)) as += eq[i] != 0.0 ? 0.0 : bterm(ss[i], sl[i], su[i], kd);
    double r2[2] = {ax, as};
    P::sum_n(r2);
    const double bx = r2[0], bs = r2[1];
    return fv + muv * (bx + bs);
  }
  // Ipm::measures: theta, the barrier function and the NaN detector of g in one pass
  DNLP_WFN DNLP_HD static WMeasures measures(WS* S, double fv, const WD* gg, const WD* xx, const WD* ss, double muv) {
    const WD *l = WV(xL), *u = WV(xU), *sl = WV(sL), *su = WV(sU), *eq = WV(eq);
    const double kd = S->opt.kappa_d;
    W_P0();
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    W_FOR(j, WK(N)) s1 += bterm(xx[j], l[j], u[j], kd);
    W_FOR(i, WK(m)) {
      s0 += fabs(eq[i] != 0.0 ? gg[i] - sl[i] : gg[i] - ss[i]);
      s2 += gg[i] - gg[i];
      if (eq[i] == 0.0) s1 += bterm(ss[i], sl[i], su[i], kd);
    }
    { double r3[3] = {s0, s1, s2}; P::sum_n(r3); s0 = r3[0]; s1 = r3[1]; s2 = r3[2]; }
    W_P1(12);
    return WMeasures{s0, fv + muv * s1, s2};
  }
  // Ipm::jty
  DNLP_HD static const WD* jty(WS* S) {
    if (!S->jty_valid) { jac_tmult(S, WV(y), WV(tN)); S->jty_valid = true; }
    return WV(tN);
  }
  // Ipm::error (dual_residuals fused into the same pass)
  DNLP_WFN DNLP_HD static WErr error(WS* S, double muv) {
    const int N = WK(N), m = WK(m);
    const WD* jt = jty(S);
    W_P0();
    WD *r = WV(rx), *q = WV(rs);
    const WD *gr = WV(grad), *a = WV(zL), *b = WV(zU), *c = WV(vL), *d = WV(vU), *yy = WV(y), *eq = WV(eq), *fm = WV(fixm), *gg = WV(g), *ss = WV(s),
             *sl = WV(sL), *su = WV(sU), *l = WV(xL), *u = WV(xU), *xx = WV(x), *sgp = WV(sg);
    double m0 = -kInf, m1 = -kInf, m2 = -kInf, m3 = -kInf, sy = 0.0, sz = 0.0;
    // (operands first, selects instead of branches: see quality())
    W_FOR(j, N) {
      const double fmj = fm[j], aj = a[j], bj = b[j], lj = l[j], uj = u[j], xj = xx[j];
      const double rfree = gr[j] + jt[j] - aj + bj;
      const double rj = fmj != 0.0 ? 0.0 : rfree;
      r[j] = rj;
      m0 = mxin(m0, fabs(rj));
      const double cl_ = fabs((xj - lj) * aj - muv), cu_ = fabs((uj - xj) * bj - muv);
      double cv = 0.0;
      cv = lj > -kInf ? fmax(cv, cl_) : cv;
      cv = uj < kInf ? fmax(cv, cu_) : cv;
      m2 = mxin(m2, cv);
      sz += fabs(aj) + fabs(bj);
    }
    W_FOR(i, m) {
      const double eqi = eq[i], yi = yy[i], ci = c[i], di = d[i], gi = gg[i], li = sl[i], ui = su[i], si = ss[i], sgi = sgp[i];
      const bool in = eqi == 0.0;
      const double qfree = -yi - ci + di;
      const double qi = in ? qfree : 0.0;
      q[i] = qi;
      m0 = mxin(m0, fabs(qi));
      const double pr = fabs(eqi != 0.0 ? gi - li : gi - si);
      m1 = mxin(m1, pr);
      m3 = mxin(m3, pr / sgi);
      const double cl_ = fabs((si - li) * ci - muv), cu_ = fabs((ui - si) * di - muv);
      double cv = 0.0;
      cv = (in && li > -kInf) ? fmax(cv, cl_) : cv;
      cv = (in && ui < kInf) ? fmax(cv, cu_) : cv;
      m2 = mxin(m2, cv);
      sy += fabs(yi) + fabs(ci) + fabs(di);
    }
    { double r4[4] = {m0, m1, m2, m3}; P::vmax_n(r4); m0 = r4[0]; m1 = r4[1]; m2 = r4[2]; m3 = r4[3]; }
    { double r2[2] = {sy, sz}; P::sum_n(r2); sy = r2[0]; sz = r2[1]; }
    P::sync();
    WErr e;
    e.dual = std::max(m0, 0.0);
    e.primal = m ? std::max(m1, 0.0) : 0.0;
    e.cmpl = std::max(m2, 0.0);
    e.primal_unscaled = m ? std::max(m3, 0.0) : 0.0;
    const double smax = 100.0;
    const i64 nb = n_bound_mults(S);
    e.sd = std::max(smax, (sy + sz) / std::max<double>(1.0, static_cast<double>(m + nb))) / smax;
    e.sc = std::max(smax, sz / std::max<double>(1.0, static_cast<double>(nb))) / smax;
    e.total = std::max(std::max(e.dual / e.sd, e.primal), e.cmpl / e.sc);
    W_P1(2);
    return e;
  }
  // Ipm::n_bound_mults
  DNLP_HD static i64 n_bound_mults(WS* S) {
    if (S->nb_cache >= 0) return S->nb_cache;
    const WD *l = WV(xL), *u = WV(xU), *sl = WV(sL), *su = WV(sU), *eq = WV(eq);
    double c1 = 0.0, c2 = 0.0;
    W_FOR(j, WK(N)) c1 += (l[j] > -kInf ? 1.0 : 0.0) + (u[j] < kInf ? 1.0 : 0.0);
    W_FOR(i, WK(m)) c2 += eq[i] != 0.0 ? 0.0 : (sl[i] > -kInf ? 1.0 : 0.0) + (su[i] < kInf ? 1.0 : 0.0);
    c1 = P::sum(c1);
    c2 = WK(m) ? P::sum(c2) : 0.0;
    S->nb_cache = static_cast<i32>(c1 + c2);
    return S->nb_cache;
  }
  // Ipm::barrier_terms
  DNLP_WFN DNLP_HD static void barrier_terms(WS* S, double muv) {
    W_P0();
    const WD* jt = jty(S);
    WD *sx = WV(Sx), *sS = WV(Ss), *r = WV(rx), *q = WV(rs), *p = WV(rp);
    const WD *l = WV(xL), *u = WV(xU), *sl = WV(sL), *su = WV(sU), *eq = WV(eq), *xx = WV(x), *ss = WV(s), *a = WV(zL), *b = WV(zU), *c = WV(vL),
             *d = WV(vU), *gr = WV(grad), *yy = WV(y), *gg = WV(g), *fm = WV(fixm);
    const double kd = S->opt.kappa_d;
    if (P::hoist) {        // (operands first, selects instead of branches: see quality())
      W_FOR(j, WK(N)) {
        const double lj = l[j], uj = u[j], xj = xx[j], aj = a[j], bj = b[j], fmj = fm[j], jtj = jt[j];
        double sig = 0.0, gphi = gr[j];
        const bool hl = lj > -kInf, hu = uj < kInf;
        const double dl = xj - lj, du = uj - xj;
        const double sl_ = aj / dl, ml_ = muv / dl, su_ = bj / du, mu_ = muv / du;
        sig = hl ? sig + sl_ : sig; gphi = hl ? gphi - ml_ : gphi;
        sig = hu ? sig + su_ : sig; gphi = hu ? gphi + mu_ : gphi;
        gphi = (hl && !hu) ? gphi + kd * muv : gphi;
        gphi = (hu && !hl) ? gphi - kd * muv : gphi;
        sx[j] = sig;
        r[j] = fmj != 0.0 ? 0.0 : gphi + jtj;
      }
      W_FOR(i, WK(m)) {
        const double li = sl[i], ui = su[i], si = ss[i], ci = c[i], di = d[i], eqi = eq[i], yi = yy[i], gi = gg[i];
        const bool in = eqi == 0.0;
        double sig = 0.0, gphi = 0.0;
        const bool hl = li > -kInf, hu = ui < kInf;
        const double dl = si - li, du = ui - si;
        const double sl_ = ci / dl, ml_ = muv / dl, su_ = di / du, mu_ = muv / du;
        sig = hl ? sig + sl_ : sig; gphi = hl ? gphi - ml_ : gphi;
        sig = hu ? sig + su_ : sig; gphi = hu ? gphi + mu_ : gphi;
        gphi = (hl && !hu) ? gphi + kd * muv : gphi;
        gphi = (hu && !hl) ? gphi - kd * muv : gphi;
        sS[i] = in ? sig : 0.0;
        q[i] = in ? gphi - yi : 0.0;
        p[i] = in ? gi - si : gi - li;
      }
    } else {
      W_FOR(j, WK(N)) {
        double sig = 0.0, gphi = gr[j];
        const bool hl = l[j] > -kInf, hu = u[j] < kInf;
        if (hl) { sig += a[j] / (xx[j] - l[j]); gphi -= muv / (xx[j] - l[j]); }
        if (hu) { sig += b[j] / (u[j] - xx[j]); gphi += muv / (u[j] - xx[j]); }
        if (hl && !hu) gphi += kd * muv;
        if (hu && !hl) gphi -= kd * muv;
        sx[j] = sig;
        r[j] = fm[j] != 0.0 ? 0.0 : gphi + jt[j];
      }
      W_FOR(i, WK(m)) {
        if (eq[i] != 0.0) { sS[i] = 0.0; q[i] = 0.0; p[i] = gg[i] - sl[i]; continue; }
        double sig = 0.0, gphi = 0.0;
        const bool hl = sl[i] > -kInf, hu = su[i] < kInf;
        if (hl) { sig += c[i] / (ss[i] - sl[i]); gphi -= muv / (ss[i] - sl[i]); }
        if (hu) { sig += d[i] / (su[i] - ss[i]); gphi += muv / (su[i] - ss[i]); }
        if (hl && !hu) gphi += kd * muv;
        if (hu && !hl) gphi -= kd * muv;
        sS[i] = sig;
        q[i] = gphi - yy[i];
        p[i] = gg[i] - ss[i];
      }
    }
    W_P1(4);
    P::sync();
  }
  // Ipm::try_factor: 0 ok, 1 wrong inertia, 2 singular
  DNLP_HD static int try_factor(WS* S, double dw, double dc) {
    int nneg = 0, nzero = 0;
    WD* dd = WV(Dd);
    const WD *sS = WV(Ss), *eq = WV(eq);
    W_FOR(i, WK(m)) dd[i] = dc + (eq[i] == 0.0 ? 1.0 / fmax(sS[i] + dw, 1e-20) : 0.0);
    P::sync();
    const bool ok = assemble_factor(S, WV(Sx), WV(Dd), dw, false, &nneg, &nzero);
    S->last_nneg = nneg;
    if (!ok) return 2;
    if (nzero > 0) return 2;
    return nneg == WK(m) ? 0 : 1;
  }
  // Ipm::factor_with_inertia (WB Algorithm IC; no Lanczos bound: host-driven large dense systems only)
  DNLP_HD static bool factor_with_inertia(WS* S, double& delta_w, double& delta_c) {
    const bool ok = factor_with_inertia_impl(S);
    delta_w = S->o_d[5]; delta_c = S->o_d[6];
    return ok;
  }
  DNLP_WFN DNLP_HD static bool factor_with_inertia_impl(WS* S) {
    auto& delta_w = S->o_d[5];
    auto& delta_c = S->o_d[6];
    const double dw_min = 1e-20, dw_0 = 1e-4, dw_max = S->opt.max_hessian_perturbation, dc_bar = 1e-8, kwp = 8.0, kwpb = 100.0, kwm = 1.0 / 3.0, kc = 0.25;
    delta_w = 0.0; delta_c = 0.0;
    const double dc_val = dc_bar * std::pow(S->mu, kc);
    int r = try_factor(S, 0.0, S->always_dc ? dc_val : 0.0);
    if (S->always_dc) delta_c = dc_val;
    const bool can_fallback = (WK(N) + WK(m)) <= S->fallback_max_n;
    if (r == 2 && !S->always_dc && can_fallback && (!S->opt.lazy_dense_fallback || S->ladder_rung > 0)) {
      ++S->sparse_singular_streak;
      if ((WK(N) + WK(m)) <= 512 || (S->iter >= 1 && S->sparse_singular_streak >= 2)) {
        S->bail = true;                    // the generic kernel's Bunch-Kaufman path takes this instance
        return false;
      }
    } else {
      S->sparse_singular_streak = 0;
    }
    if (r == 0) { S->delta_w_used_last_iter = false; if (!S->always_dc) S->dc_fixed_last = false; return true; }
    S->delta_w_used_last_iter = true;
    if (r == 2) delta_c = dc_val;
    delta_w = (S->delta_w_last == 0.0) ? dw_0 : std::max(dw_min, kwm * S->delta_w_last);
    if (delta_c == 0.0 ? (r == 2 || (r == 1 && S->dc_fixed_last)) : (r == 2 && !S->always_dc)) {
      const int r2 = try_factor(S, 0.0, dc_val);
      if (r2 == 0) { delta_c = dc_val; delta_w = 0.0; if (r == 1) dc_fixed(S); return true; }
      if (r == 1) S->dc_fixed_last = false;
    }
    const double dw_start = delta_w;
    int wrong_no_dc = 0, nneg_seen = -1;
    for (int k = 0; k < 100; ++k) {
      const int r2 = try_factor(S, delta_w, delta_c);
      if (r2 == 0) { S->delta_w_last = delta_w; if (!S->always_dc) S->dc_fixed_last = false; return true; }
      if (r2 == 2 && delta_c == 0.0) delta_c = dc_val;
      if (r2 == 1 && delta_c == 0.0) {
        if (S->last_nneg < 1000000 && (nneg_seen < 0 || S->last_nneg == nneg_seen)) ++wrong_no_dc; else wrong_no_dc = 0;
        nneg_seen = S->last_nneg < 1000000 ? S->last_nneg : -1;
        if (wrong_no_dc >= 3) {
          int r3 = try_factor(S, 0.0, dc_val);
          if (r3 == 0) { delta_c = dc_val; delta_w = 0.0; dc_fixed(S); return true; }
          r3 = try_factor(S, dw_start, dc_val);
          if (r3 == 0) { delta_c = dc_val; delta_w = S->delta_w_last = dw_start; dc_fixed(S); return true; }
          delta_c = dc_val;
        } else if (delta_w > 1e20) {
          delta_c = dc_val;
          delta_w = dw_start;
          continue;
        }
      }
      delta_w = (S->delta_w_last == 0.0) ? kwpb * delta_w : kwp * delta_w;
      if (delta_w > dw_max) return false;
    }
    return false;
  }
  DNLP_HD static void dc_fixed(WS* S) {
    S->dc_fixed_last = true;
    if (++S->dc_fixed_count >= 3 && !S->always_dc) S->always_dc = true;
  }
  // the long outputs of a product by output (more than kCooHeavy entries), by all lanes: out[g] for those only
  DNLP_HD static void coo_heavy(const WCoo ix, const WD* a, const WD* v, WD* out) {
    WI *ptr = ix.ptr, *ent = ix.ent, *src = ix.src;
    for (i32 hq = 0; hq < ix.nheavy; ++hq) {
      const i32 gq = ix.heavy[hq];
      const i32 p0 = ptr[gq], cnt = ptr[gq + 1] - p0;
      double sacc = 0.0;
      W_FOR(q, cnt) sacc += a[ent[p0 + q]] * v[src[p0 + q]];
      sacc = P::sum(sacc);
      if (P::lane() == 0) out[gq] = sacc;
    }
  }
  // one output of a product by output: its short segment walked here, or the value coo_heavy left in `pre`
  DNLP_HD static double coo_one(const WCoo& ix, const WD* a, const WD* v, const WD* pre, int gq) {
    const i32 p0 = ix.ptr[gq], p1 = ix.ptr[gq + 1];
    if (p1 - p0 > static_cast<i32>(kCooHeavy)) return pre[gq];
    double sacc = 0.0;
    for (i32 p = p0; p < p1; ++p) sacc += a[ix.ent[p]] * v[ix.src[p]];
    return sacc;
  }
  // Ipm::kkt_residual: out = rhsv - K v, max |out|, max |v|.  The three products (sym(H) v, J^T v_y, J v_x) and the
  // combination are ONE pass: the owner of an output walks its segments of the three indices itself (the sums and their
  // order are those of the separate products); only the long outputs are formed beforehand by all lanes.
  DNLP_HD static void kkt_residual(WS* S, const WD* v, double dw, const WD* rhsv, WD* out, double& en, double& sn) {
    kkt_residual_impl(S, v, dw, rhsv, out);
    en = S->o_d[1]; sn = S->o_d[2];
  }
  DNLP_WFN DNLP_HD static void kkt_residual_impl(WS* S, const WD* v, double dw, const WD* rhsv, WD* out) {
#ifdef DNLP_WAVE_GEN
    { W_P0(); double e1, s1, e2, s2; wgen::kkt_residual<P, false>(S, dw, v, rhsv, out, v, rhsv, out, e1, s1, e2, s2); S->o_d[1] = e1; S->o_d[2] = s1; W_P1(9); return; }
#endif
    auto& en = S->o_d[1];
    auto& sn = S->o_d[2];
    W_P0();
    const int N = WK(N), m = WK(m);
    const WCoo hs = WCOO(hs), jc = WCOO(jc), jr = WCOO(jr);
    const WD *Hs = WV(Hs), *jv = WV(jv);
    WD *preH = out, *preJt = WV(xt), *preJ = WV(tM);
    if (hs.nheavy | jc.nheavy | jr.nheavy) {
      coo_heavy(hs, Hs, v, preH);
      coo_heavy(jc, jv, v + N, preJt);
      coo_heavy(jr, jv, v, preJ);
      P::sync();
    }
    const WD *sx = WV(Sx), *dd = WV(Dd), *fm = WV(fixm);
    double m0 = -kInf, m1 = -kInf;
    W_FOR(k, N) {
      const double hv = coo_one(hs, Hs, v, preH, k), jt = coo_one(jc, jv, v + N, preJt, k);
      const double kv = fm[k] != 0.0 ? v[k] : hv + (sx[k] + dw) * v[k] + jt;
      const double r = rhsv[k] - kv;
      out[k] = r;
      m0 = mxin(m0, fabs(r)); m1 = mxin(m1, fabs(v[k]));
    }
    W_FOR(i, m) {
      const int k = N + i;
      const double kv = coo_one(jr, jv, v, preJ, i) - dd[i] * v[k];
      const double r = rhsv[k] - kv;
      out[k] = r;
      m0 = mxin(m0, fabs(r)); m1 = mxin(m1, fabs(v[k]));
    }
    { double r2[2] = {m0, m1}; P::vmax_n(r2); en = r2[0]; sn = r2[1]; }
    P::sync();
    W_P1(9);
  }
  // the same for two systems at once (mu oracle): one walk of the three indices; the second system's long outputs in dx / ds
  struct Res2 { double en, sn, en2, sn2; };
  DNLP_WFN DNLP_HD static Res2 kkt_residual2(WS* S, double dw, const WD* v, const WD* rhsv, WD* out, const WD* v2, const WD* rhsv2, WD* out2) {
#ifdef DNLP_WAVE_GEN
    { W_P0(); Res2 Rg; wgen::kkt_residual<P, true>(S, dw, v, rhsv, out, v2, rhsv2, out2, Rg.en, Rg.sn, Rg.en2, Rg.sn2); W_P1(9); return Rg; }
#endif
    W_P0();
    const int N = WK(N), m = WK(m);
    const WCoo hs = WCOO(hs), jc = WCOO(jc), jr = WCOO(jr);
    const WD *Hs = WV(Hs), *jv = WV(jv);
    WD *preH = out, *preJt = WV(xt), *preJ = WV(tM), *preH2 = out2, *preJt2 = WV(dx), *preJ2 = WV(ds);
    if (hs.nheavy | jc.nheavy | jr.nheavy) {
      coo_heavy(hs, Hs, v, preH); coo_heavy(jc, jv, v + N, preJt); coo_heavy(jr, jv, v, preJ);
      coo_heavy(hs, Hs, v2, preH2); coo_heavy(jc, jv, v2 + N, preJt2); coo_heavy(jr, jv, v2, preJ2);
      P::sync();
    }
    const WD *sx = WV(Sx), *dd = WV(Dd), *fm = WV(fixm);
    double m0 = -kInf, m1 = -kInf, n0 = -kInf, n1 = -kInf;
    W_FOR(k, N) {
      // both systems' segments in one walk: the entry and source indices are loaded once
      double hv, jt, hv2, jt2;
      {
        const i32 p0 = hs.ptr[k], p1 = hs.ptr[k + 1];
        if (p1 - p0 > static_cast<i32>(kCooHeavy)) { hv = preH[k]; hv2 = preH2[k]; }
        else { double a = 0.0, b = 0.0; for (i32 p = p0; p < p1; ++p) { const double c = Hs[hs.ent[p]]; const i32 u = hs.src[p]; a += c * v[u]; b += c * v2[u]; } hv = a; hv2 = b; }
      }
      {
        const i32 p0 = jc.ptr[k], p1 = jc.ptr[k + 1];
        if (p1 - p0 > static_cast<i32>(kCooHeavy)) { jt = preJt[k]; jt2 = preJt2[k]; }
        else { double a = 0.0, b = 0.0; for (i32 p = p0; p < p1; ++p) { const double c = jv[jc.ent[p]]; const i32 u = N + jc.src[p]; a += c * v[u]; b += c * v2[u]; } jt = a; jt2 = b; }
      }
      const bool fx = fm[k] != 0.0;
      const double sd = sx[k] + dw;
      const double kv = fx ? v[k] : hv + sd * v[k] + jt, kv2 = fx ? v2[k] : hv2 + sd * v2[k] + jt2;
      const double r = rhsv[k] - kv, r2 = rhsv2[k] - kv2;
      out[k] = r; out2[k] = r2;
      m0 = mxin(m0, fabs(r)); m1 = mxin(m1, fabs(v[k]));
      n0 = mxin(n0, fabs(r2)); n1 = mxin(n1, fabs(v2[k]));
    }
    W_FOR(i, m) {
      const int k = N + i;
      double jx, jx2;
      {
        const i32 p0 = jr.ptr[i], p1 = jr.ptr[i + 1];
        if (p1 - p0 > static_cast<i32>(kCooHeavy)) { jx = preJ[i]; jx2 = preJ2[i]; }
        else { double a = 0.0, b = 0.0; for (i32 p = p0; p < p1; ++p) { const double c = jv[jr.ent[p]]; const i32 u = jr.src[p]; a += c * v[u]; b += c * v2[u]; } jx = a; jx2 = b; }
      }
      const double kv = jx - dd[i] * v[k], kv2 = jx2 - dd[i] * v2[k];
      const double r = rhsv[k] - kv, r2 = rhsv2[k] - kv2;
      out[k] = r; out2[k] = r2;
      m0 = mxin(m0, fabs(r)); m1 = mxin(m1, fabs(v[k]));
      n0 = mxin(n0, fabs(r2)); n1 = mxin(n1, fabs(v2[k]));
    }
    Res2 R;
    { double r4[4] = {m0, m1, n0, n1}; P::vmax_n(r4); R.en = r4[0]; R.sn = r4[1]; R.en2 = r4[2]; R.sn2 = r4[3]; }
    P::sync();
    W_P1(9);
    return R;
  }
  // Ipm::solve_refined
  DNLP_WFN DNLP_HD static bool solve_refined(WS* S, double dw) {
    const int n = WK(N) + WK(m);
    kkt_solve(S, WV(rhs), WV(sol));
    const WD* rr = WV(rhs);
    double rn = -kInf;
    W_FOR(i, n) rn = mxin(rn, fabs(rr[i]));
    rn = P::vmax(rn);
    double best = kInf;
    bool fresh = false;
    for (int it = 0; it < S->opt.max_refine; ++it) {
      double en, sn;
      kkt_residual(S, WV(sol), dw, WV(rhs), WV(res), en, sn);
      const double ratio = en / (std::max(rn, 1e-300) + sn);
      if (!std::isfinite(en)) return false;
      S->last_ratio = std::isfinite(ratio) ? ratio : kInf;
      fresh = true;
      if (it >= S->opt.min_refine && ratio <= 1e-10) break;
      if (en >= best * 0.999 && it >= S->opt.min_refine) break;
      best = std::min(best, en);
      kkt_solve(S, WV(res), WV(cor));
      WD* sw = WV(sol);
      const WD* co = WV(cor);
      W_FOR(i, n) sw[i] += co[i];
      P::sync();
      fresh = false;
    }
    if (!fresh) {
      double en, sn;
      kkt_residual(S, WV(sol), dw, WV(rhs), WV(res), en, sn);
      S->last_ratio = en / (std::max(rn, 1e-300) + sn);
      if (!std::isfinite(S->last_ratio)) S->last_ratio = kInf;
    }
    return true;
  }
  // Ipm::solve_refined for the mu oracle's two systems side by side: K sol = rhs (affine scaling) and K sol2 = rhs2
  // (centering), each refined exactly as solve_refined refines it — the same iterates, the same stopping decisions — but
  // in joint solves and joint residual passes while both are still going.  0: both fine; 1 / 2: the first / second met a
  // non-finite residual (what makes solve_refined return false).  ratio / ratio2: their last_ratio_.
  DNLP_HD static int solve_refined2(WS* S, double dw, WD* rhs2, WD* sol2, WD* res2, double& ratio_out, double& ratio2_out) {
    const int rc = solve_refined2_impl(S, dw, rhs2, sol2, res2);
    ratio_out = S->o_d[3]; ratio2_out = S->o_d[4];
    return rc;
  }
  DNLP_WFN DNLP_HD static int solve_refined2_impl(WS* S, double dw, WD* rhs2, WD* sol2, WD* res2) {
    auto& ratio_out = S->o_d[3];
    auto& ratio2_out = S->o_d[4];
    const int n = WK(N) + WK(m);
    WD *rhs = WV(rhs), *sol = WV(sol), *res = WV(res), *cor = WV(cor);
    kkt_solve(S, rhs, sol, rhs2, sol2);
    double rn = -kInf, rn2 = -kInf;
    W_FOR(i, n) { rn = mxin(rn, fabs(rhs[i])); rn2 = mxin(rn2, fabs(rhs2[i])); }
    { double r2[2] = {rn, rn2}; P::vmax_n(r2); rn = r2[0]; rn2 = r2[1]; }
    double best = kInf, best2 = kInf, lr = 0.0, lr2 = 0.0;
    bool fresh = false, fresh2 = false, go = true, go2 = true;      // go: still inside its refinement loop
    const int max_refine = S->opt.max_refine, min_refine = S->opt.min_refine;
    for (int it = 0; it < max_refine && (go || go2); ++it) {
      double en = 0.0, sn = 0.0, en2 = 0.0, sn2 = 0.0;
      if (go && go2) { const Res2 R = kkt_residual2(S, dw, sol, rhs, res, sol2, rhs2, res2); en = R.en; sn = R.sn; en2 = R.en2; sn2 = R.sn2; }
      else if (go) kkt_residual(S, sol, dw, rhs, res, en, sn);
      else kkt_residual(S, sol2, dw, rhs2, res2, en2, sn2);
      if (go) {
        const double ratio = en / (std::max(rn, 1e-300) + sn);
        if (!std::isfinite(en)) return 1;
        lr = std::isfinite(ratio) ? ratio : kInf;
        fresh = true;
        if (it >= min_refine && ratio <= 1e-10) go = false;
        else if (en >= best * 0.999 && it >= min_refine) go = false;
        else best = std::min(best, en);
      }
      if (go2) {
        const double ratio = en2 / (std::max(rn2, 1e-300) + sn2);
        if (!std::isfinite(en2)) return 2;
        lr2 = std::isfinite(ratio) ? ratio : kInf;
        fresh2 = true;
        if (it >= min_refine && ratio <= 1e-10) go2 = false;
        else if (en2 >= best2 * 0.999 && it >= min_refine) go2 = false;
        else best2 = std::min(best2, en2);
      }
      // correction solves of the systems still going (the second one's correction in place in its residual array)
      if (go && go2) {
        kkt_solve(S, res, cor, res2, res2);
        W_FOR(i, n) { sol[i] += cor[i]; sol2[i] += res2[i]; }
        P::sync();
        fresh = fresh2 = false;
      } else if (go) {
        kkt_solve(S, res, cor);
        W_FOR(i, n) sol[i] += cor[i];
        P::sync();
        fresh = false;
      } else if (go2) {
        kkt_solve(S, res2, res2);
        W_FOR(i, n) sol2[i] += res2[i];
        P::sync();
        fresh2 = false;
      }
    }
    if (!fresh || !fresh2) {
      double en = 0.0, sn = 0.0, en2 = 0.0, sn2 = 0.0;
      if (!fresh && !fresh2) { const Res2 R = kkt_residual2(S, dw, sol, rhs, res, sol2, rhs2, res2); en = R.en; sn = R.sn; en2 = R.en2; sn2 = R.sn2; }
      else if (!fresh) kkt_residual(S, sol, dw, rhs, res, en, sn);
      else kkt_residual(S, sol2, dw, rhs2, res2, en2, sn2);
      if (!fresh) { lr = en / (std::max(rn, 1e-300) + sn); if (!std::isfinite(lr)) lr = kInf; }
      if (!fresh2) { lr2 = en2 / (std::max(rn2, 1e-300) + sn2); if (!std::isfinite(lr2)) lr2 = kInf; }
    }
    ratio_out = lr; ratio2_out = lr2;
    return 0;
  }
  // Ipm::compute_direction; `set` picks the seven output arrays (0: dx .. dvU, 1: affine-scaling, 2: centering);
  // pres == nullptr stands for the all-zero primal residual of the centering system
  DNLP_WFN DNLP_HD static bool compute_direction(WS* S, double muv, const WD* pres, double dw, bool centering, int set) {
    const int N = WK(N), m = WK(m);
    WD* r = WV(rhs);
    const WD *rxx = WV(rx), *q = WV(rs), *sS = WV(Ss), *eq = WV(eq);
    { W_P0();
    W_FOR(k, N) r[k] = -rxx[k];
    W_FOR(i, m) r[N + i] = -(pres ? pres[i] : 0.0) - (eq[i] == 0.0 ? q[i] / (sS[i] + dw) : 0.0);
    P::sync();
    W_P1(11); }
    { W_P0(); const bool oks = solve_refined(S, dw); W_P1(21); if (!oks) return false; }
    direction_outputs(S, WV(sol), WV(rs), muv, dw, centering, set);
    return true;
  }
  // the second half of Ipm::compute_direction: the seven arrays of a direction from the solution `so` of its system
  // (q: the slack residual the system was built with).  so / q may be the output arrays themselves (read before written).
  DNLP_HD static void direction_outputs(WS* S, const WD* so, const WD* q, double muv, double dw, bool centering, int set) {
    W_P0();
    const int N = WK(N), m = WK(m);
    const WD *sS = WV(Ss), *eq = WV(eq);
    WD *ddx = WDIR(set, 0), *dds = WDIR(set, 1), *ddy = WDIR(set, 2), *da = WDIR(set, 3), *db = WDIR(set, 4), *dc = WDIR(set, 5),
       *dd2 = WDIR(set, 6);
    const WD *l = WV(xL), *u = WV(xU), *sl = WV(sL), *su = WV(sU), *xx = WV(x), *ss = WV(s), *a = WV(zL), *b = WV(zU), *c = WV(vL), *d = WV(vU);
    const double keep = centering ? 0.0 : 1.0;
    if (P::hoist) {        // (operands first, selects instead of branches: see quality())
      W_FOR(j, N) {
        const double dxj = so[j], lj = l[j], uj = u[j], xj = xx[j], aj = a[j], bj = b[j];
        const double va = (muv - aj * dxj) / (xj - lj) - keep * aj, vb = (muv + bj * dxj) / (uj - xj) - keep * bj;
        ddx[j] = dxj;
        da[j] = (lj > -kInf) ? va : 0.0;
        db[j] = (uj < kInf) ? vb : 0.0;
      }
      W_FOR(i, m) {
        const double soi = so[N + i], qi = q[i], li = sl[i], ui = su[i], si = ss[i], ci = c[i], di = d[i];
        const bool in = eq[i] == 0.0;
        const double dfree = (soi - qi) / (sS[i] + dw);
        const double dsi = in ? dfree : 0.0;
        const double vc = (muv - ci * dsi) / (si - li) - keep * ci, vd = (muv + di * dsi) / (ui - si) - keep * di;
        ddy[i] = soi;
        dds[i] = dsi;
        dc[i] = (in && li > -kInf) ? vc : 0.0;
        dd2[i] = (in && ui < kInf) ? vd : 0.0;
      }
    } else {
      W_FOR(j, N) {
        const double dxj = so[j];
        ddx[j] = dxj;
        da[j] = (l[j] > -kInf) ? (muv - a[j] * dxj) / (xx[j] - l[j]) - keep * a[j] : 0.0;
        db[j] = (u[j] < kInf) ? (muv + b[j] * dxj) / (u[j] - xx[j]) - keep * b[j] : 0.0;
      }
      W_FOR(i, m) {
        const bool in = eq[i] == 0.0;
        const double soi = so[N + i], qi = q[i];
        const double dsi = in ? (soi - qi) / (sS[i] + dw) : 0.0;
        ddy[i] = soi;
        dds[i] = dsi;
        dc[i] = (in && sl[i] > -kInf) ? (muv - c[i] * dsi) / (ss[i] - sl[i]) - keep * c[i] : 0.0;
        dd2[i] = (in && su[i] < kInf) ? (muv + d[i] * dsi) / (su[i] - ss[i]) - keep * d[i] : 0.0;
      }
    }
    P::sync();
    W_P1(11);
  }
  // Ipm::max_step_primal
  DNLP_HD static double max_step_primal(WS* S, double tauv) {
    const WD *l = WV(xL), *u = WV(xU), *sl = WV(sL), *su = WV(sU), *xx = WV(x), *ss = WV(s), *ddx = WV(dx), *dds = WV(ds), *eq = WV(eq);
    double ax = -kInf, as = -kInf;
    W_FOR(j, WK(N)) {
      double a = 1.0;
      if (l[j] > -kInf && ddx[j] < 0.0) a = fmin(a, -tauv * (xx[j] - l[j]) / ddx[j]);
      if (u[j] < kInf && ddx[j] > 0.0) a = fmin(a, tauv * (u[j] - xx[j]) / ddx[j]);
      ax = mnin(ax, a);
    }
    W_FOR(i, WK(m)) {
      double a = 1.0;
      if (eq[i] == 0.0) {
        if (sl[i] > -kInf && dds[i] < 0.0) a = fmin(a, -tauv * (ss[i] - sl[i]) / dds[i]);
        if (su[i] < kInf && dds[i] > 0.0) a = fmin(a, tauv * (su[i] - ss[i]) / dds[i]);
      }
      as = mnin(as, a);
    }
    const double rx_ = -P::vmax(ax), rs_ = WK(m) ? -P::vmax(as) : 1.0;
    return std::min(1.0, std::min(rx_, rs_));
  }
  // Ipm::max_step_dual
  DNLP_HD static double max_step_dual(WS* S, double tauv) {
    const WD *a = WV(zL), *b = WV(zU), *c = WV(vL), *d = WV(vU), *da = WV(dzL), *db = WV(dzU), *dc = WV(dvL), *dd2 = WV(dvU);
    double az = -kInf, av = -kInf;
    W_FOR(j, WK(N)) {
      double t = 1.0;
      if (da[j] < 0.0) t = fmin(t, -tauv * a[j] / da[j]);
      if (db[j] < 0.0) t = fmin(t, -tauv * b[j] / db[j]);
      az = mnin(az, t);
    }
    W_FOR(i, WK(m)) {
      double t = 1.0;
      if (dc[i] < 0.0) t = fmin(t, -tauv * c[i] / dc[i]);
      if (dd2[i] < 0.0) t = fmin(t, -tauv * d[i] / dd2[i]);
      av = mnin(av, t);
    }
    const double rz = -P::vmax(az), rv = WK(m) ? -P::vmax(av) : 1.0;
    return std::min(1.0, std::min(rz, rv));
  }
  // Ipm::max_steps: both fraction-to-boundary step sizes in one pass
  DNLP_WFN DNLP_HD static D2 max_steps(WS* S, double tauv) {
    const WD *l = WV(xL), *u = WV(xU), *sl = WV(sL), *su = WV(sU), *xx = WV(x), *ss = WV(s), *ddx = WV(dx), *dds = WV(ds), *eq = WV(eq);
    const WD *a = WV(zL), *b = WV(zU), *c = WV(vL), *d = WV(vU), *da = WV(dzL), *db = WV(dzU), *dc = WV(dvL), *dd2 = WV(dvU);
    W_P0();
    double a0 = -kInf, a1 = -kInf;
    if (P::hoist) {        // (operands first, selects instead of branches: see quality())
      W_FOR(j, WK(N)) {
        const double lj = l[j], uj = u[j], xj = xx[j], dxx = ddx[j], aj = a[j], bj = b[j], daj = da[j], dbj = db[j];
        const bool lo = lj > -kInf && dxx < 0.0, up = uj < kInf && dxx > 0.0;
        const double qp = (lo ? -tauv * (xj - lj) : tauv * (uj - xj)) / dxx, qa = -tauv * aj / daj, qb = -tauv * bj / dbj;
        double tp = 1.0, td = 1.0;
        tp = (lo || up) ? fmin(tp, qp) : tp;
        td = daj < 0.0 ? fmin(td, qa) : td;
        td = dbj < 0.0 ? fmin(td, qb) : td;
        a0 = mnin(a0, tp); a1 = mnin(a1, td);
      }
      W_FOR(i, WK(m)) {
        const double li = sl[i], ui = su[i], si = ss[i], dss = dds[i], ci = c[i], di = d[i], dci = dc[i], ddi = dd2[i];
        const bool in = eq[i] == 0.0;
        const bool lo = in && li > -kInf && dss < 0.0, up = in && ui < kInf && dss > 0.0;
        const double qp = (lo ? -tauv * (si - li) : tauv * (ui - si)) / dss, qc = -tauv * ci / dci, qd = -tauv * di / ddi;
        double tp = 1.0, td = 1.0;
        tp = (lo || up) ? fmin(tp, qp) : tp;
        td = dci < 0.0 ? fmin(td, qc) : td;
        td = ddi < 0.0 ? fmin(td, qd) : td;
        a0 = mnin(a0, tp); a1 = mnin(a1, td);
      }
      { double r2[2] = {a0, a1}; P::vmax_n(r2); a0 = r2[0]; a1 = r2[1]; }
      W_P1(12);
    } else {
      W_FOR(j, WK(N)) {
        double tp = 1.0, td = 1.0;
        if (l[j] > -kInf && ddx[j] < 0.0) tp = fmin(tp, -tauv * (xx[j] - l[j]) / ddx[j]);
        if (u[j] < kInf && ddx[j] > 0.0) tp = fmin(tp, tauv * (u[j] - xx[j]) / ddx[j]);
        if (da[j] < 0.0) td = fmin(td, -tauv * a[j] / da[j]);
        if (db[j] < 0.0) td = fmin(td, -tauv * b[j] / db[j]);
        a0 = mnin(a0, tp); a1 = mnin(a1, td);
      }
      W_FOR(i, WK(m)) {
        double tp = 1.0, td = 1.0;
        if (eq[i] == 0.0) {
          if (sl[i] > -kInf && dds[i] < 0.0) tp = fmin(tp, -tauv * (ss[i] - sl[i]) / dds[i]);
          if (su[i] < kInf && dds[i] > 0.0) tp = fmin(tp, tauv * (su[i] - ss[i]) / dds[i]);
        }
        if (dc[i] < 0.0) td = fmin(td, -tauv * c[i] / dc[i]);
        if (dd2[i] < 0.0) td = fmin(td, -tauv * d[i] / dd2[i]);
        a0 = mnin(a0, tp); a1 = mnin(a1, td);
      }
      { double r2[2] = {a0, a1}; P::vmax_n(r2); a0 = r2[0]; a1 = r2[1]; }
      W_P1(12);
    }
    return D2{std::min(1.0, -a0), std::min(1.0, -a1)};
  }
  // Ipm::check_convergence
  DNLP_HD static int check_convergence(WS* S, const WErr& e0) {
    const IpmOptions& opt = S->opt;
    const double unsc_du = e0.dual / S->sf, unsc_pr = e0.primal_unscaled, unsc_co = e0.cmpl / S->sf;
    S->inf_pr = unsc_pr; S->inf_du = unsc_du; S->cmpl = unsc_co; S->nlp_error = e0.total;
    if (e0.total <= opt.tol && unsc_du <= opt.dual_inf_tol && unsc_pr <= opt.constr_viol_tol && unsc_co <= opt.compl_inf_tol)
      return Solve_Succeeded;
    const bool acc = e0.total <= opt.acceptable_tol && unsc_du <= opt.acceptable_dual_inf_tol &&
                     unsc_pr <= opt.acceptable_constr_viol_tol && unsc_co <= opt.acceptable_compl_inf_tol;
    S->acceptable_count = acc ? S->acceptable_count + 1 : 0;
    if (opt.acceptable_iter > 0 && S->acceptable_count >= opt.acceptable_iter) return Solved_To_Acceptable_Level;
    return 99;
  }
  DNLP_HD static WErr cached_err(WS* S) {
    return WErr{S->e_dual, S->e_primal, S->e_cmpl, S->e_sd, S->e_sc, S->e_total, S->e_primal_unscaled};
  }
  DNLP_HD static void cache_err(WS* S, const WErr& e) {
    S->e_dual = e.dual; S->e_primal = e.primal; S->e_cmpl = e.cmpl; S->e_sd = e.sd; S->e_sc = e.sc; S->e_total = e.total;
    S->e_primal_unscaled = e.primal_unscaled;
    S->e_cached_valid = true;
  }

  // Ipm::step
  // (inlined into its callers' loops: as a called function it saved and restored ~75 callee-saved VGPRs per iteration —
  //  150 scratch operations, 22 KB of HBM writes per iteration over the launch)
  DNLP_WINL DNLP_HD static int step(WS* S) {
    const int N = WK(N), m = WK(m);
    if (!S->initialized) return S->status = Internal_Error;
    const WErr e0 = S->e_cached_valid ? cached_err(S) : error(S, 0.0);
    S->e_cached_valid = false;
    const int cv = check_convergence(S, e0);
    if (cv != 99) return S->status = cv;
    if (S->iter >= S->opt.max_iter) return S->status = Maximum_Iterations_Exceeded;
    if (P::now() - S->t_begin > S->opt.max_wall_time) return S->status = Maximum_WallTime_Exceeded;
    {
      const WD* xx = WV(x);
      double xm = -kInf;
      W_FOR(j, N) xm = mxin(xm, fabs(xx[j]));
      xm = P::vmax(xm);
      if (!(xm <= S->opt.diverging_iterates_tol)) return S->status = Diverging_Iterates;
    }
    bool want_oracle;
    { W_P0(); want_oracle = update_mu(S, e0); W_P1(18); }
    eval_hessian(S);
    barrier_terms(S, S->mu);
    double dw = 0.0, dc = 0.0;
    { W_P0(); const bool okf = factor_with_inertia(S, dw, dc); W_P1(19); if (!okf) return S->status = Error_In_Step_Computation; }
    bool have_dir = false;
    if (want_oracle) { W_P0(); have_dir = quality_function_mu(S, dw); W_P1(20); }
    if (!have_dir) {
      barrier_terms(S, S->mu);
      if (!compute_direction(S, S->mu, WV(rp), dw, false, 0)) return S->status = Error_In_Step_Computation;
    }
    for (int tries = 0; tries < 6 && S->last_ratio > 1e-5; ++tries) {
      if (dc == 0.0) dc = 1e-8 * std::pow(S->mu, 0.25);
      dw = (dw == 0.0) ? ((S->delta_w_last == 0.0) ? 1e-4 : std::max(1e-20, S->delta_w_last / 3.0)) : 8.0 * dw;
      int r = try_factor(S, dw, dc);
      while (r != 0 && dw < S->opt.max_hessian_perturbation) { dw *= 8.0; r = try_factor(S, dw, dc); }
      if (r != 0) return S->status = Error_In_Step_Computation;
      S->delta_w_last = dw;
      barrier_terms(S, S->mu);
      if (!compute_direction(S, S->mu, WV(rp), dw, false, 0)) return S->status = Error_In_Step_Computation;
    }
    // ---- backtracking filter line search (WB Algorithm A, steps A-5) ----
    const double mu = S->mu, tau = S->tau;
    W_P0();
    const D2 steps = max_steps(S, tau);
    const double a_max = steps.first;
    double a_z = steps.second;
    const WMeasures mk = measures(S, S->f, WV(g), WV(x), WV(s), mu);
    const double theta_k = mk.theta, phi_k = mk.phi;
    double gphid;
    {
      const WD *rxx = WV(rx), *ddx = WV(dx), *q = WV(rs), *yy = WV(y), *dds = WV(ds), *eq = WV(eq);
      const WD* jt = jty(S);
      double acc = 0.0;
      W_FOR(k, N) acc += (rxx[k] - jt[k]) * ddx[k];
      W_FOR(i, m) acc += eq[i] == 0.0 ? (q[i] + yy[i]) * dds[i] : 0.0;
      gphid = P::sum(acc);
    }
    const double g_th = 1e-5, g_ph = 1e-8, dlt = 1.0, s_th = 1.1, s_ph = 2.3, eta = 1e-8, g_al = 0.05;
    double a_min;
    if (gphid < 0.0) {
      a_min = std::min(g_th, g_ph * theta_k / (-gphid));
      if (theta_k <= S->theta_min) a_min = std::min(a_min, dlt * std::pow(theta_k, s_th) / std::pow(-gphid, s_ph));
      a_min *= g_al;
    } else {
      a_min = g_al * g_th;
    }
    const double macheps = 2.220446049250313e-16;
    auto le = [&](double a, double b, double base) { return a - b <= 10.0 * macheps * std::fabs(base); };
    double alpha = a_max;
    bool accepted = false, ftype = false;
    int ls = 0;
    bool soc_tried = false;
    double th_t = 0.0, ph_t = 0.0, f_t = 0.0;
    while (true) {
      ++ls;
      bool fin;
      { W_P0();
      trial_point(S, alpha);
      fin = eval_fg(S, WV(xt), f_t, WV(gt));
      W_P1(13); }
      if (fin) {
        const WMeasures mt = measures(S, f_t, WV(gt), WV(xt), WV(st), mu);
        th_t = mt.theta;
        ph_t = mt.phi;
        fin = mt.chk == 0.0 && std::isfinite(th_t) && std::isfinite(ph_t);
      }
      if (fin && th_t <= S->theta_max && filter_ok(S, th_t, ph_t)) {
        const bool sw = gphid < 0.0 && alpha * std::pow(-gphid, s_ph) > dlt * std::pow(theta_k, s_th);
        if (theta_k <= S->theta_min && sw) {
          if (le(ph_t, phi_k + eta * alpha * gphid, phi_k)) { accepted = true; ftype = true; }
        } else {
          if (le(th_t, (1.0 - g_th) * theta_k, theta_k) || le(ph_t, phi_k - g_ph * theta_k, phi_k)) accepted = true;
        }
      }
      if (accepted) break;
      if (ls == 1 && !soc_tried && fin && th_t >= theta_k && S->opt.max_soc > 0 && m > 0) {
        soc_tried = true;
        if (second_order_correction(S, alpha, dw, theta_k, phi_k, gphid, th_t, ph_t, f_t, ftype)) {
          accepted = true;
          a_z = max_step_dual(S, tau);
          break;
        }
        compute_direction(S, mu, WV(rp), dw, false, 0);
      }
      alpha *= 0.5;
      if (alpha < a_min || ls > 60) break;
    }
    const double alpha_used = alpha;
    W_P1(23);
    if (!accepted) {
      if (S->opt.restoration && restoration_phase(S, theta_k)) {
        ++S->iter;
        (void)error(S, 0.0);             // (Ipm::step computes it for the log line; it also refreshes rx / rs)
        return 99;
      }
      if (S->bail) return S->status = Error_In_Step_Computation;
      const WD* ddx = WV(dx);
      double dn = -kInf;
      W_FOR(j, N) dn = mxin(dn, fabs(ddx[j]));
      dn = P::vmax(dn);
      if (dn < 1e-12 && theta_k < S->opt.constr_viol_tol) return S->status = Search_Direction_Becomes_Too_Small;
      return S->status = (S->resto_stationary && S->resto_theta > S->opt.constr_viol_tol) ? Infeasible_Problem_Detected : Restoration_Failed;
    }
    if (!ftype) filter_add(S, (1.0 - g_th) * theta_k, phi_k - g_ph * theta_k);
    accept_trial(S, alpha_used, a_z, f_t);
    ++S->iter;
    const bool guard_on = S->opt.stall_guard == 1 || (S->opt.stall_guard < 0 && S->in_solve && S->opt.adaptive_fallback && S->ladder_rung < 2);
    const double kStallAlpha = 1e-2;
    const int kStallSteps = 30;
    if (guard_on && alpha_used <= kStallAlpha) {
      if (S->tiny_streak == 0) { S->streak_theta0 = theta_k; S->streak_f0 = S->f; }
      ++S->tiny_streak;
      if (S->tiny_streak >= kStallSteps) {
        const bool progress = th_t < 0.9 * S->streak_theta0 ||
                              (th_t <= S->streak_theta0 && f_t < S->streak_f0 - 1e-2 * fmax(1.0, fabs(S->streak_f0)));
        if (progress) S->tiny_streak = 0;
      }
    } else {
      S->tiny_streak = 0;
    }
    if (S->tiny_streak >= kStallSteps) return S->status = Search_Direction_Becomes_Too_Small;
    const WErr e = error(S, 0.0);
    cache_err(S, e);
    return 99;
  }

  // Ipm::trial_point
  DNLP_HD static void trial_point(WS* S, double alpha) {
    WD *a = WV(xt), *b = WV(st);
    const WD *xx = WV(x), *ss = WV(s), *ddx = WV(dx), *dds = WV(ds), *eq = WV(eq), *sl = WV(sL);
    W_FOR(k, WK(N)) a[k] = xx[k] + alpha * ddx[k];
    W_FOR(i, WK(m)) b[i] = eq[i] != 0.0 ? sl[i] : ss[i] + alpha * dds[i];
    P::sync();
  }
  // Ipm::accept_trial (+ reset_bound_multipliers, WB eq. (16), after the derivatives as there)
  DNLP_WFN DNLP_HD static void accept_trial(WS* S, double alpha, double a_z, double f_new) {
    W_P0();
    const int N = WK(N), m = WK(m);
    {
      WD *xx = WV(x), *ss = WV(s), *yy = WV(y), *a = WV(zL), *b = WV(zU), *c = WV(vL), *d = WV(vU), *gg = WV(g);
      const WD *nx = WV(xt), *ns = WV(st), *ddy = WV(dy), *da = WV(dzL), *db = WV(dzU), *dc = WV(dvL), *dd2 = WV(dvU), *gn = WV(gt);
      W_FOR(j, N) { xx[j] = nx[j]; a[j] += a_z * da[j]; b[j] += a_z * db[j]; }
      W_FOR(i, m) { ss[i] = ns[i]; yy[i] += alpha * ddy[i]; c[i] += a_z * dc[i]; d[i] += a_z * dd2[i]; gg[i] = gn[i]; }
      P::sync();
    }
    S->f = f_new;
    // (Ipm::accept_trial sweeps again "in case a later trial was evaluated": the accepted point IS the last one evaluated
    //  on every path that gets here, and then z / dvals are already those of x — the same values, one sweep less)
    if (!S->swept_xt) sweep(S, WV(x), false);
    S->swept_xt = false;
    eval_derivs(S);
    reset_bound_multipliers(S);
    W_P1(14);
  }
  DNLP_HD static void reset_bound_multipliers(WS* S) {
    const double kS = 1e10, muv = S->mu;
    const WD *l = WV(xL), *u = WV(xU), *sl = WV(sL), *su = WV(sU), *xx = WV(x), *ss = WV(s), *eq = WV(eq);
    WD *a = WV(zL), *b = WV(zU), *c = WV(vL), *d = WV(vU);
    if (P::hoist) {        // (operands first, selects instead of branches: see quality())
      W_FOR(j, WK(N)) {
        const double lj = l[j], uj = u[j], xj = xx[j], aj = a[j], bj = b[j];
        const double tl = xj - lj, tu = uj - xj;
        const double na = fmax(fmin(aj, kS * muv / tl), muv / (kS * tl)), nb = fmax(fmin(bj, kS * muv / tu), muv / (kS * tu));
        a[j] = lj > -kInf ? na : aj;
        b[j] = uj < kInf ? nb : bj;
      }
      W_FOR(i, WK(m)) {
        const double li = sl[i], ui = su[i], si = ss[i], ci = c[i], di = d[i];
        const bool in = eq[i] == 0.0;
        const double tl = si - li, tu = ui - si;
        const double nc = fmax(fmin(ci, kS * muv / tl), muv / (kS * tl)), nd = fmax(fmin(di, kS * muv / tu), muv / (kS * tu));
        c[i] = (in && li > -kInf) ? nc : ci;
        d[i] = (in && ui < kInf) ? nd : di;
      }
    } else {
      W_FOR(j, WK(N)) {
        if (l[j] > -kInf) { const double t = xx[j] - l[j]; a[j] = fmax(fmin(a[j], kS * muv / t), muv / (kS * t)); }
        if (u[j] < kInf) { const double t = u[j] - xx[j]; b[j] = fmax(fmin(b[j], kS * muv / t), muv / (kS * t)); }
      }
      W_FOR(i, WK(m)) {
        if (eq[i] != 0.0) continue;
        if (sl[i] > -kInf) { const double t = ss[i] - sl[i]; c[i] = fmax(fmin(c[i], kS * muv / t), muv / (kS * t)); }
        if (su[i] < kInf) { const double t = su[i] - ss[i]; d[i] = fmax(fmin(d[i], kS * muv / t), muv / (kS * t)); }
      }
    }
    P::sync();
  }
  // Ipm::second_order_correction (WB section 2.4)
  DNLP_HD static bool second_order_correction(WS* S, double alpha, double dw, double theta_k, double phi_k, double gphid,
                                              double& th_t, double& ph_t, double& f_t, bool& ftype) {
    S->o_d[7] = th_t; S->o_d[8] = ph_t; S->o_d[9] = f_t; S->o_i[0] = ftype ? 1 : 0;
    const bool ok = second_order_correction_impl(S, alpha, dw, theta_k, phi_k, gphid);
    th_t = S->o_d[7]; ph_t = S->o_d[8]; f_t = S->o_d[9]; ftype = S->o_i[0] != 0;
    return ok;
  }
  DNLP_WFN DNLP_HD static bool second_order_correction_impl(WS* S, double alpha, double dw, double theta_k, double phi_k, double gphid) {
    auto& th_t = S->o_d[7];
    auto& ph_t = S->o_d[8];
    auto& f_t = S->o_d[9];
    auto& ftype_i = S->o_i[0];
    const double k_soc = 0.99, g_th = 1e-5, g_ph = 1e-8, dlt = 1.0, s_th = 1.1, s_ph = 2.3, eta = 1e-8;
    const double macheps = 2.220446049250313e-16;
    auto le = [&](double a, double b, double base) { return a - b <= 10.0 * macheps * std::fabs(base); };
    const int m = WK(m);
    WD* cs = WV(csoc);
    const WD *p = WV(rp), *eq = WV(eq), *sl = WV(sL);
    {
      const WD *gg = WV(gt), *ss = WV(st);
      W_FOR(i, m) cs[i] = alpha * p[i] + (eq[i] != 0.0 ? gg[i] - sl[i] : gg[i] - ss[i]);
      P::sync();
    }
    double th_old = th_t;
    for (int k = 0; k < S->opt.max_soc; ++k) {
      if (!compute_direction(S, S->mu, WV(csoc), dw, false, 0)) return false;
      const double a_soc = max_step_primal(S, S->tau);
      trial_point(S, a_soc);
      double fv;
      if (!eval_fg(S, WV(xt), fv, WV(gt)) || nan_check(S, WV(gt)) != 0.0) return false;
      const double th = theta_at(S, WV(gt), WV(st)), ph = barrier_at(S, fv, WV(xt), WV(st), S->mu);
      if (!std::isfinite(th) || !std::isfinite(ph)) return false;
      if (th <= S->theta_max && filter_ok(S, th, ph)) {
        const bool sw = gphid < 0.0 && alpha * std::pow(-gphid, s_ph) > dlt * std::pow(theta_k, s_th);
        bool ok = false;
        if (theta_k <= S->theta_min && sw) {
          if (le(ph, phi_k + eta * alpha * gphid, phi_k)) { ok = true; ftype_i = 1; }
        } else if (le(th, (1.0 - g_th) * theta_k, theta_k) || le(ph, phi_k - g_ph * theta_k, phi_k)) {
          ok = true;
        }
        if (ok) { th_t = th; ph_t = ph; f_t = fv; return true; }
      }
      if (th > k_soc * th_old) return false;
      th_old = th;
      const WD *gg2 = WV(gt), *ss2 = WV(st);
      W_FOR(i, m) cs[i] = a_soc * cs[i] + (eq[i] != 0.0 ? gg2[i] - sl[i] : gg2[i] - ss2[i]);
      P::sync();
    }
    return false;
  }

  // Ipm::avg_complementarity
  DNLP_HD static double avg_complementarity(WS* S) {
    const i64 nb = n_bound_mults(S);
    if (nb == 0) return 0.0;
    const WD *l = WV(xL), *u = WV(xU), *sl = WV(sL), *su = WV(sU), *xx = WV(x), *ss = WV(s), *a = WV(zL), *b = WV(zU), *c = WV(vL), *d = WV(vU), *eq = WV(eq);
    double acc = 0.0;
    W_FOR(j, WK(N)) {
      double v = 0.0;
      if (l[j] > -kInf) v += (xx[j] - l[j]) * a[j];
      if (u[j] < kInf) v += (u[j] - xx[j]) * b[j];
      acc += v;
    }
    W_FOR(i, WK(m)) {
      double v = 0.0;
      if (eq[i] == 0.0) {
        if (sl[i] > -kInf) v += (ss[i] - sl[i]) * c[i];
        if (su[i] < kInf) v += (su[i] - ss[i]) * d[i];
      }
      acc += v;
    }
    return P::sum(acc) / static_cast<double>(nb);
  }
  DNLP_HD static double mu_floor_now(WS* S) {
    const double tol = S->opt.tol, ctol = S->opt.compl_inf_tol, mu_min = S->opt.mu_min;
    const double t = std::min(tol, ctol);
    if (S->opt.mu_strategy == 0) return std::max(mu_min, t / 11.0);
    return std::min(mu_min, 0.5 * t);
  }
  // Ipm::monotone_update
  DNLP_WFN DNLP_HD static void monotone_update(WS* S) {
    const double k_eps = 10.0, k_mu = 0.2, th_mu = 1.5;
    const double mu_floor = mu_floor_now(S);
    for (int k = 0; k < 50; ++k) {
      const WErr e = error(S, S->mu);
      if (e.total <= k_eps * S->mu && S->mu > mu_floor) {
        const double nm = std::max(mu_floor, std::min(k_mu * S->mu, std::pow(S->mu, th_mu)));
        if (nm >= S->mu) break;
        S->mu = nm;
        S->tau = std::max(0.99, 1.0 - S->mu);
        S->nfilt = 0;
      } else {
        break;
      }
    }
  }
  DNLP_HD static void hist_push(WS* S, double v) {
    if (S->n_hist == 4) { for (int k = 1; k < 4; ++k) S->kkt_hist[k - 1] = S->kkt_hist[k]; --S->n_hist; }
    S->kkt_hist[S->n_hist++] = v;
  }
  // Ipm::update_mu
  DNLP_HD static bool update_mu(WS* S, const WErr& e0) {
    if (n_bound_mults(S) == 0) { S->tau = 0.99; return false; }
    if (S->opt.mu_strategy == 0) { monotone_update(S); return false; }
    const double mu_floor = mu_floor_now(S);
    const double kkt = e0.dual + e0.primal + e0.cmpl;
    if (!S->fixed_mode) {
      bool ok = S->n_hist == 0;
      for (int k = 0; k < S->n_hist; ++k) if (kkt <= 0.9999 * S->kkt_hist[k]) ok = true;
      if (ok) {
        hist_push(S, kkt);
      } else {
        S->fixed_mode = true;
        S->mu = std::max(mu_floor, std::min(0.8 * avg_complementarity(S), 1e5));
        S->tau = std::max(0.99, 1.0 - S->mu);
        S->nfilt = 0;
      }
    } else {
      bool ok = false;
      for (int k = 0; k < S->n_hist; ++k) if (kkt <= 0.9999 * S->kkt_hist[k]) ok = true;
      if (ok || S->n_hist == 0) {
        S->fixed_mode = false;
        hist_push(S, kkt);
      }
    }
    if (S->fixed_mode) { monotone_update(S); return false; }
    return true;
  }

  // one evaluation of the quality function (the lambda qf of Ipm::quality_function_mu)
  DNLP_WFN DNLP_HD static double quality(WS* S, double sigma, double avg, double nd2, double np2, double n_dual, double n_pri, i64 nb) {
    W_P0();
    const double mus = sigma * avg;
    const double tv = std::max(0.99, 1.0 - mus);
    const WD *ax = WDIR(1, 0), *as = WDIR(1, 1), *aa = WDIR(1, 3), *ab = WDIR(1, 4), *ac = WDIR(1, 5), *ad = WDIR(1, 6);
    const WD *cx = WDIR(2, 0), *cs = WDIR(2, 1), *ca = WDIR(2, 3), *cb = WDIR(2, 4), *cc = WDIR(2, 5), *cd = WDIR(2, 6);
    const WD *l = WV(xL), *u = WV(xU), *sl = WV(sL), *su = WV(sU), *eq = WV(eq), *xx = WV(x), *ss = WV(s), *a = WV(zL), *b = WV(zU), *c = WV(vL), *d = WV(vU);
    const int N = WK(N), m = WK(m);
    // Two forms of the same values.  P::hoist (a policy whose vectors live in GLOBAL memory: the workgroup-per-instance
    // kernel): every operand is loaded before the first use and every "if" is a select, so that the loads of a lane leave
    // together and a trip waits for memory ONCE — the compiler does not move a load across a branch, and a branch per
    // bound is a chain of a dozen memory round trips per evaluation (12 evaluations per iteration: 240 k of 1 860 k cycles
    // per iteration of path planning); a select picks what the branch would have computed, the two primal bounds of an
    // entry share one division (d < 0 can only meet the lower bound, d > 0 only the upper).  With the vectors in LDS the
    // branches win: a round trip is short, a branch skips its division for the whole wavefront where no lane has the
    // bound (measured on localization: 9.4 k cycles per iteration with branches, 11.5 k with selects).
    double a0 = -kInf, a1 = -kInf;
    double comp = 0.0;
    if (P::hoist) {
      W_FOR(j, N) {
        const double lj = l[j], uj = u[j], xj = xx[j], aj = a[j], bj = b[j];
        const double dxx = ax[j] + mus * cx[j], da = aa[j] + mus * ca[j], db = ab[j] + mus * cb[j];
        const bool lo = lj > -kInf && dxx < 0.0, up = uj < kInf && dxx > 0.0;
        const double qp = (lo ? -tv * (xj - lj) : tv * (uj - xj)) / dxx, qa = -tv * aj / da, qb = -tv * bj / db;
        double tp = 1.0, td = 1.0;
        tp = (lo || up) ? fmin(tp, qp) : tp;
        td = da < 0.0 ? fmin(td, qa) : td;
        td = db < 0.0 ? fmin(td, qb) : td;
        a0 = mnin(a0, tp); a1 = mnin(a1, td);
      }
      W_FOR(i, m) {
        const double li = sl[i], ui = su[i], si = ss[i], ci = c[i], di = d[i];
        const bool in = eq[i] == 0.0;
        const double dss = as[i] + mus * cs[i], dc = ac[i] + mus * cc[i], dd2 = ad[i] + mus * cd[i];
        const bool lo = in && li > -kInf && dss < 0.0, up = in && ui < kInf && dss > 0.0;
        const double qp = (lo ? -tv * (si - li) : tv * (ui - si)) / dss, qc = -tv * ci / dc, qd = -tv * di / dd2;
        double tp = 1.0, td = 1.0;
        tp = (lo || up) ? fmin(tp, qp) : tp;
        td = dc < 0.0 ? fmin(td, qc) : td;
        td = dd2 < 0.0 ? fmin(td, qd) : td;
        a0 = mnin(a0, tp); a1 = mnin(a1, td);
      }
    } else {
      W_FOR(j, N) {
        double tp = 1.0, td = 1.0;
        const double dxx = ax[j] + mus * cx[j];
        if (l[j] > -kInf && dxx < 0.0) tp = fmin(tp, -tv * (xx[j] - l[j]) / dxx);
        if (u[j] < kInf && dxx > 0.0) tp = fmin(tp, tv * (u[j] - xx[j]) / dxx);
        const double da = aa[j] + mus * ca[j], db = ab[j] + mus * cb[j];
        if (da < 0.0) td = fmin(td, -tv * a[j] / da);
        if (db < 0.0) td = fmin(td, -tv * b[j] / db);
        a0 = mnin(a0, tp); a1 = mnin(a1, td);
      }
      W_FOR(i, m) {
        double tp = 1.0, td = 1.0;
        if (eq[i] == 0.0) {
          const double dss = as[i] + mus * cs[i];
          if (sl[i] > -kInf && dss < 0.0) tp = fmin(tp, -tv * (ss[i] - sl[i]) / dss);
          if (su[i] < kInf && dss > 0.0) tp = fmin(tp, tv * (su[i] - ss[i]) / dss);
        }
        const double dc = ac[i] + mus * cc[i], dd2 = ad[i] + mus * cd[i];
        if (dc < 0.0) td = fmin(td, -tv * c[i] / dc);
        if (dd2 < 0.0) td = fmin(td, -tv * d[i] / dd2);
        a0 = mnin(a0, tp); a1 = mnin(a1, td);
      }
    }
    { double r2[2] = {a0, a1}; P::vmax_n(r2); a0 = r2[0]; a1 = r2[1]; }
    const double apv = std::min(1.0, -a0), adv = std::min(1.0, -a1);
    if (P::hoist) {
      W_FOR(j, N) {
        const double lj = l[j], uj = u[j], xj = xx[j];
        const double dxx = ax[j] + mus * cx[j];
        const double t1 = (xj - lj + apv * dxx) * (a[j] + adv * (aa[j] + mus * ca[j]));
        const double t2 = (uj - xj - apv * dxx) * (b[j] + adv * (ab[j] + mus * cb[j]));
        double v = 0.0;
        v = lj > -kInf ? v + t1 * t1 : v;
        v = uj < kInf ? v + t2 * t2 : v;
        comp += v;
      }
      W_FOR(i, m) {
        const double li = sl[i], ui = su[i], si = ss[i];
        const bool in = eq[i] == 0.0;
        const double dss = as[i] + mus * cs[i];
        const double t1 = (si - li + apv * dss) * (c[i] + adv * (ac[i] + mus * cc[i]));
        const double t2 = (ui - si - apv * dss) * (d[i] + adv * (ad[i] + mus * cd[i]));
        double v = 0.0;
        v = (in && li > -kInf) ? v + t1 * t1 : v;
        v = (in && ui < kInf) ? v + t2 * t2 : v;
        comp += v;
      }
    } else {
      W_FOR(j, N) {
        double v = 0.0;
        const double dxx = ax[j] + mus * cx[j];
        if (l[j] > -kInf) { const double t = (xx[j] - l[j] + apv * dxx) * (a[j] + adv * (aa[j] + mus * ca[j])); v += t * t; }
        if (u[j] < kInf) { const double t = (u[j] - xx[j] - apv * dxx) * (b[j] + adv * (ab[j] + mus * cb[j])); v += t * t; }
        comp += v;
      }
      W_FOR(i, m) {
        double v = 0.0;
        if (eq[i] == 0.0) {
          const double dss = as[i] + mus * cs[i];
          if (sl[i] > -kInf) { const double t = (ss[i] - sl[i] + apv * dss) * (c[i] + adv * (ac[i] + mus * cc[i])); v += t * t; }
          if (su[i] < kInf) { const double t = (su[i] - ss[i] - apv * dss) * (d[i] + adv * (ad[i] + mus * cd[i])); v += t * t; }
        }
        comp += v;
      }
    }
    comp = P::sum(comp);
    W_P1(10);
    return (1.0 - adv) * (1.0 - adv) * nd2 / n_dual + (1.0 - apv) * (1.0 - apv) * np2 / n_pri + comp / static_cast<double>(nb);
  }
#ifdef DNLP_WAVE_SPEC
  // ---- the quality function over operands held in REGISTERS (P::hoist: the workgroup-per-instance kernel) --------------------
  // The mu oracle evaluates quality() 10 to 30 times per iteration on the same 23 vectors (only sigma changes), and with the
  // vectors in global memory every evaluation waits for them twice (a round trip is 1 us: 10 k cycles per evaluation, 122 k of
  // 1 070 k per iteration on path planning).  A lane's share is ceil(N / lanes) + ceil(m / lanes) entries — two and two at 512
  // lanes: 46 doubles — so the oracle loads them ONCE (qf_load) and every evaluation (quality_regs: the same expressions in the
  // same order as quality()'s P::hoist form, hence the same bits) is arithmetic and two block reductions.  The golden section
  // has ONE evaluation site here (section_regs: the order of evaluations of section(), written as a loop) so that the inlined
  // evaluation is not repeated six times in the code.
  static constexpr int kQfTN = (wspec::k_N + P::lanes - 1) / P::lanes > 0 ? (wspec::k_N + P::lanes - 1) / P::lanes : 1;
  static constexpr int kQfTM = (wspec::k_m + P::lanes - 1) / P::lanes > 0 ? (wspec::k_m + P::lanes - 1) / P::lanes : 1;
  struct QfRegs {
    double l[kQfTN], u[kQfTN], x[kQfTN], a[kQfTN], b[kQfTN], ax[kQfTN], cx[kQfTN], aa[kQfTN], ca[kQfTN], ab[kQfTN], cb[kQfTN];
    double sl[kQfTM], su[kQfTM], s[kQfTM], c[kQfTM], d[kQfTM], eq[kQfTM], as[kQfTM], cs[kQfTM], ac[kQfTM], cc[kQfTM], ad[kQfTM], cd[kQfTM];
  };
  DNLP_WINL DNLP_HD static void qf_load(WS* S, QfRegs& R) {
    const WD *ax = WDIR(1, 0), *as = WDIR(1, 1), *aa = WDIR(1, 3), *ab = WDIR(1, 4), *ac = WDIR(1, 5), *ad = WDIR(1, 6);
    const WD *cx = WDIR(2, 0), *cs = WDIR(2, 1), *ca = WDIR(2, 3), *cb = WDIR(2, 4), *cc = WDIR(2, 5), *cd = WDIR(2, 6);
    const WD *l = WV(xL), *u = WV(xU), *sl = WV(sL), *su = WV(sU), *eq = WV(eq), *xx = WV(x), *ss = WV(s), *a = WV(zL), *b = WV(zU), *c = WV(vL), *d = WV(vU);
    constexpr int N = WK(N), m = WK(m);
#pragma unroll
    for (int t = 0; t < kQfTN; ++t) {
      const int j0 = P::lane() + t * P::lanes, j = j0 < N ? j0 : 0;      // (an entry past the end reads entry 0: never used)
      if (N > 0) {
        R.l[t] = l[j]; R.u[t] = u[j]; R.x[t] = xx[j]; R.a[t] = a[j]; R.b[t] = b[j];
        R.ax[t] = ax[j]; R.cx[t] = cx[j]; R.aa[t] = aa[j]; R.ca[t] = ca[j]; R.ab[t] = ab[j]; R.cb[t] = cb[j];
      }
    }
#pragma unroll
    for (int t = 0; t < kQfTM; ++t) {
      const int i0 = P::lane() + t * P::lanes, i = i0 < m ? i0 : 0;
      if (m > 0) {
        R.sl[t] = sl[i]; R.su[t] = su[i]; R.s[t] = ss[i]; R.c[t] = c[i]; R.d[t] = d[i]; R.eq[t] = eq[i];
        R.as[t] = as[i]; R.cs[t] = cs[i]; R.ac[t] = ac[i]; R.cc[t] = cc[i]; R.ad[t] = ad[i]; R.cd[t] = cd[i];
      }
    }
  }
  DNLP_WINL DNLP_HD static double quality_regs(WS* S, const QfRegs& R, double sigma) {
    W_P0();
    const double avg = S->qf_avg, nd2 = S->qf_nd2, np2 = S->qf_np2, n_dual = S->qf_n_dual, n_pri = S->qf_n_pri;
    const i64 nb = S->qf_nb;
    const double mus = sigma * avg;
    const double tv = std::max(0.99, 1.0 - mus);
    constexpr int N = WK(N), m = WK(m);
    double a0 = -kInf, a1 = -kInf;
    double comp = 0.0;
#pragma unroll
    for (int t = 0; t < kQfTN; ++t) {
      if (P::lane() + t * P::lanes < N) {
        const double lj = R.l[t], uj = R.u[t], xj = R.x[t], aj = R.a[t], bj = R.b[t];
        const double dxx = R.ax[t] + mus * R.cx[t], da = R.aa[t] + mus * R.ca[t], db = R.ab[t] + mus * R.cb[t];
        const bool lo = lj > -kInf && dxx < 0.0, up = uj < kInf && dxx > 0.0;
        const double qp = (lo ? -tv * (xj - lj) : tv * (uj - xj)) / dxx, qa = -tv * aj / da, qb = -tv * bj / db;
        double tp = 1.0, td = 1.0;
        tp = (lo || up) ? fmin(tp, qp) : tp;
        td = da < 0.0 ? fmin(td, qa) : td;
        td = db < 0.0 ? fmin(td, qb) : td;
        a0 = mnin(a0, tp); a1 = mnin(a1, td);
      }
    }
#pragma unroll
    for (int t = 0; t < kQfTM; ++t) {
      if (P::lane() + t * P::lanes < m) {
        const double li = R.sl[t], ui = R.su[t], si = R.s[t], ci = R.c[t], di = R.d[t];
        const bool in = R.eq[t] == 0.0;
        const double dss = R.as[t] + mus * R.cs[t], dc = R.ac[t] + mus * R.cc[t], dd2 = R.ad[t] + mus * R.cd[t];
        const bool lo = in && li > -kInf && dss < 0.0, up = in && ui < kInf && dss > 0.0;
        const double qp = (lo ? -tv * (si - li) : tv * (ui - si)) / dss, qc = -tv * ci / dc, qd = -tv * di / dd2;
        double tp = 1.0, td = 1.0;
        tp = (lo || up) ? fmin(tp, qp) : tp;
        td = dc < 0.0 ? fmin(td, qc) : td;
        td = dd2 < 0.0 ? fmin(td, qd) : td;
        a0 = mnin(a0, tp); a1 = mnin(a1, td);
      }
    }
    { double r2[2] = {a0, a1}; P::vmax_n(r2); a0 = r2[0]; a1 = r2[1]; }
    const double apv = std::min(1.0, -a0), adv = std::min(1.0, -a1);
#pragma unroll
    for (int t = 0; t < kQfTN; ++t) {
      if (P::lane() + t * P::lanes < N) {
        const double lj = R.l[t], uj = R.u[t], xj = R.x[t];
        const double dxx = R.ax[t] + mus * R.cx[t];
        const double t1 = (xj - lj + apv * dxx) * (R.a[t] + adv * (R.aa[t] + mus * R.ca[t]));
        const double t2 = (uj - xj - apv * dxx) * (R.b[t] + adv * (R.ab[t] + mus * R.cb[t]));
        double v = 0.0;
        v = lj > -kInf ? v + t1 * t1 : v;
        v = uj < kInf ? v + t2 * t2 : v;
        comp += v;
      }
    }
#pragma unroll
    for (int t = 0; t < kQfTM; ++t) {
      if (P::lane() + t * P::lanes < m) {
        const double li = R.sl[t], ui = R.su[t], si = R.s[t];
        const bool in = R.eq[t] == 0.0;
        const double dss = R.as[t] + mus * R.cs[t];
        const double t1 = (si - li + apv * dss) * (R.c[t] + adv * (R.ac[t] + mus * R.cc[t]));
        const double t2 = (ui - si - apv * dss) * (R.d[t] + adv * (R.ad[t] + mus * R.cd[t]));
        double v = 0.0;
        v = (in && li > -kInf) ? v + t1 * t1 : v;
        v = (in && ui < kInf) ? v + t2 * t2 : v;
        comp += v;
      }
    }
    comp = P::sum(comp);
    W_P1(10);
    return (1.0 - adv) * (1.0 - adv) * nd2 / n_dual + (1.0 - apv) * (1.0 - apv) * np2 / n_pri + comp / static_cast<double>(nb);
  }
  // section() with one evaluation site: the same points in the same order (m1, m2, one new point per step, slo, sup)
  DNLP_WINL DNLP_HD static double section_regs(WS* S, const QfRegs& R, double slo, double sup) {
    const double gr = 0.5 * (3.0 - std::sqrt(5.0));
    double la = std::log(slo), lb = std::log(std::max(sup, slo * (1 + 1e-12)));
    double m1 = la + gr * (lb - la), m2 = lb - gr * (lb - la);
    double f1 = 0.0, f2 = 0.0, qlo = 0.0, qup = 0.0, sg = 0.0, fsel = 0.0;
    int it = 0, stage = 0;          // 0: f1 at m1, 1: f2 at m2, 2: the steps, 3: qlo, 4: qup
    bool second = false;            // a step's new value is f2 (else f1)
    while (true) {
      double at;
      if (stage == 0) at = std::exp(m1);
      else if (stage == 1) at = std::exp(m2);
      else if (stage == 2) {
        if (it < 8 && (lb - la) > 1e-2 * std::fabs(lb) + 1e-12) {
          ++it;
          if (f1 > f2) { la = m1; m1 = m2; f1 = f2; m2 = lb - gr * (lb - la); second = true; at = std::exp(m2); }
          else { lb = m2; m2 = m1; f2 = f1; m1 = la + gr * (lb - la); second = false; at = std::exp(m1); }
        } else {
          sg = std::exp(f1 < f2 ? m1 : m2);
          fsel = std::min(f1, f2);
          stage = 3; at = slo;
        }
      } else at = sup;
      const double f = quality_regs(S, R, at);
      if (stage == 0) { f1 = f; stage = 1; }
      else if (stage == 1) { f2 = f; stage = 2; }
      else if (stage == 2) { if (second) f2 = f; else f1 = f; }
      else if (stage == 3) { qlo = f; stage = 4; }
      else { qup = f; break; }
    }
    bool endpoint = false;
    if (qlo < fsel && qlo <= qup) { sg = slo; fsel = qlo; endpoint = true; }
    else if (qup < fsel) { sg = sup; fsel = qup; endpoint = true; }
    S->qf_fsel = fsel;
    S->qf_endpoint = endpoint ? 1 : 0;
    return sg;
  }
#endif
  // golden section in log(sigma) + IPOPT's end-point check (the lambda `section` of Ipm::quality_function_mu)
  DNLP_WFN DNLP_HD static double section(WS* S, double slo, double sup) {
    auto qf = [&](double sg) { return quality(S, sg, S->qf_avg, S->qf_nd2, S->qf_np2, S->qf_n_dual, S->qf_n_pri, S->qf_nb); };
    const double gr = 0.5 * (3.0 - std::sqrt(5.0));
    double la = std::log(slo), lb = std::log(std::max(sup, slo * (1 + 1e-12)));
    double m1 = la + gr * (lb - la), m2 = lb - gr * (lb - la);
    double f1 = qf(std::exp(m1)), f2 = qf(std::exp(m2));
    for (int it = 0; it < 8 && (lb - la) > 1e-2 * std::fabs(lb) + 1e-12; ++it) {
      if (f1 > f2) { la = m1; m1 = m2; f1 = f2; m2 = lb - gr * (lb - la); f2 = qf(std::exp(m2)); }
      else { lb = m2; m2 = m1; f2 = f1; m1 = la + gr * (lb - la); f1 = qf(std::exp(m1)); }
    }
    double sg = std::exp(f1 < f2 ? m1 : m2);
    double fsel = std::min(f1, f2);
    const double qlo = qf(slo), qup = qf(sup);
    bool endpoint = false;
    if (qlo < fsel && qlo <= qup) { sg = slo; fsel = qlo; endpoint = true; }
    else if (qup < fsel) { sg = sup; fsel = qup; endpoint = true; }
    S->qf_fsel = fsel;
    S->qf_endpoint = endpoint ? 1 : 0;
    return sg;
  }
  // the oracle's search for sigma (Ipm::quality_function_mu): qf = one evaluation, sec = the golden section between two bounds
  template <class QF, class SEC>
  DNLP_WINL DNLP_HD static double sigma_search(WS* S, double s_lo, double s_up, QF qf, SEC sec) {
    double sigma;
    if (s_lo >= s_up) {
      sigma = s_lo;
    } else {
      const double q1 = qf(1.0), s1m = 1.0 - 1e-2, q1m = qf(std::max(s_lo, s1m));
      double lo, up;
      if (q1m > q1 && s_up > 1.0) { lo = 1.0; up = s_up; } else { lo = s_lo; up = std::min(std::max(s_lo, s1m), s_up); }
      sigma = sec(lo, up);
      double fsel = S->qf_fsel;
      if (S->qf_endpoint != 0 && up > lo * 10.0) {
        const double grid[6] = {lo, 1e-4, 1e-2, 1e-1, 0.5, up};
        double gs[6], gq[6];
        int ng = 0;
        for (int k = 0; k < 6; ++k) {
          if (grid[k] < lo || grid[k] > up || (ng > 0 && grid[k] <= gs[ng - 1])) continue;
          gs[ng] = grid[k]; gq[ng] = qf(grid[k]); ++ng;
        }
        int best = 0;
        for (int k = 1; k < ng; ++k) if (gq[k] < gq[best]) best = k;
        if (gq[best] < fsel && best > 0 && best + 1 < ng) {
          const double f0 = fsel, s0 = sigma;
          sigma = sec(gs[best - 1], gs[best + 1]);
          fsel = S->qf_fsel;
          if (!(fsel < f0)) sigma = s0;
        }
      }
    }
    return sigma;
  }
  // Ipm::quality_function_mu
  DNLP_WFN DNLP_HD static bool quality_function_mu(WS* S, double dw) {
    const int N = WK(N), m = WK(m);
    W_P0();
    const double avg = avg_complementarity(S);
    const i64 nb = n_bound_mults(S);
    if (!(avg > 0.0) || nb == 0) return false;
    const double mu_floor = mu_floor_now(S);
    barrier_terms(S, 0.0);
    double nd2, np2;
    {
      const WD *rxx = WV(rx), *rss = WV(rs), *rpp = WV(rp);
      double s0 = 0.0, s1 = 0.0;
      W_FOR(k, N) s0 += rxx[k] * rxx[k];
      W_FOR(i, m) { s0 += rss[i] * rss[i]; s1 += rpp[i] * rpp[i]; }
      double r2[2] = {s0, s1};
      P::sum_n(r2);
      nd2 = r2[0];
      np2 = m ? r2[1] : 0.0;
    }
    W_P1(17);
    // The two systems of the oracle — affine scaling (mu = 0, residuals rx / rs / rp) and centering (unit mu, the
    // derivative of the barrier terms) — share the factor: both right-hand sides first, ONE refined joint solve
    // (solve_refined2), then the two directions.  The centering system lives in the arrays of its own direction until
    // that is written: rhs2 = [czL | cvL], sol2 = [cx | cs], res2 = [czU | cvU], its slack residual in cy.
    WD *rhs2 = WDIR(2, 3), *sol2 = WDIR(2, 0), *res2 = WDIR(2, 4), *q2 = WDIR(2, 2);
    {
      W_P0();
      WD* r = WV(rhs);
      const WD *rxx = WV(rx), *q = WV(rs), *pres = WV(rp), *sS = WV(Ss), *eq = WV(eq);
      const WD *l = WV(xL), *u = WV(xU), *sl = WV(sL), *su = WV(sU), *xx = WV(x), *ss = WV(s), *fm = WV(fixm);
      const double kd = S->opt.kappa_d;
      W_FOR(j, N) {
        r[j] = -rxx[j];
        double c = 0.0;
        const bool hl = l[j] > -kInf, hu = u[j] < kInf;
        if (hl) c -= 1.0 / (xx[j] - l[j]);
        if (hu) c += 1.0 / (u[j] - xx[j]);
        if (hl && !hu) c += kd;
        if (hu && !hl) c -= kd;
        const double rc = fm[j] != 0.0 ? 0.0 : c;
        rhs2[j] = -rc;
      }
      W_FOR(i, m) {
        r[N + i] = -pres[i] - (eq[i] == 0.0 ? q[i] / (sS[i] + dw) : 0.0);
        double c = 0.0;
        if (eq[i] == 0.0) {
          const bool hl = sl[i] > -kInf, hu = su[i] < kInf;
          if (hl) c -= 1.0 / (ss[i] - sl[i]);
          if (hu) c += 1.0 / (su[i] - ss[i]);
          if (hl && !hu) c += kd;
          if (hu && !hl) c -= kd;
        }
        q2[i] = c;
        rhs2[N + i] = -0.0 - (eq[i] == 0.0 ? c / (sS[i] + dw) : 0.0);
      }
      P::sync();
      W_P1(11);
    }
    double ratio_aff = 0.0, ratio_cen = 0.0;
    {
      W_P0();
      const int rc = solve_refined2(S, dw, rhs2, sol2, res2, ratio_aff, ratio_cen);
      W_P1(21);
      if (rc != 0) return false;
    }
    direction_outputs(S, WV(sol), WV(rs), 0.0, dw, false, 1);
    direction_outputs(S, sol2, q2, 1.0, dw, true, 2);
    S->last_ratio = ratio_cen;
    if (ratio_aff > S->last_ratio) S->last_ratio = ratio_aff;
    const i64 n_ineq = m - S->n_eq;
    S->qf_avg = avg; S->qf_nd2 = nd2; S->qf_np2 = np2; S->qf_nb = nb;
    S->qf_n_dual = static_cast<double>(N + n_ineq); S->qf_n_pri = static_cast<double>(m > 0 ? m : 1);
    const double mu_max = S->opt.mu_max_fact * avg;
    const double s_lo = std::max(1e-6, mu_floor / avg), s_up = std::min(1e2, mu_max / avg);
    double sigma;
#ifdef DNLP_WAVE_SPEC
    if constexpr (P::hoist) {
      QfRegs R;
      qf_load(S, R);
      sigma = sigma_search(S, s_lo, s_up, [&](double sg) DNLP_WINL { return quality_regs(S, R, sg); },
                           [&](double lo, double up) DNLP_WINL { return section_regs(S, R, lo, up); });
    } else
#endif
    sigma = sigma_search(S, s_lo, s_up, [&](double sg) { return quality(S, sg, S->qf_avg, S->qf_nd2, S->qf_np2, S->qf_n_dual, S->qf_n_pri, S->qf_nb); },
                         [&](double lo, double up) { return section(S, lo, up); });
    const double nm = std::max(mu_floor, std::min(sigma * avg, mu_max));
    if (!std::isfinite(nm)) return false;
    S->mu = nm;
    S->tau = std::max(0.99, 1.0 - S->mu);
    S->nfilt = 0;
    barrier_terms(S, S->mu);
    const double muv = S->mu;
    {
      for (int k = 0; k < 7; ++k) {
        WD* o = WDIR(0, k);
        const WD *av = WDIR(1, k), *cv = WDIR(2, k);
        const int n = (k == 0 || k == 3 || k == 4) ? N : m;
        W_FOR(i, n) o[i] = av[i] + muv * cv[i];
      }
      P::sync();
    }
    return true;
  }

  // Ipm::restoration_phase
  DNLP_WFN DNLP_HD static bool restoration_phase(WS* S, double theta_k) {
    const int N = WK(N), m = WK(m);
    const double phi_k = barrier_at(S, S->f, WV(x), WV(s), S->mu);
    filter_add(S, (1.0 - 1e-5) * theta_k, phi_k - 1e-8 * theta_k);
    double th_cur = theta_k;
    double zeta = std::sqrt(S->mu);
    S->resto_stationary = false;
    S->resto_theta = theta_k;
    const WD *eq = WV(eq), *sl = WV(sL);
    for (int it = 0; it < 100; ++it) {
      WD *sx = WV(Sx), *dd = WV(Dd);
      const double zz = zeta;
      W_FOR(j, N) sx[j] = zz;
      W_FOR(i, m) dd[i] = 1.0 + (eq[i] == 0.0 ? 1.0 / zz : 0.0);
      { WD* hs = WV(Hs); W_FOR(p, WK(nnzH)) hs[p] = 0.0; }
      P::sync();
      int nneg = 0, nzero = 0;
      if (!assemble_factor(S, WV(Sx), WV(Dd), 0.0, true, &nneg, &nzero)) return false;
      WD* r = WV(rhs);
      const WD *gg = WV(g), *ss = WV(s);
      W_FOR(j, N) r[j] = 0.0;
      W_FOR(i, m) r[N + i] = -(eq[i] != 0.0 ? gg[i] - sl[i] : gg[i] - ss[i]);
      P::sync();
      kkt_solve(S, WV(rhs), WV(sol));
      const WD* so = WV(sol);
      WD *ddx = WV(dx), *dds = WV(ds);
      W_FOR(j, N) ddx[j] = so[j];
      W_FOR(i, m) dds[i] = eq[i] == 0.0 ? so[N + i] / zz : 0.0;
      P::sync();
      double a = max_step_primal(S, S->tau);
      bool moved = false;
      for (int bt = 0; bt < 30; ++bt) {
        trial_point(S, a);
        double fv;
        if (eval_fg(S, WV(xt), fv, WV(gt)) && nan_check(S, WV(gt)) == 0.0) {
          const double th = theta_at(S, WV(gt), WV(st));
          if (std::isfinite(th) && th < (1.0 - 1e-4 * a) * th_cur) {
            WD *xx = WV(x), *sv = WV(s), *gv = WV(g);
            const WD *nx = WV(xt), *ns = WV(st), *gn = WV(gt);
            W_FOR(j, N) xx[j] = nx[j];
            W_FOR(i, m) { sv[i] = ns[i]; gv[i] = gn[i]; }
            P::sync();
            S->f = fv;
            th_cur = th;
            moved = true;
            break;
          }
        }
        a *= 0.5;
      }
      if (!moved) { zeta *= 10.0; if (zeta > 1e8) { S->resto_stationary = true; S->resto_theta = th_cur; return false; } continue; }
      S->resto_theta = th_cur;
      sweep(S, WV(x), false);
      eval_derivs(S);
      const double ph = barrier_at(S, S->f, WV(x), WV(s), S->mu);
      if (th_cur <= 0.9 * theta_k && th_cur <= S->theta_max && filter_ok(S, th_cur, ph)) {
        S->jty_valid = false;
        const double zi = 1.0;
        const WD *l = WV(xL), *u = WV(xU), *su = WV(sU);
        WD *za = WV(zL), *zb = WV(zU), *c = WV(vL), *d = WV(vU), *yy = WV(y);
        W_FOR(j, N) { za[j] = (l[j] > -kInf) ? zi : 0.0; zb[j] = (u[j] < kInf) ? zi : 0.0; }
        W_FOR(i, m) {
          yy[i] = 0.0;
          c[i] = (eq[i] == 0.0 && sl[i] > -kInf) ? zi : 0.0;
          d[i] = (eq[i] == 0.0 && su[i] < kInf) ? zi : 0.0;
        }
        P::sync();
        if (m > 0) init_multipliers_ls(S);
        return true;
      }
      if (th_cur < 1e-13) return false;
    }
    return false;
  }

  // Ipm::polish
  DNLP_WFN DNLP_HD static void polish(WS* S) {
    const int N = WK(N), m = WK(m);
    WD* sv[7] = {WV(x), WV(s), WV(y), WV(zL), WV(zU), WV(vL), WV(vU)};
    const int sz[7] = {N, m, m, N, N, m, m};
    const int slot[7] = {0, 1, 2, 3, 4, 5, 6};
    int poff[8];
    poff[0] = 0;
    for (int k = 0; k < 7; ++k) poff[k + 1] = poff[k] + ((sz[k] + 1) & ~1);
    (void)slot;
    for (int k = 0; k < 7; ++k) { double* dst = S->park + poff[k]; const WD* src = sv[k]; W_FOR(i, sz[k]) dst[i] = src[i]; }
    P::sync();
    const double tol0 = S->opt.tol, mu0 = S->mu, tau0 = S->tau;
    const int maxit0 = S->opt.max_iter;
    S->opt.tol = tol0 * 1e-3;
    S->opt.max_iter = S->iter + 12;
    while (step(S) == 99) {}
    S->opt.tol = tol0;
    S->opt.max_iter = maxit0;
    if (S->status == Solve_Succeeded) return;
    if (S->bail) return;
    {
      const WErr e = error(S, 0.0);
      if (check_convergence(S, e) == Solve_Succeeded) { S->status = Solve_Succeeded; return; }
    }
    for (int k = 0; k < 7; ++k) { const double* src = S->park + poff[k]; WD* dst = sv[k]; W_FOR(i, sz[k]) dst[i] = src[i]; }
    P::sync();
    S->mu = mu0; S->tau = tau0;
    (void)eval_fg(S, WV(x), S->f, WV(g));
    eval_derivs(S);
    S->e_cached_valid = false;
    S->acceptable_count = 0;
    (void)check_convergence(S, error(S, 0.0));
    S->status = Solve_Succeeded;
  }

  // Ipm::solve (the retry ladder included)
  DNLP_WFN DNLP_HD static int solve(WS* S) {
    W_P0();
    const double t_all = P::now();
    S->in_solve = true;
    S->bail = false;
    S->ladder_rung = 0;
    S->sparse_singular_streak = 0;
    S->delta_w_used_last_iter = false;
    S->last_ratio = 0.0;
    S->initialized = false;
    S->iter = 0;
    S->f = 0.0;
    int rc = begin(S);
    if (rc != 0) { S->in_solve = false; return rc; }
    while (step(S) == 99) {}
    if (S->opt.adaptive_fallback && !S->bail) {
      const int strategy0 = S->opt.mu_strategy;
      const double mu_init0 = S->opt.mu_init;
      const int max_iter0 = S->opt.max_iter;
      const double max_wall0 = S->opt.max_wall_time;
      for (int rung = 1; rung <= 2; ++rung) {
        const int status = S->status;
        const bool failed = status == Infeasible_Problem_Detected || status == Restoration_Failed || status == Error_In_Step_Computation ||
                            status == Search_Direction_Becomes_Too_Small || status == Diverging_Iterates;
        if (!failed || S->bail) break;
        if (rung == 1 && strategy0 != 1) continue;
        const int it_first = S->iter;
        if (it_first >= max_iter0) { S->status = Maximum_Iterations_Exceeded; break; }
        if (P::now() - t_all > max_wall0) { S->status = Maximum_WallTime_Exceeded; break; }
        S->opt.max_iter = max_iter0 - it_first;
        S->opt.max_wall_time = max_wall0 - (P::now() - t_all);
        S->ladder_rung = rung;
        S->opt.mu_strategy = 0;
        if (rung == 2) S->opt.mu_init = mu_init0 * 10.0 > 1.0 ? mu_init0 * 10.0 : 1.0;
        rc = begin(S);
        if (rc == 0) while (step(S) == 99) {}
        if (S->status == Solve_Succeeded && !S->bail) polish(S);
        S->iter += it_first;
      }
      S->opt.mu_strategy = strategy0;
      S->opt.mu_init = mu_init0;
      S->opt.max_iter = max_iter0;
      S->opt.max_wall_time = max_wall0;
      S->ladder_rung = 0;
    }
    S->in_solve = false;
    S->wall = P::now() - t_all;
    if (S->bail) S->status = kWaveNeedsGeneric;
    W_P1(0);
    return S->status;
  }
#undef W_FOR
};

}  // namespace dnlp
