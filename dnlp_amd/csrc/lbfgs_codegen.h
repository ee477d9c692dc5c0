// Device-resident L-BFGS for generated fused objectives (BASELINE config C2 at its stated size,
// n = 1e5: "tape f / grad f eval + line search only").
//
// The host-driven loop of lbfgs_core.h costs ~6 stream synchronisations per iteration (each a launch
// + a scalar read-back, 30-60 us): 0.45 ms per iteration for kernels that run 2-6 us.  Here every
// decision (Armijo test, curvature test, restart, convergence) is taken ON THE DEVICE; the host only
// enqueues a fixed four-kernel "slot" over and over — kernels turn into no-ops once the state says
// `done` — and looks at the state once per batch of slots.  On MI355X a dependent kernel boundary costs
// ~1.5 us (MI355X_MICROARCH.md, price list "boundary") against 4-7 us for a grid barrier across the 8
// non-coherent XCD L2s, so a chain of small kernels beats one persistent kernel with grid barriers for
// this size class.
//
// One slot = one trial point of the line search:
//   lb_eval    (grid)   first trial of an iteration: dir = sum_j coef_j B_j on the fly (owner entries
//                       stored); window of the trial point x + step dir in registers; generated element
//                       code (fused_codegen.h: owner computes, no atomics) -> grad at the trial point,
//                       f partials
//   lb_accept  (1 WG)   f = sum of partials; Armijo test; on failure step /= 2 (next slot tries again)
//   lb_update  (grid)   on acceptance: x += s, history rows s, y, new gradient; partial dots of the three
//                       new rows against the whole basis B = [s_0.. | y_0.. | g] (vector-free two-loop,
//                       Chen et al. 2014, as lbfgs_core.h)
//   lb_control (1 WG)   Gram rows, curvature test, rho, convergence test, two-loop recursion on the
//                       2M+1 coefficients -> coef, g'd for the next iteration
// Same mathematics and the same decisions as ReducedLbfgs::solve (lbfgs_core.h); only where they are
// taken differs.
#pragma once
#include <string>

#include "fused_codegen.h"

namespace dnlp {

constexpr int kLbMaxM = 15;
constexpr int kLbMaxNB = 2 * kLbMaxM + 1;

// The state lives in device memory; this text is compiled BOTH here (host driver reads it back) and
// inside the generated translation unit, so the two layouts cannot drift.
#define DNLP_LB_STR_(...) #__VA_ARGS__
#define DNLP_LB_STR(...) DNLP_LB_STR_(__VA_ARGS__)
#define DNLP_LB_STATE_BODY                                                                                  \
  double f; double fn; double step; double gd; double gn; double tol;                                       \
  double rho[15]; double alpha[15]; double coef[31]; double G[31 * 31];                                      \
  int iter; int evals; int head; int stored; int ls; int phase; int accept; int done; int max_iter; int M;   \
  int nblocks; int gcur;
struct LbfgsState { DNLP_LB_STATE_BODY };

// M (history length) is a generation-time constant: the loops over the 2M+1 basis rows unroll, which is
// what lets the single-lane two-loop recursion of lb_control pipeline its LDS reads.
// Control block of the persistent kernel (device memory, zeroed by the host before every launch).
struct LbPersistCtl {
  unsigned arrive; unsigned abort; unsigned pad[30];        // top barrier counter on a line of its own
  unsigned grp[8][32];                                      // arrival counter of workgroup group g = blockIdx % 8, a line each
  unsigned gen[8][32];                                      // generation word the members of group g wait on, a line each
  double acc[3][96 * 16];                                   // rotating accumulators, one per 128-byte line; since round 4 only
                                                            // entry 95 is live: max |g| (an atomic max: order-independent)
  // Order-fixed sums across the workgroups (round 4: the atomic adds that used to land here in arrival order made C2 the
  // one path whose bits could change from run to run): a workgroup stores its 95 partial sums in its row of `part`, the
  // LAST arriver of a barrier group adds its group's rows in member order into gsum[parity][group], and after the
  // barrier every workgroup adds the eight group sums in group order.
  double gsum[2][8][96];
  double part[256][96];
};
constexpr int kLbPersistMaxWgs = 256;                       // rows of LbPersistCtl::part

// `per` > 0 adds the persistent kernel dnlp_lb_persist for workgroups that own `per` consecutive variables each.
// `mode`: where a workgroup keeps its slice — 0 everything in LDS (slices up to ~700 variables at M = 10), 1 the 2M
// history rows in a per-workgroup strip of global memory (L2 / Infinity-Cache resident: 48 MB at n = 3e5) and the five
// working vectors in LDS, 2 everything in that strip (slices that exceed LDS altogether: n beyond ~9e5 on 256 CUs).
inline std::string lbfgs_codegen_source(const std::vector<FusedSlotProg>& progs, const FusedCodegenInfo& info, int M,
                                        long long per = 0, int mode = 0) {
  std::string s = fused_codegen_preamble(info.E);
  s += "#define DNLP_M " + std::to_string(M) + "\n#define DNLP_NB " + std::to_string(2 * M + 1) + "\n";
  if (per > 0) s += "#define DNLP_PER " + std::to_string(per) + "\n#define DNLP_PMODE " + std::to_string(mode) + "\n";
  if (const char* ev = std::getenv("DNLP_LBFGS_FULL_FENCE"); ev && std::atoi(ev) != 0) s += "#define DNLP_LB_FULL_FENCE 1\n";     // (A/B of the barrier's fences)
  if (const char* ev = std::getenv("DNLP_LBFGS_ATOMIC_SUMS"); ev && std::atoi(ev) != 0) s += "#define DNLP_LB_ATOMIC_SUMS 1\n";   // (the arrival-order sums of rounds 3-4: not reproducible)
  s += fused_codegen_chunk(progs, info);
  s += "struct LbfgsState { " DNLP_LB_STR(DNLP_LB_STATE_BODY) " };\n";
  s += R"DNLPLB(
#define DNLP_MAXNB 31
#define DNLP_NV (3 * DNLP_MAXNB + 1)
// Partial results travel between the grid kernels and the one-workgroup control kernels as COLUMNS:
// value k of block b at part[k * ldp + b], so that one wavefront reads 64 blocks' values of one quantity
// in one coalesced load and folds them with shuffles.
// (Host emulation, DNLP_EMULATE: the lanes run one after the other, so "reduce over the wavefront and let
// lane 0 store" becomes "lane 0 clears, every lane folds in", and a column reduction is a serial loop.)
#ifndef DNLP_EMULATE
__device__ __forceinline__ double dnlp_wave_max(double v) {
  v = fmax(v, dnlp_dpp_f64<0x111, 0xf>(v, v));
  v = fmax(v, dnlp_dpp_f64<0x112, 0xf>(v, v));
  v = fmax(v, dnlp_dpp_f64<0x114, 0xf>(v, v));
  v = fmax(v, dnlp_dpp_f64<0x118, 0xf>(v, v));
  v = fmax(v, dnlp_dpp_f64<0x142, 0xa>(v, v));
  v = fmax(v, dnlp_dpp_f64<0x143, 0xc>(v, v));
  return dnlp_lane63(v);
}
__device__ __forceinline__ void dnlp_wave_store_sum(double* __restrict__ dst, double v) {
  v = dnlp_wave_sum(v);
  if ((threadIdx.x & 63) == 0) *dst = v;
}
__device__ __forceinline__ void dnlp_wave_store_max(double* __restrict__ dst, double v) {
  v = dnlp_wave_max(v);
  if ((threadIdx.x & 63) == 0) *dst = v;
}
// reduction of col[0..n) by the calling wavefront in two steps, so that a wavefront can put the loads of
// MANY columns in flight before it folds any of them (one memory latency instead of one per column):
// dnlp_col_partial = this lane's share, dnlp_col_finish = fold across the wavefront (every lane gets it)
__device__ __forceinline__ double dnlp_col_partial(const double* __restrict__ col, int n, bool is_max) {
  const int lane = threadIdx.x & 63;
  double a = is_max ? -1.0 : 0.0;
  for (int b = lane; b < n; b += 64) { const double v = col[b]; a = is_max ? fmax(a, v) : a + v; }
  return a;
}
__device__ __forceinline__ double dnlp_col_finish(double a, bool is_max) { return is_max ? dnlp_wave_max(a) : dnlp_wave_sum(a); }
#define DNLP_SYNC() __syncthreads()
#else
inline void dnlp_wave_store_sum(double* dst, double v) { if ((threadIdx.x & 63) == 0) *dst = 0.0; *dst += v; }
inline void dnlp_wave_store_max(double* dst, double v) { if ((threadIdx.x & 63) == 0) *dst = -1.0; *dst = fmax(*dst, v); }
inline double dnlp_col_partial(const double* col, int n, bool is_max) {
  double a = is_max ? -1.0 : 0.0;
  for (int b = 0; b < n; ++b) a = is_max ? fmax(a, col[b]) : a + col[b];
  return a;
}
inline double dnlp_col_finish(double a, bool) { return a; }
#define DNLP_SYNC()
#endif

// Row j of the basis B = [s_0 .. s_{M-1} | y_0 .. y_{M-1} | g]: the current gradient is one of two
// buffers that swap roles on acceptance (lb_eval writes the trial gradient into the other one), so the
// update kernel never writes a row another wavefront still reads.
__device__ __forceinline__ const double* dnlp_row(const double* __restrict__ BV, const double* __restrict__ gcur, int j, int GR, i64 nf) {
  return j == GR ? gcur : BV + static_cast<i64>(j) * nf;
}

// ---- trial point: gradient and f partials at x + step * dir -------------------------------------
extern "C" __global__ void __launch_bounds__(256) dnlp_lb_eval(const LbfgsState* __restrict__ S, const double* __restrict__ x,
    const double* __restrict__ BV, double* __restrict__ gbuf0, double* __restrict__ gbuf1, double* __restrict__ dir,
    const double* __restrict__ consts, double* __restrict__ fpart, const i64 nf, const i64 nchunks, const int ldp) {
  double facc = 0.0, chk = 0.0;                       // f partial, NaN / inf detector of the gradient
  if (S->done == 0) {
    constexpr int GR = 2 * DNLP_M, nb = DNLP_NB;
    const double* __restrict__ gcur = S->gcur ? gbuf1 : gbuf0;
    double* __restrict__ gt = S->gcur ? gbuf0 : gbuf1;
    const int phase = S->phase;
    const double step = phase == 0 ? 0.0 : S->step;
    const bool fresh = phase != 0 && S->ls == 0;      // first trial of an iteration: dir is not built yet
    for (i64 q = static_cast<i64>(blockIdx.x) * 256 + threadIdx.x; q < nchunks; q += static_cast<i64>(gridDim.x) * 256) {
      const i64 c = q * DNLP_E;
      double xr[DNLP_NX], dw[DNLP_NX];
#pragma unroll
      for (int k = 0; k < DNLP_NX; ++k) dw[k] = 0.0;
      if (fresh) {
#pragma unroll 1
        for (int j = 0; j < nb; ++j) {
          const double cj = S->coef[j];
          if (cj == 0.0) continue;                   // uniform: rows not yet in the history cost nothing
          const double* __restrict__ row = dnlp_row(BV, gcur, j, GR, nf);
#pragma unroll
          for (int k = 0; k < DNLP_NX; ++k) { const i64 i = c - DNLP_W + k; if (i >= 0 && i < nf) dw[k] += cj * row[i]; }
        }
#pragma unroll
        for (int t = 0; t < DNLP_E; ++t) if (c + t < nf) dir[c + t] = dw[DNLP_W + t];
      } else if (phase != 0) {
#pragma unroll
        for (int k = 0; k < DNLP_NX; ++k) { const i64 i = c - DNLP_W + k; if (i >= 0 && i < nf) dw[k] = dir[i]; }
      }
#pragma unroll
      for (int k = 0; k < DNLP_NX; ++k) { const i64 i = c - DNLP_W + k; xr[k] = (i >= 0 && i < nf) ? x[i] + step * dw[k] : 0.0; }
      double g[DNLP_E];
#pragma unroll
      for (int t = 0; t < DNLP_E; ++t) g[t] = 0.0;
      if (DNLP_INTERIOR(c, nf)) dnlp_chunk_w<false>(c, xr, consts, g, facc);
      else dnlp_chunk_w<true>(c, xr, consts, g, facc);
#pragma unroll
      for (int t = 0; t < DNLP_E; ++t) if (c + t < nf) { gt[c + t] = g[t]; chk += g[t] - g[t]; }
    }
  }
  // one partial per wavefront: column index 4 * block + wave
  const int col = 4 * blockIdx.x + (threadIdx.x >> 6);
  dnlp_wave_store_sum(fpart + col, facc);
  dnlp_wave_store_sum(fpart + static_cast<i64>(ldp) * 4 + col, chk);
}

// ---- Armijo test (one wavefront) --------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(64) dnlp_lb_accept(LbfgsState* __restrict__ S, const double* __restrict__ fpart,
                                                                const double c0, const int ldp) {
  if (S->done != 0) return;
  const int ncol = 4 * S->nblocks;
  const double p0 = dnlp_col_partial(fpart, ncol, false);
  const double p1 = dnlp_col_partial(fpart + static_cast<i64>(ldp) * 4, ncol, false);
  const double fsum = dnlp_col_finish(p0, false), chk = dnlp_col_finish(p1, false);
  if (threadIdx.x != 63) return;                    // (the LAST lane decides: host emulation runs lanes in order)
  const double fn = c0 + fsum;
  S->evals += 1;
  const bool finite = (fn - fn == 0.0) && chk == 0.0;
  S->fn = fn;
  if (S->phase == 0) {
    if (finite) S->accept = 1; else S->done = 4;    // invalid number at the start
    return;
  }
  if (finite && fn <= S->f + 1e-4 * S->step * S->gd) { S->accept = 1; return; }
  S->accept = 0;
  S->step *= 0.5;
  S->ls += 1;
  if (S->ls >= 60) S->done = 2;                      // line search stuck
}

// ---- accepted step: x, history rows s and y, partial Gram rows --------------------------------------
// A block owns a contiguous range of elements; every wavefront holds the range's s, y, g_new in
// registers (tiles of 1024 elements), wavefront w writes the elements of tile slots t = w (mod 4) and
// owns the basis rows j = w (mod 4): it streams row j once against the three vectors, so a row's three
// dots are reduced inside ONE wavefront and stored by its lane 0 — no cross-wavefront reduction, and no
// row is written that another wavefront still reads (rows head / M + head come from registers, the
// gradient row is the buffer lb_eval just filled).
#define DNLP_UT 16
extern "C" __global__ void __launch_bounds__(256) dnlp_lb_update(const LbfgsState* __restrict__ S, double* __restrict__ x,
    double* __restrict__ BV, const double* __restrict__ gbuf0, const double* __restrict__ gbuf1,
    const double* __restrict__ dir, double* __restrict__ upart, const i64 nf, const int ldp) {
  if (S->done != 0 || S->accept == 0) return;
  constexpr int M = DNLP_M, nb = DNLP_NB, GR = 2 * DNLP_M;
  const int head = S->head;
  const bool run = S->phase != 0;
  const double step = S->step;
  const double* __restrict__ gold = S->gcur ? gbuf1 : gbuf0;
  const double* __restrict__ gnew = S->gcur ? gbuf0 : gbuf1;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const i64 per = ((nf + gridDim.x - 1) / gridDim.x + 63) / 64 * 64;
  const i64 b0 = static_cast<i64>(blockIdx.x) * per, b1 = b0 + per < nf ? b0 + per : nf;
  double as[8], ay[8], ag[8];                       // dots of this wavefront's rows (at most ceil(31 / 4) = 8)
#pragma unroll
  for (int r = 0; r < 8; ++r) as[r] = ay[r] = ag[r] = 0.0;
  double gmax = 0.0;
  for (i64 t0 = b0; t0 < b1; t0 += 64 * DNLP_UT) {
    double sv[DNLP_UT], yv[DNLP_UT], gv[DNLP_UT];
#pragma unroll
    for (int t = 0; t < DNLP_UT; ++t) {
      const i64 i = t0 + lane + 64 * t;
      const bool in = i < b1;
      gv[t] = in ? gnew[i] : 0.0;
      sv[t] = (in && run) ? step * dir[i] : 0.0;
      yv[t] = (in && run) ? gv[t] - gold[i] : 0.0;
      gmax = fmax(gmax, fabs(gv[t]));
      if (in && run && (t & 3) == wave) {
        x[i] += sv[t];
        BV[static_cast<i64>(head) * nf + i] = sv[t];
        BV[static_cast<i64>(M + head) * nf + i] = yv[t];
      }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int j = wave + 4 * r;
      if (j < nb) {
        const double* __restrict__ row = dnlp_row(BV, gnew, j, GR, nf);
        const bool is_s = run && j == head, is_y = run && j == M + head, is_g = j == GR;
#pragma unroll
        for (int t = 0; t < DNLP_UT; ++t) {
          const i64 i = t0 + lane + 64 * t;
          const double v = is_s ? sv[t] : is_y ? yv[t] : is_g ? gv[t] : (i < b1 ? row[i] : 0.0);
          as[r] += sv[t] * v;
          ay[r] += yv[t] * v;
          ag[r] += gv[t] * v;
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int j = wave + 4 * r;
    if (j < nb) {
      dnlp_wave_store_sum(upart + static_cast<i64>(j) * ldp + blockIdx.x, as[r]);
      dnlp_wave_store_sum(upart + static_cast<i64>(DNLP_MAXNB + j) * ldp + blockIdx.x, ay[r]);
      dnlp_wave_store_sum(upart + static_cast<i64>(2 * DNLP_MAXNB + j) * ldp + blockIdx.x, ag[r]);
    }
  }
  if (wave == 0) dnlp_wave_store_max(upart + static_cast<i64>(3 * DNLP_MAXNB) * ldp + blockIdx.x, gmax);
}

// ---- Gram rows, history bookkeeping, convergence, two-loop recursion (one workgroup of 256) -----------
// The scalar part runs on ONE lane; everything it touches repeatedly (Gram matrix, coefficients) is staged
// in LDS first — a dependent chain of global loads costs ~1 us per link, an LDS one ~0.05.
extern "C" __global__ void __launch_bounds__(256) dnlp_lb_control(LbfgsState* __restrict__ S, const double* __restrict__ upart,
                                                                  const int ldp) {
  __shared__ double red[DNLP_NV];
  __shared__ double Gs[DNLP_MAXNB * DNLP_MAXNB];
  __shared__ double cf[DNLP_MAXNB], rh[16], al[16];
  if (S->done != 0 || S->accept == 0) return;
  const int nblocks = S->nblocks;
  {
    // wavefront w folds the columns k = w (mod 4): all their loads first, then the shuffles
    constexpr int nb0 = DNLP_NB;
    const int wave = threadIdx.x >> 6;
    constexpr int NC = (DNLP_NV + 3) / 4;
    double part[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int k = wave + 4 * c;
      const bool used = k < DNLP_NV && (k >= 3 * DNLP_MAXNB || (k % DNLP_MAXNB) < nb0);   // rows beyond the basis were never written
      part[c] = used ? dnlp_col_partial(upart + static_cast<i64>(k) * ldp, nblocks, k == DNLP_NV - 1) : 0.0;
    }
    for (int k = threadIdx.x; k < DNLP_MAXNB * DNLP_MAXNB; k += 256) Gs[k] = S->G[k];
    if (threadIdx.x < 15) rh[threadIdx.x] = S->rho[threadIdx.x];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int k = wave + 4 * c;
      if (k < DNLP_NV) {
        const double r = dnlp_col_finish(part[c], k == DNLP_NV - 1);
        if ((threadIdx.x & 63) == 63) red[k] = r;
      }
    }
  }
  DNLP_SYNC();
#ifndef DNLP_EMULATE
  // The scalar part on ONE WAVEFRONT, lane q owning coefficient q of the 2M+1 basis rows: every inner product of the
  // two-loop recursion is one DPP reduction instead of a 21-link chain of dependent LDS reads on a single lane
  // (17 us of the 55 us slot in round 2).  Same arithmetic order as the serial text below is NOT kept (a tree sum
  // replaces the left-to-right sum): iterates agree to rounding, decisions to the last bit only by luck.
  if (threadIdx.x >= 64) return;
  {
    const int lane = threadIdx.x;
    constexpr int M = DNLP_M, nb = DNLP_NB, GR = 2 * DNLP_M;
    constexpr int ld = DNLP_MAXNB;
    const int NV = DNLP_NV;
    int head = S->head, stored = S->stored, iter = S->iter;
    const bool run = S->phase != 0;
    const double tol = S->tol, fnew = S->fn;
    const int max_iter = S->max_iter;
    double* G = S->G;
    const bool in = lane < nb;
    if (in) {
      if (run) {
        Gs[head * ld + lane] = Gs[lane * ld + head] = red[lane];
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (in && run) { Gs[(M + head) * ld + lane] = Gs[lane * ld + (M + head)] = red[DNLP_MAXNB + lane]; }
    __builtin_amdgcn_wave_barrier();
    if (in) { Gs[GR * ld + lane] = Gs[lane * ld + GR] = red[2 * DNLP_MAXNB + lane]; }
    __builtin_amdgcn_wave_barrier();
    // write the new rows / columns through
    if (in) {
      if (run) {
        G[head * ld + lane] = Gs[head * ld + lane]; G[lane * ld + head] = Gs[lane * ld + head];
        G[(M + head) * ld + lane] = Gs[(M + head) * ld + lane]; G[lane * ld + (M + head)] = Gs[lane * ld + (M + head)];
      }
      G[GR * ld + lane] = Gs[GR * ld + lane]; G[lane * ld + GR] = Gs[lane * ld + GR];
    }
    if (run) {
      const double sy = Gs[head * ld + (M + head)], ss = Gs[head * ld + head], yy = Gs[(M + head) * ld + (M + head)];
      if (sy > 1e-10 * sqrt(ss) * sqrt(yy)) {
        if (lane == 0) { rh[head] = 1.0 / sy; S->rho[head] = 1.0 / sy; }
        head = (head + 1) % M;
        if (stored < M) ++stored;
      }
      iter += 1;
    }
    __builtin_amdgcn_wave_barrier();
    const double gn = red[NV - 1];
    if (lane == 0) {
      S->iter = iter; S->head = head; S->f = fnew; S->phase = 1; S->accept = 0; S->ls = 0; S->gcur ^= 1; S->gn = gn;
    }
    if (gn <= tol * fmax(1.0, fabs(fnew))) { if (lane == 0) { S->stored = stored; S->done = 1; } return; }
    if (iter >= max_iter) { if (lane == 0) { S->stored = stored; S->done = 3; } return; }
    double c = (lane == GR) ? 1.0 : 0.0;
    double myal = 0.0;                               // alpha of history slot `lane` (lanes 0 .. M-1)
    for (int j = 0; j < stored; ++j) {
      const int idx = (head - 1 - j + 2 * M) % M;
      const double v = dnlp_wave_sum(in ? c * Gs[idx * ld + lane] : 0.0);
      const double a = rh[idx] * v;
      if (lane == idx) myal = a;
      if (lane == M + idx) c -= a;
    }
    if (stored > 0) {
      const int idx = (head - 1 + M) % M;
      const double gam = Gs[idx * ld + (M + idx)] / Gs[(M + idx) * ld + (M + idx)];
      c *= gam;
    }
    for (int j = stored - 1; j >= 0; --j) {
      const int idx = (head - 1 - j + 2 * M) % M;
      const double v = dnlp_wave_sum(in ? c * Gs[(M + idx) * ld + lane] : 0.0);
      if (lane == idx) c += myal - rh[idx] * v;
    }
    c = -c;
    double gd = dnlp_wave_sum(in ? c * Gs[GR * ld + lane] : 0.0);
    if (!(gd < 0.0)) {                                 // not a descent direction: steepest descent, history dropped
      stored = 0;
      c = (lane == GR) ? -1.0 : 0.0;
      gd = -Gs[GR * ld + GR];
    }
    if (in) S->coef[lane] = c;
    if (lane == 0) {
      S->stored = stored;
      S->gd = gd;
      S->step = (iter == 0 && stored == 0) ? fmin(1.0, 1.0 / fmax(gn, 1e-300)) : 1.0;
    }
  }
#else
  if (threadIdx.x != 255) return;                    // (the LAST lane: host emulation runs the lanes in order)
  const int NV = DNLP_NV;
  constexpr int M = DNLP_M, nb = DNLP_NB, GR = 2 * DNLP_M;
  constexpr int ld = DNLP_MAXNB;
  int head = S->head, stored = S->stored, iter = S->iter;
  const bool run = S->phase != 0;
  const double tol = S->tol, fnew = S->fn;
  const int max_iter = S->max_iter;
  double* G = S->G;
  if (run) {
    for (int j = 0; j < nb; ++j) { Gs[head * ld + j] = Gs[j * ld + head] = red[j]; }
    for (int j = 0; j < nb; ++j) { Gs[(M + head) * ld + j] = Gs[j * ld + (M + head)] = red[DNLP_MAXNB + j]; }
  }
  for (int j = 0; j < nb; ++j) Gs[GR * ld + j] = Gs[j * ld + GR] = red[2 * DNLP_MAXNB + j];
  // write the new rows / columns through (in the order above, so that shared entries end up as in LDS)
  if (run) {
    for (int j = 0; j < nb; ++j) { G[head * ld + j] = Gs[head * ld + j]; G[j * ld + head] = Gs[j * ld + head]; }
    for (int j = 0; j < nb; ++j) { G[(M + head) * ld + j] = Gs[(M + head) * ld + j]; G[j * ld + (M + head)] = Gs[j * ld + (M + head)]; }
  }
  for (int j = 0; j < nb; ++j) { G[GR * ld + j] = Gs[GR * ld + j]; G[j * ld + GR] = Gs[j * ld + GR]; }
  if (run) {
    const double sy = Gs[head * ld + (M + head)], ss = Gs[head * ld + head], yy = Gs[(M + head) * ld + (M + head)];
    if (sy > 1e-10 * sqrt(ss) * sqrt(yy)) {
      rh[head] = 1.0 / sy;
      S->rho[head] = rh[head];
      head = (head + 1) % M;
      if (stored < M) ++stored;
    }
    iter += 1;
  }
  S->iter = iter;
  S->head = head;
  S->f = fnew;
  S->phase = 1;
  S->accept = 0;
  S->ls = 0;
  S->gcur ^= 1;                                      // the trial gradient is the current one from here on
  const double gn = red[NV - 1];
  S->gn = gn;
  if (gn <= tol * fmax(1.0, fabs(fnew))) { S->stored = stored; S->done = 1; return; }
  if (iter >= max_iter) { S->stored = stored; S->done = 3; return; }
  // two-loop recursion on the coefficients of q in the basis (lbfgs_core.h)
  for (int j = 0; j < nb; ++j) cf[j] = 0.0;
  cf[GR] = 1.0;
  for (int j = 0; j < stored; ++j) {
    const int idx = (head - 1 - j + 2 * M) % M;
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < nb; ++q) v += cf[q] * Gs[idx * ld + q];
    const double a = rh[idx] * v;
    al[idx] = a;
    cf[M + idx] -= a;
  }
  if (stored > 0) {
    const int idx = (head - 1 + M) % M;
    const double gam = Gs[idx * ld + (M + idx)] / Gs[(M + idx) * ld + (M + idx)];
    for (int j = 0; j < nb; ++j) cf[j] *= gam;
  }
  for (int j = stored - 1; j >= 0; --j) {
    const int idx = (head - 1 - j + 2 * M) % M;
    double v = 0.0;
#pragma unroll
    for (int q = 0; q < nb; ++q) v += cf[q] * Gs[(M + idx) * ld + q];
    cf[idx] += al[idx] - rh[idx] * v;
  }
  for (int j = 0; j < nb; ++j) cf[j] = -cf[j];
  double gd = 0.0;
#pragma unroll
  for (int q = 0; q < nb; ++q) gd += cf[q] * Gs[GR * ld + q];
  if (!(gd < 0.0)) {                                 // not a descent direction: steepest descent, history dropped
    stored = 0;
    for (int j = 0; j < nb; ++j) cf[j] = 0.0;
    cf[GR] = -1.0;
    gd = -Gs[GR * ld + GR];
  }
  for (int j = 0; j < nb; ++j) S->coef[j] = cf[j];
  S->stored = stored;
  S->gd = gd;
  S->step = (iter == 0 && stored == 0) ? fmin(1.0, 1.0 / fmax(gn, 1e-300)) : 1.0;
#endif
}

#if defined(DNLP_PER) && !defined(DNLP_EMULATE)
// ---- the whole solve in ONE launch (BASELINE C2 at its stated size) ---------------------------------------------------
// One workgroup per compute unit owns DNLP_PER consecutive variables and keeps, in LDS for the whole solve, its slice
// (plus a halo of DNLP_W on either side) of x, the direction, the gradients and all 2M history rows: a trial point, the
// direction, the history update and the 3 (2M+1) inner products of the vector-free two-loop recursion never touch
// HBM.  Workgroups meet twice per iteration (once per rejected trial) at a grid barrier; what crosses it is tiny —
// f and the inner products through double-precision atomic adds into rotating accumulators, the boundary entries
// of the trial gradient for the neighbours' halos — and every workgroup then takes the same decisions from the same
// numbers (Armijo test, curvature test, two-loop recursion on its own LDS copy of the Gram matrix).  The four-kernel
// slot above spends its time in dependent global loads (state, coefficients, rows): 50 us per slot at n = 1e5.
#define DNLP_PL (DNLP_PER + 2 * DNLP_W)
struct LbPersistCtl { unsigned arrive; unsigned abort; unsigned pad[30]; unsigned grp[8][32]; unsigned gen[8][32]; double acc[3][96 * 16];
                      double gsum[2][8][96]; double part[256][96]; };
#define DNLP_ACC(p, k) ((p) + 16 * (k))

__device__ __forceinline__ double dnlp_agent_load(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Everything that crosses workgroups in this kernel travels as device-scope atomic loads / stores (write-through, cache-
// bypassing).  What the barrier's release side needs is COMPLETION of those stores — in EVERY wavefront that issued some,
// not only in the one that arrives: a workgroup-scope fence lowers to s_waitcnt lgkmcnt(0) alone on gfx950, and
// `global_store ... sc1; s_barrier; global_atomic_add` then lets the arrive overtake another wavefront's part[] / gsum /
// halo stores.  DNLP_RELEASE_ALL() is executed by every lane before the __syncthreads that precedes the arrive: an order
// fence for the compiler + s_waitcnt vmcnt(0) (gfx9 counts stores in vmcnt), i.e. all of this wavefront's write-through
// stores are acknowledged; it skips only the L2 write-back (buffer_wbl2) of an agent-scope release, which the plain
// (cached) stores of this kernel do not need because none of them is read by another workgroup.
// DNLP_LB_FULL_FENCE=1 makes both sides __threadfence().
#ifdef DNLP_LB_FULL_FENCE
#define DNLP_ORDER_FENCE() __threadfence()
#define DNLP_RELEASE_ALL() __threadfence()
#else
#define DNLP_ORDER_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_s_waitcnt(0x0F70); } while (0)
#define DNLP_RELEASE_ALL() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_waitcnt(0x0F70); } while (0)
#endif
// Two-level counter barrier over the nwg co-resident workgroups (grid <= compute units: every workgroup is resident).
// 256 arrivals on ONE device-scope counter serialise in its memory channel (~12 ns each: 3 us before the last arriver
// is even counted) and 256 pollers hammer the same line; here a workgroup arrives on the counter of its group
// g = blockIdx % 8 (32 arrivals per line, the eight lines in parallel), the LAST arriver of a group arrives on the top
// counter, waits for the eight groups and bumps its group's generation word, which the other members poll.
// A spin is bounded: a stranded workgroup raises `abort` and everybody leaves instead of hanging the device.
__device__ __forceinline__ bool dnlp_grid_barrier(LbPersistCtl* ctl, unsigned& epoch, const unsigned nwg, int* s_flag) {
  ++epoch;                                             // (uniform: every lane counts the barriers)
  DNLP_RELEASE_ALL();                                  // every wavefront: its device-scope stores have completed
  __syncthreads();
  if (threadIdx.x == 0) {
    DNLP_ORDER_FENCE();                                   // release: this workgroup's stores and atomics before the arrive
    const unsigned g = blockIdx.x & 7u, ngroups = nwg < 8u ? nwg : 8u;
    const unsigned gsize = (nwg + 7u - g) >> 3;        // workgroups with blockIdx % 8 == g
    const unsigned mine = atomicAdd(&ctl->grp[g][0], 1u) + 1u;
    unsigned spins = 0;
    int ok = 1;
    const bool leader = mine == epoch * gsize;
    if (leader) atomicAdd(&ctl->arrive, 1u);
    unsigned* word = leader ? &ctl->arrive : &ctl->gen[g][0];
    const unsigned target = leader ? epoch * ngroups : epoch;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 1023u) == 0u) {
        if (spins > (1u << 22) || __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
    }
    if (leader && ok) __hip_atomic_store(&ctl->gen[g][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DNLP_ORDER_FENCE();                                   // acquire
    *s_flag = ok;
  }
  __syncthreads();
  return *s_flag != 0;
}

// The same barrier carrying the order-fixed sums of the 95 per-workgroup partials in wpart (LDS) into red (LDS):
// see LbPersistCtl::part / gsum.  scr: 192 doubles of LDS.
// The same barrier carrying the order-fixed sums of the 95 per-workgroup partials in wpart (LDS) into red (LDS):
// see LbPersistCtl::part / gsum.  scr: 192 doubles of LDS.
__device__ __noinline__ bool dnlp_grid_barrier_sum(LbPersistCtl* ctl, unsigned& epoch, const unsigned nwg, int* s_flag, int* s_lead,
                                                      const double* wpart, double* red, double* scr, const int parity) {
  ++epoch;                                             // (uniform: every lane counts the barriers)
  const int tid = threadIdx.x;
  const unsigned g = blockIdx.x & 7u, ngroups = nwg < 8u ? nwg : 8u;
  const unsigned gsize = (nwg + 7u - g) >> 3;          // workgroups with blockIdx % 8 == g
  DNLP_RELEASE_ALL();                                  // (the halo entries any wavefront stored before the call)
  __syncthreads();                                     // wpart is complete
  if (tid < 95) __hip_atomic_store(&ctl->part[blockIdx.x][tid], wpart[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  DNLP_RELEASE_ALL();                                  // BOTH storing wavefronts (tid 0..94): rows and halo entries acknowledged
  __syncthreads();
  if (tid == 0) {
    DNLP_ORDER_FENCE();                                   // release: this workgroup's stores before the arrive
    const unsigned mine = atomicAdd(&ctl->grp[g][0], 1u) + 1u;
    *s_lead = (mine == epoch * gsize) ? 1 : 0;
  }
  __syncthreads();
  const bool leader = *s_lead != 0;
  if (leader) {
    DNLP_ORDER_FENCE();                                   // acquire: the members' rows
    if (tid < 192) {
      // two half-sums per value, members in index order, then the halves: the same tree whoever arrives last
      const int v = tid % 96, c = tid / 96;
      const unsigned half = (gsize + 1u) >> 1, k0 = c ? half : 0u, k1 = c ? gsize : half;
      // (the loads of a half issued together, THEN added in member order: a load-add loop is a chain of round trips)
      double t[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) t[q] = (v < 95 && k0 + q < k1) ? dnlp_agent_load(&ctl->part[g + 8u * (k0 + q)][v]) : 0.0;
      double sacc = 0.0;
#pragma unroll
      for (int q = 0; q < 16; ++q) sacc += t[q];
      scr[c * 96 + v] = sacc;
    }
    __syncthreads();
    if (tid < 95) __hip_atomic_store(&ctl->gsum[parity][g][tid], scr[tid] + scr[96 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    DNLP_RELEASE_ALL();
    __syncthreads();
  }
  if (tid == 0) {
    unsigned spins = 0;
    int ok = 1;
    if (leader) { DNLP_ORDER_FENCE(); atomicAdd(&ctl->arrive, 1u); }
    unsigned* word = leader ? &ctl->arrive : &ctl->gen[g][0];
    const unsigned target = leader ? epoch * ngroups : epoch;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 1023u) == 0u) {
        if (spins > (1u << 22) || __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
          __hip_atomic_store(&ctl->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
      }
    }
    DNLP_ORDER_FENCE();                                   // acquire what the other groups released; ordered before the release below
    if (leader && ok) __hip_atomic_store(&ctl->gen[g][0], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *s_flag = ok;
  }
  __syncthreads();
  if (*s_flag == 0) return false;
  if (tid < 95) {
    double t[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) t[h] = (static_cast<unsigned>(h) < ngroups) ? dnlp_agent_load(&ctl->gsum[parity][h][tid]) : 0.0;
    double sacc = 0.0;
#pragma unroll
    for (int h = 0; h < 8; ++h) sacc += t[h];
    red[tid] = sacc;
  }
  return true;                                         // (the caller's next __syncthreads publishes red)
}

extern "C" __global__ void __launch_bounds__(256) dnlp_lb_persist(LbfgsState* __restrict__ S, double* __restrict__ x,
    const double* __restrict__ consts, LbPersistCtl* __restrict__ ctl, double* __restrict__ halo, const double c0, const i64 nf,
    double* __restrict__ strip) {
  constexpr int M = DNLP_M, nb = DNLP_NB, GR = 2 * DNLP_M, W = DNLP_W, PL = DNLP_PL, PER = DNLP_PER;
  constexpr int ldg = DNLP_MAXNB;
  // the slice of this workgroup: in LDS, or (DNLP_PMODE) in its strip of global memory — (2M + 5) PL doubles per
  // workgroup, read and written by this workgroup only, coalesced; the same indexing either way
#if DNLP_PMODE == 0
  __shared__ double xs[PL], ds[PL], xt[PL], gold[PL], gnew[PL];
  __shared__ double rows[2 * M][PL];                       // s_0 .. s_{M-1}, y_0 .. y_{M-1}
#else
  double* const mine = strip + static_cast<i64>(blockIdx.x) * (2 * M + 5) * PL;
  double (*rows)[PL] = reinterpret_cast<double (*)[PL]>(mine);
#if DNLP_PMODE == 1
  __shared__ double xs[PL], ds[PL], xt[PL], gold[PL], gnew[PL];
#else
  double* const xs = mine + static_cast<i64>(2 * M) * PL;
  double* const ds = xs + PL;
  double* const xt = ds + PL;
  double* const gold = xt + PL;
  double* const gnew = gold + PL;
#endif
#endif
  __shared__ double Gs[DNLP_MAXNB * DNLP_MAXNB];
  __shared__ double cf[DNLP_MAXNB], rh[16], red[96], wred[4][4], wpart[96], scr[192];
  __shared__ double sc[8];                                  // gd, step, (spare)
  __shared__ int sci[8];                                    // head, stored, iter, done, flag
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned nwg = gridDim.x;
  const i64 e0 = static_cast<i64>(blockIdx.x) * PER;       // first owned variable
  const i64 e1 = e0 + PER < nf ? e0 + PER : nf;
  const int own = static_cast<int>(e1 > e0 ? e1 - e0 : 0);
  unsigned epoch = 0;
  // ---- load the slice, clear the history ----
  for (int li = tid; li < PL; li += 256) {
    const i64 i = e0 - W + li;
    xs[li] = (i >= 0 && i < nf) ? x[i] : 0.0;
    ds[li] = 0.0; gold[li] = 0.0; gnew[li] = 0.0;
  }
  for (int k = tid; k < 2 * M * PL; k += 256) (&rows[0][0])[k] = 0.0;
  for (int k = tid; k < DNLP_MAXNB * DNLP_MAXNB; k += 256) Gs[k] = 0.0;
  if (tid < DNLP_MAXNB) cf[tid] = 0.0;
  if (tid < 16) rh[tid] = 0.0;
  if (tid < 96) wpart[tid] = 0.0;
  __syncthreads();
  // replicated state (every lane of every workgroup holds the same values)
  double f = 0.0, step = 0.0, gd = 0.0, gn = 0.0;
  int iter = 0, evals = 0, head = 0, stored = 0, ls = 0, phase = 0, done = 0, slot = 0;
  bool fresh = false;
  const double tol = S->tol;
  const int max_iter = S->max_iter;
  while (done == 0) {
    // ---- trial point xt = xs + step ds (phase 0: the start point itself) ----
    if (fresh) {
#pragma unroll 2
      for (int li = tid; li < PL; li += 256) {
        double d = cf[GR] * gold[li];
#pragma unroll
        for (int j = 0; j < 2 * M; ++j) d = fma(cf[j], rows[j][li], d);
        ds[li] = d;
      }
      __syncthreads();
    }
    const double stp = phase == 0 ? 0.0 : step;
    for (int li = tid; li < PL; li += 256) xt[li] = fma(stp, ds[li], xs[li]);
    __syncthreads();
    double facc = 0.0, chk = 0.0;
    for (int q = tid; q * DNLP_E < own; q += 256) {
      const i64 c = e0 + static_cast<i64>(q) * DNLP_E;
      double xr[DNLP_NX];
#pragma unroll
      for (int k = 0; k < DNLP_NX; ++k) xr[k] = xt[q * DNLP_E + k];        // xt[li] holds variable e0 - W + li
      double g[DNLP_E];
#pragma unroll
      for (int t = 0; t < DNLP_E; ++t) g[t] = 0.0;
      if (DNLP_INTERIOR(c, nf)) dnlp_chunk_w<false>(c, xr, consts, g, facc);
      else dnlp_chunk_w<true>(c, xr, consts, g, facc);
#pragma unroll
      for (int t = 0; t < DNLP_E; ++t) if (q * DNLP_E + t < own) { gnew[W + q * DNLP_E + t] = g[t]; chk += g[t] - g[t]; }
    }
    // ---- what an ACCEPTED trial needs from the other workgroups travels with f through the same barrier: the inner
    // products of the would-be history rows s = step d, y = g_new - g_old and of g_new against the whole basis, over the
    // owned entries (a rejected trial has computed them for nothing: LDS work of a microsecond against a second
    // grid barrier per iteration)
    const bool run = phase != 0;
    double gmax = 0.0;
    {
      // wavefront w owns the basis rows j = w (mod 4): three inner products each
      double as_[8], ay_[8], ag_[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) as_[r] = ay_[r] = ag_[r] = 0.0;
      __syncthreads();                                  // gnew of the owned entries is complete
      // (unrolled: with the history in the global strip the loads of four strides are in flight together)
#pragma unroll 4
      for (int li = W + lane; li < W + own; li += 64) {
        const double gv = gnew[li];
        const double sv = run ? stp * ds[li] : 0.0, yv = run ? gv - gold[li] : 0.0;
        gmax = fmax(gmax, fabs(gv));
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const int j = wave + 4 * r;
          if (j < nb) {
            const double v = j == GR ? gv : (run && j == head) ? sv : (run && j == M + head) ? yv : rows[j < 2 * M ? j : 0][li];
            as_[r] = fma(sv, v, as_[r]);
            ay_[r] = fma(yv, v, ay_[r]);
            ag_[r] = fma(gv, v, ag_[r]);
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int j = wave + 4 * r;
        if (j < nb) {
          const double a = dnlp_wave_sum(as_[r]), b = dnlp_wave_sum(ay_[r]), c = dnlp_wave_sum(ag_[r]);
          if (lane == 0) {                              // (row j belongs to this wavefront alone: plain LDS stores)
#ifdef DNLP_LB_ATOMIC_SUMS
            double* accd = ctl->acc[slot % 3];
            atomicAdd(DNLP_ACC(accd, 2 + j), a);
            atomicAdd(DNLP_ACC(accd, 2 + DNLP_MAXNB + j), b);
            atomicAdd(DNLP_ACC(accd, 2 + 2 * DNLP_MAXNB + j), c);
#else
            wpart[2 + j] = a;
            wpart[2 + DNLP_MAXNB + j] = b;
            wpart[2 + 2 * DNLP_MAXNB + j] = c;
#endif
          }
        }
      }
    }
    facc = dnlp_wave_sum(facc);
    chk = dnlp_wave_sum(chk);
    gmax = dnlp_wave_max(gmax);
    if (lane == 0) { wred[wave][0] = facc; wred[wave][1] = chk; wred[wave][2] = gmax; }
    __syncthreads();
    double* acc = ctl->acc[slot % 3];
    if (tid == 0) {
#ifdef DNLP_LB_ATOMIC_SUMS
      atomicAdd(DNLP_ACC(acc, 0), (wred[0][0] + wred[1][0]) + (wred[2][0] + wred[3][0]));
      atomicAdd(DNLP_ACC(acc, 1), (wred[0][1] + wred[1][1]) + (wred[2][1] + wred[3][1]));
#else
      wpart[0] = (wred[0][0] + wred[1][0]) + (wred[2][0] + wred[3][0]);
      wpart[1] = (wred[0][1] + wred[1][1]) + (wred[2][1] + wred[3][1]);
#endif
      const double g4 = fmax(fmax(wred[0][2], wred[1][2]), fmax(wred[2][2], wred[3][2]));
      // max of non-negative doubles = max of their bit patterns as unsigned integers
      atomicMax(reinterpret_cast<unsigned long long*>(DNLP_ACC(acc, 95)), static_cast<unsigned long long>(__double_as_longlong(g4)));
    }
    if (blockIdx.x == 0 && tid < 96) __hip_atomic_store(DNLP_ACC(ctl->acc[(slot + 1) % 3], tid), 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // boundary entries of the trial gradient for the neighbours: halo[parity][wg][0 .. W) = first W, [W .. 2W) = last W
    double* hb = halo + (static_cast<i64>(slot & 1) * nwg + blockIdx.x) * (2 * W);
    if (tid < W) {
      __hip_atomic_store(hb + tid, own > tid ? gnew[W + tid] : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(hb + W + tid, own >= W ? gnew[W + own - W + tid] : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef DNLP_LB_ATOMIC_SUMS
    if (!dnlp_grid_barrier(ctl, epoch, nwg, &sci[4])) { done = 5; break; }
    if (tid < 96) red[tid] = dnlp_agent_load(DNLP_ACC(acc, tid));
#else
    if (!dnlp_grid_barrier_sum(ctl, epoch, nwg, &sci[4], &sci[5], wpart, red, scr, slot & 1)) { done = 5; break; }
    if (tid == 95) red[95] = dnlp_agent_load(DNLP_ACC(acc, 95));
#endif
    if (tid >= 128 && tid < 128 + W) {
      // left halo = the last W entries of the left neighbour, right halo = the first W of the right one
      const int t = tid - 128;
      const double* hp = halo + static_cast<i64>(slot & 1) * nwg * (2 * W);
      gnew[t] = blockIdx.x > 0 ? dnlp_agent_load(hp + static_cast<i64>(blockIdx.x - 1) * (2 * W) + W + t) : 0.0;
      gnew[W + own + t] = (blockIdx.x + 1 < nwg && own == PER) ? dnlp_agent_load(hp + static_cast<i64>(blockIdx.x + 1) * (2 * W) + t) : 0.0;
    }
    __syncthreads();
    const double fn = c0 + red[0];
    const double chks = red[1];
    ++slot;
    ++evals;
    const bool finite = (fn - fn == 0.0) && chks == 0.0;
    bool accept;
    if (phase == 0) { accept = finite; if (!finite) { done = 4; break; } }
    else accept = finite && fn <= f + 1e-4 * step * gd;
    if (!accept) {
      step *= 0.5;
      ++ls;
      fresh = false;
      if (ls >= 60) done = 2;
      __syncthreads();
      continue;
    }
    // ---- accepted: history rows, x, gradient (own entries and halo: all local) ----
    for (int li = tid; li < PL; li += 256) {
      if (run) {
        const double sv = step * ds[li];
        rows[head][li] = sv;
        rows[M + head][li] = gnew[li] - gold[li];
        xs[li] += sv;
      }
      gold[li] = gnew[li];
    }
    __syncthreads();
    // ---- Gram matrix, curvature test, convergence, two-loop recursion: wavefront 0, a coefficient per lane ----
    if (tid < 64) {
      const bool in = lane < nb;
      if (in && run) { Gs[head * ldg + lane] = Gs[lane * ldg + head] = red[2 + lane]; }
      __builtin_amdgcn_wave_barrier();
      if (in && run) { Gs[(M + head) * ldg + lane] = Gs[lane * ldg + (M + head)] = red[2 + DNLP_MAXNB + lane]; }
      __builtin_amdgcn_wave_barrier();
      if (in) { Gs[GR * ldg + lane] = Gs[lane * ldg + GR] = red[2 + 2 * DNLP_MAXNB + lane]; }
      __builtin_amdgcn_wave_barrier();
      int h2 = head, st2 = stored, it2 = iter;
      if (run) {
        const double sy = Gs[head * ldg + (M + head)], ss = Gs[head * ldg + head], yy = Gs[(M + head) * ldg + (M + head)];
        if (sy > 1e-10 * sqrt(ss) * sqrt(yy)) {
          if (lane == 0) rh[head] = 1.0 / sy;
          h2 = (head + 1) % M;
          if (st2 < M) ++st2;
        }
        it2 += 1;
      }
      __builtin_amdgcn_wave_barrier();
      const double gn2 = red[95];                     // (the largest bit pattern IS the largest non-negative double)
      int dn = 0;
      if (gn2 <= tol * fmax(1.0, fabs(fn))) dn = 1;
      else if (it2 >= max_iter) dn = 3;
      double c = (lane == GR) ? 1.0 : 0.0, myal = 0.0, gd2 = 0.0;
      if (dn == 0) {
        for (int j = 0; j < st2; ++j) {
          const int idx = (h2 - 1 - j + 2 * M) % M;
          const double v = dnlp_wave_sum(in ? c * Gs[idx * ldg + lane] : 0.0);
          const double a = rh[idx] * v;
          if (lane == idx) myal = a;
          if (lane == M + idx) c -= a;
        }
        if (st2 > 0) {
          const int idx = (h2 - 1 + M) % M;
          c *= Gs[idx * ldg + (M + idx)] / Gs[(M + idx) * ldg + (M + idx)];
        }
        for (int j = st2 - 1; j >= 0; --j) {
          const int idx = (h2 - 1 - j + 2 * M) % M;
          const double v = dnlp_wave_sum(in ? c * Gs[(M + idx) * ldg + lane] : 0.0);
          if (lane == idx) c += myal - rh[idx] * v;
        }
        c = -c;
        gd2 = dnlp_wave_sum(in ? c * Gs[GR * ldg + lane] : 0.0);
        if (!(gd2 < 0.0)) { st2 = 0; c = (lane == GR) ? -1.0 : 0.0; gd2 = -Gs[GR * ldg + GR]; }
        if (in) cf[lane] = c;
      }
      if (lane == 0) {
        sc[0] = gd2; sc[1] = gn2;
        sc[2] = (it2 == 0 && st2 == 0) ? fmin(1.0, 1.0 / fmax(gn2, 1e-300)) : 1.0;
        sci[0] = h2; sci[1] = st2; sci[2] = it2; sci[3] = dn;
      }
    }
    __syncthreads();
    gd = sc[0]; gn = sc[1]; step = sc[2];
    head = sci[0]; stored = sci[1]; iter = sci[2]; done = sci[3];
    f = fn; phase = 1; ls = 0; fresh = true;
    __syncthreads();
  }
  // ---- results: the owned entries of x, the state ----
  for (int li = W + tid; li < W + own; li += 256) x[e0 + (li - W)] = xs[li];
  if (blockIdx.x == 0 && tid == 0) {
    S->f = f; S->fn = f; S->gn = gn; S->iter = iter; S->evals = evals; S->done = done; S->head = head; S->stored = stored;
    S->step = step; S->gd = gd; S->phase = phase; S->ls = ls;
  }
}
#endif
)DNLPLB";
  return s;
}

}  // namespace dnlp
