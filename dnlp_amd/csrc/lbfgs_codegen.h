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
inline std::string lbfgs_codegen_source(const std::vector<FusedSlotProg>& progs, const FusedCodegenInfo& info, int M) {
  std::string s = fused_codegen_preamble(info.E);
  s += "#define DNLP_M " + std::to_string(M) + "\n#define DNLP_NB " + std::to_string(2 * M + 1) + "\n";
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
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
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
}
)DNLPLB";
  return s;
}

}  // namespace dnlp
