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
  int nblocks; int pad;
struct LbfgsState { DNLP_LB_STATE_BODY };

inline std::string lbfgs_codegen_source(const std::vector<FusedSlotProg>& progs, const FusedCodegenInfo& info) {
  std::string s = fused_codegen_preamble(info.E);
  s += fused_codegen_chunk(progs, info);
  s += "struct LbfgsState { " DNLP_LB_STR(DNLP_LB_STATE_BODY) " };\n";
  s += R"DNLPLB(
#define DNLP_MAXNB 31
// block-wide reduction of `n` per-lane values into dst[0..n): the first n - ntail by sum, the last ntail
// by max.  (Host emulation runs the lanes one after the other: lane 0 clears, every lane folds in.)
#ifndef DNLP_EMULATE
__device__ __forceinline__ double dnlp_wave_max(double v) {
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}
template <int N>
__device__ __forceinline__ void dnlp_block_reduce_store(const double (&v)[N], int ntail, double* __restrict__ dst) {
  __shared__ double red[4][N];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double r = (k >= N - ntail) ? dnlp_wave_max(v[k]) : dnlp_wave_sum(v[k]);
    if (lane == 0) red[wave][k] = r;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < N; k += 256)
    dst[k] = (k >= N - ntail) ? fmax(fmax(red[0][k], red[1][k]), fmax(red[2][k], red[3][k]))
                              : (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
}
#define DNLP_SYNC() __syncthreads()
#else
template <int N>
inline void dnlp_block_reduce_store(const double (&v)[N], int ntail, double* dst) {
  for (int k = 0; k < N; ++k) {
    if (threadIdx.x == 0) dst[k] = (k >= N - ntail) ? -1.0 : 0.0;
    if (k >= N - ntail) dst[k] = fmax(dst[k], v[k]); else dst[k] += v[k];
  }
}
#define DNLP_SYNC()
#endif

// ---- trial point: gradient and f partials at x + step * dir -------------------------------------
extern "C" __global__ void __launch_bounds__(256) dnlp_lb_eval(LbfgsState* __restrict__ S, const double* __restrict__ x,
    const double* __restrict__ BV, double* __restrict__ dir, double* __restrict__ gt, const double* __restrict__ consts,
    double* __restrict__ fpart, const i64 nf, const i64 nchunks) {
  double vals[2] = {0.0, 0.0};                      // f partial, NaN / inf detector of the gradient
  if (S->done == 0) {
    const int nb = 2 * S->M + 1;
    const double step = S->phase == 0 ? 0.0 : S->step;
    const bool fresh = S->phase != 0 && S->ls == 0;   // first trial of an iteration: dir is not built yet
    for (i64 q = static_cast<i64>(blockIdx.x) * 256 + threadIdx.x; q < nchunks; q += static_cast<i64>(gridDim.x) * 256) {
      const i64 c = q * DNLP_E;
      double xr[DNLP_NX], dw[DNLP_NX];
#pragma unroll
      for (int k = 0; k < DNLP_NX; ++k) dw[k] = 0.0;
      if (fresh) {
        for (int j = 0; j < nb; ++j) {
          const double cj = S->coef[j];
          if (cj == 0.0) continue;                   // uniform: rows not yet in the history cost nothing
          const double* __restrict__ row = BV + static_cast<i64>(j) * nf;
#pragma unroll
          for (int k = 0; k < DNLP_NX; ++k) { const i64 i = c - DNLP_W + k; if (i >= 0 && i < nf) dw[k] += cj * row[i]; }
        }
#pragma unroll
        for (int t = 0; t < DNLP_E; ++t) if (c + t < nf) dir[c + t] = dw[DNLP_W + t];
      } else if (S->phase != 0) {
#pragma unroll
        for (int k = 0; k < DNLP_NX; ++k) { const i64 i = c - DNLP_W + k; if (i >= 0 && i < nf) dw[k] = dir[i]; }
      }
#pragma unroll
      for (int k = 0; k < DNLP_NX; ++k) { const i64 i = c - DNLP_W + k; xr[k] = (i >= 0 && i < nf) ? x[i] + step * dw[k] : 0.0; }
      double g[DNLP_E];
#pragma unroll
      for (int t = 0; t < DNLP_E; ++t) g[t] = 0.0;
      if (DNLP_INTERIOR(c, nf)) dnlp_chunk_w<false>(c, xr, consts, g, vals[0]);
      else dnlp_chunk_w<true>(c, xr, consts, g, vals[0]);
#pragma unroll
      for (int t = 0; t < DNLP_E; ++t) if (c + t < nf) { gt[c + t] = g[t]; vals[1] += g[t] - g[t]; }
    }
  }
  dnlp_block_reduce_store<2>(vals, 0, fpart + 2 * static_cast<i64>(blockIdx.x));
}

// ---- Armijo test (one workgroup of 64) ------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(64) dnlp_lb_accept(LbfgsState* __restrict__ S, const double* __restrict__ fpart,
                                                                const double c0) {
  __shared__ double red[2][64];
  if (S->done != 0) return;
  double a = 0.0, b = 0.0;
  for (int k = threadIdx.x; k < S->nblocks; k += 64) { a += fpart[2 * k]; b += fpart[2 * k + 1]; }
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = b;
  DNLP_SYNC();
  if (threadIdx.x != 63) return;                    // (the LAST lane decides: host emulation runs lanes in order)
  double fn = c0, chk = 0.0;
  for (int k = 0; k < 64; ++k) { fn += red[0][k]; chk += red[1][k]; }
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

// ---- accepted step: x, history rows, partial Gram rows ----------------------------------------------
extern "C" __global__ void __launch_bounds__(256) dnlp_lb_update(const LbfgsState* __restrict__ S, double* __restrict__ x,
    double* __restrict__ BV, const double* __restrict__ dir, const double* __restrict__ gt, double* __restrict__ upart,
    const i64 nf) {
  double acc[3 * DNLP_MAXNB + 1];
#pragma unroll
  for (int k = 0; k < 3 * DNLP_MAXNB + 1; ++k) acc[k] = 0.0;
  const bool act = S->done == 0 && S->accept != 0;
  if (act) {
    const int M = S->M, nb = 2 * M + 1, GR = 2 * M, head = S->head;
    const bool run = S->phase != 0;
    const double step = S->step;
    for (i64 i = static_cast<i64>(blockIdx.x) * 256 + threadIdx.x; i < nf; i += static_cast<i64>(gridDim.x) * 256) {
      const double gnew = gt[i];
      const double sv = run ? step * dir[i] : 0.0;
      const double yv = run ? gnew - BV[static_cast<i64>(GR) * nf + i] : 0.0;
      if (run) {
        x[i] += sv;
        BV[static_cast<i64>(head) * nf + i] = sv;
        BV[static_cast<i64>(M + head) * nf + i] = yv;
      }
      BV[static_cast<i64>(GR) * nf + i] = gnew;
#pragma unroll
      for (int j = 0; j < DNLP_MAXNB; ++j) {
        if (j < nb) {
          const double r = (run && j == head) ? sv : (run && j == M + head) ? yv : (j == GR) ? gnew : BV[static_cast<i64>(j) * nf + i];
          acc[j] += sv * r;
          acc[DNLP_MAXNB + j] += yv * r;
          acc[2 * DNLP_MAXNB + j] += gnew * r;
        }
      }
      acc[3 * DNLP_MAXNB] = fmax(acc[3 * DNLP_MAXNB], fabs(gnew));
    }
  }
  dnlp_block_reduce_store<3 * DNLP_MAXNB + 1>(acc, 1, upart + static_cast<i64>(3 * DNLP_MAXNB + 1) * blockIdx.x);
}

// ---- Gram rows, history bookkeeping, convergence, two-loop recursion (one workgroup of 128) -----------
extern "C" __global__ void __launch_bounds__(128) dnlp_lb_control(LbfgsState* __restrict__ S, const double* __restrict__ upart) {
  __shared__ double red[3 * DNLP_MAXNB + 1];
  if (S->done != 0 || S->accept == 0) return;
  const int NV = 3 * DNLP_MAXNB + 1;
  if (static_cast<int>(threadIdx.x) < NV) {
    const int k = threadIdx.x;
    double a = (k == NV - 1) ? -1.0 : 0.0;
    for (int b = 0; b < S->nblocks; ++b) { const double v = upart[static_cast<i64>(NV) * b + k]; a = (k == NV - 1) ? fmax(a, v) : a + v; }
    red[k] = a;
  }
  DNLP_SYNC();
  if (threadIdx.x != 127) return;
  const int M = S->M, nb = 2 * M + 1, GR = 2 * M;
  const int ld = DNLP_MAXNB;
  double* G = S->G;
  int head = S->head, stored = S->stored;
  const bool run = S->phase != 0;
  if (run) {
    for (int j = 0; j < nb; ++j) {
      G[head * ld + j] = G[j * ld + head] = red[j];
    }
    for (int j = 0; j < nb; ++j) {
      G[(M + head) * ld + j] = G[j * ld + (M + head)] = red[DNLP_MAXNB + j];
    }
  }
  for (int j = 0; j < nb; ++j) G[GR * ld + j] = G[j * ld + GR] = red[2 * DNLP_MAXNB + j];
  if (run) {
    const double sy = G[head * ld + (M + head)], ss = G[head * ld + head], yy = G[(M + head) * ld + (M + head)];
    if (sy > 1e-10 * sqrt(ss) * sqrt(yy)) {
      S->rho[head] = 1.0 / sy;
      head = (head + 1) % M;
      if (stored < M) ++stored;
    }
    S->iter += 1;
  }
  S->head = head;
  S->stored = stored;
  S->f = S->fn;
  S->phase = 1;
  S->accept = 0;
  S->ls = 0;
  const double gn = red[NV - 1], f = S->f;
  S->gn = gn;
  if (gn <= S->tol * fmax(1.0, fabs(f))) { S->done = 1; return; }
  if (S->iter >= S->max_iter) { S->done = 3; return; }
  // two-loop recursion on the coefficients of q in the basis (lbfgs_core.h)
  double* coef = S->coef;
  for (int j = 0; j < nb; ++j) coef[j] = 0.0;
  coef[GR] = 1.0;
  for (int j = 0; j < stored; ++j) {
    const int idx = (head - 1 - j + 2 * M) % M;
    double v = 0.0;
    for (int q = 0; q < nb; ++q) v += coef[q] * G[idx * ld + q];
    const double a = S->rho[idx] * v;
    S->alpha[idx] = a;
    coef[M + idx] -= a;
  }
  if (stored > 0) {
    const int idx = (head - 1 + M) % M;
    const double gam = G[idx * ld + (M + idx)] / G[(M + idx) * ld + (M + idx)];
    for (int j = 0; j < nb; ++j) coef[j] *= gam;
  }
  for (int j = stored - 1; j >= 0; --j) {
    const int idx = (head - 1 - j + 2 * M) % M;
    double v = 0.0;
    for (int q = 0; q < nb; ++q) v += coef[q] * G[(M + idx) * ld + q];
    coef[idx] += S->alpha[idx] - S->rho[idx] * v;
  }
  for (int j = 0; j < nb; ++j) coef[j] = -coef[j];
  double gd = 0.0;
  for (int q = 0; q < nb; ++q) gd += coef[q] * G[GR * ld + q];
  if (!(gd < 0.0)) {                                 // not a descent direction: steepest descent, history dropped
    S->stored = stored = 0;
    for (int j = 0; j < nb; ++j) coef[j] = 0.0;
    coef[GR] = -1.0;
    gd = -G[GR * ld + GR];
  }
  S->gd = gd;
  S->step = (S->iter == 0 && stored == 0) ? fmin(1.0, 1.0 / fmax(gn, 1e-300)) : 1.0;
}
)DNLPLB";
  return s;
}

}  // namespace dnlp
