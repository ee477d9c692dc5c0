// HIP execution space for gfx950 (MI355X): the only execution space of libdnlp_hip.so.
//
// * map / sum / max / min instantiate the single-source DNLP_HD lambdas of model.h and
//   ipm_core.h as grid-wide kernels (64-wide wavefront shuffles -> LDS -> per-block partial ->
//   pinned host finish; one stream, no host sync except the scalar read).
// * gemv_sym, dense_block_add, coo_* are hand-written bandwidth kernels (16-B/lane coalesced
//   column-major reads, deterministic two-stage reduction instead of float atomics for the
//   dense product).
// * ldlt_* : Bunch-Kaufman LDL^T for small/medium orders (per-step pivot kernel + grid-wide
//   trailing update), blocked unpivoted LDL^T with an FP64-MFMA Schur-complement update for
//   large orders (ldlt_blocked.h).
#pragma once
#include <hip/hip_runtime.h>

#include <stdexcept>
#include <string>
#include <functional>
#include <vector>
#include <map>
#include <tuple>

#include "atom_math.h"
#include "exec.h"
#include "wave_ops.h"
#include "bk_panel.h"
#include <chrono>
#include "fused_obj.h"
#include "fused_codegen.h"
#include "lbfgs_codegen.h"
#include "fused_rtc.h"
#include "sparse_ldl.h"

namespace dnlp {

#define DNLP_HIP_CHECK(call)                                                                   \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      throw std::runtime_error(std::string("HIP error ") + hipGetErrorString(e_) + " at " +    \
                               __FILE__ + ":" + std::to_string(__LINE__));                    \
  } while (0)

constexpr int kBlock = 256;
constexpr int kMaxPartials = 2048;

// HIP caps gridDim.x * blockDim.x below 2^32 threads per launch: large index spaces are
// covered by several launches with an element offset.
constexpr i64 kMaxLaunchElems = (static_cast<i64>(1) << 31);

#define DNLP_LAUNCH_CHECK() DNLP_HIP_CHECK(hipGetLastError())

template <class F>
__global__ void __launch_bounds__(kBlock) map_kernel(i64 off, i64 n, F f) {
  const i64 i = off + static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x;
  if (i < n) f(i);
}

__device__ inline double wave_sum(double v) { return wave_all_sum(v); }
__device__ inline double wave_max(double v) { return wave_all_max(v); }

// mode 0 sum, 1 max (NaN -> +inf), 2 min (NaN -> -inf)
template <int MODE, class F>
__global__ void __launch_bounds__(kBlock) reduce_kernel(i64 n, F f, double* partial) {
  __shared__ double sm[kBlock / 64];
  double acc = MODE == 0 ? 0.0 : (MODE == 1 ? -kInf : kInf);
  for (i64 i = static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x; i < n;
       i += static_cast<i64>(gridDim.x) * kBlock) {
    double v = f(i);
    if (MODE == 0) acc += v;
    else if (MODE == 1) acc = fmax(acc, v != v ? kInf : v);
    else acc = fmin(acc, v != v ? -kInf : v);
  }
  if (MODE == 2) acc = -acc;                       // min via max
  double w = MODE == 0 ? wave_sum(acc) : wave_max(acc);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) sm[wid] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    double r = sm[0];
    for (int k = 1; k < kBlock / 64; ++k) r = MODE == 0 ? r + sm[k] : fmax(r, sm[k]);
    partial[blockIdx.x] = MODE == 2 ? -r : r;
  }
}

// NM maxima and NS sums in one pass; partial[(NM + NS) b + k]
template <int NM, int NS, class F>
__global__ void __launch_bounds__(kBlock) reduce_multi_kernel(i64 n, F f, double* partial) {
  __shared__ double sm[(NM + NS) * (kBlock / 64)];
  double am[NM > 0 ? NM : 1], as[NS > 0 ? NS : 1];
#pragma unroll
  for (int k = 0; k < NM; ++k) am[k] = -kInf;
#pragma unroll
  for (int k = 0; k < NS; ++k) as[k] = 0.0;
  for (i64 i = static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<i64>(gridDim.x) * kBlock) {
    const RMulti v = f(i);
#pragma unroll
    for (int k = 0; k < NM; ++k) am[k] = fmax(am[k], v.mx[k] != v.mx[k] ? kInf : v.mx[k]);
#pragma unroll
    for (int k = 0; k < NS; ++k) as[k] += v.sm[k];
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NM; ++k) { const double w = wave_max(am[k]); if (lane == 0) sm[wid * (NM + NS) + k] = w; }
#pragma unroll
  for (int k = 0; k < NS; ++k) { const double w = wave_sum(as[k]); if (lane == 0) sm[wid * (NM + NS) + NM + k] = w; }
  __syncthreads();
  if (threadIdx.x < NM + NS) {
    const int k = threadIdx.x;
    double r = sm[k];
    for (int w = 1; w < kBlock / 64; ++w) r = k < NM ? fmax(r, sm[w * (NM + NS) + k]) : r + sm[w * (NM + NS) + k];
    partial[(NM + NS) * blockIdx.x + k] = r;
  }
}

// two minima in one pass (NaN -> -inf, as mode 2 above); partial[2 b], partial[2 b + 1]
template <class F>
__global__ void __launch_bounds__(kBlock) reduce_min2_kernel(i64 n, F f, double* partial) {
  __shared__ double sm[2 * (kBlock / 64)];
  double a0 = -kInf, a1 = -kInf;
  for (i64 i = static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x; i < n; i += static_cast<i64>(gridDim.x) * kBlock) {
    const D2 v = f(i);
    a0 = fmax(a0, v.first != v.first ? kInf : -v.first);
    a1 = fmax(a1, v.second != v.second ? kInf : -v.second);
  }
  a0 = wave_max(a0);
  a1 = wave_max(a1);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sm[2 * wid] = a0; sm[2 * wid + 1] = a1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double r0 = sm[0], r1 = sm[1];
    for (int k = 1; k < kBlock / 64; ++k) { r0 = fmax(r0, sm[2 * k]); r1 = fmax(r1, sm[2 * k + 1]); }
    partial[2 * blockIdx.x] = -r0;
    partial[2 * blockIdx.x + 1] = -r1;
  }
}

// ---- dense symmetric product y = A x, column-major ----------------------------------------
// Stage 1: block (rb, cb) owns 512 rows x GEMV_CB columns; each lane streams two adjacent rows
// (16 B per lane, 1 KiB per wave instruction, fully coalesced down a column) and keeps the two
// running sums in registers; x of the chunk is wave-uniform.  Partial sums go to part[cb][row].
// Stage 2 sums the column chunks in a fixed order (bitwise reproducible, no float atomics).
constexpr int GEMV_CB = 256;
__global__ void __launch_bounds__(kBlock) gemv_stage1(i64 n, const double* __restrict__ A, i64 ld,
                                                      const double* __restrict__ x,
                                                      double* __restrict__ part, i64 nrb) {
  const i64 rb = blockIdx.x % nrb, cb = blockIdx.x / nrb;
  const i64 r0 = rb * 512 + 2 * static_cast<i64>(threadIdx.x);
  const i64 c0 = cb * GEMV_CB;
  const i64 c1 = c0 + GEMV_CB < n ? c0 + GEMV_CB : n;
  double a0 = 0.0, a1 = 0.0;
  if (r0 + 1 < n && (ld & 1) == 0 && ((reinterpret_cast<uintptr_t>(A) & 15) == 0)) {
    const double* col = A + r0 + c0 * ld;
    i64 j = c0;
    for (; j + 4 <= c1; j += 4) {
      const double2 v0 = *reinterpret_cast<const double2*>(col);
      const double2 v1 = *reinterpret_cast<const double2*>(col + ld);
      const double2 v2 = *reinterpret_cast<const double2*>(col + 2 * ld);
      const double2 v3 = *reinterpret_cast<const double2*>(col + 3 * ld);
      const double x0 = x[j], x1 = x[j + 1], x2 = x[j + 2], x3 = x[j + 3];
      a0 += v0.x * x0 + v1.x * x1 + v2.x * x2 + v3.x * x3;
      a1 += v0.y * x0 + v1.y * x1 + v2.y * x2 + v3.y * x3;
      col += 4 * ld;
    }
    for (; j < c1; ++j) {
      const double2 v = *reinterpret_cast<const double2*>(col);
      a0 += v.x * x[j];
      a1 += v.y * x[j];
      col += ld;
    }
  } else {
    for (i64 j = c0; j < c1; ++j) {
      if (r0 < n) a0 += A[r0 + j * ld] * x[j];
      if (r0 + 1 < n) a1 += A[r0 + 1 + j * ld] * x[j];
    }
  }
  if (r0 < n) part[cb * n + r0] = a0;
  if (r0 + 1 < n) part[cb * n + r0 + 1] = a1;
}
__global__ void __launch_bounds__(kBlock) gemv_stage2(i64 n, i64 ncb, const double* __restrict__ part,
                                                      double* __restrict__ y) {
  const i64 r = static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x;
  if (r >= n) return;
  double s = 0.0;
  for (i64 c = 0; c < ncb; ++c) s += part[c * n + r];
  y[r] = s;
}

// The same product from the LOWER triangle only (A symmetric, stored in full): half the bytes.  A block of 512 rows x
// GEMV_CB columns on or below the diagonal gives BOTH contributions of its entries: the row sums y_i += a_ij x_j as
// above, and the column sums y_j += a_ij x_i — per column a DPP wave reduction of the two products a lane holds
// (the VALU work grows from 2 to ~22 instructions per 1 KiB of matrix, still a third of what the memory rate allows),
// kept by lane j mod 64, the four wavefronts' sums combined through LDS into part2[rb][column].  Blocks that the
// diagonal crosses predicate per element (i >= j for the row sum, i > j for the column sum); blocks above it exit.
// Stage 2 adds, in a fixed order, the row-sum partials of the column blocks left of the diagonal and the column-sum
// partials of the row blocks below it: bitwise reproducible like the full product.
__global__ void __launch_bounds__(kBlock) gemv_sym_stage1(i64 n, const double* __restrict__ A, i64 ld,
                                                          const double* __restrict__ x, double* __restrict__ part,
                                                          double* __restrict__ part2, i64 nrb) {
  __shared__ double tl[kBlock / 64][GEMV_CB];
  const i64 rb = blockIdx.x % nrb, cb = blockIdx.x / nrb;
  const i64 row0 = rb * 512, c0 = cb * GEMV_CB;
  if (row0 + 512 <= c0) return;                                   // every row above every column of the block
  const i64 c1 = c0 + GEMV_CB < n ? c0 + GEMV_CB : n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const i64 r0 = row0 + 2 * static_cast<i64>(threadIdx.x);
  const bool in0 = r0 < n, in1 = r0 + 1 < n;
  const double xr0 = in0 ? x[r0] : 0.0, xr1 = in1 ? x[r0 + 1] : 0.0;
  const bool crossing = row0 < c0 + GEMV_CB;                      // some entries of the block lie above the diagonal
  double a0 = 0.0, a1 = 0.0;
  double keep[GEMV_CB / 64];
#pragma unroll
  for (int q = 0; q < GEMV_CB / 64; ++q) keep[q] = 0.0;
  const double* col = A + (in0 ? r0 : 0) + c0 * ld;
  // eight columns per trip, their loads issued together (one 16-byte load per lane and trip in flight drew 2.0 TB/s)
  constexpr int U = 8;
  for (i64 j0 = c0; j0 < c1; j0 += U, col += U * ld) {
    double2 v[U];
    double xj[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool live = j0 + u < c1;
      v[u] = double2{0.0, 0.0};
      if (live && in1) v[u] = *reinterpret_cast<const double2*>(col + u * ld);          // (every lane of the wavefront stays in the
      else if (live && in0) v[u].x = col[u * ld];                                         //  loop: the reduction below is wave-wide)
      xj[u] = live ? x[j0 + u] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const i64 j = j0 + u;
      double t;
      if (!crossing) {
        a0 = fma(v[u].x, xj[u], a0);
        a1 = fma(v[u].y, xj[u], a1);
        t = fma(v[u].x, xr0, v[u].y * xr1);
      } else {
        a0 += (r0 >= j) ? v[u].x * xj[u] : 0.0;
        a1 += (r0 + 1 >= j) ? v[u].y * xj[u] : 0.0;
        t = ((r0 > j) ? v[u].x * xr0 : 0.0) + ((r0 + 1 > j) ? v[u].y * xr1 : 0.0);
      }
      const double T = wave_all_sum(t);
      const int jj = static_cast<int>(j - c0);
#pragma unroll
      for (int q = 0; q < GEMV_CB / 64; ++q) if ((jj >> 6) == q && lane == (jj & 63)) keep[q] = T;
    }
  }
  if (in0) part[cb * n + r0] = a0;
  if (in1) part[cb * n + r0 + 1] = a1;
#pragma unroll
  for (int q = 0; q < GEMV_CB / 64; ++q) tl[wave][lane + 64 * q] = keep[q];
  __syncthreads();
  const i64 jc = c0 + threadIdx.x;
  if (jc < n) {
    double sacc = tl[0][threadIdx.x];
    for (int w2 = 1; w2 < kBlock / 64; ++w2) sacc += tl[w2][threadIdx.x];
    part2[rb * n + jc] = sacc;
  }
}
__global__ void __launch_bounds__(kBlock) gemv_sym_stage2(i64 n, i64 nrb, const double* __restrict__ part,
                                                          const double* __restrict__ part2, double* __restrict__ y) {
  const i64 r = static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x;
  if (r >= n) return;
  double s = 0.0;
  const i64 cmax = ((r / 512) * 512 + 511) / GEMV_CB;             // last column block the row's row block touched
  for (i64 c = 0; c <= cmax && c * GEMV_CB < n; ++c) s += part[c * n + r];
  const i64 bmin = ((r / GEMV_CB) * GEMV_CB) / 512;               // first row block that touched the column's column block
  for (i64 b = bmin; b < nrb; ++b) s += part2[b * n + r];
  y[r] = s;
}

// K[x0+r, x0+c] (+)= w P[r,c] on r >= c.  grid.x walks 512-row tiles (two rows per lane, 16-B
// accesses down the column), grid.y strides over columns; tiles above the diagonal exit.
__global__ void __launch_bounds__(kBlock) dense_block_add_kernel(double* __restrict__ K, i64 ldk, i64 x0,
                                                                 const double* __restrict__ P, i64 ldp,
                                                                 i64 nb, double w, int set) {
  const i64 r = static_cast<i64>(blockIdx.x) * 512 + 2 * static_cast<i64>(threadIdx.x);
  const bool vec = ((ldk | ldp | x0) & 1) == 0 && ((reinterpret_cast<uintptr_t>(K) | reinterpret_cast<uintptr_t>(P)) & 15) == 0;
  for (i64 c = blockIdx.y; c < nb; c += gridDim.y) {
    if (static_cast<i64>(blockIdx.x) * 512 + 511 < c) continue;
    if (r >= nb) continue;
    double* dst = K + (x0 + r) + (x0 + c) * ldk;
    const double* src = P + r + c * ldp;
    if (vec && r >= c && r + 1 < nb) {
      const double2 pv = *reinterpret_cast<const double2*>(src);
      double2 kv = set ? double2{0.0, 0.0} : *reinterpret_cast<const double2*>(dst);
      kv.x += w * pv.x;
      kv.y += w * pv.y;
      *reinterpret_cast<double2*>(dst) = kv;
    } else {
      if (r >= c) dst[0] = (set ? 0.0 : dst[0]) + w * src[0];
      if (r + 1 >= c && r + 1 < nb) dst[1] = (set ? 0.0 : dst[1]) + w * src[1];
    }
  }
}

// COO products on the row-major sorted patterns of the tape (Jacobian, lower Hessian): the
// contributions to out[r] of one wavefront are contiguous runs of equal rows, so they are summed
// with a segmented shuffle scan and leave as ONE atomic per run (a dense 1e4-entry Jacobian row
// would otherwise serialise 64 same-address atomics per wavefront); the out[c] side (transpose,
// symmetric mirror) has distinct neighbouring columns and goes out as plain atomics.
__device__ inline void coo_row_add(double* out, i32 row, double val, bool active) {
  const int lane = threadIdx.x & 63;
  if (!active) row = -1 - lane;                 // unique, never merged, never written
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const double tv = __shfl_up(val, o, 64);
    const i32 tr = __shfl_up(row, o, 64);
    if (lane >= o && tr == row) val += tv;
  }
  const i32 nxt = __shfl_down(row, 1, 64);
  if (active && (lane == 63 || nxt != row)) unsafeAtomicAdd(&out[row], val);
}

__global__ void __launch_bounds__(kBlock) coo_mult_kernel(i64 nnz, const i32* __restrict__ r,
                                                          const i32* __restrict__ c, const double* __restrict__ a,
                                                          const double* __restrict__ v, double* out, int mode) {
  const i64 p = static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x;
  const bool active = p < nnz;
  const double av = active ? a[p] : 0.0;
  const i32 rp = active ? r[p] : 0, cp = active ? c[p] : 0;
  if (mode == 0) {
    coo_row_add(out, rp, active ? av * v[cp] : 0.0, active);
  } else if (mode == 1) {
    if (active) unsafeAtomicAdd(&out[cp], av * v[rp]);
  } else {
    coo_row_add(out, rp, active ? av * v[cp] : 0.0, active);
    if (active && rp != cp) unsafeAtomicAdd(&out[cp], av * v[rp]);
  }
}

// Order-fixed COO product: the entries that feed one output are a segment of an index built once per pattern
// (tape.h CooIdx: counting sort by output, storage order inside a segment), sixteen lanes walk a segment in
// strides and reduce in a fixed tree, ONE lane assigns the sum to out[g].  No atomics: the sum of an output is rounded
// in the same order on every run (with atomics portfolio construction ended after 22 to 72 iterations from run to
// run, with this after 22 every time: profiles/r03_determinism.txt).
__global__ void __launch_bounds__(kBlock) coo_rows_kernel(i64 nout, const i32* __restrict__ ptr, const i32* __restrict__ ent,
                                                          const i32* __restrict__ src, const double* __restrict__ a,
                                                          const double* __restrict__ v, double* out) {
  const i64 g = (static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x) >> 4;
  const int l = threadIdx.x & 15;
  double s = 0.0;
  i64 p0 = 0, p1 = 0;
  if (g < nout) { p0 = ptr[g]; p1 = ptr[g + 1]; }
  for (i64 p = p0 + l; p < p1; p += 16) s += a[ent[p]] * v[src[p]];
  s += __shfl_xor(s, 8, 16);
  s += __shfl_xor(s, 4, 16);
  s += __shfl_xor(s, 2, 16);
  s += __shfl_xor(s, 1, 16);
  if (l == 0 && g < nout) out[g] = s;      // assigned (an output without entries becomes 0): no memset before the product
}

// Products with a rectangular Jacobian stored row-major (tape.h jac_rect_cols: every row carries the same L columns),
// order-fixed without an index.  out[r] += sum_k a[r L + k] v[col[k]]: one wavefront per row, coalesced, DPP tree.
__global__ void __launch_bounds__(kBlock) rect_mult_kernel(i64 rows, i64 L, const i32* __restrict__ col, const double* __restrict__ a,
                                                           const double* __restrict__ v, double* out) {
  const i64 r = static_cast<i64>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const double* ar = a + r * L;
  double s0 = 0.0, s1 = 0.0;
  i64 k = lane;
  for (; k + 64 < L; k += 128) { s0 += ar[k] * v[col[k]]; s1 += ar[k + 64] * v[col[k + 64]]; }
  if (k < L) s0 += ar[k] * v[col[k]];
  const double s = wave_sum(s0 + s1);
  if (lane == 0) out[r] += s;
}
// out[col[k]] += sum_r a[r L + k] v[r]: a lane owns a column position k (coalesced along k), grid.y walks chunks of
// rows into partial[chunk][k]; the finish kernel adds the chunks in order.
__global__ void __launch_bounds__(kBlock) rect_tmult_partial_kernel(i64 rows, i64 L, int rows_per_chunk, const double* __restrict__ a,
                                                                    const double* __restrict__ v, double* __restrict__ partial) {
  const i64 k = static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x;
  if (k >= L) return;
  const i64 r0 = static_cast<i64>(blockIdx.y) * rows_per_chunk;
  i64 r1 = r0 + rows_per_chunk;
  if (r1 > rows) r1 = rows;
  double s0 = 0.0, s1 = 0.0;
  i64 r = r0;
  for (; r + 1 < r1; r += 2) { s0 += a[r * L + k] * v[r]; s1 += a[(r + 1) * L + k] * v[r + 1]; }
  if (r < r1) s0 += a[r * L + k] * v[r];
  partial[static_cast<i64>(blockIdx.y) * L + k] = s0 + s1;
}
__global__ void __launch_bounds__(kBlock) rect_tmult_finish_kernel(i64 L, int nchunks, const i32* __restrict__ col,
                                                                   const double* __restrict__ partial, double* out) {
  const i64 k = static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x;
  if (k >= L) return;
  double s = 0.0;
  for (int c = 0; c < nchunks; ++c) s += partial[static_cast<i64>(c) * L + k];
  out[col[k]] += s;
}

// ---- Bunch-Kaufman LDL^T (DSYTF2 semantics, lower): state, argmax and the panel body live in bk_panel.h ----
constexpr int BK_T = 1024;

__global__ void __launch_bounds__(BK_T) bk_pivot_kernel(double* A, int n, i64 ld, int* ipiv, BkState* st) {
  __shared__ double sv[BK_T / 64];
  __shared__ int si[BK_T / 64];
  __shared__ int s_k, s_kstep, s_kp, s_imax;
  __shared__ double s_colmax, s_rowmax;
  const int tid = threadIdx.x;
  const double alpha = 0.6403882032022076;   // (1 + sqrt(17)) / 8
  int k = st->k;
  if (st->pending) {
    const int ks = st->kstep;
    if (ks == 1) {
      const double inv = 1.0 / st->d11;
      for (int i = k + 1 + tid; i < n; i += BK_T) A[i + static_cast<i64>(k) * ld] *= inv;
    } else {
      const double d11 = st->d11, d22 = st->d22, d21 = st->d21;
      for (int j = k + 2 + tid; j < n; j += BK_T) {
        const double ajk = A[j + static_cast<i64>(k) * ld], ajk1 = A[j + static_cast<i64>(k + 1) * ld];
        A[j + static_cast<i64>(k) * ld] = d21 * (d11 * ajk - ajk1);
        A[j + static_cast<i64>(k + 1) * ld] = d21 * (d22 * ajk1 - ajk);
      }
    }
    k += ks;
    __syncthreads();
  }
  if (k >= n) {
    if (tid == 0) { st->k = k; st->pending = 0; st->kstep = 0; }
    return;
  }
  // column maximum below the diagonal
  double v = -1.0;
  int idx = n;
  for (int i = k + 1 + tid; i < n; i += BK_T) {
    const double a = fabs(A[i + static_cast<i64>(k) * ld]);
    if (a > v || (a == v && i < idx)) { v = a; idx = i; }
  }
  double colmax;
  int imax;
  bk_argmax(v, idx, sv, si, colmax, imax);
  if (colmax < 0.0) { colmax = 0.0; imax = k; }
  const double akk = A[k + static_cast<i64>(k) * ld];
  const double absakk = fabs(akk);
  int kstep = 1, kp = k;
  bool zero_piv = false, bad = !(absakk == absakk) || !(colmax == colmax);
  if (!bad && fmax(absakk, colmax) == 0.0) {
    zero_piv = true;
  } else if (!bad && absakk < alpha * colmax) {
    // row maximum of row/column imax in the trailing matrix
    double rv = 0.0;
    for (int j = k + tid; j < imax; j += BK_T) rv = fmax(rv, fabs(A[imax + static_cast<i64>(j) * ld]));
    for (int i = imax + 1 + tid; i < n; i += BK_T) rv = fmax(rv, fabs(A[i + static_cast<i64>(imax) * ld]));
    double rowmax;
    int dummy;
    bk_argmax(rv, tid, sv, si, rowmax, dummy);
    const double aii = fabs(A[imax + static_cast<i64>(imax) * ld]);
    if (absakk >= alpha * colmax * (colmax / rowmax)) kp = k;
    else if (aii >= alpha * rowmax) kp = imax;
    else { kp = imax; kstep = 2; }
  }
  if (bad) {
    if (tid == 0) { st->fail = 1; st->k = n; st->pending = 0; st->kstep = 0; }
    return;
  }
  const int kk = k + kstep - 1;
  if (!zero_piv && kp != kk) {
    // symmetric interchange of rows/columns kk and kp in the trailing matrix
    for (int i = kp + 1 + tid; i < n; i += BK_T) {
      const double t = A[i + static_cast<i64>(kk) * ld];
      A[i + static_cast<i64>(kk) * ld] = A[i + static_cast<i64>(kp) * ld];
      A[i + static_cast<i64>(kp) * ld] = t;
    }
    for (int j = kk + 1 + tid; j < kp; j += BK_T) {
      const double t = A[j + static_cast<i64>(kk) * ld];
      A[j + static_cast<i64>(kk) * ld] = A[kp + static_cast<i64>(j) * ld];
      A[kp + static_cast<i64>(j) * ld] = t;
    }
    // ... and in the columns already factored, so that L ends as ONE unit-lower factor of P A P^T
    // (the blocked solve below needs no interchange between its column blocks)
    for (int j = tid; j < k; j += BK_T) {
      const double t = A[kk + static_cast<i64>(j) * ld];
      A[kk + static_cast<i64>(j) * ld] = A[kp + static_cast<i64>(j) * ld];
      A[kp + static_cast<i64>(j) * ld] = t;
    }
    __syncthreads();
    if (tid == 0) {
      const double t = A[kk + static_cast<i64>(kk) * ld];
      A[kk + static_cast<i64>(kk) * ld] = A[kp + static_cast<i64>(kp) * ld];
      A[kp + static_cast<i64>(kp) * ld] = t;
      if (kstep == 2) {
        const double t2 = A[k + 1 + static_cast<i64>(k) * ld];
        A[k + 1 + static_cast<i64>(k) * ld] = A[kp + static_cast<i64>(k) * ld];
        A[kp + static_cast<i64>(k) * ld] = t2;
      }
    }
    __syncthreads();
  }
  if (tid == 0) {
    st->k = k;
    st->kstep = kstep;
    st->kp = kp;
    st->pending = 1;
    if (zero_piv) {
      A[k + static_cast<i64>(k) * ld] = 1e-20;
      st->nzero += 1;
      st->d11 = 1e-20;
      ipiv[k] = k + 1;
    } else if (kstep == 1) {
      const double d = A[k + static_cast<i64>(k) * ld];
      st->d11 = d;
      if (d < 0.0) st->nneg += 1;
      if (fabs(d) < 1e-300) st->nzero += 1;
      ipiv[k] = kp + 1;
    } else {
      double d21 = A[k + 1 + static_cast<i64>(k) * ld];
      const double d11 = A[k + 1 + static_cast<i64>(k + 1) * ld] / d21;
      const double d22 = A[k + static_cast<i64>(k) * ld] / d21;
      const double t = 1.0 / (d11 * d22 - 1.0);
      d21 = t / d21;
      st->d11 = d11; st->d22 = d22; st->d21 = d21;
      st->nneg += 1;
      ipiv[k] = -(kp + 1);
      ipiv[k + 1] = -(kp + 1);
    }
  }
}

// Trailing update of one pivot step over the whole chip: tile = 256 rows x 16 columns.
__global__ void __launch_bounds__(kBlock) bk_update_kernel(double* A, int n, i64 ld, const BkState* st) {
  const int k = st->k, ks = st->kstep;
  if (!st->pending || k >= n) return;
  const int j0 = k + ks;                       // first trailing column
  const int t = n - j0;
  if (t <= 0) return;
  const int ntr = (t + kBlock - 1) / kBlock;
  const int rb = blockIdx.x % ntr, cbk = blockIdx.x / ntr;
  const int cstart = j0 + cbk * 16;
  if (cstart >= n) return;
  const int i = j0 + rb * kBlock + threadIdx.x;
  if (j0 + rb * kBlock + kBlock - 1 < cstart) return;   // tile above the diagonal
  if (i >= n) return;
  if (ks == 1) {
    const double inv = 1.0 / st->d11;
    const double ai = A[i + static_cast<i64>(k) * ld];
    if (ai == 0.0) return;
#pragma unroll 4
    for (int j = cstart; j < cstart + 16 && j < n; ++j) {
      if (i < j) break;
      const double wj = A[j + static_cast<i64>(k) * ld] * inv;
      A[i + static_cast<i64>(j) * ld] -= ai * wj;
    }
  } else {
    const double d11 = st->d11, d22 = st->d22, d21 = st->d21;
    const double ai0 = A[i + static_cast<i64>(k) * ld], ai1 = A[i + static_cast<i64>(k + 1) * ld];
    for (int j = cstart; j < cstart + 16 && j < n; ++j) {
      if (i < j) break;
      const double ajk = A[j + static_cast<i64>(k) * ld], ajk1 = A[j + static_cast<i64>(k + 1) * ld];
      const double wk = d21 * (d11 * ajk - ajk1), wkp1 = d21 * (d22 * ajk1 - ajk);
      A[i + static_cast<i64>(j) * ld] -= ai0 * wk + ai1 * wkp1;
    }
  }
}

template <int ROWS, int NBP>
__global__ void __launch_bounds__(BK_PT) bk_panel_kernel(double* A, int n, i64 ld, int* ipiv, BkState* st, double* Wg,
                                                        i64 ldw, BkPanelSwaps* swaps) {
  bk_panel_body<ROWS, NBP, BK_PT>(A, n, ld, ipiv, st, Wg, ldw, swaps);
}

// After a panel: A22 -= W21 L21^T (columns st->kp .. st->kp + st->kstep just finished; tile = 256 rows x 16
// columns, the row's W in registers, the tile's multipliers in LDS), and the panel's interchanges applied to
// the rows of the columns of EARLIER panels (one lane per column, the swaps in order): blockIdx.y == 1.
template <int NBP>
__global__ void __launch_bounds__(kBlock) bk_panel_update_kernel(double* __restrict__ A, int n, i64 ld,
                                                                 const double* __restrict__ Wg, i64 ldw,
                                                                 const BkState* __restrict__ st,
                                                                 const BkPanelSwaps* __restrict__ swaps) {
  __shared__ double Ls[16][NBP + 1];
  const int k0 = st->kp, cnt = st->kstep, j0 = k0 + cnt;
  if (!st->pending || cnt <= 0) return;
  if (blockIdx.y == 1) {
    const int q = blockIdx.x * kBlock + threadIdx.x;
    const int ns = swaps->count;
    if (q < k0 && ns > 0) {
      double* col = A + static_cast<i64>(q) * ld;
      for (int e = 0; e < ns; ++e) {
        const int ra = swaps->rows[2 * e], rb = swaps->rows[2 * e + 1];
        const double t = col[ra];
        col[ra] = col[rb];
        col[rb] = t;
      }
    }
    return;
  }
  if (st->fail) return;
  const int t = n - j0;
  if (t <= 0) return;
  const int ntr = (t + kBlock - 1) / kBlock;
  const int rb = blockIdx.x % ntr, cbk = blockIdx.x / ntr;
  const int cstart = j0 + cbk * 16;
  if (cstart >= n) return;
  if (j0 + rb * kBlock + kBlock - 1 < cstart) return;   // tile above the diagonal
  for (int e = threadIdx.x; e < 16 * NBP; e += kBlock) {
    const int cc = e % 16, i = e / 16;
    Ls[cc][i] = (cstart + cc < n && i < cnt) ? A[(cstart + cc) + static_cast<i64>(k0 + i) * ld] : 0.0;
  }
  __syncthreads();
  const int r = j0 + rb * kBlock + threadIdx.x;
  if (r >= n || r < cstart) return;
  double w[NBP];
#pragma unroll
  for (int i = 0; i < NBP; ++i) w[i] = Wg[r + static_cast<i64>(i) * ldw];
#pragma unroll 4
  for (int cc = 0; cc < 16; ++cc) {
    const int cidx = cstart + cc;
    if (cidx > r || cidx >= n) break;
    double sacc = 0.0;
#pragma unroll
    for (int i = 0; i < NBP; ++i) sacc += w[i] * Ls[cc][i];
    A[r + static_cast<i64>(cidx) * ld] -= sacc;
  }
}

// After the last pivot: the interchanges as ONE permutation (x = b[perm]) and the pivot structure
// (0: 1x1, 1 / 2: first / second column of a 2x2 block), both by a sequential pass in LDS.
constexpr int BK_NMAX = 4096;            // largest order the one-workgroup solve keeps in LDS
__global__ void __launch_bounds__(BK_T) bk_finish_kernel(const int* __restrict__ ipiv, int n, int* __restrict__ perm,
                                                         int* __restrict__ dtype) {
  __shared__ int sp[BK_NMAX], sv[BK_NMAX], st[BK_NMAX];
  const int tid = threadIdx.x;
  for (int i = tid; i < n; i += BK_T) { sp[i] = i; sv[i] = ipiv[i]; }
  __syncthreads();
  if (tid == 0) {
    int k = 0;
    while (k < n) {
      if (sv[k] > 0) {
        const int kp = sv[k] - 1;
        if (kp != k) { const int t = sp[k]; sp[k] = sp[kp]; sp[kp] = t; }
        st[k] = 0;
        k += 1;
      } else {
        const int kp = -sv[k] - 1;
        if (kp != k + 1) { const int t = sp[k + 1]; sp[k + 1] = sp[kp]; sp[kp] = t; }
        st[k] = 1;
        st[k + 1] = 2;
        k += 2;
      }
    }
  }
  __syncthreads();
  for (int i = tid; i < n; i += BK_T) { perm[i] = sp[i]; dtype[i] = st[i]; }
}

// P A P^T = L D L^T solve in one workgroup with the vector in LDS: 32-column blocks, the diagonal block
// by one wavefront (lane = row, v_readlane broadcasts), the rows below / the columns' dot products by all
// sixteen; three barriers per block instead of two per column.
__global__ void __launch_bounds__(BK_T) bk_solve_kernel(const double* __restrict__ A, int n, i64 ld,
                                                        const int* __restrict__ perm, const int* __restrict__ dtype,
                                                        double* __restrict__ b) {
  __shared__ double x[BK_NMAX];
  __shared__ double Lb[32][33];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < n; i += BK_T) x[i] = b[perm[i]];
  __syncthreads();
  auto stage_diag = [&](int j0, int jb) {
    for (int e = tid; e < jb * jb; e += BK_T) {
      const int r = e % jb, c = e / jb;
      double v = (r > c) ? A[(j0 + r) + static_cast<i64>(j0 + c) * ld] : 0.0;
      if (r == c + 1 && dtype[j0 + c] == 1) v = 0.0;          // the off-diagonal of a 2x2 pivot is D, not L
      Lb[r][c] = v;
    }
  };
  // forward: L y = P b
  for (int j0 = 0; j0 < n; j0 += 32) {
    const int jb = min(32, n - j0);
    stage_diag(j0, jb);
    __syncthreads();
    if (wave == 0) {
      double y = lane < jb ? x[j0 + lane] : 0.0;
      for (int c = 0; c < jb; ++c) {
        const double yc = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(y), c),
                                           __builtin_amdgcn_readlane(__double2loint(y), c));
        if (lane > c && lane < jb) y -= Lb[lane][c] * yc;
      }
      if (lane < jb) x[j0 + lane] = y;
    }
    __syncthreads();
    const int r0 = j0 + jb;
    const bool straddle = dtype[r0 - 1] == 1;                  // 2x2 pivot across the block boundary
    for (int r = r0 + tid; r < n; r += BK_T) {
      double s = 0.0;
#pragma unroll 8
      for (int c = 0; c < jb; ++c) s += A[r + static_cast<i64>(j0 + c) * ld] * x[j0 + c];
      if (straddle && r == r0) s -= A[r + static_cast<i64>(r0 - 1) * ld] * x[r0 - 1];
      x[r] -= s;
    }
    __syncthreads();
  }
  // D z = y
  for (int k = tid; k < n; k += BK_T) {
    const int ty = dtype[k];
    if (ty == 0) {
      x[k] /= A[k + static_cast<i64>(k) * ld];
    } else if (ty == 1) {
      const double akm1k = A[k + 1 + static_cast<i64>(k) * ld];
      const double akm1 = A[k + static_cast<i64>(k) * ld] / akm1k, ak = A[k + 1 + static_cast<i64>(k + 1) * ld] / akm1k;
      const double denom = akm1 * ak - 1.0, bkm1 = x[k] / akm1k, bkk = x[k + 1] / akm1k;
      x[k] = (ak * bkm1 - bkk) / denom;
      x[k + 1] = (akm1 * bkk - bkm1) / denom;
    }
  }
  __syncthreads();
  // backward: L^T w = z
  for (int j0 = (n - 1) / 32 * 32; j0 >= 0; j0 -= 32) {
    const int jb = min(32, n - j0);
    const int r0 = j0 + jb;
    stage_diag(j0, jb);
    const bool straddle = r0 < n && dtype[r0 - 1] == 1;
    for (int c = wave; c < jb; c += BK_T / 64) {
      double s = 0.0;
      for (int r = r0 + lane; r < n; r += 64) s += A[r + static_cast<i64>(j0 + c) * ld] * x[r];
      s = wave_sum(s);
      if (lane == 0) {
        if (straddle && c == jb - 1) s -= A[r0 + static_cast<i64>(r0 - 1) * ld] * x[r0];
        x[j0 + c] -= s;
      }
    }
    __syncthreads();
    if (wave == 0) {
      double w = lane < jb ? x[j0 + lane] : 0.0;
      for (int c = jb - 1; c >= 0; --c) {
        const double wc = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(w), c),
                                           __builtin_amdgcn_readlane(__double2loint(w), c));
        if (lane < c) w -= Lb[c][lane] * wc;
      }
      if (lane < jb) x[j0 + lane] = w;
    }
    __syncthreads();
  }
  for (int i = tid; i < n; i += BK_T) b[perm[i]] = x[i];
}

// ---- tape sweep over the elementwise-class segments (hand-written form of Model::sweep_flat) ---
// 512 work units per workgroup, two consecutive units per lane.  The segment of the block's
// first unit is found once (binary search by lane 0, broadcast through LDS); lanes advance from
// there.  When both units of a lane sit in one unary segment with a contiguous, 16-B aligned
// argument the lane moves double2's (x read, z / dvals / hvals written): fully coalesced 1-KiB
// wave transactions.  Everything else takes the scalar path with identical arithmetic.
struct FlatTable {
  i64 nflat, total;
  const i64* start;
  const i32* op;
  const i64 *a0b, *a0o, *a1b, *a1o, *zoff, *doff, *hoff, *n, *d0, *d1, *d2;
  const double *p, *p2;
  const i32* gidx;
};

__device__ inline void sweep_one_unit(const FlatTable& t, i64 s, i64 i, const double* __restrict__ x,
                                      double* __restrict__ z, double* __restrict__ dv, double* __restrict__ hv,
                                      const double* __restrict__ ww, bool with_h) {
  const int op = t.op[s];
  const i64 n = t.n[s];
  if (op < OP_MUL) {
    const i64 xi = t.a0b[s] >= 0 ? t.a0b[s] + i : t.gidx[t.a0o[s] + i];
    double val, g1, g2;
    unary_rules(op, x[xi], t.p[s], t.p2[s], val, g1, g2);
    z[t.zoff[s] + i] = val;
    dv[t.doff[s] + i] = g1;
    if (with_h) hv[t.hoff[s] + i] = ww[t.zoff[s] + i] * g2;
  } else if (op == OP_MUL) {
    const i64 xi = t.a0b[s] >= 0 ? t.a0b[s] + i : t.gidx[t.a0o[s] + i];
    const i64 yi = t.a1b[s] >= 0 ? t.a1b[s] + i : t.gidx[t.a1o[s] + i];
    const double u = x[xi], v = x[yi];
    z[t.zoff[s] + i] = u * v;
    dv[t.doff[s] + i] = v;
    dv[t.doff[s] + n + i] = u;
    if (with_h) hv[t.hoff[s] + i] = ww[t.zoff[s] + i];
  } else if (op == OP_REL_ENTR) {
    const i64 xi = t.a0b[s] >= 0 ? t.a0b[s] + i : t.gidx[t.a0o[s] + i];
    const i64 yi = t.a1b[s] >= 0 ? t.a1b[s] + i : t.gidx[t.a1o[s] + i];
    const double u = x[xi], v = x[yi];
    const double lr = log(u / v);
    z[t.zoff[s] + i] = u * lr;
    dv[t.doff[s] + i] = lr + 1.0;
    dv[t.doff[s] + n + i] = -u / v;
    if (with_h) {
      const double wi = ww[t.zoff[s] + i];
      hv[t.hoff[s] + i] = wi / u;
      hv[t.hoff[s] + n + i] = wi * u / (v * v);
      hv[t.hoff[s] + 2 * n + i] = -wi / v;
    }
  } else {   // OP_MATMUL
    const i64 mm = t.d0[s], kk = t.d1[s];
    const i64 r = i % mm, cidx = i / mm;
    double acc = 0.0;
    const i64 dbase = t.doff[s] + i * kk, cnt = mm * t.d2[s] * kk;
    for (i64 l = 0; l < kk; ++l) {
      const double u = x[t.gidx[t.a0o[s] + r + l * mm]], v = x[t.gidx[t.a1o[s] + l + cidx * kk]];
      acc += u * v;
      dv[dbase + l] = v;
      dv[dbase + cnt + l] = u;
      if (with_h) hv[t.hoff[s] + i * kk + l] = ww[t.zoff[s] + i];
    }
    z[t.zoff[s] + i] = acc;
  }
}

__global__ void __launch_bounds__(kBlock) sweep_flat_kernel(FlatTable t, const double* __restrict__ x,
                                                            double* __restrict__ z, double* __restrict__ dv,
                                                            double* __restrict__ hv, const double* __restrict__ ww,
                                                            int with_h) {
  __shared__ i64 s_first;
  const i64 e_blk = static_cast<i64>(blockIdx.x) * (2 * kBlock);
  if (threadIdx.x == 0) {
    i64 lo = 0, hi = t.nflat;
    while (hi - lo > 1) {
      const i64 mid = (lo + hi) >> 1;
      if (t.start[mid] <= e_blk) lo = mid; else hi = mid;
    }
    s_first = lo;
  }
  __syncthreads();
  const i64 e0 = e_blk + 2 * static_cast<i64>(threadIdx.x);
  if (e0 >= t.total) return;
  i64 s = s_first;
  while (t.start[s + 1] <= e0) ++s;
  const i64 i = e0 - t.start[s];
  const bool pair = (e0 + 1 < t.start[s + 1]);
  const int op = t.op[s];
  if (pair && op < OP_MUL && t.a0b[s] >= 0) {
    const i64 xi = t.a0b[s] + i, zi = t.zoff[s] + i, di = t.doff[s] + i, hi2 = t.hoff[s] + i;
    if (((xi | zi | di | (with_h ? hi2 : 0)) & 1) == 0) {
      const double2 u = *reinterpret_cast<const double2*>(x + xi);
      double2 val, g1, g2;
      const double p = t.p[s], p2 = t.p2[s];
      unary_rules(op, u.x, p, p2, val.x, g1.x, g2.x);
      unary_rules(op, u.y, p, p2, val.y, g1.y, g2.y);
      *reinterpret_cast<double2*>(z + zi) = val;
      *reinterpret_cast<double2*>(dv + di) = g1;
      if (with_h) {
        const double2 w2 = *reinterpret_cast<const double2*>(ww + zi);
        *reinterpret_cast<double2*>(hv + hi2) = double2{w2.x * g2.x, w2.y * g2.y};
      }
      return;
    }
  }
  sweep_one_unit(t, s, i, x, z, dv, hv, ww, with_h != 0);
  if (e0 + 1 < t.total) {
    i64 s1 = s;
    while (t.start[s1 + 1] <= e0 + 1) ++s1;
    sweep_one_unit(t, s1, e0 + 1 - t.start[s1], x, z, dv, hv, ww, with_h != 0);
  }
}

// part[q * 1024 + block] = this block's share of sum_i V[q*N + i] * w[i] for q < k (k <= 32): all k dot products in one
// sweep of w; vt_dot_finish_kernel adds the blocks' shares in block order (round 4: an atomic add per block landed in
// arrival order — the Lanczos bound of C4 and the host-driven L-BFGS's Gram rows could differ in the last bits)
__global__ void __launch_bounds__(kBlock) vt_dot_kernel(int k, const double* __restrict__ V, i64 N,
                                                        const double* __restrict__ w, double* __restrict__ part) {
  __shared__ double red[kBlock / 64][32];
  double acc[32];
#pragma unroll
  for (int q = 0; q < 32; ++q) acc[q] = 0.0;
  for (i64 i = static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x; i < N; i += static_cast<i64>(gridDim.x) * kBlock) {
    const double wi = w[i];
#pragma unroll
    for (int q = 0; q < 32; ++q)
      if (q < k) acc[q] += V[static_cast<i64>(q) * N + i] * wi;
  }
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 32; ++q) {
    if (q < k) {
      const double s = wave_sum(acc[q]);
      if (lane == 0) red[wid][q] = s;
    }
  }
  __syncthreads();
  if (threadIdx.x < k) {
    double t = 0.0;
    for (int w2 = 0; w2 < kBlock / 64; ++w2) t += red[w2][threadIdx.x];
    part[static_cast<i64>(threadIdx.x) * 1024 + blockIdx.x] = t;
  }
}
__global__ void __launch_bounds__(64) vt_dot_finish_kernel(int nblocks, const double* __restrict__ part, double* __restrict__ out) {
  // one wavefront per dot product: lanes take the blocks round robin, the fixed wavefront tree adds the lanes
  const double* row = part + static_cast<i64>(blockIdx.x) * 1024;
  double s = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 64) s += row[b];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[blockIdx.x] = s;
}
// w[i] -= sum_q c[q] V[q*N + i]
__global__ void __launch_bounds__(kBlock) v_comb_kernel(int k, const double* __restrict__ V, i64 N,
                                                        const double* __restrict__ c, double* __restrict__ out) {
  const i64 i = static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= N) return;
  double s = 0.0;
  for (int q = 0; q < k; ++q) s += c[q] * V[static_cast<i64>(q) * N + i];
  out[i] = s;
}
__global__ void __launch_bounds__(kBlock) v_axpy_kernel(int k, const double* __restrict__ V, i64 N,
                                                        const double* __restrict__ c, double* __restrict__ w) {
  const i64 i = static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x;
  if (i >= N) return;
  double s = 0.0;
  for (int q = 0; q < k; ++q) s += c[q] * V[static_cast<i64>(q) * N + i];
  w[i] -= s;
}

// y = base + M v for CSR matrices with long rows: one wavefront per row, lanes stride over the row's
// entries (coalesced index / value reads), shuffle reduction
__global__ void __launch_bounds__(kBlock) spmv_long_kernel(i64 rows, const i64* __restrict__ ptr, const i32* __restrict__ idx,
                                                           const double* __restrict__ val, const double* __restrict__ v,
                                                           const double* __restrict__ base, double* __restrict__ y) {
  const i64 r = static_cast<i64>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  double s0 = 0.0, s1 = 0.0;
  const i64 b = ptr[r], e = ptr[r + 1];
  i64 k = b + lane;
  for (; k + 64 < e; k += 128) { s0 += val[k] * v[idx[k]]; s1 += val[k + 64] * v[idx[k + 64]]; }
  if (k < e) s0 += val[k] * v[idx[k]];
  const double s = wave_sum(s0 + s1);
  if (lane == 0) y[r] = s + (base ? base[r] : 0.0);
}

// ---- fused element programs (fused_obj.h): the interpreter's register file lives in LDS --------
// slot[k][lane] (k < P.n, 256 lanes): consecutive lanes hit consecutive banks.  The program sits
// in the kernel arguments, so opcode dispatch is scalar and uniform.  f: per-lane partial ->
// wavefront shuffle -> LDS -> partial[block]; grad: hardware FP64 atomic adds (L2).
// NE elements per lane and opcode decode, WPE = occupancy target (wavefronts per SIMD) that caps
// the register budget: the interpreter is latency-bound (decode -> LDS read -> ALU -> LDS write per
// instruction), so resident wavefronts matter more than registers for the libm paths.
template <int NE, int WPE>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
fused_eval_kernel(FusedSlotProg P, const double* __restrict__ x, const double* __restrict__ consts,
                  double* __restrict__ grad, double* __restrict__ partial) {
  // dynamic LDS: [slots: nslots x NE x 256 doubles][gradient window of the current tile: 256 NE doubles]
  // — nothing static, so that 4 slots x 4 elements + window = exactly 40 KB = four workgroups per CU
  extern __shared__ double fz_slots[];
  double* mine = fz_slots + threadIdx.x;
  const i64 tile = static_cast<i64>(kBlock) * NE;
  double* gwin = fz_slots + static_cast<size_t>(P.nslots) * tile;
  double acc = 0.0;
  // adjoints that land in [tile start + win_lo, + tile) are summed in LDS; the few beyond (offsets
  // reach win_extra entries further) go out as global atomics
  const int win = P.win_extra >= 0 ? static_cast<int>(tile) : 0;
  for (int t = threadIdx.x; t < win; t += kBlock) gwin[t] = 0.0;
  __syncthreads();
  for (i64 base = static_cast<i64>(blockIdx.x) * tile; base < P.nelem; base += static_cast<i64>(gridDim.x) * tile) {
    // lane l handles elements base + l + 256 e: every load / atomic of an instruction is a
    // fully coalesced wavefront access
    bool valid[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) valid[e] = base + threadIdx.x + static_cast<i64>(e) * kBlock < P.nelem;
    const i64 wbase = base + P.win_lo;
    auto slots = [=](int k, int e) -> double& { return mine[(k * NE + e) * kBlock]; };   // slot-major: conflict-free
    auto scat = [=](i64 idx, double v) {
      const i64 t = idx - wbase;
      if (t >= 0 && t < win) unsafeAtomicAdd(&gwin[t], v);      // LDS
      else unsafeAtomicAdd(&grad[idx], v);
    };
    if (base + tile <= P.nelem) acc += fused_elements<NE, true>(P, base + threadIdx.x, kBlock, valid, x, consts, slots, scat);
    else acc += fused_elements<NE, false>(P, base + threadIdx.x, kBlock, valid, x, consts, slots, scat);
    if (win) {
      __syncthreads();
      for (int t = threadIdx.x; t < win; t += kBlock) {
        const double v = gwin[t];
        if (v != 0.0) { unsafeAtomicAdd(&grad[wbase + t], v); gwin[t] = 0.0; }
      }
      __syncthreads();
    }
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) partial[blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)] = acc;
}

// ---- static-pattern sparse LDL^T (sparse_ldl.h) driven from the host: one workgroup walks the
// pivot blocks in order (the update triples of a block are spread over the 256 lanes).
struct WgPar {
  double* red;
  template <class T> __device__ T* vec(T* p) const { return p; }
  __device__ int lanes() const { return kBlock; }
  __device__ int lane() const { return static_cast<int>(threadIdx.x); }
  __device__ void sync() const { __syncthreads(); }
  __device__ double sum(double v) const {
    v = wave_sum(v);
    __syncthreads();                      // red may still be read from the previous call
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = red[0];
    for (int k = 1; k < kBlock / 64; ++k) r += red[k];
    return r;
  }
};
struct SparseInfo { int ok, nneg, nzero, pad; };
__global__ void __launch_bounds__(kBlock) sparse_factor_kernel(SparsePlan pl, double* vals, double* w, SparseInfo* info) {
  __shared__ double red[kBlock / 64];
  int nneg = 0, nzero = 0;
  const bool ok = sparse_ldl_factor(pl, vals, w, &nneg, &nzero, WgPar{red});
  if (threadIdx.x == 0) { info->ok = ok ? 1 : 0; info->nneg = nneg; info->nzero = nzero; }
}
__global__ void __launch_bounds__(kBlock) sparse_solve_kernel(SparsePlan pl, const double* vals, double* x) {
  __shared__ double red[kBlock / 64];
  sparse_ldl_solve(pl, vals, x, WgPar{red});
}
// Wide levels (large sparse systems: 7e5-order chains have levels of 1e5 independent pivot
// blocks): one grid-wide kernel per level phase instead of one workgroup walking everything.
__global__ void sp_info_reset_kernel(SparseInfo* info) { *info = SparseInfo{1, 0, 0, 0}; }
__global__ void __launch_bounds__(kBlock) sp_pivot_kernel(SparsePlan pl, double* vals, double* dinv, i64 b0, i64 b1, SparseInfo* info) {
  const i64 k = b0 + static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x;
  if (k >= b1) return;
  double nneg = 0.0, nzero = 0.0, bad = 0.0;
  sp_pivot(pl, vals, dinv, k, nneg, nzero, bad);
  if (nneg != 0.0) atomicAdd(&info->nneg, static_cast<int>(nneg));
  if (nzero != 0.0) atomicAdd(&info->nzero, static_cast<int>(nzero));
  if (bad != 0.0) atomicExch(&info->ok, 0);
}
// One lane per block, pivot and scaling in sequence: for levels whose blocks have short structs (chain-like
// patterns: three rows per block in the Rosenbrock chain) the two phases of a block are a handful of operations.
__global__ void __launch_bounds__(kBlock) sp_pivot_scale_kernel(SparsePlan pl, double* vals, double* w, double* dinv, i64 b0, i64 b1,
                                                                SparseInfo* info) {
  const i64 k = b0 + static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x;
  if (k >= b1) return;
  double nneg = 0.0, nzero = 0.0, bad = 0.0;
  sp_pivot(pl, vals, dinv, k, nneg, nzero, bad);
  if (nneg != 0.0) atomicAdd(&info->nneg, static_cast<int>(nneg));
  if (nzero != 0.0) atomicAdd(&info->nzero, static_cast<int>(nzero));
  if (bad != 0.0) atomicExch(&info->ok, 0);
  for (i64 r = pl.soff[k]; r < pl.soff[k + 1]; ++r) sp_scale(pl, vals, w, dinv, r);
}
__global__ void __launch_bounds__(kBlock) sp_scale_kernel(SparsePlan pl, double* vals, double* w, const double* dinv, i64 r0, i64 r1) {
  const i64 r = r0 + static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x;
  if (r < r1) sp_scale(pl, vals, w, dinv, r);
}
// Schur-complement updates of a level, ORDER-FIXED: the level's triples are stored sorted by destination
// (sparse_plan.h gdst / goff); sixteen lanes walk a destination's triples in strides, a fixed tree reduces them and
// ONE lane subtracts the sum.  No floating-point atomics (they made power flow take 16 to 18 iterations from run to run).
__global__ void __launch_bounds__(kBlock) sp_update_gather_kernel(SparsePlan pl, double* vals, const double* w, i64 g0, i64 g1) {
  const i64 g = g0 + ((static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x) >> 4);
  const int l = threadIdx.x & 15;
  double s = 0.0;
  i64 q0 = 0, q1 = 0;
  if (g < g1) { q0 = pl.goff[g]; q1 = pl.goff[g + 1]; }
  for (i64 q = q0 + l; q < q1; q += 16) s += sp_update(pl, vals, w, q);
  s += __shfl_xor(s, 8, 16);
  s += __shfl_xor(s, 4, 16);
  s += __shfl_xor(s, 2, 16);
  s += __shfl_xor(s, 1, 16);
  if (l == 0 && q1 > q0) vals[pl.gdst[g]] -= s;
}
// tail rows of a level's panel blocks into the two dense panels (l and w = l D^-1)
__global__ void __launch_bounds__(kBlock) sp_panel_gather_kernel(SparsePlan pl, const double* vals, const double* w, double* Pl, double* Pw,
                                                                 i64 q0, i64 q1) {
  const i64 q = q0 + static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x;
  if (q >= q1) return;
  Pl[pl.pg_dst[q]] = w[pl.pg_src[q]];          // (sp_scale leaves the unscaled l in w and L = l D^-1 in vals)
  Pw[pl.pg_dst[q]] = vals[pl.pg_src[q]];
}
// ... and a whole workgroup per destination for levels whose groups are long (the separator of the NMF example: 300
// tail nodes under 1 200 blocks each): wavefront sums by DPP, the four partials added in order
__global__ void __launch_bounds__(kBlock) sp_update_gather_wg_kernel(SparsePlan pl, double* vals, const double* w, i64 g0) {
  __shared__ double part[kBlock / 64];
  const i64 g = g0 + blockIdx.x;
  const i64 q0 = pl.goff[g], q1 = pl.goff[g + 1];
  double s = 0.0;
  for (i64 q = q0 + threadIdx.x; q < q1; q += kBlock) s += sp_update(pl, vals, w, q);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { double t = part[0]; for (int k = 1; k < kBlock / 64; ++k) t += part[k]; vals[pl.gdst[g]] -= t; }
}
__global__ void __launch_bounds__(kBlock) sp_fwd_gather_wg_kernel(SparsePlan pl, const double* vals, double* x, i64 h0) {
  __shared__ double part[kBlock / 64];
  const i64 h = h0 + blockIdx.x;
  const i64 q0 = pl.foff[h], q1 = pl.fend ? pl.fend[h] : pl.foff[h + 1];
  double s = 0.0;
  for (i64 q = q0 + threadIdx.x; q < q1; q += kBlock) s += sp_fwd(pl, vals, x, q);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) { double t = part[0]; for (int k = 1; k < kBlock / 64; ++k) t += part[k]; x[pl.fnode[h]] -= t; }
}
// forward substitution in gather form (sparse_plan.h fnode / foff / frow): sixteen lanes per target node, fixed tree
__global__ void __launch_bounds__(kBlock) sp_fwd_gather_kernel(SparsePlan pl, const double* vals, double* x, i64 h0, i64 h1) {
  const i64 h = h0 + ((static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x) >> 4);
  const int l = threadIdx.x & 15;
  double s = 0.0;
  i64 q0 = 0, q1 = 0;
  if (h < h1) { q0 = pl.foff[h]; q1 = pl.fend ? pl.fend[h] : pl.foff[h + 1]; }
  for (i64 q = q0 + l; q < q1; q += 16) s += sp_fwd(pl, vals, x, q);
  s += __shfl_xor(s, 8, 16);
  s += __shfl_xor(s, 4, 16);
  s += __shfl_xor(s, 2, 16);
  s += __shfl_xor(s, 1, 16);
  if (l == 0 && q1 > q0) x[pl.fnode[h]] -= s;
}
__global__ void __launch_bounds__(kBlock) sp_dsolve_kernel(SparsePlan pl, const double* vals, double* x) {
  const i64 k = static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x;
  if (k < pl.nblk_run) sp_dsolve(pl, vals, x, k);
}
// backward: one wavefront per block of the level (shuffle reduction over the block's struct)
__global__ void __launch_bounds__(kBlock) sp_bwd_kernel(SparsePlan pl, const double* vals, double* x, i64 b0, i64 b1) {
  const i64 k = b0 + static_cast<i64>(blockIdx.x) * (kBlock / 64) + (threadIdx.x >> 6);
  if (k >= b1) return;
  const int lane = threadIdx.x & 63;
  const i64 s0 = pl.soff[k], s = pl.soff[k + 1] - s0;
  const i32 u0 = pl.bnode[2 * k], u1 = pl.bnode[2 * k + 1];
  const double* Lk = vals + pl.loff[k];
  double a0 = 0.0, a1 = 0.0;
  if (u1 < 0) {
    for (i64 i = lane; i < s; i += 64) a0 += Lk[i] * x[pl.sidx[s0 + i]];
  } else {
    for (i64 i = lane; i < s; i += 64) { const double xi = x[pl.sidx[s0 + i]]; a0 += Lk[2 * i] * xi; a1 += Lk[2 * i + 1] * xi; }
  }
  a0 = wave_sum(a0);
  a1 = wave_sum(a1);
  if (lane == 0) { x[u0] -= a0; if (u1 >= 0) x[u1] -= a1; }
}

// levels whose structs are short (a few entries: chain-like problems): one lane per block, serial dot
__global__ void __launch_bounds__(kBlock) sp_bwd_thread_kernel(SparsePlan pl, const double* vals, double* x, i64 b0, i64 b1) {
  const i64 k = b0 + static_cast<i64>(blockIdx.x) * kBlock + threadIdx.x;
  if (k >= b1) return;
  const i64 s0 = pl.soff[k], s = pl.soff[k + 1] - s0;
  if (s == 0) return;
  const i32 u0 = pl.bnode[2 * k], u1 = pl.bnode[2 * k + 1];
  const double* Lk = vals + pl.loff[k];
  double a0 = 0.0, a1 = 0.0;
  if (u1 < 0) {
    for (i64 i = 0; i < s; ++i) a0 += Lk[i] * x[pl.sidx[s0 + i]];
    x[u0] -= a0;
  } else {
    for (i64 i = 0; i < s; ++i) { const double xi = x[pl.sidx[s0 + i]]; a0 += Lk[2 * i] * xi; a1 += Lk[2 * i + 1] * xi; }
    x[u0] -= a0; x[u1] -= a1;
  }
}

struct BlockedLdlt;   // ldlt_blocked.h

struct HipExec : HostControlled {
  static constexpr long long kPivotedMaxOrder = 4096;      // = BK_NMAX: bk_solve_kernel / bk_finish_kernel LDS arrays
  static constexpr bool has_condensed_ls = true;
  static constexpr bool is_device = true;
  using FlatTableT = FlatTable;
  int device = 0;
  hipStream_t stream = nullptr;
  double* d_partial = nullptr;
  double* h_partial = nullptr;
  double* r_partial = nullptr;   // where the reductions' kernels write: the pinned host buffer itself (device view), or d_partial
  double* gemv_part = nullptr;
  struct SparseInfo* sparse_info = nullptr;
  size_t gemv_part_cap = 0;

  struct LdltWork {
    BkState* st = nullptr;
    i64 bk_n = 0;                // order the Bunch-Kaufman workspace below was sized for
    i32* bk_perm = nullptr;      // Bunch-Kaufman: the interchanges as one permutation, and the pivot structure
    i32* bk_dtype = nullptr;
    double* bk_w = nullptr;      // panel workspace W = L D of bk_panel_kernel (n x 16)
    struct BkPanelSwaps* bk_swaps = nullptr;
    bool bk_panels = true;       // DNLP_BK_PANELS=0: the unblocked two-launches-per-column pair
    BlockedLdlt* blocked = nullptr;
    int expect_neg = -1;         // inertia the caller needs (early exit of hopeless attempts)
    bool time_updates = false;
    bool padded = false;         // matrix allocation carries >= 128 doubles of slack
  };
  // kernel statistics of the dominant (MFMA Schur update) kernel: seconds, flops, launches
  void ldlt_stats(LdltWork& w, double* out3);

  explicit HipExec(int dev = 0) : device(dev) {
    DNLP_HIP_CHECK(hipSetDevice(device));
    DNLP_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    // (the fused objective kernel writes one partial per wavefront: 4 per workgroup)
    DNLP_HIP_CHECK(hipMalloc(&d_partial, sizeof(double) * kMaxPartials * 4));
    DNLP_HIP_CHECK(hipHostMalloc(&h_partial, sizeof(double) * kMaxPartials * 4));
    // A reduction's partials (a few doubles per workgroup) are written straight into the pinned host buffer: the
    // device-to-host copy behind every reduction was a blit kernel of its own (3-4 us and a launch) on the
    // host-driven loop's ~50 reductions per iteration.  DNLP_REDUCE_COPY=1 restores the copy.
    r_partial = d_partial;
    if (!std::getenv("DNLP_REDUCE_COPY") || std::atoi(std::getenv("DNLP_REDUCE_COPY")) == 0) {
      void* dv = nullptr;
      if (hipHostGetDevicePointer(&dv, h_partial, 0) == hipSuccess && dv) r_partial = static_cast<double*>(dv);
      else (void)hipGetLastError();
    }
    if (const char* v = std::getenv("DNLP_FUSED_NE")) fused_ne_override = std::atoi(v);
  }
  ~HipExec() {
    hipSetDevice(device);
    for (auto& f : at_exit_) f();                 // host objects that hold streams / events on this device
    for (const LevelGraph& g : level_graphs_) hipGraphExecDestroy(g.exec);
    for (void* p : owned_) hipFree(p);
    // device-resident L-BFGS workspace (lbfgs_generated_solve): (2M + 3) nfree doubles per handle
    lb_graph_reset();
    if (lb_state) hipFree(lb_state);
    if (lb_ctl) hipFree(lb_ctl);
    if (vt_part) hipFree(vt_part);
    if (lb_halo) hipFree(lb_halo);
    if (lb_xsave) hipFree(lb_xsave);
    if (lb_strip) hipFree(lb_strip);
    if (lb_host) hipHostFree(lb_host);
    if (lb_fpart) hipFree(lb_fpart);
    if (lb_upart) hipFree(lb_upart);
    if (lb_BV) hipFree(lb_BV);
    if (lb_dir) hipFree(lb_dir);
    if (lb_gt) hipFree(lb_gt);
    if (gemv_part) hipFree(gemv_part);
    if (d_partial) hipFree(d_partial);
    if (h_partial) hipHostFree(h_partial);
    if (stream) hipStreamDestroy(stream);
  }
  HipExec(const HipExec&) = delete;
  HipExec& operator=(const HipExec&) = delete;

  template <class T> T* alloc(size_t n) {
    DNLP_HIP_CHECK(hipSetDevice(device));
    void* p = nullptr;
    const size_t bytes = (n ? n : 1) * sizeof(T);
    DNLP_HIP_CHECK(hipMalloc(&p, bytes));
    DNLP_HIP_CHECK(hipMemsetAsync(p, 0, bytes, stream));
    owned_.push_back(p);
    return static_cast<T*>(p);
  }
  void release(void* p) {
    for (auto& q : owned_) if (q == p) { hipFree(p); q = nullptr; }
  }
  void h2d(void* dst, const void* src, size_t bytes) {
    if (!bytes) return;
    DNLP_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, stream));
    DNLP_HIP_CHECK(hipStreamSynchronize(stream));   // src is pageable host memory
  }
  void d2h(void* dst, const void* src, size_t bytes) {
    if (!bytes) return;
    DNLP_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream));
    DNLP_HIP_CHECK(hipStreamSynchronize(stream));
  }
  void d2d(void* dst, const void* src, size_t bytes) {
    if (bytes) DNLP_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream));
  }
  void zero(void* p, size_t bytes) {
    if (bytes) DNLP_HIP_CHECK(hipMemsetAsync(p, 0, bytes, stream));
  }
  void sync() { DNLP_HIP_CHECK(hipStreamSynchronize(stream)); }

  template <class F> void map(i64 n, F f) {
    for (i64 off = 0; off < n; off += kMaxLaunchElems) {
      const i64 cnt = (n - off < kMaxLaunchElems) ? n - off : kMaxLaunchElems;
      const i64 grid = (cnt + kBlock - 1) / kBlock;
      hipLaunchKernelGGL(map_kernel<F>, dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, stream, off, n, f);
    }
    DNLP_LAUNCH_CHECK();
  }
  template <int MODE, class F> double reduce(i64 n, F f) {
    if (n <= 0) return MODE == 0 ? 0.0 : (MODE == 1 ? -kInf : kInf);
    i64 grid = (n + kBlock - 1) / kBlock;
    if (grid > kMaxPartials) grid = kMaxPartials;
    hipLaunchKernelGGL((reduce_kernel<MODE, F>), dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, stream, n, f, r_partial);
    DNLP_LAUNCH_CHECK();
    if (r_partial == d_partial) DNLP_HIP_CHECK(hipMemcpyAsync(h_partial, d_partial, sizeof(double) * grid, hipMemcpyDeviceToHost, stream));
    DNLP_HIP_CHECK(hipStreamSynchronize(stream));
    double r = h_partial[0];
    for (i64 k = 1; k < grid; ++k) {
      if (MODE == 0) r += h_partial[k];
      else if (MODE == 1) r = std::fmax(r, h_partial[k]);
      else r = std::fmin(r, h_partial[k]);
    }
    return r;
  }
  template <class F> double sum(i64 n, F f) { return reduce<0>(n, f); }
  template <class F> double max(i64 n, F f) { return reduce<1>(n, f); }
  template <class F> double min(i64 n, F f) { return reduce<2>(n, f); }
  // NM maxima and NS sums with ONE launch and ONE read-back (every separate reduction costs a launch and a
  // stream synchronisation: the optimality error of an iteration was nine of them)
  template <int NM, int NS, class F> RMulti reduce_multi(i64 n, F f) {
    RMulti r;
    for (int k = 0; k < 4; ++k) { r.mx[k] = -kInf; r.sm[k] = 0.0; }
    if (n <= 0) return r;
    i64 grid = (n + kBlock - 1) / kBlock;
    if (grid > kMaxPartials * 4 / (NM + NS)) grid = kMaxPartials * 4 / (NM + NS);
    hipLaunchKernelGGL((reduce_multi_kernel<NM, NS, F>), dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, stream, n, f, r_partial);
    DNLP_LAUNCH_CHECK();
    if (r_partial == d_partial) DNLP_HIP_CHECK(hipMemcpyAsync(h_partial, d_partial, sizeof(double) * (NM + NS) * grid, hipMemcpyDeviceToHost, stream));
    DNLP_HIP_CHECK(hipStreamSynchronize(stream));
    for (i64 b = 0; b < grid; ++b) {
      for (int k = 0; k < NM; ++k) r.mx[k] = std::fmax(r.mx[k], h_partial[(NM + NS) * b + k]);
      for (int k = 0; k < NS; ++k) r.sm[k] += h_partial[(NM + NS) * b + NM + k];
    }
    return r;
  }
  template <class F> D2 min2(i64 n, F f) {
    if (n <= 0) return D2{kInf, kInf};
    i64 grid = (n + kBlock - 1) / kBlock;
    if (grid > kMaxPartials) grid = kMaxPartials;
    hipLaunchKernelGGL((reduce_min2_kernel<F>), dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, stream, n, f, r_partial);
    DNLP_LAUNCH_CHECK();
    if (r_partial == d_partial) DNLP_HIP_CHECK(hipMemcpyAsync(h_partial, d_partial, sizeof(double) * 2 * grid, hipMemcpyDeviceToHost, stream));
    DNLP_HIP_CHECK(hipStreamSynchronize(stream));
    D2 r{h_partial[0], h_partial[1]};
    for (i64 k = 1; k < grid; ++k) { r.first = std::fmin(r.first, h_partial[2 * k]); r.second = std::fmin(r.second, h_partial[2 * k + 1]); }
    return r;
  }

  void spmv_long_rows(const Csr& M, const double* v, const double* base, double* y) {
    const i64 grid = (M.rows + kBlock / 64 - 1) / (kBlock / 64);
    hipLaunchKernelGGL(spmv_long_kernel, dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, stream, M.rows, M.ptr, M.idx, M.val,
                       v, base, y);
    DNLP_LAUNCH_CHECK();
  }
  void gemv_sym(i64 n, const double* P, i64 ld, const double* u, double* out) {
    const i64 nrb = (n + 511) / 512, ncb = (n + GEMV_CB - 1) / GEMV_CB;
    static const bool lower_only = std::getenv("DNLP_GEMV_FULL") == nullptr;
    const bool tri = lower_only && n >= 4096 && (ld & 1) == 0 && (reinterpret_cast<uintptr_t>(P) & 15) == 0 && kBlock == 256;
    const size_t need = (static_cast<size_t>(ncb) + (tri ? static_cast<size_t>(nrb) : 0)) * static_cast<size_t>(n);
    if (need > gemv_part_cap) {
      if (gemv_part) DNLP_HIP_CHECK(hipFree(gemv_part));
      DNLP_HIP_CHECK(hipMalloc(&gemv_part, need * sizeof(double)));
      gemv_part_cap = need;
    }
    if (tri) {
      // the lower triangle only: half the bytes of the full product (A is symmetric and stored in full)
      double* part2 = gemv_part + static_cast<size_t>(ncb) * static_cast<size_t>(n);
      hipLaunchKernelGGL(gemv_sym_stage1, dim3(static_cast<unsigned>(nrb * ncb)), dim3(kBlock), 0, stream, n, P, ld, u, gemv_part, part2, nrb);
      hipLaunchKernelGGL(gemv_sym_stage2, dim3(static_cast<unsigned>((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, n, nrb, gemv_part, part2, out);
      DNLP_LAUNCH_CHECK();
      return;
    }
    hipLaunchKernelGGL(gemv_stage1, dim3(static_cast<unsigned>(nrb * ncb)), dim3(kBlock), 0, stream, n, P, ld, u, gemv_part, nrb);
    hipLaunchKernelGGL(gemv_stage2, dim3(static_cast<unsigned>((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, n, ncb, gemv_part, out);
    DNLP_LAUNCH_CHECK();
  }
  // Gram-Schmidt step against k stored vectors: c = V^T w (returned on the host), w -= V c
  // c = V^T w (k <= 32 vectors, one fused pass over w and V, one scalar read-back)
  void vt_dot(int k, const double* V, i64 N, const double* w, double* c_host) {
    if (k <= 0) return;
    double* dc = d_partial;
    i64 grid = (N + kBlock - 1) / kBlock;
    if (grid > 1024) grid = 1024;
    if (!vt_part) DNLP_HIP_CHECK(hipMalloc(&vt_part, sizeof(double) * 32 * 1024));
    hipLaunchKernelGGL(vt_dot_kernel, dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, stream, k, V, N, w, vt_part);
    hipLaunchKernelGGL(vt_dot_finish_kernel, dim3(static_cast<unsigned>(k)), dim3(64), 0, stream, static_cast<int>(grid), vt_part, dc);
    DNLP_HIP_CHECK(hipMemcpyAsync(c_host, dc, sizeof(double) * k, hipMemcpyDeviceToHost, stream));
    DNLP_HIP_CHECK(hipStreamSynchronize(stream));
    DNLP_LAUNCH_CHECK();
  }
  // out = sum_q c_q V_q (coefficients from the host, no read-back)
  void v_comb(int k, const double* V, i64 N, const double* c_host, double* out) {
    double* dc = d_partial + 64;
    DNLP_HIP_CHECK(hipMemcpyAsync(dc, c_host, sizeof(double) * k, hipMemcpyHostToDevice, stream));
    DNLP_HIP_CHECK(hipStreamSynchronize(stream));      // c_host is pageable and may be reused by the caller
    hipLaunchKernelGGL(v_comb_kernel, dim3(static_cast<unsigned>((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, k, V, N, dc, out);
    DNLP_LAUNCH_CHECK();
  }
  void orthogonalize(int k, const double* V, i64 N, double* w, double* c_host) {
    if (k <= 0) return;
    double* dc = d_partial;      // k <= 32 doubles of the reduction scratch
    i64 grid = (N + kBlock - 1) / kBlock;
    if (grid > 1024) grid = 1024;
    if (!vt_part) DNLP_HIP_CHECK(hipMalloc(&vt_part, sizeof(double) * 32 * 1024));
    hipLaunchKernelGGL(vt_dot_kernel, dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, stream, k, V, N, w, vt_part);
    hipLaunchKernelGGL(vt_dot_finish_kernel, dim3(static_cast<unsigned>(k)), dim3(64), 0, stream, static_cast<int>(grid), vt_part, dc);
    hipLaunchKernelGGL(v_axpy_kernel, dim3(static_cast<unsigned>((N + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream, k, V, N, dc, w);
    DNLP_HIP_CHECK(hipMemcpyAsync(c_host, dc, sizeof(double) * k, hipMemcpyDeviceToHost, stream));
    DNLP_HIP_CHECK(hipStreamSynchronize(stream));
    DNLP_LAUNCH_CHECK();
  }
  void sweep_flat(const FlatTable& t, const double* x, double* z, double* dv, double* hv, const double* w, bool with_h) {
    if (t.total <= 0) return;
    const i64 grid = (t.total + 2 * kBlock - 1) / (2 * kBlock);
    hipLaunchKernelGGL(sweep_flat_kernel, dim3(static_cast<unsigned>(grid)), dim3(kBlock), 0, stream, t, x, z, dv, hv, w,
                       with_h ? 1 : 0);
    DNLP_LAUNCH_CHECK();
  }
  // Small systems: one workgroup walks all levels (one launch).  Large ones (>= 8192 pivot
  // blocks): one grid-wide kernel per level phase, sized by that level's blocks / rows / triples.
  // one workgroup walks the whole plan (no launches) below these sizes; above, one kernel per level phase
  // over the whole chip (a plan with few blocks but a long update program -- dense-ish fronts, 1e6 triples
  // at order 1e4 in the small NMF example -- is compute-bound on one CU)
  static constexpr i64 kSparseGridMin = 8192;
  static constexpr i64 kSparseGridMinTriples = 200000;
  static bool sparse_grid_path(const SparsePlan& pl) {
    // (a plan with a dense tail always: its panel products are the MFMA kernel's, not one workgroup's loops — phase
    //  retrieval, 5.7e4 triples left beside a 128-node tail: 7 ms per factorisation in one workgroup, 0.25 ms here)
    return pl.h_lev_blk && (pl.nblk >= kSparseGridMin || pl.ntrip >= kSparseGridMinTriples || pl.tail_n > 0);
  }
  // The level loops of the static-pattern factorisation / solves are the same launch sequence every time
  // (a plan with 300 levels is 1 200 + 600 launches of ~2 us kernels: host launch overhead, ~16 us each,
  // was 80 % of the NMF example).  They are captured ONCE per (plan, buffers) into a HIP graph and replayed.
  struct LevelGraph { const void* k0; const void* k1; const void* k2; int kind; hipGraphExec_t exec; };
  std::vector<std::function<void()>> at_exit_;   // run by the destructor before the device memory goes
  std::vector<LevelGraph> level_graphs_;
  int level_graphs_on_ = -1;
  bool level_fusion_ = std::getenv("DNLP_LEVEL_FUSION") == nullptr || std::atoi(std::getenv("DNLP_LEVEL_FUSION")) != 0;
  template <class F>
  void replay_levels(int kind, const void* k0, const void* k1, const void* k2, F&& launches) {
    if (level_graphs_on_ < 0) {
      const char* ev = std::getenv("DNLP_LEVEL_GRAPHS");
      level_graphs_on_ = (ev && std::atoi(ev) == 0) ? 0 : 1;
    }
    if (!level_graphs_on_) { launches(); return; }
    for (const LevelGraph& g : level_graphs_)
      if (g.kind == kind && g.k0 == k0 && g.k1 == k1 && g.k2 == k2) {
        DNLP_HIP_CHECK(hipGraphLaunch(g.exec, stream));
        return;
      }
    // a runtime that cannot capture or instantiate runs the launches directly from then on
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      (void)hipGetLastError();
      level_graphs_on_ = 0;
      launches();
      return;
    }
    launches();
    DNLP_HIP_CHECK(hipStreamEndCapture(stream, &graph));
    hipGraphExec_t exec = nullptr;
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
      (void)hipGetLastError();
      hipGraphDestroy(graph);
      level_graphs_on_ = 0;
      launches();
      return;
    }
    DNLP_HIP_CHECK(hipGraphDestroy(graph));
    if (level_graphs_.size() >= 24) {              // buffers of retired solver objects: oldest out
      hipGraphExecDestroy(level_graphs_.front().exec);
      level_graphs_.erase(level_graphs_.begin());
    }
    level_graphs_.push_back(LevelGraph{k0, k1, k2, kind, exec});
    DNLP_HIP_CHECK(hipGraphLaunch(exec, stream));
  }
  void sparse_tail_gemm(double* T, i64 ldt, const double* Pl, const double* Pw, int r, int cols);   // ldlt_blocked.h kernels
  bool sparse_factor(const SparsePlan& pl, double* vals, double* w, int* nneg, int* nzero) {
    if (!sparse_info) { sparse_info = alloc<SparseInfo>(1); }
    auto grid = [](i64 items) { return dim3(static_cast<unsigned>((items + kBlock - 1) / kBlock)); };
    if (!sparse_grid_path(pl)) {
      hipLaunchKernelGGL(sparse_factor_kernel, dim3(1), dim3(kBlock), 0, stream, pl, vals, w, sparse_info);
      DNLP_LAUNCH_CHECK();
    } else {
      double* dinv = w + pl.nvals;
      SparseInfo* info = sparse_info;
      double* Tacc = w + sparse_ldl_tail_acc_offset(pl);
      double* Pl = Tacc + pl.tail_ld * pl.tail_n;
      double* Pw = Pl + pl.tail_ld * pl.pg_maxcols;
      replay_levels(0, pl.soff, vals, w, [&] {
        hipLaunchKernelGGL(sp_info_reset_kernel, dim3(1), dim3(1), 0, stream, info);
        if (pl.tail_n > 0) DNLP_HIP_CHECK(hipMemsetAsync(Tacc, 0, sizeof(double) * static_cast<size_t>(pl.tail_ld * pl.tail_n), stream));
        for (i64 lev = 0; lev < pl.nlev_run; ++lev) {
          const i64 b0 = pl.h_lev_blk[lev], b1 = pl.h_lev_blk[lev + 1], r0 = pl.h_lev_row[lev], r1 = pl.h_lev_row[lev + 1];
          const i64 g0 = pl.h_lev_g[lev], g1 = pl.h_lev_g[lev + 1];
          if (level_fusion_ && (r1 - r0) <= 8 * (b1 - b0)) {
            hipLaunchKernelGGL(sp_pivot_scale_kernel, grid(b1 - b0), dim3(kBlock), 0, stream, pl, vals, w, dinv, b0, b1, info);
          } else {
            hipLaunchKernelGGL(sp_pivot_kernel, grid(b1 - b0), dim3(kBlock), 0, stream, pl, vals, dinv, b0, b1, info);
            if (r1 > r0) hipLaunchKernelGGL(sp_scale_kernel, grid(r1 - r0), dim3(kBlock), 0, stream, pl, vals, w, dinv, r0, r1);
          }
          if (pl.tail_n > 0 && pl.h_pg_cols[lev] > 0) {
            // the level's panel blocks: T -= Pl Pw^T on the FP64 MFMA kernel (lower triangle, tail_n x tail_n x cols)
            const i64 cols = pl.h_pg_cols[lev], q0 = pl.h_pg_off[lev], q1 = pl.h_pg_off[lev + 1];
            DNLP_HIP_CHECK(hipMemsetAsync(Pl, 0, sizeof(double) * static_cast<size_t>(2 * pl.tail_ld * pl.pg_maxcols), stream));
            hipLaunchKernelGGL(sp_panel_gather_kernel, grid(q1 - q0), dim3(kBlock), 0, stream, pl, vals, w, Pl, Pw, q0, q1);
            sparse_tail_gemm(Tacc, pl.tail_ld, Pl, Pw, static_cast<int>(pl.tail_n), static_cast<int>(cols));
          }
          if (g1 > g0) {
            if (pl.h_lev_trip[lev + 1] - pl.h_lev_trip[lev] >= 128 * (g1 - g0))
              hipLaunchKernelGGL(sp_update_gather_wg_kernel, dim3(static_cast<unsigned>(g1 - g0)), dim3(kBlock), 0, stream, pl, vals, w, g0);
            else
              hipLaunchKernelGGL(sp_update_gather_kernel, grid(16 * (g1 - g0)), dim3(kBlock), 0, stream, pl, vals, w, g0, g1);
          }
        }
      });
      DNLP_LAUNCH_CHECK();
    }
    SparseInfo h;
    DNLP_HIP_CHECK(hipMemcpyAsync(&h, sparse_info, sizeof h, hipMemcpyDeviceToHost, stream));
    DNLP_HIP_CHECK(hipStreamSynchronize(stream));
    *nneg = h.nneg;
    *nzero = h.nzero;
    return h.ok != 0;
  }
  void sparse_solve(const SparsePlan& pl, const double* vals, double* x) {
    auto grid = [](i64 items) { return dim3(static_cast<unsigned>((items + kBlock - 1) / kBlock)); };
    if (!sparse_grid_path(pl)) {
      hipLaunchKernelGGL(sparse_solve_kernel, dim3(1), dim3(kBlock), 0, stream, pl, vals, x);
    } else {
      replay_levels(1 + 2 * pl.solve_phase, pl.soff, vals, x, [&] {
        if (pl.solve_phase != 2) {
          for (i64 lev = 1; lev <= pl.nlev_run; ++lev) {
            const bool last = lev == pl.nlev_run;
            if (last && pl.nlev_run == pl.nlev) break;
            const i64 h0 = pl.h_lev_f[lev], h1 = last ? pl.h_lev_f[pl.nlev] : pl.h_lev_f[lev + 1];
            if (h1 > h0) {
              if (pl.h_fwd_rows[h1] - pl.h_fwd_rows[h0] >= 128 * (h1 - h0))
                hipLaunchKernelGGL(sp_fwd_gather_wg_kernel, dim3(static_cast<unsigned>(h1 - h0)), dim3(kBlock), 0, stream, pl, vals, x, h0);
              else
                hipLaunchKernelGGL(sp_fwd_gather_kernel, grid(16 * (h1 - h0)), dim3(kBlock), 0, stream, pl, vals, x, h0, h1);
            }
          }
          hipLaunchKernelGGL(sp_dsolve_kernel, grid(pl.nblk_run), dim3(kBlock), 0, stream, pl, vals, x);
        }
        for (i64 lev = pl.solve_phase == 1 ? -1 : pl.nlev_run - 1; lev >= 0; --lev) {
          const i64 b0 = pl.h_lev_blk[lev], b1 = pl.h_lev_blk[lev + 1];
          if (pl.h_lev_row[lev + 1] == pl.h_lev_row[lev]) continue;       // root blocks: empty structs
          // average struct length of the level decides: a wavefront per block only pays for long structs
          if ((pl.h_lev_row[lev + 1] - pl.h_lev_row[lev]) < 16 * (b1 - b0))
            hipLaunchKernelGGL(sp_bwd_thread_kernel, grid(b1 - b0), dim3(kBlock), 0, stream, pl, vals, x, b0, b1);
          else
            hipLaunchKernelGGL(sp_bwd_kernel, dim3(static_cast<unsigned>((b1 - b0 + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0,
                               stream, pl, vals, x, b0, b1);
        }
      });
    }
    DNLP_LAUNCH_CHECK();
  }
  template <int NE, int WPE>
  void fused_launch(const FusedSlotProg& P, const double* x, const double* consts, double* grad, i64 blocks) {
    const size_t lds = static_cast<size_t>(P.nslots + 1) * NE * kBlock * sizeof(double);      // slots + gradient window
    hipLaunchKernelGGL((fused_eval_kernel<NE, WPE>), dim3(static_cast<unsigned>(blocks)), dim3(kBlock), lds, stream, P, x,
                       consts, grad, d_partial);
  }
  double fused_eval(const FusedSlotProg& P, const double* x, const double* consts, double* grad) {
    if (P.nelem <= 0) return 0.0;
    // Elements per lane: the interpreter is bound by its per-op decode / branch / LDS round trip,
    // which NE independent element chains share; the slot form keeps the register file small
    // enough (slots x NE x 2 KB per workgroup) for that without giving up resident wavefronts.
    // Small problems keep one element per lane (more workgroups than CUs matters more there).
    // Measured at n = 1e8 (Rosenbrock, 22 ops over 5 slots, 16 n measure): NE 1: 442, 2: 507,
    // 4: 702 (4 wavefronts/SIMD; 6: 556, 3: 694), 8: 398 GB/s; the same interpreter with its slots in
    // VGPRs (s_set_gpr_idx relative addressing, 162-214 registers) reached 284-324 GB/s.  With the
    // 14-op program and the predicate-free full-tile body: NE 4 at 3 / 4 / 5 wavefronts per SIMD 1025 /
    // 1082 / 1042, NE 2: 937, NE 8: 697 GB/s.
    int ne = 1;
    if (P.nelem >= (static_cast<i64>(1) << 19) && (P.nslots + 1) * 4 * kBlock * 8 <= 56 * 1024) ne = 4;
    else if (P.nelem >= (static_cast<i64>(1) << 18) && (P.nslots + 1) * 2 * kBlock * 8 <= 56 * 1024) ne = 2;
    if (fused_ne_override == 1 || fused_ne_override == 2 || (fused_ne_override == 4 && (P.nslots + 1) * 4 * kBlock * 8 <= 64 * 1024))
      ne = fused_ne_override;
    const i64 tile = static_cast<i64>(kBlock) * ne;
    i64 blocks = (P.nelem + tile - 1) / tile;
    if (blocks > kMaxPartials) blocks = kMaxPartials;
    if (ne == 4) fused_launch<4, 4>(P, x, consts, grad, blocks);
    else if (ne == 2) fused_launch<2, 8>(P, x, consts, grad, blocks);
    else fused_launch<1, 6>(P, x, consts, grad, blocks);
    DNLP_LAUNCH_CHECK();
    const i64 np = blocks * (kBlock / 64);
    DNLP_HIP_CHECK(hipMemcpyAsync(h_partial, d_partial, sizeof(double) * static_cast<size_t>(np), hipMemcpyDeviceToHost, stream));
    DNLP_HIP_CHECK(hipStreamSynchronize(stream));
    double f = 0.0;
    for (i64 k = 0; k < np; ++k) f += h_partial[k];
    return f;
  }
  int fused_ne_override = 0;     // DNLP_FUSED_NE = 1 | 2 | 4 (measurement sweeps)

  // ---- generated fused-objective kernel (fused_codegen.h + fused_rtc.h) -------------------------
  bool fused_codegen = true;     // option fused_codegen=no keeps the interpreter
  int fused_E = 4;               // grad entries per lane of the generated kernel (DNLP_FUSED_E)
  RtcKernel fused_rtc;
  const void* fused_rtc_key = nullptr;
  bool fused_generated_eval(const std::vector<FusedSlotProg>& progs, const double* x, const double* consts, double* grad,
                            i64 nfree, double& f) {
    if (!fused_codegen || progs.empty() || nfree <= 0) return false;
    if (fused_rtc_key != static_cast<const void*>(&progs)) {
      fused_rtc_key = &progs;
      if (const char* v = std::getenv("DNLP_FUSED_E")) { const int e = std::atoi(v); if (e >= 1 && e <= 16) fused_E = e; }
      const FusedCodegenInfo info = fused_codegen_plan(progs, fused_E);
      if (info.ok) {
        const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        fused_rtc.load(fused_codegen_eval_source(progs, info), "dnlp_fused_eval");
        fused_rtc.compile_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0;
      } else {
        fused_rtc.tried = true;
        fused_rtc.log = info.why;
      }
    }
    if (!fused_rtc.ok) return false;
    const i64 nchunks = (nfree + fused_E - 1) / fused_E;
    i64 blocks = (nchunks + kBlock - 1) / kBlock;
    if (blocks > kMaxPartials) blocks = kMaxPartials;
    i64 nf = nfree, nc = nchunks;
    void* args[] = {&x, &consts, &grad, &d_partial, &nf, &nc};
    DNLP_HIP_CHECK(hipModuleLaunchKernel(fused_rtc.fn, static_cast<unsigned>(blocks), 1, 1, kBlock, 1, 1, 0, stream, args, nullptr));
    const i64 np = blocks * (kBlock / 64);
    DNLP_HIP_CHECK(hipMemcpyAsync(h_partial, d_partial, sizeof(double) * static_cast<size_t>(np), hipMemcpyDeviceToHost, stream));
    DNLP_HIP_CHECK(hipStreamSynchronize(stream));
    f = 0.0;
    for (i64 k = 0; k < np; ++k) f += h_partial[k];
    return true;
  }
  // ---- device-resident L-BFGS over the generated objective (lbfgs_codegen.h) -------------------------
  struct LbfgsResult { int status = -199, iterations = 0, evaluations = 0, slots = 0; double f = 0.0, gnorm = 0.0, seconds = 0.0; bool persistent = false; };
  RtcKernel lb_rtc;
  hipFunction_t lb_eval = nullptr, lb_accept = nullptr, lb_update = nullptr, lb_control = nullptr, lb_persist = nullptr;
  LbPersistCtl* lb_ctl = nullptr;       // control block of the persistent kernel
  double* vt_part = nullptr;            // per-block shares of vt_dot (32 x 1024)
  double* lb_halo = nullptr;
  i64 lb_key_nf = -1, lb_per = 0;
  int lb_persist_wgs = 0;
  bool lb_persist_failed_ = false;      // the persistent kernel could not be made co-resident once: slot kernels from then on
  int lb_persist_fallbacks = 0;
  double* lb_strip = nullptr;           // per-workgroup slices of the persistent kernel when they exceed LDS (lbfgs_codegen.h mode 1 / 2)
  size_t lb_strip_cap = 0;
  int lb_key_mode = 0;
  double* lb_xsave = nullptr;           // start point of a persistent launch (restored when the launch is given up)
  i64 lb_xsave_cap = 0;
  const void* lb_key = nullptr;
  int lb_key_M = 0, lb_key_E = 0;
  hipGraphExec_t lb_graph_exec = nullptr;      // one batch of slots, captured once per argument set
  const void* lb_graph_key[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  double lb_graph_c0 = 0.0;
  int lb_graph_on_ = -1;
  void lb_graph_reset() { if (lb_graph_exec) { hipGraphExecDestroy(lb_graph_exec); lb_graph_exec = nullptr; } }
  LbfgsState* lb_state = nullptr;       // device
  LbfgsState* lb_host = nullptr;        // pinned
  double *lb_BV = nullptr, *lb_dir = nullptr, *lb_gt = nullptr, *lb_fpart = nullptr, *lb_upart = nullptr;
  i64 lb_cap_nf = 0;
  int lb_cap_M = 0, lb_slots_per_batch = 16;
  // x: exec space, nfree entries, start point in / solution out.  false = no generated form (caller
  // keeps the host-driven loop).
  bool lbfgs_generated_solve(const std::vector<FusedSlotProg>& progs, const double* consts, double c0, i64 nfree, double* x,
                             int M, double tol, int max_iter, LbfgsResult& out) {
    if (!fused_codegen || progs.empty() || nfree <= 0) return false;
    if (const char* v = std::getenv("DNLP_LBFGS_DEVICE")) if (std::atoi(v) == 0) return false;
    if (M < 1) M = 1;
    if (M > kLbMaxM) M = kLbMaxM;
    // Elements per lane of the L-BFGS kernels = the evaluation kernel's (four).  Fewer per lane were measured at the
    // stated size of BASELINE C2 (n = 1e5, slot kernels): 4 -> 4.0 ms, 2 -> 4.6 ms, 1 -> 6.3 ms per solve: the slot
    // kernels are bound by their dependent global loads, not by idle compute units.
    int lbE = fused_E;
    if (const char* v = std::getenv("DNLP_LBFGS_E")) { const int e = std::atoi(v); if (e >= 1 && e <= 16) lbE = e; }
    // Persistent single-launch form (one workgroup per compute unit, slices of x and of the whole history in LDS):
    // when a slice of ceil(nfree / CUs) variables with its 2M + 5 vectors fits the LDS of a compute unit
    int ncu = 0;
    DNLP_HIP_CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device));
    i64 per = 0;
    bool want_persist = true;
    if (const char* v = std::getenv("DNLP_LBFGS_PERSIST")) want_persist = std::atoi(v) != 0;
    int pmode = 0;
    if (want_persist && ncu >= 2 && nfree >= 4 * static_cast<i64>(ncu)) {
      per = ((nfree + ncu - 1) / ncu + lbE - 1) / lbE * lbE;
      // where a workgroup's slice lives (lbfgs_codegen.h): everything in LDS, the history in its global strip, or all of it
      const i64 all_lds = (2 * static_cast<i64>(M) + 5) * (per + 64) * 8 + 12 * 1024;     // (+ halo, Gram matrix, scratch)
      const i64 work_lds = 5 * (per + 64) * 8 + 12 * 1024;
      pmode = all_lds <= 150 * 1024 ? 0 : work_lds <= 150 * 1024 ? 1 : 2;
      // Measured (profiles/r04_c2_n_sweep.jsonl, per trial point): n = 3e5 mode 1 46 us against 70 us of the four-kernel
      // slots; n = 1e6 mode 2 203 us against 166 us — a workgroup per compute unit streams its strip with 256 lanes, the
      // slot kernels with the whole chip's — so a slice takes the strip while it is short (<= 2400 variables: n <= 6e5 on
      // 256 compute units) and the slot kernels beyond; DNLP_LBFGS_PERSIST_MODE = 1 / 2 forces a mode (tests).
      bool forced = false;
      if (const char* v = std::getenv("DNLP_LBFGS_PERSIST_MODE")) { const int w = std::atoi(v); if (w >= pmode && w <= 2) { pmode = w; forced = true; } }
      if (!forced && (pmode == 2 || (pmode == 1 && per > 2400))) per = 0;
      if (pmode == 2 && per > 65536) per = 0;
    }
    if (lb_key != static_cast<const void*>(&progs) || lb_key_M != M || lb_key_E != lbE || lb_key_nf != (per ? nfree : -1) || lb_key_mode != pmode) {
      lb_key_mode = pmode;
      lb_key = &progs;
      lb_key_M = M;
      lb_key_E = lbE;
      lb_key_nf = per ? nfree : -1;
      lb_per = per;
      lb_persist = nullptr;
      lb_graph_reset();
      if (lb_rtc.mod) { hipModuleUnload(lb_rtc.mod); lb_rtc.mod = nullptr; }
      lb_rtc.ok = false;
      if (const char* v = std::getenv("DNLP_FUSED_E")) { const int e = std::atoi(v); if (e >= 1 && e <= 16) fused_E = e; }
      const FusedCodegenInfo info = fused_codegen_plan(progs, lbE);
      if (info.ok && per > 0 && (info.hi - info.lo) < 1) { per = 0; lb_per = 0; }     // (no neighbour coupling: nothing to exchange; the slot kernels do)
      if (info.ok && lb_rtc.load(lbfgs_codegen_source(progs, info, M, per, pmode), "dnlp_lb_eval")) {
        lb_eval = lb_rtc.fn;
        lb_persist = per > 0 ? lb_rtc.get("dnlp_lb_persist") : nullptr;
        lb_persist_wgs = per > 0 ? static_cast<int>((nfree + per - 1) / per) : 0;
        lb_accept = lb_rtc.get("dnlp_lb_accept");
        lb_update = lb_rtc.get("dnlp_lb_update");
        lb_control = lb_rtc.get("dnlp_lb_control");
        if (!lb_accept || !lb_update || !lb_control) lb_rtc.ok = false;
      }
    }
    if (!lb_rtc.ok) return false;
    const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    const i64 nchunks = (nfree + lbE - 1) / lbE;
    i64 blocks = (nchunks + kBlock - 1) / kBlock;
    if (blocks > 1024) blocks = 1024;
    constexpr int ldp = 1024;                      // leading dimension of the partial-result columns (>= blocks)
    if (!lb_state) {
      DNLP_HIP_CHECK(hipMalloc(&lb_state, sizeof(LbfgsState)));
      DNLP_HIP_CHECK(hipHostMalloc(&lb_host, sizeof(LbfgsState)));
      DNLP_HIP_CHECK(hipMalloc(&lb_fpart, sizeof(double) * 8 * ldp));
      DNLP_HIP_CHECK(hipMalloc(&lb_upart, sizeof(double) * (3 * kLbMaxNB + 1) * ldp));
    }
    if (nfree > lb_cap_nf || M > lb_cap_M) {
      if (lb_BV) { hipFree(lb_BV); hipFree(lb_dir); hipFree(lb_gt); }
      DNLP_HIP_CHECK(hipMalloc(&lb_BV, sizeof(double) * static_cast<size_t>(2 * M) * static_cast<size_t>(nfree)));
      DNLP_HIP_CHECK(hipMalloc(&lb_dir, sizeof(double) * static_cast<size_t>(nfree)));
      DNLP_HIP_CHECK(hipMalloc(&lb_gt, sizeof(double) * 2 * static_cast<size_t>(nfree)));      // two gradient buffers
      lb_cap_nf = nfree; lb_cap_M = M;
    }
    DNLP_HIP_CHECK(hipMemsetAsync(lb_BV, 0, sizeof(double) * static_cast<size_t>(2 * M) * static_cast<size_t>(nfree), stream));
    DNLP_HIP_CHECK(hipMemsetAsync(lb_gt, 0, sizeof(double) * 2 * static_cast<size_t>(nfree), stream));
    std::memset(lb_host, 0, sizeof(LbfgsState));
    lb_host->tol = tol; lb_host->max_iter = max_iter; lb_host->M = M; lb_host->nblocks = static_cast<int>(blocks);
    DNLP_HIP_CHECK(hipMemcpyAsync(lb_state, lb_host, sizeof(LbfgsState), hipMemcpyHostToDevice, stream));
    if (lb_persist && lb_persist_wgs >= 2 && lb_persist_wgs <= ncu && lb_persist_wgs <= kLbPersistMaxWgs && !lb_persist_failed_) {
      // ONE launch: the state goes in zeroed, comes back final.  The kernel's grid barrier needs every workgroup
      // resident at once: the launch is COOPERATIVE (the runtime refuses it when the grid cannot be co-resident with the
      // ~150 KB of LDS per workgroup — another stream, handle or process holding LDS or compute units), and x is saved
      // first: a refused launch or a barrier that timed out all the same (done == 5) restores x and takes the slot
      // kernels below, which need no co-residency.  It never throws and never leaves a half-written x behind.
      if (!lb_ctl) {
        DNLP_HIP_CHECK(hipMalloc(&lb_ctl, sizeof(LbPersistCtl)));
        DNLP_HIP_CHECK(hipMalloc(&lb_halo, sizeof(double) * 2 * 1024 * 128));
      }
      if (nfree > lb_xsave_cap) {
        if (lb_xsave) hipFree(lb_xsave);
        lb_xsave = nullptr;
        DNLP_HIP_CHECK(hipMalloc(&lb_xsave, sizeof(double) * static_cast<size_t>(nfree)));
        lb_xsave_cap = nfree;
      }
      DNLP_HIP_CHECK(hipMemcpyAsync(lb_xsave, x, sizeof(double) * static_cast<size_t>(nfree), hipMemcpyDeviceToDevice, stream));
      DNLP_HIP_CHECK(hipMemsetAsync(lb_ctl, 0, sizeof(LbPersistCtl), stream));
      i64 nfp = nfree;
      double c0p = c0;
      if (lb_key_mode != 0) {
        // (2M + 5) (per + 2 W) doubles per workgroup; W <= 64 (fused_codegen_plan's halo limit)
        const size_t need = static_cast<size_t>(lb_persist_wgs) * static_cast<size_t>(2 * M + 5) * static_cast<size_t>(lb_per + 128);
        if (need > lb_strip_cap) {
          if (lb_strip) hipFree(lb_strip);
          lb_strip = nullptr;
          DNLP_HIP_CHECK(hipMalloc(&lb_strip, need * sizeof(double)));
          lb_strip_cap = need;
        }
      }
      void* a_p[] = {&lb_state, &x, &consts, &lb_ctl, &lb_halo, &c0p, &nfp, &lb_strip};
      bool launched = true;
      static const bool plain_launch = [] { const char* e = std::getenv("DNLP_LBFGS_COOPERATIVE"); return e && std::atoi(e) == 0; }();
      if (plain_launch) {
        DNLP_HIP_CHECK(hipModuleLaunchKernel(lb_persist, static_cast<unsigned>(lb_persist_wgs), 1, 1, kBlock, 1, 1, 0, stream, a_p, nullptr));
      } else if (hipModuleLaunchCooperativeKernel(lb_persist, static_cast<unsigned>(lb_persist_wgs), 1, 1, kBlock, 1, 1, 0, stream, a_p) != hipSuccess) {
        (void)hipGetLastError();
        launched = false;
      }
      if (launched) {
        DNLP_HIP_CHECK(hipMemcpyAsync(lb_host, lb_state, sizeof(LbfgsState), hipMemcpyDeviceToHost, stream));
        DNLP_HIP_CHECK(hipStreamSynchronize(stream));
      }
      if (launched && lb_host->done != 5) {
        out.slots = lb_host->evals;
        out.iterations = lb_host->iter;
        out.evaluations = lb_host->evals;
        out.f = lb_host->f;
        out.gnorm = lb_host->gn;
        out.status = lb_host->done == 1 ? 0 : lb_host->done == 2 ? 3 : lb_host->done == 4 ? -13 : -1;
        out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0;
        out.persistent = true;
        return true;
      }
      // not co-resident: the start point back, a clean state, and the slot kernels from here on for this handle
      lb_persist_failed_ = true;
      ++lb_persist_fallbacks;
      DNLP_HIP_CHECK(hipMemcpyAsync(x, lb_xsave, sizeof(double) * static_cast<size_t>(nfree), hipMemcpyDeviceToDevice, stream));
      DNLP_HIP_CHECK(hipMemsetAsync(lb_BV, 0, sizeof(double) * static_cast<size_t>(2 * M) * static_cast<size_t>(nfree), stream));
      DNLP_HIP_CHECK(hipMemsetAsync(lb_gt, 0, sizeof(double) * 2 * static_cast<size_t>(nfree), stream));
      std::memset(lb_host, 0, sizeof(LbfgsState));
      lb_host->tol = tol; lb_host->max_iter = max_iter; lb_host->M = M; lb_host->nblocks = static_cast<int>(blocks);
      DNLP_HIP_CHECK(hipMemcpyAsync(lb_state, lb_host, sizeof(LbfgsState), hipMemcpyHostToDevice, stream));
    }
    i64 nf = nfree, nc = nchunks;
    double c0v = c0;
    int ldpv = ldp;
    double* g0 = lb_gt;
    double* g1 = lb_gt + nfree;
    void* a_eval[] = {&lb_state, &x, &lb_BV, &g0, &g1, &lb_dir, &consts, &lb_fpart, &nf, &nc, &ldpv};
    void* a_acc[] = {&lb_state, &lb_fpart, &c0v, &ldpv};
    void* a_upd[] = {&lb_state, &x, &lb_BV, &g0, &g1, &lb_dir, &lb_upart, &nf, &ldpv};
    void* a_ctl[] = {&lb_state, &lb_upart, &ldpv};
    auto slot = [&]() {
      DNLP_HIP_CHECK(hipModuleLaunchKernel(lb_eval, static_cast<unsigned>(blocks), 1, 1, kBlock, 1, 1, 0, stream, a_eval, nullptr));
      DNLP_HIP_CHECK(hipModuleLaunchKernel(lb_accept, 1, 1, 1, 64, 1, 1, 0, stream, a_acc, nullptr));
      DNLP_HIP_CHECK(hipModuleLaunchKernel(lb_update, static_cast<unsigned>(blocks), 1, 1, kBlock, 1, 1, 0, stream, a_upd, nullptr));
      DNLP_HIP_CHECK(hipModuleLaunchKernel(lb_control, 1, 1, 1, kBlock, 1, 1, 0, stream, a_ctl, nullptr));
    };
    // A batch of slots is one graph launch: the 64 kernel nodes are enqueued by the GPU's command processor back to
    // back (a dependent kernel boundary is ~1.5 us there; 64 eager hipModuleLaunchKernel calls keep a host thread busy
    // for longer than the kernels run).  The graph is captured once per argument set and replayed.
    const void* gkey[6] = {x, consts, lb_BV, lb_gt, reinterpret_cast<const void*>(static_cast<uintptr_t>(nfree)),
                           reinterpret_cast<const void*>(static_cast<uintptr_t>(blocks))};
    bool use_graph = lb_graph_on_ != 0;
    if (lb_graph_on_ < 0) { const char* ev = std::getenv("DNLP_LBFGS_GRAPH"); use_graph = !(ev && std::atoi(ev) == 0); lb_graph_on_ = use_graph ? 1 : 0; }
    if (use_graph && (!lb_graph_exec || std::memcmp(gkey, lb_graph_key, sizeof gkey) != 0 || lb_graph_c0 != c0)) {
      lb_graph_reset();
      hipGraph_t graph = nullptr;
      if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        for (int k = 0; k < lb_slots_per_batch; ++k) slot();
        if (hipStreamEndCapture(stream, &graph) == hipSuccess && graph &&
            hipGraphInstantiate(&lb_graph_exec, graph, nullptr, nullptr, 0) == hipSuccess) {
          std::memcpy(lb_graph_key, gkey, sizeof gkey);
          lb_graph_c0 = c0;
        } else {
          (void)hipGetLastError();
          lb_graph_exec = nullptr;
        }
        if (graph) hipGraphDestroy(graph);
      } else {
        (void)hipGetLastError();
      }
      if (!lb_graph_exec) { lb_graph_on_ = 0; use_graph = false; }     // a runtime that cannot capture: eager launches
    }
    int slots = 0;
    const long max_slots = static_cast<long>(max_iter) * 4 + 256;
    while (true) {
      if (use_graph && lb_graph_exec) { DNLP_HIP_CHECK(hipGraphLaunch(lb_graph_exec, stream)); slots += lb_slots_per_batch; }
      else for (int k = 0; k < lb_slots_per_batch; ++k, ++slots) slot();
      DNLP_HIP_CHECK(hipMemcpyAsync(lb_host, lb_state, sizeof(LbfgsState), hipMemcpyDeviceToHost, stream));
      DNLP_HIP_CHECK(hipStreamSynchronize(stream));
      if (lb_host->done != 0 || slots > max_slots) break;
    }
    out.slots = slots;
    out.iterations = lb_host->iter;
    out.evaluations = lb_host->evals;
    out.f = lb_host->f;
    out.gnorm = lb_host->gn;
    out.status = lb_host->done == 1 ? 0 : lb_host->done == 2 ? 3 : lb_host->done == 4 ? -13 : -1;
    out.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0;
    return true;
  }

  // launch only (timing loops: no read-back)
  bool fused_generated_launch(const std::vector<FusedSlotProg>& progs, const double* x, const double* consts, double* grad,
                              i64 nfree) {
    if (!fused_rtc.ok || fused_rtc_key != static_cast<const void*>(&progs)) return false;
    const i64 nchunks = (nfree + fused_E - 1) / fused_E;
    i64 blocks = (nchunks + kBlock - 1) / kBlock;
    if (blocks > kMaxPartials) blocks = kMaxPartials;
    i64 nf = nfree, nc = nchunks;
    void* args[] = {&x, &consts, &grad, &d_partial, &nf, &nc};
    DNLP_HIP_CHECK(hipModuleLaunchKernel(fused_rtc.fn, static_cast<unsigned>(blocks), 1, 1, kBlock, 1, 1, 0, stream, args, nullptr));
    return true;
  }
  // Order-fixed products through the tape's index by output (tape.h CooIdx, built when the tape is loaded — no lazy
  // index keyed by device pointers, nothing built inside a product): sixteen lanes per output, fixed reduction tree.
  void coo_gather(const CooIdx& ix, const double* a, const double* v, double* out) {
    if (ix.nout <= 0) return;
    if (ix.total <= 0) { DNLP_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(double) * static_cast<size_t>(ix.nout), stream)); return; }   // (the product ASSIGNS)
    hipLaunchKernelGGL(coo_rows_kernel, dim3(static_cast<unsigned>((ix.nout * 16 + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                       ix.nout, ix.ptr, ix.ent, ix.src, a, v, out);
  }
  // rectangular row-major Jacobian (BASELINE C3's dense constraint block): see rect_mult_kernel
  double* rect_part = nullptr;
  size_t rect_part_cap = 0;
  void rect_mult(i64 rows, i64 L, const i32* col, const double* a, const double* v, double* out) {
    hipLaunchKernelGGL(rect_mult_kernel, dim3(static_cast<unsigned>((rows + kBlock / 64 - 1) / (kBlock / 64))), dim3(kBlock), 0, stream,
                       rows, L, col, a, v, out);
  }
  void rect_tmult(i64 rows, i64 L, const i32* col, const double* a, const double* v, double* out) {
    // enough chunks to fill the chip: ceil(L / 256) workgroups per chunk, ~1024 workgroups in all
    const i64 wg = (L + kBlock - 1) / kBlock;
    int nchunks = static_cast<int>(std::min<i64>(std::max<i64>(1, 1024 / wg), (rows + 15) / 16));
    const int per = static_cast<int>((rows + nchunks - 1) / nchunks);
    nchunks = static_cast<int>((rows + per - 1) / per);
    const size_t need = static_cast<size_t>(nchunks) * static_cast<size_t>(L);
    if (need > rect_part_cap) { rect_part = alloc<double>(need); rect_part_cap = need; }     // (grow-only; the old buffer goes with the handle)
    hipLaunchKernelGGL(rect_tmult_partial_kernel, dim3(static_cast<unsigned>(wg), static_cast<unsigned>(nchunks)), dim3(kBlock), 0, stream,
                       rows, L, per, a, v, rect_part);
    hipLaunchKernelGGL(rect_tmult_finish_kernel, dim3(static_cast<unsigned>(wg)), dim3(kBlock), 0, stream, L, nchunks, col, rect_part, out);
  }
  // scatter products with floating-point atomics: patterns the tape did not index (above Tape::coo_index_max entries)
  void coo_product(i64 nnz, const i32* r, const i32* c, const double* a, const double* v, double* out, int mode) {
    if (nnz <= 0) return;
    hipLaunchKernelGGL(coo_mult_kernel, dim3(static_cast<unsigned>((nnz + kBlock - 1) / kBlock)), dim3(kBlock), 0, stream,
                       nnz, r, c, a, v, out, mode);
  }
  void coo_mult(i64 nnz, const i32* r, const i32* c, const double* a, const double* v, double* out, bool trans) {
    coo_product(nnz, r, c, a, v, out, trans ? 1 : 0);
  }
  void coo_sym_mult(i64 nnz, const i32* r, const i32* c, const double* a, const double* v, double* out) {
    coo_product(nnz, r, c, a, v, out, 2);
  }
  void dense_block_add(double* K, i64 ldk, i64 x0, const double* P, i64 ldp, i64 nb, double w, bool set) {
    const i64 nrb = (nb + 511) / 512;
    const unsigned gy = static_cast<unsigned>(nb < 16384 ? nb : 16384);
    hipLaunchKernelGGL(dense_block_add_kernel, dim3(static_cast<unsigned>(nrb), gy), dim3(kBlock), 0, stream,
                       K, ldk, x0, P, ldp, nb, w, set ? 1 : 0);
    DNLP_LAUNCH_CHECK();
  }

  // ---- factorisation (ldlt_blocked.h supplies the large-order path) ----
  void ldlt_prepare(LdltWork& w, i64 n, i64 ld, bool pivoted);
  bool ldlt_factor(LdltWork& w, double* A, i64 n, i64 ld, i32* ipiv, bool pivoted, int* nneg, int* nzero);
  void ldlt_solve(LdltWork& w, const double* A, i64 n, i64 ld, const i32* ipiv, bool pivoted, double* b);
  // least-squares multipliers of [I J^T; J -D][x; y] = [rx; ry] through the m x m Schur complement
  // S = -(D + J J^T): one K = N pass of the MFMA update kernel, an order-m LDL^T, one solve (ldlt_blocked.h).
  // jd_rhs: J rx (m values); y receives the multipliers.  False: not applicable / not definite.
  struct CondensedLs { BlockedLdlt* ldlt = nullptr; double* Jd = nullptr; double* S = nullptr; double* parts = nullptr; size_t parts_cap = 0; i64 N = 0, m = 0, Npad = 0, lds = 0; };
  CondensedLs cls_;
  bool condensed_ls(i64 N, i64 m, i64 nnzJ, const i32* jr, const i32* jc, const double* jv, const double* fixmask,
                    const double* Dd, const double* ry_minus_Jrx, double* y);

  // panel-blocked Bunch-Kaufman: ROWS x 1024 rows, NBP-column panels (bk_panel_kernel)
  template <int ROWS, int NBP>
  void bk_factor_panels(LdltWork& w, double* A, int ni, i64 ld, i32* ipiv) {
    static_assert(NBP <= 16, "BkPanelSwaps holds 16 interchanges");
    const i64 ldw = (ni + 7) / 8 * 8;
    const int rounds = (ni + NBP - 2) / (NBP - 1) + 1;       // a panel takes at least NBP - 1 columns
    const unsigned swap_blocks = static_cast<unsigned>((ni + kBlock - 1) / kBlock);
    for (int step = 0; step < rounds; ++step) {
      hipLaunchKernelGGL((bk_panel_kernel<ROWS, NBP>), dim3(1), dim3(BK_PT), 0, stream, A, ni, ld, ipiv, w.st, w.bk_w, ldw,
                         w.bk_swaps);
      const int t = ni - (step + 1) * (NBP - 1);             // upper bound of the trailing order after this panel
      const int ntr = t > 0 ? (t + kBlock - 1) / kBlock : 0, ntc = t > 0 ? (t + 15) / 16 : 0;
      const unsigned gx = std::max<unsigned>(static_cast<unsigned>(ntr * ntc), swap_blocks);
      hipLaunchKernelGGL((bk_panel_update_kernel<NBP>), dim3(gx, 2), dim3(kBlock), 0, stream, A, ni, ld, w.bk_w, ldw, w.st,
                         w.bk_swaps);
    }
  }

  bool bk_factor(LdltWork& w, double* A, i64 n, i64 ld, i32* ipiv, int* nneg, int* nzero) {
    BkState init;
    std::memset(&init, 0, sizeof init);
    DNLP_HIP_CHECK(hipMemcpyAsync(w.st, &init, sizeof init, hipMemcpyHostToDevice, stream));
    DNLP_HIP_CHECK(hipStreamSynchronize(stream));
    const int ni = static_cast<int>(n);
    if (w.bk_panels && ni > 32) {
      // rows per lane sized to the order: the register tile W (ROWS x 16 doubles) and the per-column work follow
      if (ni <= BK_PT) bk_factor_panels<1, 16>(w, A, ni, ld, ipiv);
      else if (ni <= 2 * BK_PT) bk_factor_panels<2, 16>(w, A, ni, ld, ipiv);
      else if (ni <= 3 * BK_PT) bk_factor_panels<3, 16>(w, A, ni, ld, ipiv);
      else if (ni <= 4 * BK_PT) bk_factor_panels<4, 8>(w, A, ni, ld, ipiv);
      else if (ni <= 6 * BK_PT) bk_factor_panels<6, 8>(w, A, ni, ld, ipiv);
      else bk_factor_panels<8, 8>(w, A, ni, ld, ipiv);
      hipLaunchKernelGGL(bk_finish_kernel, dim3(1), dim3(BK_T), 0, stream, ipiv, ni, w.bk_perm, w.bk_dtype);
      BkState out;
      DNLP_HIP_CHECK(hipMemcpyAsync(&out, w.st, sizeof out, hipMemcpyDeviceToHost, stream));
      DNLP_HIP_CHECK(hipStreamSynchronize(stream));
      *nneg = out.nneg;
      *nzero = out.nzero;
      return out.fail == 0;
    }
    for (int step = 0; step < ni; ++step) {
      hipLaunchKernelGGL(bk_pivot_kernel, dim3(1), dim3(BK_T), 0, stream, A, ni, ld, ipiv, w.st);
      const int t = ni - step - 1;   // upper bound of the trailing order at this step
      if (t > 0) {
        const int ntr = (t + kBlock - 1) / kBlock, ntc = (t + 15) / 16;
        // grid sized for the worst case (k = step); tiles beyond the actual trailing block exit
        hipLaunchKernelGGL(bk_update_kernel, dim3(static_cast<unsigned>(ntr * ntc)), dim3(kBlock), 0, stream, A, ni, ld, w.st);
      }
    }
    hipLaunchKernelGGL(bk_pivot_kernel, dim3(1), dim3(BK_T), 0, stream, A, ni, ld, ipiv, w.st);
    hipLaunchKernelGGL(bk_finish_kernel, dim3(1), dim3(BK_T), 0, stream, ipiv, ni, w.bk_perm, w.bk_dtype);
    BkState out;
    DNLP_HIP_CHECK(hipMemcpyAsync(&out, w.st, sizeof out, hipMemcpyDeviceToHost, stream));
    DNLP_HIP_CHECK(hipStreamSynchronize(stream));
    *nneg = out.nneg;
    *nzero = out.nzero;
    return out.fail == 0;
  }

 private:
  std::vector<void*> owned_;
};

}  // namespace dnlp
