// Blocked right-looking LDL^T (no pivoting) for large dense KKT matrices on gfx950.
//
//   for each outer panel of NB columns (512; 1024 from order 32768 on):
//     for each 128-column sub-panel (the 32-column chain below serves what is left of a panel):
//       ldlt_top128_kernel   — 128 x 128 diagonal block in the registers of one workgroup
//       ldlt_rows128_kernel  — rows below on FP64 MFMA, the row block transposed in the accumulators;
//                              stores L21 in place and W21 = L21 D in the panel workspace
//       gemm_nt_update_fast  — rest of the panel, K = 128
//     (32-column chain: ldlt_diag_kernel — 32 x 32 block by one wavefront in registers —,
//      ldlt_trsm_kernel — per-row forward substitution —, gemm_nt_update, K = 32)
//     gemm_nt_update_fast    — trailing matrix (Schur complement): A22 -= W21 L21^T, lower tiles only, on its
//                              own stream with one panel of look-ahead.  This is the n^3/3 part:
//                              v_mfma_f64_16x16x4_f64, 128 x 128 tile per 512-thread workgroup, a wavefront
//                              owns 64 x 32 (2 x 4 MFMA tiles, 64 accumulator VGPRs, four wavefronts per SIMD,
//                              two workgroups per CU), operand k-tiles go global -> LDS directly
//                              (global_load_lds_dwordx4), double buffered, one barrier per 16-deep k-tile,
//                              LDS rows padded by 16 doubles.  Edge tiles, unaligned operands and K not a
//                              multiple of 16 take gemm_nt_update: four wavefronts of 64 x 64, operands staged
//                              through registers.
//   The MFMA computes D[row=j][col=i] so that lane&15 walks the contiguous (row) direction
//   of the column-major C tile in the read-modify-write epilogue.
//
// Inertia is read off D (Sylvester); zero / tiny pivots are counted and replaced so the
// caller's regularisation loop (WB Algorithm IC) can react.
#pragma once
#include "exec_hip.h"

namespace dnlp {

typedef double mfma_d4 __attribute__((ext_vector_type(4)));

constexpr int LD_NB_MAX = 1024; // largest outer panel width (workspace)

constexpr int LD_nb = 32;      // inner block
constexpr int GM_BM = 128, GM_BN = 128, GM_BK = 16, GM_PAD = 16;

struct LdltInfo { int nneg, nzero, fail, pad; double dmax; };

// ---- diagonal block ----------------------------------------------------------------------
// One wavefront, lane = row, the row's 32 entries in registers (fully unrolled: static register
// indices); pivots and the multipliers of the pivot column travel by v_readlane, so a step is a
// handful of scalar broadcasts and FMAs with no LDS, no barrier and no global access: ~3 us per
// block against ~30 us for the LDS / barrier form it replaces (the panel chain is latency-bound:
// n / 32 dependent diag -> trsm -> update rounds).
__device__ inline double ldlt_bcast(double v, int srclane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
  return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(64) ldlt_diag_kernel(double* A, i64 ld, int j0, int jb, LdltInfo* info,
                                                       double tiny) {
  const int lane = threadIdx.x;
  const int row = lane < jb ? lane : jb - 1;            // clamped: loads stay inside the block
  double S[LD_nb];
#pragma unroll
  for (int c = 0; c < LD_nb; ++c) {
    const int cc = c < jb ? c : jb - 1;
    const double v = A[(j0 + row) + static_cast<i64>(j0 + cc) * ld];
    // rows / columns beyond jb: identity padding (never stored, never counted)
    S[c] = (lane < jb && c < jb) ? (lane >= c ? v : 0.0) : (lane == c ? 1.0 : 0.0);
  }
  int nneg = 0, nzero = 0, fail = 0;
#pragma unroll
  for (int k = 0; k < LD_nb; ++k) {
    double d = ldlt_bcast(S[k], k);
    if (k < jb) {
      if (!(d == d)) { fail = 1; d = 1.0; }
      if (fabs(d) <= tiny) { nzero += 1; d = (d < 0.0 ? -tiny : tiny); if (d == 0.0) d = 1e-300; }
      if (d < 0.0) nneg += 1;
    }
    const double di = 1.0 / d;
    const double mine = S[k];                            // unscaled pivot-column entry of this row
#pragma unroll
    for (int c = k + 1; c < LD_nb; ++c) S[c] -= mine * (ldlt_bcast(S[k], c) * di);
    S[k] = lane == k ? d : mine * di;
  }
  if (lane < jb) {
#pragma unroll
    for (int c = 0; c < LD_nb; ++c)
      if (c < jb && lane >= c) A[(j0 + lane) + static_cast<i64>(j0 + c) * ld] = S[c];
  }
  if (lane == 0 && (nneg | nzero | fail)) {
    info->nneg += nneg;
    info->nzero += nzero;
    if (fail) info->fail = 1;
  }
}

// ---- panel rows: forward substitution per row ----------------------------------------------
__global__ void __launch_bounds__(256) ldlt_trsm_kernel(double* A, i64 ld, int j0, int jb, int n,
                                                        double* Wp, i64 ldw, int wcol0) {
  __shared__ double Lb[LD_nb][LD_nb + 1];
  __shared__ double dv[LD_nb];
  const int tid = threadIdx.x;
  for (int e = tid; e < jb * jb; e += 256) {
    const int r = e % jb, c = e / jb;
    Lb[r][c] = (r > c) ? A[(j0 + r) + static_cast<i64>(j0 + c) * ld] : 0.0;
  }
  if (tid < jb) dv[tid] = A[(j0 + tid) + static_cast<i64>(j0 + tid) * ld];
  __syncthreads();
  const i64 r = static_cast<i64>(j0) + jb + static_cast<i64>(blockIdx.x) * 256 + tid;
  if (r >= n) return;
  double x[LD_nb];
#pragma unroll
  for (int c = 0; c < LD_nb; ++c) x[c] = (c < jb) ? A[r + static_cast<i64>(j0 + c) * ld] : 0.0;
#pragma unroll
  for (int c = 1; c < LD_nb; ++c) {
    if (c < jb) {
      double s = x[c];
#pragma unroll
      for (int t = 0; t < c; ++t) s -= x[t] * Lb[c][t];
      x[c] = s;
    }
  }
#pragma unroll
  for (int c = 0; c < LD_nb; ++c) {
    if (c < jb) {
      Wp[r + static_cast<i64>(wcol0 + c) * ldw] = x[c];
      A[r + static_cast<i64>(j0 + c) * ld] = x[c] / dv[c];
    }
  }
}

// ---- 128-column sub-panels --------------------------------------------------------------------
// The 32-column chain above costs a diag -> trsm -> update round of ~45 us of pure latency per 32
// columns whatever the row count.  For orders where the panel chain, not the MFMA update, is the
// critical path (a few thousand to a few ten thousand rows) a panel is taken 128 columns at a
// time instead:
//   ldlt_top128_kernel  — ONE workgroup factors the 128 x 128 diagonal block: the lower triangle
//                         lives in registers as 528 tiles of 4 x 4 (one tile per lane, static
//                         register indices), a step publishes the pivot column in LDS (two
//                         buffers: one barrier per column) and every tile right of it takes its
//                         rank-1 update from 8 LDS words.  It leaves a packed operand copy behind
//                         (LD_TOP_WS doubles): the 28 strictly-lower 16 x 16 blocks negated, the
//                         inverses of the 8 unit-lower diagonal 16 x 16 blocks, and 1 / d.
//   ldlt_rows128_kernel — rows below on FP64 MFMA, 16 rows per wavefront, the row block TRANSPOSED
//                         in the accumulators: X^T_c = Linv_cc (A^T_c - sum_{p<c} L_cp X^T_p) per
//                         16-column block c.  In the 16x16x4 layout an accumulator register of
//                         block p IS the B operand of k-slab s = register index, so the chain needs
//                         no shuffle, no LDS and no barrier; the A operands are 512-B coalesced reads
//                         of the packed copy (144 MFMAs per 16 rows).
constexpr int LD_T = 128;
constexpr int LD_TB = LD_T / 16;               // 16-column blocks of a sub-panel
constexpr int LD_TOP_THREADS = 576;            // 528 lower 4 x 4 tiles, whole wavefronts
constexpr int LD_TOP_NEG = 0;                                          // [pair(c,p)][k][i] = -L[16c+i][16p+k]
constexpr int LD_TOP_INV = (LD_TB * (LD_TB - 1) / 2) * 256;            // [c][k][i] = inv(L_cc)[i][k]
constexpr int LD_TOP_DINV = LD_TOP_INV + LD_TB * 256;                  // [col] = 1 / d
constexpr int LD_TOP_WS = LD_TOP_DINV + LD_T;

// 1 / d to full double precision without the IEEE division sequence (v_rcp_f64 + two Newton steps): the
// reciprocals of the four pivots of a tile sit back to back on the critical path of the top-block kernel
__device__ inline double ldlt_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(r, fma(-d, r, 1.0), r);
  r = fma(r, fma(-d, r, 1.0), r);
  return r;
}

__global__ void __launch_bounds__(LD_TOP_THREADS) ldlt_top128_kernel(double* __restrict__ A, i64 ld, int j0,
                                                                     LdltInfo* info, double tiny,
                                                                     double* __restrict__ Ltop) {
  // per tile column (4 matrix columns): the diagonal tile is factored by its lane alone and published
  // (barrier), the tiles below it finish their four columns from it and publish W = L D and L
  // (barrier), every tile to the right takes one rank-4 update.  Two barriers per four columns.
  __shared__ __attribute__((aligned(16))) double Wb[LD_T][4];
  __shared__ __attribute__((aligned(16))) double Lb[LD_T][4];
  __shared__ double dgL[4][4];
  __shared__ double dgI[4];
  __shared__ double Ld[LD_TB][16][17];          // unit-lower diagonal 16 x 16 blocks, for their inverses
  const int t = threadIdx.x;
  // tile columns in order, the tiles of a column top to bottom: finished columns are a thread prefix
  int tj = 0, rem = t;
  while (tj < LD_T / 4 && rem >= LD_T / 4 - tj) { rem -= LD_T / 4 - tj; ++tj; }
  const bool valid = tj < LD_T / 4;
  const int ti = valid ? tj + rem : 0;
  if (!valid) tj = -1;                          // never a pivot column, never right of one
  double T[4][4];
  {
    const double* src = A + (j0 + 4 * ti) + static_cast<i64>(j0 + 4 * (valid ? tj : 0)) * ld;
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int a = 0; a < 4; ++a) T[a][b] = src[a + static_cast<i64>(b) * ld];
  }
  double dinv[4] = {1.0, 1.0, 1.0, 1.0};
  int nneg = 0, nzero = 0, fail = 0;
  for (int kb = 0; kb < LD_T / 4; ++kb) {
    if (tj == kb && ti == kb) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        // the reciprocal starts on the raw pivot; the (rare) repair of a NaN / tiny pivot redoes it
        double d = T[kk][kk];
        double di = ldlt_rcp(d);
        if (!(fabs(d) > tiny)) {
          if (!(d == d)) { fail = 1; d = 1.0; }
          else { nzero += 1; d = (d < 0.0 ? -tiny : tiny); if (d == 0.0) d = 1e-300; }
          di = ldlt_rcp(d);
          T[kk][kk] = d;
        }
        nneg += d < 0.0 ? 1 : 0;
        dinv[kk] = di;
        dgI[kk] = di;
#pragma unroll
        for (int a = kk + 1; a < 4; ++a) {
          const double l = T[a][kk] * di;
#pragma unroll
          for (int b = kk + 1; b <= a; ++b) T[a][b] -= l * T[b][kk];
        }
#pragma unroll
        for (int a = kk + 1; a < 4; ++a) {
          T[a][kk] *= di;
          dgL[a][kk] = T[a][kk];
        }
      }
    }
    __syncthreads();
    if (tj == kb && ti > kb) {
      double w[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          double v = T[a][kk];
#pragma unroll
          for (int q = 0; q < kk; ++q) v -= w[a][q] * dgL[kk][q];
          w[a][kk] = v;
          T[a][kk] = v * dgI[kk];
        }
        *reinterpret_cast<double2*>(&Wb[4 * ti + a][0]) = double2{w[a][0], w[a][1]};
        *reinterpret_cast<double2*>(&Wb[4 * ti + a][2]) = double2{w[a][2], w[a][3]};
        *reinterpret_cast<double2*>(&Lb[4 * ti + a][0]) = double2{T[a][0], T[a][1]};
        *reinterpret_cast<double2*>(&Lb[4 * ti + a][2]) = double2{T[a][2], T[a][3]};
      }
    }
    __syncthreads();
    if (tj > kb) {
      double wi[4][4], lj[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const double2 w0v = *reinterpret_cast<const double2*>(&Wb[4 * ti + a][0]);
        const double2 w1v = *reinterpret_cast<const double2*>(&Wb[4 * ti + a][2]);
        const double2 l0v = *reinterpret_cast<const double2*>(&Lb[4 * tj + a][0]);
        const double2 l1v = *reinterpret_cast<const double2*>(&Lb[4 * tj + a][2]);
        wi[a][0] = w0v.x; wi[a][1] = w0v.y; wi[a][2] = w1v.x; wi[a][3] = w1v.y;
        lj[a][0] = l0v.x; lj[a][1] = l0v.y; lj[a][2] = l1v.x; lj[a][3] = l1v.y;
      }
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) T[a][b] -= wi[a][kk] * lj[b][kk];
    }
  }
  if (valid) {
    const int cb = ti >> 2, pb = tj >> 2;       // 16-blocks of the tile's rows / columns
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int r = 4 * ti + a, c = 4 * tj + b;
        if (r >= c) A[(j0 + r) + static_cast<i64>(j0 + c) * ld] = T[a][b];
        if (cb > pb) Ltop[LD_TOP_NEG + ((cb * (cb - 1) / 2 + pb) * 16 + (c & 15)) * 16 + (r & 15)] = -T[a][b];
        else if (r > c) Ld[cb][r & 15][c & 15] = T[a][b];
        else if (r == c) Ltop[LD_TOP_DINV + c] = dinv[a];
      }
  }
  __syncthreads();
  if (t < LD_T) {
    // column j of the inverse of the unit-lower block c: forward substitution on e_j
    const int c = t >> 4, j = t & 15;
    double y[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      double v = (i == j) ? 1.0 : 0.0;
#pragma unroll
      for (int k = 0; k < i; ++k) v -= Ld[c][i][k] * y[k];
      y[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) Ltop[LD_TOP_INV + (c * 16 + j) * 16 + i] = y[i];
  }
  if (nneg) atomicAdd(&info->nneg, nneg);
  if (nzero) atomicAdd(&info->nzero, nzero);
  if (fail) atomicExch(&info->fail, 1);
}

__global__ void __launch_bounds__(256) ldlt_rows128_kernel(double* __restrict__ A, i64 ld, int j0, int n,
                                                           double* __restrict__ Wp, i64 ldw, int wcol0,
                                                           const double* __restrict__ Ltop) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const i64 row0 = static_cast<i64>(j0) + LD_T + (static_cast<i64>(blockIdx.x) * 4 + wave) * 16;
  if (row0 >= n) return;                                  // whole wavefront; the kernel has no barrier
  const int lr = lane & 15, lq = lane >> 4;
  const i64 row = row0 + lr;
  const bool ok = row < n;
  const i64 rr = ok ? row : static_cast<i64>(n) - 1;      // clamped loads, predicated stores
  mfma_d4 D[LD_TB];
#pragma unroll
  for (int c = 0; c < LD_TB; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) D[c][r] = A[rr + static_cast<i64>(j0 + 16 * c + lq + 4 * r) * ld];
  const double* opn = Ltop + LD_TOP_NEG + lq * 16 + lr;
  const double* opi = Ltop + LD_TOP_INV + lq * 16 + lr;
#pragma unroll
  for (int c = 0; c < LD_TB; ++c) {
#pragma unroll
    for (int p = 0; p < c; ++p)
#pragma unroll
      for (int s = 0; s < 4; ++s)
        D[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(opn[((c * (c - 1) / 2 + p) * 16 + 4 * s) * 16], D[p][s], D[c], 0, 0, 0);
    mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s)
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(opi[(c * 16 + 4 * s) * 16], D[c][s], acc, 0, 0, 0);
    D[c] = acc;
  }
  if (ok) {
#pragma unroll
    for (int c = 0; c < LD_TB; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = 16 * c + lq + 4 * r;
        Wp[row + static_cast<i64>(wcol0 + col) * ldw] = D[c][r];
        A[row + static_cast<i64>(j0 + col) * ld] = D[c][r] * Ltop[LD_TOP_DINV + col];
      }
  }
}


// ---- a whole 512-column panel below its diagonal block in ONE launch (round 5) -------------------------------------
// Once the 512 x 512 diagonal block of a panel is factored (the sub-panel chain above on its own 512 rows: launches of
// 1-6 workgroups), every block of 16 rows below it is independent of every other: X^T_C = inv(L_CC) (A^T_C -
// sum_{P<C} L_CP X^T_P) over the panel's thirty-two 16-column blocks.  The sub-panel chain did that as 4 x (rows128 +
// in-panel update over ALL rows below) — eight chip-wide launches per panel on the critical path, each waiting for
// compute units beside the trailing update.  Here a wavefront carries its 16 rows through the four sub-panels alone:
// the eight accumulator blocks of the current sub-panel first take the products with the earlier sub-panels' W (read
// back from the panel workspace: the lane that wrote an entry is the lane that reads it — no fence, no barrier), then
// the chain of ldlt_rows128_kernel.  2112 MFMAs per 16 rows, 128 registers: it shares a compute unit with the
// trailing update's workgroups instead of waiting for an empty one.
//   Ltop4: the four packed operand copies ldlt_top128_*_kernel leave behind (LD_TOP_WS doubles apart)
//   Lpk:   the six off-diagonal 128 x 128 blocks of the diagonal block's L, packed by ldlt_pack512_kernel:
//          [(j (j - 1) / 2 + i) * 64 + c * 8 + p][k][i'] = -L[128 j + 16 c + i'][128 i + 16 p + k]
constexpr int LD_P = 512;
constexpr int LD_PK_WS = 6 * 64 * 256;

__global__ void __launch_bounds__(256) ldlt_pack512_kernel(const double* __restrict__ A, i64 ld, int K0, double* __restrict__ Lpk) {
  const int pair = blockIdx.x >> 6, c = (blockIdx.x >> 3) & 7, p = blockIdx.x & 7;
  int j = 1, base = 0;
  while (base + j <= pair) { base += j; ++j; }
  const int i = pair - base;
  const int t = threadIdx.x, ii = t & 15, k = t >> 4;
  Lpk[static_cast<i64>(blockIdx.x) * 256 + k * 16 + ii] =
      -A[(K0 + 128 * j + 16 * c + ii) + static_cast<i64>(K0 + 128 * i + 16 * p + k) * ld];
}

// Sixteen rows (one wavefront) through the sub-panels 0 .. nsub - 1 of a 512-column panel.  PACK: the rows belong to
// the diagonal block itself (ldlt_diag512_kernel) — their -L blocks also go to `pack` in operand layout.
struct LdltNoHook { __device__ void operator()(int) const {} };
// (pre / mid / post: called by the whole workgroup before sub-panel j's products, before its chain, after its stores —
//  the waits and the diagonal-block update of ldlt_diag512_kernel)
template <bool PACK, class Pre = LdltNoHook, class Mid = LdltNoHook, class Post = LdltNoHook>
__device__ __forceinline__ void ldlt_trsm16(double* __restrict__ A, i64 ld, int K0, i64 row, i64 rr, bool ok, double* __restrict__ Wp,
                                            i64 ldw, const double* __restrict__ Ltop4, const double* __restrict__ Lpk, int nsub,
                                            double* __restrict__ pack, Pre pre = Pre(), Mid mid = Mid(), Post post = Post()) {
  const int lane = threadIdx.x & 63;
  const int lr = lane & 15, lq = lane >> 4;
  // Operands travel in groups of 32 (one per MFMA of the group), two groups in flight: the loads of the next group are
  // issued BEFORE the products of the current one (the compiler, left alone, waits for every load right in front of its
  // MFMA — measured 470 cycles per product that way, 150 with the groups).
  double a0[32], a1[32], b0[4], b1[4];
#pragma unroll 1
  for (int j = 0; j < nsub; ++j) {
    const int j0 = K0 + LD_T * j;
    pre(j);
    mfma_d4 D[LD_TB];
#pragma unroll
    for (int c = 0; c < LD_TB; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) D[c][r] = A[rr + static_cast<i64>(j0 + 16 * c + lq + 4 * r) * ld];
    // the earlier sub-panels' share: D_c += (-L_(j,c),(i,p)) W^T_(i,p), step t = 8 i + p
    const double* pkj = Lpk + static_cast<i64>(j * (j - 1) / 2) * 64 * 256 + lq * 16 + lr;
    const double* wsrc = Wp + rr + static_cast<i64>(lq) * ldw;
    auto fetch = [&](double (&a)[32], double (&b)[4], int t) {
      const double* pk = pkj + static_cast<i64>(t >> 3) * 64 * 256 + (t & 7) * 256;
#pragma unroll
      for (int c = 0; c < LD_TB; ++c)
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) a[4 * c + sidx] = pk[c * 8 * 256 + 64 * sidx];
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) b[sidx] = wsrc[static_cast<i64>(16 * t + 4 * sidx) * ldw];
    };
    auto products = [&](const double (&a)[32], const double (&b)[4]) {
#pragma unroll
      for (int c = 0; c < LD_TB; ++c)
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) D[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[4 * c + sidx], b[sidx], D[c], 0, 0, 0);
    };
    const int nt = LD_TB * j;
    if (nt > 0) fetch(a0, b0, 0);
#pragma unroll 1
    for (int t = 0; t < nt; t += 2) {
      fetch(a1, b1, t + 1);
      asm volatile("" ::: "memory");
      products(a0, b0);
      if (t + 2 < nt) fetch(a0, b0, t + 2);
      asm volatile("" ::: "memory");
      products(a1, b1);
    }
    mid(j);
    // the sub-panel's own chain (ldlt_rows128_kernel), block c's operands fetched under block c - 1's products
    const double* top = Ltop4 + static_cast<i64>(j) * LD_TOP_WS;
    const double* opn = top + LD_TOP_NEG + lq * 16 + lr;
    const double* opi = top + LD_TOP_INV + lq * 16 + lr;
    auto fetch_c = [&](double (&a)[32], int c) {
#pragma unroll
      for (int p = 0; p < LD_TB - 1; ++p)
        if (p < c) {
#pragma unroll
          for (int sidx = 0; sidx < 4; ++sidx) a[4 * p + sidx] = opn[((c * (c - 1) / 2 + p) * 16 + 4 * sidx) * 16];
        }
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) a[28 + sidx] = opi[(c * 16 + 4 * sidx) * 16];
    };
    fetch_c(a0, 0);
#pragma unroll
    for (int c = 0; c < LD_TB; ++c) {
      double (&cur)[32] = (c & 1) ? a1 : a0;
      double (&nxt)[32] = (c & 1) ? a0 : a1;
      if (c + 1 < LD_TB) fetch_c(nxt, c + 1);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int p = 0; p < c; ++p)
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) D[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[4 * p + sidx], D[p][sidx], D[c], 0, 0, 0);
      mfma_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int sidx = 0; sidx < 4; ++sidx) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[28 + sidx], D[c][sidx], acc, 0, 0, 0);
      D[c] = acc;
    }
    if (ok) {
#pragma unroll
      for (int c = 0; c < LD_TB; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int col = 16 * c + lq + 4 * r;
          const double l = D[c][r] * top[LD_TOP_DINV + col];
          Wp[row + static_cast<i64>(LD_T * j + col) * ldw] = D[c][r];
          A[row + static_cast<i64>(j0 + col) * ld] = l;
          if (PACK) pack[static_cast<i64>(j) * 64 * 256 + c * 256 + 64 * r + lane] = -l;
        }
    }
    post(j);
  }
}

__global__ void __launch_bounds__(256, 2) ldlt_rows512_kernel(double* __restrict__ A, i64 ld, int K0, int n, double* __restrict__ Wp,
                                                              i64 ldw, const double* __restrict__ Ltop4, const double* __restrict__ Lpk) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const i64 row0 = static_cast<i64>(K0) + LD_P + (static_cast<i64>(blockIdx.x) * 4 + wave) * 16;
  if (row0 >= n) return;                                  // whole wavefront; the kernel has no barrier
  const i64 row = row0 + (lane & 15);
  const bool ok = row < n;
  const i64 rr = ok ? row : static_cast<i64>(n) - 1;      // clamped loads, predicated stores
  ldlt_trsm16<false>(A, ld, K0, row, rr, ok, Wp, ldw, Ltop4, Lpk, LD_P / LD_T, nullptr);
}

}  // namespace dnlp
#include "ldlt_top_mfma.h"
namespace dnlp {

// ---- the 512 x 512 diagonal block of a panel in ONE launch of four workgroups (round 5) ------------------------------
// Left-looking over its four 128-column sub-panels: workgroup j (eight wavefronts, wavefront w = its block row w) owns
// block row j.  For every sub-panel i < j:
//   1. its 16 rows through sub-panel i (ldlt_trsm16): the products with the sub-panels left of i as soon as workgroup i
//      has published ITS rows' packed -L (pk_done[i]), the chain as soon as workgroup i has factored its diagonal block
//      (top_done[i]); W to the panel workspace, L to the matrix, -L packed for ldlt_rows512_kernel and for step 2;
//   2. barrier; its block row of the 128 x 128 diagonal block takes A_(w,w-q)^T += (-L_(w-q),(i,p)) W_(w),(i,p)^T — the
//      B operands its own W read back, the A operands the packed copies the other wavefronts just wrote;
// then the block factorisation of ldlt_top128_mfma_kernel on the registers that hold the block row.  Behind the last
// top block of the chain there are only one chain (144 MFMAs), one update (32 (w + 1)) and the next top block; everything
// else runs under the top blocks of the workgroups above.  A workgroup waits only for LOWER workgroup indices (dispatch
// order: no co-residency assumption).  One workgroup for the whole block was measured first: 45 MFLOP on one compute
// unit's four FP64 matrix pipes (64 cycles per 16 x 16 x 4 on this part) — 460 us, the ten-launch chain 315.
struct PanelCtl { unsigned top_done[4]; unsigned pk_done[4]; unsigned abort; unsigned pad[23]; };

// all threads; false = gave up (a publisher that never came: the launch reports `fail` instead of hanging)
__device__ inline bool panel_wait(unsigned* flag, unsigned epoch, PanelCtl* ctl, int* s_ok) {
  if (threadIdx.x == 0) {
    unsigned spins = 0;
    int good = 1;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 1023u) == 0u && (spins > (1u << 22) || __hip_atomic_load(&ctl->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch)) {
        __hip_atomic_store(&ctl->abort, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        good = 0;
        break;
      }
    }
    *s_ok = good;
  }
  __syncthreads();
  __threadfence();          // (acquire side: what the publisher wrote before its flag)
  const bool ok = *s_ok != 0;
  __syncthreads();          // (every thread has read s_ok before thread 0 of the NEXT wait may write it: a slow wavefront must
                            //  not see a later give-up as the answer of this wait and leave the workgroup's barriers uneven)
  return ok;
}
__device__ inline void panel_publish(unsigned* flag, unsigned epoch) {
  __threadfence();          // every storing thread: its stores are out of this XCD's caches before the flag
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void __launch_bounds__(LD_TOPM_THREADS) ldlt_diag512_kernel(double* __restrict__ A, i64 ld, int K0, LdltInfo* info, double tiny,
                                                                      double* __restrict__ Ltop4, double* __restrict__ Lpk,
                                                                      double* __restrict__ Wp, i64 ldw, PanelCtl* ctl, unsigned epoch) {
  __shared__ __attribute__((aligned(16))) double negL[LD_TB][256];
  __shared__ double invS[16 * 17];
  __shared__ double dinvS[16];
  __shared__ int s_ok;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = static_cast<int>(blockIdx.x);
  const int j0 = K0 + LD_T * j;
  int nneg = 0, nzero = 0, fail = 0;
  bool alive = true;
  mfma_d4 T[LD_TB];
  if (j > 0) {
    const i64 row = static_cast<i64>(j0) + 16 * w + (lane & 15);
    double* pack = Lpk + (static_cast<i64>(j * (j - 1) / 2) * 64 + w * 8) * 256;
    const double* pkj = Lpk + static_cast<i64>(j * (j - 1) / 2) * 64 * 256 + lane;
    const double* wsrc = Wp + row + static_cast<i64>(lane >> 4) * ldw;
    double* tdst = A + (j0 + 16 * w + (lane & 15)) + static_cast<i64>(j0 + 16 * w + (lane >> 4)) * ld;
    auto pre = [&](int i) { if (i > 0 && alive) alive = panel_wait(&ctl->pk_done[i], epoch, ctl, &s_ok); };
    auto mid = [&](int i) { if (alive) alive = panel_wait(&ctl->top_done[i], epoch, ctl, &s_ok); };
    auto post = [&](int i) {
      // the packed -L of (j, i) is complete: later workgroups may take their products with it; this one its diagonal block
      if (i == j - 1) panel_publish(&ctl->pk_done[j], epoch);
      else __syncthreads();
      ldlt_top128_load(T, A, ld, j0);
      double a0[16], a1[16], bb[32];
#pragma unroll
      for (int pp = 0; pp < LD_TB; ++pp)
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) bb[4 * pp + sidx] = wsrc[static_cast<i64>(LD_T * i + 16 * pp + 4 * sidx) * ldw];
      // half a group (four column blocks of one q) in flight beside the half being multiplied
      auto fetchg = [&](double (&a)[16], int q, int h) {
        const double* pk = pkj + (static_cast<i64>(i) * 64 + (w - q) * 8 + 4 * h) * 256;
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
          for (int sidx = 0; sidx < 4; ++sidx) a[4 * pp + sidx] = pk[pp * 256 + 64 * sidx];
      };
      fetchg(a0, 0, 0);
#pragma unroll
      for (int q = 0; q < LD_TB; ++q)
        if (q <= w) {
          fetchg(a1, q, 1);
          asm volatile("" ::: "memory");
#pragma unroll
          for (int e = 0; e < 16; ++e) T[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[e], bb[e], T[q], 0, 0, 0);
          if (q + 1 < LD_TB && q + 1 <= w) fetchg(a0, q + 1, 0);
          asm volatile("" ::: "memory");
#pragma unroll
          for (int e = 0; e < 16; ++e) T[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[e], bb[16 + e], T[q], 0, 0, 0);
        }
      if (i < j - 1) {
        // back to the matrix until the next sub-panel's turn (the lane that stores an entry is the lane that reloads it)
#pragma unroll
        for (int q = 0; q < LD_TB; ++q)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if ((q > 0 && q <= w) || (q == 0 && (lane >> 4) + 4 * r <= (lane & 15))) tdst[static_cast<i64>(4 * r - 16 * q) * ld] = T[q][r];
      }
    };
    ldlt_trsm16<true>(A, ld, K0, row, row, true, Wp, ldw, Ltop4, Lpk, j, pack, pre, mid, post);
    // (the products filled the whole of T[0]; the factorisation wants zeros above its diagonal)
#pragma unroll
    for (int r = 0; r < 4; ++r) T[0][r] = ((lane >> 4) + 4 * r <= (lane & 15)) ? T[0][r] : 0.0;
  } else {
    ldlt_top128_load(T, A, ld, j0);
  }
  ldlt_top128_body(T, A, ld, j0, tiny, Ltop4 + static_cast<i64>(j) * LD_TOP_WS, nneg, nzero, fail, negL, invS, dinvS);
  panel_publish(&ctl->top_done[j], epoch);
  if (!alive) fail = 1;
  if (lane == 0) {
    if (nneg) atomicAdd(&info->nneg, nneg);
    if (nzero) atomicAdd(&info->nzero, nzero);
    if (fail) atomicExch(&info->fail, 1);
  }
}

// ---- C -= W L^T on FP64 MFMA ---------------------------------------------------------------
// Interior tiles (full 128x128, strictly below the diagonal, K a multiple of 16, 16-B aligned
// operands) take the fast body: the C tile is loaded straight into the MFMA accumulators
// (D = (-L) W^T + C, so the epilogue is a plain store), operands move global -> registers ->
// LDS as 16-B lanes with two LDS buffers and ONE barrier per 16-deep k-tile.  Edge / diagonal
// tiles take the guarded body.
__device__ inline void gemm_body_guarded(double* __restrict__ C, i64 ldc, const double* __restrict__ W, i64 ldw,
                                         const double* __restrict__ L, i64 ldl, int M, int Nc, int Kd, int lower,
                                         int tm, int tn, int vec_ok, double* smem) {
  double (*Ws)[GM_BM + GM_PAD] = reinterpret_cast<double (*)[GM_BM + GM_PAD]>(smem);
  double (*Ls)[GM_BN + GM_PAD] = reinterpret_cast<double (*)[GM_BN + GM_PAD]>(smem + GM_BK * (GM_BM + GM_PAD));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  mfma_d4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = mfma_d4{0.0, 0.0, 0.0, 0.0};
  const int srow = 2 * (tid & 63), sk = tid >> 6;
  const i64 gi = static_cast<i64>(tm) * GM_BM + srow;
  const i64 gj = static_cast<i64>(tn) * GM_BN + srow;
  double2 rw[4], rl[4];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = kt * GM_BK + sk + 4 * q;
      double2 vw = {0.0, 0.0}, vl = {0.0, 0.0};
      if (k < Kd) {
        const double* pw = W + gi + static_cast<i64>(k) * ldw;
        const double* pl = L + gj + static_cast<i64>(k) * ldl;
        if (vec_ok && gi + 1 < M) vw = *reinterpret_cast<const double2*>(pw);
        else { if (gi < M) vw.x = pw[0]; if (gi + 1 < M) vw.y = pw[1]; }
        if (vec_ok && gj + 1 < Nc) vl = *reinterpret_cast<const double2*>(pl);
        else { if (gj < Nc) vl.x = pl[0]; if (gj + 1 < Nc) vl.y = pl[1]; }
      }
      rw[q] = vw;
      rl[q] = vl;
    }
  };
  const int nkt = (Kd + GM_BK - 1) / GM_BK;
  load_tile(0);
  for (int kt = 0; kt < nkt; ++kt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<double2*>(&Ws[sk + 4 * q][srow]) = rw[q];
      *reinterpret_cast<double2*>(&Ls[sk + 4 * q][srow]) = rl[q];
    }
    __syncthreads();
    if (kt + 1 < nkt) load_tile(kt + 1);
#pragma unroll
    for (int kk = 0; kk < GM_BK; kk += 4) {
      double a[4], b[4];
      const int kr = kk + (lane >> 4), lc = lane & 15;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        a[t] = Ls[kr][wn * 64 + t * 16 + lc];
        b[t] = Ws[kr][wm * 64 + t * 16 + lc];
      }
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ni], b[mi], acc[ni][mi], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
      const i64 i = static_cast<i64>(tm) * GM_BM + wm * 64 + mi * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const i64 j = static_cast<i64>(tn) * GM_BN + wn * 64 + ni * 16 + (lane >> 4) + 4 * r;
        if (i < M && j < Nc && (!lower || i >= j)) C[i + j * ldc] -= acc[ni][mi][r];
      }
    }
}

// 512 threads = 8 waves; wave (wm, wn) owns rows wm*64..+64 (i) x cols wn*32..+32 (j):
// 2 x 4 MFMA tiles = 64 accumulator VGPRs, so four waves fit per SIMD.  Operand tiles go
// global -> LDS directly (global_load_lds_dwordx4: one 1-KiB k-row per wave instruction, no
// staging registers), double buffered, one barrier per 16-deep k-tile.  The accumulators
// start at -C and the result is stored negated (D = L W^T - C), so no operand needs a sign.
#define DNLP_GLDS(src, dst)                                                                         \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),            \
                                   (__attribute__((address_space(3))) void*)(dst), 16, 0, 0)

// MASKED = tile on the diagonal or on the last (partial) tile row / column: C accesses are
// predicated; operand rows past M / Nc are read from the (padded) allocations and only feed
// accumulators that are never stored.
template <bool MASKED>
__device__ inline void gemm_body_fast(double* __restrict__ C, i64 ldc, const double* __restrict__ W, i64 ldw,
                                      const double* __restrict__ L, i64 ldl, int Kd, int tm, int tn, double* smem,
                                      int M, int Nc, int lower) {
  constexpr int LDT = GM_BM + GM_PAD;                 // padded LDS row (doubles)
  constexpr int TILE = GM_BK * LDT;                    // doubles per operand tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
  // this wave moves k-rows `wave` and `wave + 8` of both operand tiles
  const double* pw = W + static_cast<i64>(tm) * GM_BM + 2 * lane + static_cast<i64>(wave) * ldw;
  const double* pl = L + static_cast<i64>(tn) * GM_BN + 2 * lane + static_cast<i64>(wave) * ldl;
  mfma_d4 acc[2][4];
  double* cbase = C + static_cast<i64>(tm) * GM_BM + wm * 64 + (lane & 15) +
                  (static_cast<i64>(tn) * GM_BN + wn * 32 + (lane >> 4)) * ldc;
  const int i_base = tm * GM_BM + wm * 64 + (lane & 15);      // + mi*16
  const int j_base = tn * GM_BN + wn * 32 + (lane >> 4);      // + ni*16 + 4r
  {
    const double* cp = cbase;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          if (MASKED) {
            const int i = i_base + mi * 16, j = j_base + ni * 16 + 4 * r;
            acc[ni][mi][r] = (i < M && j < Nc && (!lower || i >= j)) ? -cp[mi * 16] : 0.0;
          } else {
            acc[ni][mi][r] = -cp[mi * 16];
          }
        }
        cp += 4 * ldc;
      }
  }
  const int nkt = Kd / GM_BK;
  {
    double* ws = smem;
    double* ls = smem + TILE;
    DNLP_GLDS(pw, ws + wave * LDT);
    DNLP_GLDS(pw + 8 * ldw, ws + (wave + 8) * LDT);
    DNLP_GLDS(pl, ls + wave * LDT);
    DNLP_GLDS(pl + 8 * ldl, ls + (wave + 8) * LDT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const double* ws = smem + (kt & 1) * 2 * TILE;
    const double* ls = ws + TILE;
    if (kt + 1 < nkt) {
      double* wd = smem + ((kt + 1) & 1) * 2 * TILE;
      double* ldst = wd + TILE;
      const i64 ko = static_cast<i64>(kt + 1) * GM_BK;
      DNLP_GLDS(pw + ko * ldw, wd + wave * LDT);
      DNLP_GLDS(pw + (ko + 8) * ldw, wd + (wave + 8) * LDT);
      DNLP_GLDS(pl + ko * ldl, ldst + wave * LDT);
      DNLP_GLDS(pl + (ko + 8) * ldl, ldst + (wave + 8) * LDT);
    }
#pragma unroll
    for (int kk = 0; kk < GM_BK; kk += 4) {
      double a[2], b[4];
      const int kr = kk + (lane >> 4), lc = lane & 15;
#pragma unroll
      for (int t = 0; t < 2; ++t) a[t] = ls[kr * LDT + wn * 32 + t * 16 + lc];
#pragma unroll
      for (int t = 0; t < 4; ++t) b[t] = ws[kr * LDT + wm * 64 + t * 16 + lc];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ni], b[mi], acc[ni][mi], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  {
    double* cp = cbase;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
          if (MASKED) {
            const int i = i_base + mi * 16, j = j_base + ni * 16 + 4 * r;
            if (i < M && j < Nc && (!lower || i >= j)) cp[mi * 16] = -acc[ni][mi][r];
          } else {
            cp[mi * 16] = -acc[ni][mi][r];
          }
        }
        cp += 4 * ldc;
      }
  }
}

__device__ inline bool gemm_tile_interior(int tm, int tn, int M, int Nc, int Kd, int lower, int vec_ok,
                                          const double* C, i64 ldc) {
  return vec_ok && (Kd % GM_BK == 0) && (tm + 1) * GM_BM <= M && (tn + 1) * GM_BN <= Nc &&
         (!lower || tm * GM_BM >= (tn + 1) * GM_BN) && ((ldc & 1) == 0) &&
         ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
}

// all tiles when the operands are 16-B aligned and K is a multiple of 16 (host-side check)
__global__ void __launch_bounds__(512, 4) gemm_nt_update_fast(double* __restrict__ C, i64 ldc,
                                                              const double* __restrict__ W, i64 ldw,
                                                              const double* __restrict__ L, i64 ldl, int M,
                                                              int Nc, int Kd, int lower, int ntm, int vec_ok) {
  int tm, tn;
  if (vec_ok & 2) {
    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so the 64
    // workgroups resident on one XCD (equal id % 8, consecutive id / 8) are mapped onto one
    // 8 x 8 super-tile and share its 8 W strips and 8 L strips through that XCD's L2 instead
    // of each streaming its own 512-KB W strip from HBM (speed only; any placement is correct)
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int within = seq & 63, super = (seq >> 6) * 8 + xcd;
    const int nsm = (ntm + 7) >> 3;
    tm = (super % nsm) * 8 + (within & 7);
    tn = (super / nsm) * 8 + (within >> 3);
    if (tm >= ntm || tn * GM_BN >= Nc) return;
  } else {
    tm = blockIdx.x % ntm;
    tn = blockIdx.x / ntm;
  }
  if (lower && (tm * GM_BM + GM_BM - 1 < tn * GM_BN)) return;
  __shared__ __attribute__((aligned(16))) double smem[4 * GM_BK * (GM_BM + GM_PAD)];
  if (gemm_tile_interior(tm, tn, M, Nc, Kd, lower, vec_ok, C, ldc))
    gemm_body_fast<false>(C, ldc, W, ldw, L, ldl, Kd, tm, tn, smem, M, Nc, lower);
  else
    gemm_body_fast<true>(C, ldc, W, ldw, L, ldl, Kd, tm, tn, smem, M, Nc, lower);
}

// every tile through the guarded path (unaligned operands or K not a multiple of 16)
__global__ void __launch_bounds__(256, 2) gemm_nt_update(double* __restrict__ C, i64 ldc,
                                                         const double* __restrict__ W, i64 ldw,
                                                         const double* __restrict__ L, i64 ldl, int M,
                                                         int Nc, int Kd, int lower, int ntm, int vec_ok) {
  const int tm = blockIdx.x % ntm, tn = blockIdx.x / ntm;
  if (lower && (tm * GM_BM + GM_BM - 1 < tn * GM_BN)) return;
  __shared__ __attribute__((aligned(16))) double smem[2 * GM_BK * (GM_BM + GM_PAD)];
  gemm_body_guarded(C, ldc, W, ldw, L, ldl, M, Nc, Kd, lower, tm, tn, vec_ok, smem);
}

// 64 x 64 tiles, four wavefronts (2 x 2, each 32 x 32 = 2 x 2 MFMA tiles), register-staged operands in two LDS
// buffers with one barrier per 16-deep k-tile.  For launches whose 128 x 128 tiling gives fewer workgroups than the
// chip has room for (in-panel updates of mid-size orders: 80-240 tiles for 256 CUs).  A workgroup's k-loop is a
// chain of global round trips (16 MFMAs per wavefront per k-tile against ~1.5 us of latency), so operand tiles are
// fetched TWO k-tiles ahead into two register sets.  GUARD = edge / diagonal tile or unaligned operands.
constexpr int GS_B = 64, GS_PAD = 16;   // k-rows 640 B apart: the two k-rows a half-wavefront reads fall on disjoint LDS banks
template <bool GUARD>
__device__ inline void gemm_body_small(double* __restrict__ C, i64 ldc, const double* __restrict__ W, i64 ldw,
                                       const double* __restrict__ L, i64 ldl, int M, int Nc, int Kd, int lower,
                                       int tm, int tn, int vec_ok, double (*Ws)[GM_BK][GS_B + GS_PAD],
                                       double (*Ls)[GM_BK][GS_B + GS_PAD]) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int srow = 2 * (tid & 31), sk = tid >> 5;
  const i64 gi = static_cast<i64>(tm) * GS_B + srow;
  const i64 gj = static_cast<i64>(tn) * GS_B + srow;
  const int nkt = (Kd + GM_BK - 1) / GM_BK;
  // two register sets, named (not indexed, not passed by address) so that they stay in registers
  double2 rw0a, rw0b, rl0a, rl0b, rw1a, rw1b, rl1a, rl1b;
  auto load_one = [&](int k, double2& vw, double2& vl) {
    if (!GUARD) {
      vw = *reinterpret_cast<const double2*>(W + gi + static_cast<i64>(k) * ldw);
      vl = *reinterpret_cast<const double2*>(L + gj + static_cast<i64>(k) * ldl);
    } else {
      double wx = 0.0, wy = 0.0, lx = 0.0, ly = 0.0;
      if (k < Kd) {
        const double* pw = W + gi + static_cast<i64>(k) * ldw;
        const double* pl = L + gj + static_cast<i64>(k) * ldl;
        if (vec_ok && gi + 1 < M) { const double2 v = *reinterpret_cast<const double2*>(pw); wx = v.x; wy = v.y; }
        else { if (gi < M) wx = pw[0]; if (gi + 1 < M) wy = pw[1]; }
        if (vec_ok && gj + 1 < Nc) { const double2 v = *reinterpret_cast<const double2*>(pl); lx = v.x; ly = v.y; }
        else { if (gj < Nc) lx = pl[0]; if (gj + 1 < Nc) ly = pl[1]; }
      }
      vw = double2{wx, wy};
      vl = double2{lx, ly};
    }
  };
#define DNLP_GS_LOAD(kt, S)                                   \
  do {                                                        \
    load_one((kt) * GM_BK + sk, rw##S##a, rl##S##a);          \
    load_one((kt) * GM_BK + sk + 8, rw##S##b, rl##S##b);      \
  } while (0)
#define DNLP_GS_STORE(buf, S)                                              \
  do {                                                                     \
    *reinterpret_cast<double2*>(&Ws[buf][sk][srow]) = rw##S##a;            \
    *reinterpret_cast<double2*>(&Ls[buf][sk][srow]) = rl##S##a;            \
    *reinterpret_cast<double2*>(&Ws[buf][sk + 8][srow]) = rw##S##b;        \
    *reinterpret_cast<double2*>(&Ls[buf][sk + 8][srow]) = rl##S##b;        \
  } while (0)
  DNLP_GS_LOAD(0, 0);
  DNLP_GS_LOAD(nkt > 1 ? 1 : 0, 1);
  const int i_base = tm * GS_B + wm * 32 + (lane & 15);
  const int j_base = tn * GS_B + wn * 32 + (lane >> 4);
  mfma_d4 acc[2][2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i_base + mi * 16, j = j_base + ni * 16 + 4 * r;
        if (GUARD) acc[ni][mi][r] = (i < M && j < Nc && (!lower || i >= j)) ? -C[i + static_cast<i64>(j) * ldc] : 0.0;
        else acc[ni][mi][r] = -C[i + static_cast<i64>(j) * ldc];
      }
  auto compute = [&](int buf) {
#pragma unroll
    for (int kk = 0; kk < GM_BK; kk += 4) {
      double a[2], b[2];
      const int kr = kk + (lane >> 4), lc = lane & 15;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        a[t] = Ls[buf][kr][wn * 32 + t * 16 + lc];
        b[t] = Ws[buf][kr][wm * 32 + t * 16 + lc];
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ni], b[mi], acc[ni][mi], 0, 0, 0);
    }
  };
  DNLP_GS_STORE(0, 0);
  __syncthreads();
  // k-tile kt is computed from LDS buffer kt & 1; tile kt + 1 waits in a register set, tile kt + 2 is in flight
  for (int kt = 0; kt < nkt; kt += 2) {
    DNLP_GS_LOAD(kt + 2 < nkt ? kt + 2 : nkt - 1, 0);   // unconditional (clamped): a branch here would make the
                                                        // compiler wait for every outstanding load at the store
    compute(0);
    DNLP_GS_STORE(1, 1);
    __syncthreads();
    if (kt + 1 >= nkt) break;
    DNLP_GS_LOAD(kt + 3 < nkt ? kt + 3 : nkt - 1, 1);
    compute(1);
    DNLP_GS_STORE(0, 0);
    __syncthreads();
  }
#undef DNLP_GS_LOAD
#undef DNLP_GS_STORE
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i_base + mi * 16, j = j_base + ni * 16 + 4 * r;
        if (!GUARD || (i < M && j < Nc && (!lower || i >= j))) C[i + static_cast<i64>(j) * ldc] = -acc[ni][mi][r];
      }
}

__global__ void __launch_bounds__(256, 3) gemm_nt_update_small(double* __restrict__ C, i64 ldc,
                                                            const double* __restrict__ W, i64 ldw,
                                                            const double* __restrict__ L, i64 ldl, int M, int Nc,
                                                            int Kd, int lower, int ntm, int vec_ok) {
  const int tm = blockIdx.x % ntm, tn = blockIdx.x / ntm;
  if (lower && (tm * GS_B + GS_B - 1 < tn * GS_B)) return;
  __shared__ __attribute__((aligned(16))) double Ws[2][GM_BK][GS_B + GS_PAD];
  __shared__ __attribute__((aligned(16))) double Ls[2][GM_BK][GS_B + GS_PAD];
  const bool interior = vec_ok && (Kd % GM_BK == 0) && (tm + 1) * GS_B <= M && (tn + 1) * GS_B <= Nc &&
                        (!lower || tm * GS_B >= (tn + 1) * GS_B);
  if (interior) gemm_body_small<false>(C, ldc, W, ldw, L, ldl, M, Nc, Kd, lower, tm, tn, vec_ok, Ws, Ls);
  else gemm_body_small<true>(C, ldc, W, ldw, L, ldl, M, Nc, Kd, lower, tm, tn, vec_ok, Ws, Ls);
}

// The same tiles with the k range cut into gridDim.y slices, slice y accumulating into its own copy of C (cslice doubles
// apart; the caller adds the copies in slice order: reproducible, no atomics).  For products with few output tiles and a
// long inner dimension (J J^T of the least-squares multipliers: 136 tiles x K = 10 000 kept 136 workgroups busy for 0.92 ms).
__global__ void __launch_bounds__(256, 3) gemm_nt_update_small_splitk(double* __restrict__ C, i64 ldc, i64 cslice,
                                                                   const double* __restrict__ W, i64 ldw,
                                                                   const double* __restrict__ L, i64 ldl, int M, int Nc,
                                                                   int Kd, int kslice, int lower, int ntm, int vec_ok) {
  const int tm = blockIdx.x % ntm, tn = blockIdx.x / ntm;
  if (lower && (tm * GS_B + GS_B - 1 < tn * GS_B)) return;
  const int k0 = static_cast<int>(blockIdx.y) * kslice;
  if (k0 >= Kd) return;
  const int kd = (Kd - k0 < kslice) ? Kd - k0 : kslice;
  C += static_cast<i64>(blockIdx.y) * cslice;
  W += static_cast<i64>(k0) * ldw;
  L += static_cast<i64>(k0) * ldl;
  __shared__ __attribute__((aligned(16))) double Ws[2][GM_BK][GS_B + GS_PAD];
  __shared__ __attribute__((aligned(16))) double Ls[2][GM_BK][GS_B + GS_PAD];
  const bool interior = vec_ok && (kd % GM_BK == 0) && (tm + 1) * GS_B <= M && (tn + 1) * GS_B <= Nc &&
                        (!lower || tm * GS_B >= (tn + 1) * GS_B);
  if (interior) gemm_body_small<false>(C, ldc, W, ldw, L, ldl, M, Nc, kd, lower, tm, tn, vec_ok, Ws, Ls);
  else gemm_body_small<true>(C, ldc, W, ldw, L, ldl, M, Nc, kd, lower, tm, tn, vec_ok, Ws, Ls);
}

// ---- triangular solves with the unit-lower factor ---------------------------------------------
// Blocks of SV_B = 256 columns: the diagonal block is solved by one workgroup (32-wide
// wave-shuffle substitutions, no barrier inside a sub-block), the panel below / the panel
// transposed are bandwidth kernels that stream the factor once per solve (n^2/2 doubles).
constexpr int SV_B = 256;

// Every round s0 needs, for row t >= s0, the 32 entries L[t, s0 .. s0+32): the diagonal sub-block
// rows use them in a shuffle substitution, the rows below in a 32-term update.  They are fetched one
// round AHEAD (unconditional, clamped addresses, so the compiler issues them back to back), which
// takes the global round trip out of the 8-round dependency chain.
__global__ void __launch_bounds__(SV_B) ldlt_fwd_diag(const double* __restrict__ A, i64 ld, int j0, int jb,
                                                      double* __restrict__ b) {
  __shared__ double y[SV_B];
  const int t = threadIdx.x;
  const int tr = t < jb ? t : jb - 1;
  double v = (t < jb) ? b[j0 + t] : 0.0;
  double cur[32], nxt[32];
  const double* row = A + (j0 + tr) + static_cast<i64>(j0) * ld;
#pragma unroll
  for (int kk = 0; kk < 32; ++kk) cur[kk] = row[static_cast<i64>(kk < jb ? kk : jb - 1) * ld];
  for (int s0 = 0; s0 < jb; s0 += 32) {
    const int s1 = (s0 + 32 < jb) ? s0 + 32 : jb;
    // rows above a strip hold no entry of it (upper triangle): a wavefront whose 64 rows all lie above skips the
    // strip — the kernel is bound by the issue rate of these loads on its one CU
    if (s1 < jb && (t | 63) >= s1) {
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) nxt[kk] = row[static_cast<i64>(s1 + kk < jb ? s1 + kk : jb - 1) * ld];
    }
    if ((t >> 6) == (s0 >> 6)) {
      const bool mine = t >= s0 && t < s1;
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) {
        // the masked multiplier does not depend on v: the chain is v_readlane -> one FMA per step
        const double ck = (mine && s0 + kk < t) ? -cur[kk] : 0.0;
        const double yk = ldlt_bcast(v, __builtin_amdgcn_readfirstlane((s0 + kk) & 63));   // uniform lane: v_readlane, not ds_bpermute
        v = fma(ck, yk, v);
      }
      if (mine) y[t] = v;
    }
    __syncthreads();
    if (t >= s1 && t < jb) {
      double sa = 0.0, sb = 0.0;
#pragma unroll
      for (int kk = 0; kk < 32; kk += 2) {
        sa += (s0 + kk < s1) ? cur[kk] * y[s0 + kk] : 0.0;
        sb += (s0 + kk + 1 < s1) ? cur[kk + 1] * y[s0 + kk + 1] : 0.0;
      }
      v -= sa + sb;
    }
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) cur[kk] = nxt[kk];
  }
  if (t < jb) b[j0 + t] = v;
}
// rows below the block: b[r] -= sum_t A[r, j0+t] y[t].  A workgroup takes 64 rows; its four wavefronts
// split the block's columns (64 loads per lane instead of 256, four times the workgroups: the kernel is
// bound by the per-lane load chain, not by bandwidth) and meet in LDS.
__global__ void __launch_bounds__(256) ldlt_fwd_update(const double* __restrict__ A, i64 ld, int j0, int jb,
                                                       int n, double* __restrict__ b) {
  __shared__ double y[SV_B];
  __shared__ double part[4][64];
  if (threadIdx.x < jb) y[threadIdx.x] = b[j0 + threadIdx.x];
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const i64 r = static_cast<i64>(j0) + jb + static_cast<i64>(blockIdx.x) * 64 + lane;
  const int c0 = w * 64, c1 = (c0 + 64 < jb) ? c0 + 64 : jb;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (r < n) {
    const double* row = A + r + static_cast<i64>(j0) * ld;
    int t = c0;
    for (; t + 4 <= c1; t += 4) {
      s0 += row[static_cast<i64>(t) * ld] * y[t];
      s1 += row[static_cast<i64>(t + 1) * ld] * y[t + 1];
      s2 += row[static_cast<i64>(t + 2) * ld] * y[t + 2];
      s3 += row[static_cast<i64>(t + 3) * ld] * y[t + 3];
    }
    for (; t < c1; ++t) s0 += row[static_cast<i64>(t) * ld] * y[t];
  }
  part[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && r < n) b[r] -= (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
}
__global__ void __launch_bounds__(256) ldlt_diag_scale(const double* A, i64 ld, int n, double* b) {
  const i64 i = static_cast<i64>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) b[i] /= A[i + i * ld];
}
// transposed panel: acc[t] += sum_{r in chunk} A[r, j0+t] x[r]; one wave per column, lanes
// walk the rows (coalesced), grid.y walks row chunks.  acc is zeroed by the diagonal kernel.
__global__ void __launch_bounds__(256) ldlt_bwd_update(const double* __restrict__ A, i64 ld, int j0, int jb,
                                                       int n, const double* __restrict__ b,
                                                       double* __restrict__ acc, int rows_per_chunk) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int t = blockIdx.x * 4 + wid;
  if (t >= jb) return;
  const i64 rbeg = static_cast<i64>(j0) + jb + static_cast<i64>(blockIdx.y) * rows_per_chunk;
  i64 rend = rbeg + rows_per_chunk;
  if (rend > n) rend = n;
  const double* col = A + static_cast<i64>(j0 + t) * ld;
  double s0 = 0.0, s1 = 0.0;
  i64 r = rbeg + lane;
  for (; r + 64 < rend; r += 128) {
    s0 += col[r] * b[r];
    s1 += col[r + 64] * b[r + 64];
  }
  if (r < rend) s0 += col[r] * b[r];
  const double w = wave_sum(s0 + s1);
  if (lane == 0) unsafeAtomicAdd(&acc[t], w);
}
// diagonal block, transposed: x = L11^-T (b1 - acc); clears acc for the next block
__global__ void __launch_bounds__(SV_B) ldlt_bwd_diag(const double* __restrict__ A, i64 ld, int j0, int jb,
                                                      double* __restrict__ b, double* __restrict__ acc) {
  __shared__ double xs[SV_B];
  const int t = threadIdx.x;
  const int tc = t < jb ? t : jb - 1;
  double v = (t < jb) ? b[j0 + t] - acc[t] : 0.0;
  if (t < SV_B) acc[t] = 0.0;
  const int nsb = (jb + 31) / 32;
  // column t of the block, rows s0 .. s0+32 of every round, fetched one round ahead (see ldlt_fwd_diag)
  const double* colt = A + j0 + static_cast<i64>(j0 + tc) * ld;
  double cur[32], nxt[32];
  {
    const int s0 = (nsb - 1) * 32;
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) cur[kk] = colt[s0 + kk < jb ? s0 + kk : jb - 1];
  }
  for (int sbk = nsb - 1; sbk >= 0; --sbk) {
    const int s0 = sbk * 32, s1 = (s0 + 32 < jb) ? s0 + 32 : jb;
    // (columns right of a strip's rows hold no entry of it either)
    if (sbk > 0 && (t & ~63) < s0) {
#pragma unroll
      for (int kk = 0; kk < 32; ++kk) nxt[kk] = colt[s0 - 32 + kk];
    }
    if ((t >> 6) == (s0 >> 6)) {
      const bool mine = t >= s0 && t < s1;
#pragma unroll
      for (int kk = 31; kk >= 0; --kk) {
        const double ck = (mine && s0 + kk > t && s0 + kk < s1) ? -cur[kk] : 0.0;
        const double xk = ldlt_bcast(v, __builtin_amdgcn_readfirstlane((s0 + kk) & 63));
        v = fma(ck, xk, v);
      }
      if (mine) xs[t] = v;
    }
    __syncthreads();
    if (t < s0) {
      double sa = 0.0, sb = 0.0;
#pragma unroll
      for (int kk = 0; kk < 32; kk += 2) {
        sa += (s0 + kk < s1) ? cur[kk] * xs[s0 + kk] : 0.0;
        sb += (s0 + kk + 1 < s1) ? cur[kk + 1] * xs[s0 + kk + 1] : 0.0;
      }
      v -= sa + sb;
    }
#pragma unroll
    for (int kk = 0; kk < 32; ++kk) cur[kk] = nxt[kk];
  }
  if (t < jb) b[j0 + t] = v;
}

// ---- triangular solves on inverted diagonal blocks (round 3) ---------------------------------------------------
// The 256-wide diagonal solves above are chains of dependent substitutions on ONE compute unit (28 us per block,
// 30 % of a C3 solve).  After a factorisation the unit-lower 128 x 128 diagonal blocks are inverted once
// (ldlt_inv128_kernel, all blocks in parallel, off the solves' path); a 256-column block step of a solve is then
// products only — inverse x vector, 128 x 128 coupling block x vector, inverse x vector — and ONE launch per
// step: workgroup 0 brings the NEXT block's right-hand side up to date and solves it while the other workgroups
// update the rows (columns, in the transposed sweep) further away with the block that is already final.
constexpr int SI_B = 256;           // columns per step
constexpr int SI_H = 128;           // order of an inverted diagonal block
constexpr int SI_T = 1024;          // threads per workgroup of the step kernels

// inv / invT: per 128-block 128 x 128 doubles, column-major, X = L_bb^-1 and its transpose.  One workgroup of 128
// lanes per block: each wavefront inverts one 64 x 64 diagonal half with a column per lane in registers (static
// indices; L is read as LDS broadcasts), then M = -X1 (C X0) for the coupling block C = L[64.., ..64) in two
// products whose operands a lane holds as a register row / column.  Rows and columns past n are identity.
__global__ void __launch_bounds__(128) ldlt_inv128_kernel(const double* __restrict__ A, i64 ld, int n,
                                                          double* __restrict__ inv, double* __restrict__ invT) {
  extern __shared__ __align__(16) double si_lds[];
  double* Ls = si_lds;                 // [2][64 * 64]: the two diagonal halves, later their inverses
  double* Cs = si_lds + 2 * 4096;      // [64 * 64]: C, then T = C X0, then M
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j0 = blockIdx.x * SI_H;
  {
    // strictly lower part of diagonal half w (unit diagonal implied), row `lane`
    double* Lw = Ls + w * 4096;
    const int gi = j0 + 64 * w + lane;
    for (int j = 0; j < 64; ++j) {
      const int gj = j0 + 64 * w + j;
      Lw[lane * 64 + j] = (j < lane && gi < n) ? A[gi + static_cast<i64>(gj) * ld] : 0.0;
    }
    // coupling block: row `lane`, columns 32 w .. 32 w + 32
    const int ci = j0 + 64 + lane;
    for (int j = 32 * w; j < 32 * w + 32; ++j)
      Cs[lane * 64 + j] = (ci < n) ? A[ci + static_cast<i64>(j0 + j) * ld] : 0.0;
  }
  __syncthreads();
  double x[64];
  {
    const double* Lw = Ls + w * 4096;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      double sacc = (i == lane) ? 1.0 : 0.0;
#pragma unroll
      for (int j = 0; j < i; ++j) sacc = fma(-Lw[i * 64 + j], x[j], sacc);
      x[i] = sacc;
    }
  }
  __syncthreads();                      // every lane is done reading L
  {
    double* Xw = Ls + w * 4096;         // X_w[i][t] at i * 64 + t
#pragma unroll
    for (int i = 0; i < 64; ++i) Xw[i * 64 + lane] = x[i];
  }
  // T = C X0: lane owns row `lane` of C (registers) and 32 columns of the result
#pragma unroll
  for (int k = 0; k < 64; ++k) x[k] = Cs[lane * 64 + k];
  __syncthreads();                      // X0, X1 in LDS; every lane holds its row of C
  {
    const double* X0 = Ls;
    for (int j = 32 * w; j < 32 * w + 32; ++j) {
      double sacc = 0.0;
#pragma unroll
      for (int k = 0; k < 64; ++k) sacc = fma(x[k], X0[k * 64 + j], sacc);   // (X0[k][j] = 0 for k < j)
      Cs[lane * 64 + j] = sacc;         // row `lane` is this lane pair's own: no other lane reads it before the barrier
    }
  }
  __syncthreads();
  // M = -X1 T: lane owns column `lane` of T (registers) and 32 rows of the result
#pragma unroll
  for (int k = 0; k < 64; ++k) x[k] = Cs[k * 64 + lane];
  __syncthreads();
  {
    const double* X1 = Ls + 4096;
    for (int i = 32 * w; i < 32 * w + 32; ++i) {
      double sacc = 0.0;
#pragma unroll
      for (int k = 0; k < 64; ++k) sacc = fma(X1[i * 64 + k], x[k], sacc);   // (X1[i][k] = 0 for k > i)
      Cs[i * 64 + lane] = -sacc;
    }
  }
  __syncthreads();
  double* out = inv + static_cast<i64>(blockIdx.x) * (SI_H * SI_H);
  double* outT = invT + static_cast<i64>(blockIdx.x) * (SI_H * SI_H);
  auto elem = [&](int r, int c) -> double {
    if (r < 64) return c < 64 ? Ls[r * 64 + c] : 0.0;
    return c < 64 ? Cs[(r - 64) * 64 + c] : Ls[4096 + (r - 64) * 64 + (c - 64)];
  };
  for (int idx = tid; idx < SI_H * SI_H; idx += 128) {
    const int lo = idx & (SI_H - 1), hi = idx >> 7;
    out[idx] = elem(lo, hi);            // column-major X: idx = r + 128 c
    outT[idx] = elem(hi, lo);           // column-major X^T: idx = c + 128 r
  }
}

}  // namespace dnlp
#include "ldlt_inv_mfma.h"
namespace dnlp {

// Operands of a 128 x 128 product held by the SI_T lanes: lane (i = tid & 127, ch = tid >> 7) keeps the sixteen
// entries M[i + ldm (16 ch + c)], c < 16 — fetched at kernel start, long before the vector they multiply exists,
// so the dependent part of a step is LDS traffic only.
// Entry `elem` (in doubles) behind a base that is uniform across the wavefront: the load is scalar base + 32-bit lane
// offset.  (With 64-bit index arithmetic every load of an unrolled group keeps its own 64-bit address pair in VGPRs:
// 64 registers for 32 loads, and these kernels have 128 per lane.)  Valid while elem * 8 < 2^32: the largest use is
// 255 columns x ld + 255 rows, i.e. ld < 2^21.
__device__ inline double si_ld(const double* base, unsigned elem) {
  return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(base) + elem * 8u);
}
struct SiTile { double v[16]; };
__device__ inline void si_tile_load(SiTile& t, const double* __restrict__ M, i64 ldm, int tid) {
  // (the chunk index is the same for a whole wavefront: said so, the column bases live in scalar registers and a
  //  load is scalar base + lane offset instead of a 64-bit address pair per load)
  const int ch = __builtin_amdgcn_readfirstlane(tid >> 7);
  const double* col = M + static_cast<i64>(16 * ch) * ldm;
  const unsigned i = tid & (SI_H - 1), l32 = static_cast<unsigned>(ldm);
#pragma unroll
  for (int c = 0; c < 16; ++c) t.v[c] = si_ld(col, c * l32 + i);
}
// part[ch][i] = sum_c tile[c] v[16 ch + c]; ends with a barrier
__device__ inline void si_tile_vec(const SiTile& t, const double* v, double* part, int tid) {
  const int ch = __builtin_amdgcn_readfirstlane(tid >> 7);
  double s0 = 0.0, s1 = 0.0;
#pragma unroll
  for (int c = 0; c < 16; c += 2) {
    s0 = fma(t.v[c], v[16 * ch + c], s0);
    s1 = fma(t.v[c + 1], v[16 * ch + c + 1], s1);
  }
  part[ch * SI_H + (tid & (SI_H - 1))] = s0 + s1;
  __syncthreads();
}
__device__ inline double si_part_sum(const double* part, int i) {
  double s = 0.0;
#pragma unroll
  for (int ch = 0; ch < 8; ++ch) s += part[ch * SI_H + i];
  return s;
}

// One step of L y = b.  j0 = block whose y is final in b (j0 < 0: prologue, only block 0 is solved).
// Workgroup 0: rows of the next block j1 = j0 + 256: b1 -= L[j1.., j0..) y, then y1 = L11^-1 b1 through the two
// inverted halves.  Workgroups g >= 1: 64 rows from j1 + 256 + 64 (g - 1): b[r] -= L[r, j0..) y.
// Workgroup 0 of a step is bound by what ONE compute unit pulls cold from HBM (896 KB: 22 us).  The launch of the step
// BEFORE it carries one more workgroup, at a block index that is a multiple of 8 — the XCD of workgroup 0 — which does
// nothing but touch every 128-byte line of that data, so the next step finds it in its XCD's L2 (SI_PF such workgroups,
// 8 block indices apart, share the lines).
constexpr int SI_PF = 1;              // prefetching workgroups per step launch (block indices 8 apart: the same XCD); four of
                                      // them sharing the lines measured the same as one (C3 t_solve 4.21 vs 4.19 ms, 5.25 without)
__device__ inline double si_touch_cols(const double* base, i64 ld, int ncols, int nrows, int tid) {
  const int lpc = (nrows + 15) / 16, total = ncols * lpc;
  double acc = 0.0;
  for (int idx = tid; idx < total; idx += SI_T * SI_PF) {
    const int c = idx / lpc, l = idx - c * lpc;
    acc += base[static_cast<i64>(c) * ld + 16 * l];
  }
  return acc;
}
__global__ void __launch_bounds__(SI_T) ldlt_fwd_step_kernel(const double* __restrict__ A, i64 ld, int n, int j0,
                                                             double* __restrict__ b, const double* __restrict__ inv,
                                                             int grid_real, double* __restrict__ sink) {
  __shared__ double y[SI_B];
  __shared__ double part[SI_T];
  __shared__ double bn[SI_B];
  const int tid = threadIdx.x;
  if (static_cast<int>(blockIdx.x) >= grid_real) {
    // prefetch for the next step (j0 + SI_B): its coupling block, its two inverse tiles and the tile between them
    const int j0n = j0 + SI_B, j1n = j0n + SI_B;
    const int g8 = (grid_real + 7) / 8 * 8, off = static_cast<int>(blockIdx.x) - g8;
    if (off < 0 || (off & 7) != 0 || j1n >= n) return;
    const int pt = tid + SI_T * (off >> 3);            // this workgroup's share of the lines
    const int jbn = (n - j1n < SI_B) ? n - j1n : SI_B;
    double acc = si_touch_cols(A + static_cast<i64>(j0n) * ld + j1n, ld, SI_B, jbn, pt);
    const double* X0n = inv + static_cast<i64>(j1n / SI_H) * (SI_H * SI_H);
    acc += si_touch_cols(X0n, SI_H, SI_H, SI_H, pt);
    if (jbn > SI_H) {
      acc += si_touch_cols(X0n + SI_H * SI_H, SI_H, SI_H, SI_H, pt);
      acc += si_touch_cols(A + (j1n + SI_H) + static_cast<i64>(j1n) * ld, ld, SI_H, jbn - SI_H, pt);
    }
    if (acc == 1.2345678e-301) sink[0] = acc;          // (keeps the loads)
    return;
  }
  const int j1 = j0 + SI_B;
  if (blockIdx.x == 0) {
    const int jbn = (n - j1 < SI_B) ? n - j1 : SI_B;
    const double* X0 = inv + static_cast<i64>(j1 / SI_H) * (SI_H * SI_H);
    // One compute unit pulls this workgroup's 896 KB (coupling block 512, three operand tiles 384): more than its
    // registers and LDS hold, so the coupling block is consumed in two halves and the tiles follow it in issue order
    SiTile t0, tc, t1;
    if (j0 >= 0) {
      // 256 x 256 coupling block: lane = row, four column quarters
      const int r = tid & (SI_B - 1), q = __builtin_amdgcn_readfirstlane(tid >> 8);
      const double* row = A + static_cast<i64>(j0 + 64 * q) * ld + j1;      // wavefront-uniform base ...
      const unsigned ro = r < jbn ? r : 0, l32 = static_cast<unsigned>(ld);   // ... plus the lane's row
      double a[32];
#pragma unroll
      for (int c = 0; c < 32; ++c) a[c] = si_ld(row, c * l32 + ro);
      if (tid < SI_B) y[tid] = b[j0 + tid];
      __syncthreads();
      const double* yq = y + 64 * q;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int c = 0; c < 32; c += 4) {
        s0 = fma(a[c], yq[c], s0);
        s1 = fma(a[c + 1], yq[c + 1], s1);
        s2 = fma(a[c + 2], yq[c + 2], s2);
        s3 = fma(a[c + 3], yq[c + 3], s3);
      }
      // the second half's loads are issued here — not hoisted above, and the first half's products are done (the sums
      // are operands of the statement): 128 VGPRs per lane
      asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : : "memory");
#pragma unroll
      for (int c = 0; c < 32; ++c) a[c] = si_ld(row, (32 + c) * l32 + ro);
      si_tile_load(t0, X0, SI_H, tid);
#pragma unroll
      for (int c = 0; c < 32; c += 4) {
        s0 = fma(a[c], yq[32 + c], s0);
        s1 = fma(a[c + 1], yq[32 + c + 1], s1);
        s2 = fma(a[c + 2], yq[32 + c + 2], s2);
        s3 = fma(a[c + 3], yq[32 + c + 3], s3);
      }
      asm volatile("" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) : : "memory");
      if (jbn > SI_H) {
        si_tile_load(tc, A + (j1 + SI_H) + static_cast<i64>(j1) * ld, ld, tid);   // (rows past n: inside the padded allocation, unused)
        si_tile_load(t1, X0 + SI_H * SI_H, SI_H, tid);
      }
      part[q * SI_B + r] = (s0 + s1) + (s2 + s3);
      __syncthreads();
      if (tid < SI_B)
        bn[tid] = (tid < jbn) ? b[j1 + tid] - ((part[tid] + part[SI_B + tid]) + (part[2 * SI_B + tid] + part[3 * SI_B + tid])) : 0.0;
    } else {
      si_tile_load(t0, X0, SI_H, tid);
      if (jbn > SI_H) {
        si_tile_load(tc, A + (j1 + SI_H) + static_cast<i64>(j1) * ld, ld, tid);
        si_tile_load(t1, X0 + SI_H * SI_H, SI_H, tid);
      }
      if (tid < SI_B) bn[tid] = (tid < jbn) ? b[j1 + tid] : 0.0;
    }
    __syncthreads();
    si_tile_vec(t0, bn, part, tid);
    if (tid < SI_H) {
      const double v = si_part_sum(part, tid);
      bn[tid] = v;
      if (tid < jbn) b[j1 + tid] = v;
    }
    __syncthreads();
    if (jbn > SI_H) {
      si_tile_vec(tc, bn, part, tid);
      if (tid < SI_H) bn[SI_H + tid] -= (SI_H + tid < jbn) ? si_part_sum(part, tid) : 0.0;
      __syncthreads();
      si_tile_vec(t1, bn + SI_H, part, tid);
      if (tid < SI_H && SI_H + tid < jbn) b[j1 + SI_H + tid] = si_part_sum(part, tid);
    }
  } else {
    const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);           // sixteen column chunks of 16
    const i64 r = static_cast<i64>(j1) + SI_B + 64 * static_cast<i64>(blockIdx.x - 1) + lane;
    double a[16];
    const double* row = A + static_cast<i64>(j0 + 16 * w) * ld;
    const unsigned ro = static_cast<unsigned>(r < n ? r : static_cast<i64>(n) - 1), l32 = static_cast<unsigned>(ld);
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = si_ld(row, c * l32 + ro);
    if (tid < SI_B) y[tid] = b[j0 + tid];
    __syncthreads();
    const double* yw = y + 16 * w;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
      s0 = fma(a[c], yw[c], s0);
      s1 = fma(a[c + 1], yw[c + 1], s1);
    }
    part[w * 64 + lane] = s0 + s1;
    __syncthreads();
    if (w == 0 && r < n) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) s += part[k * 64 + lane];
      b[r] -= s;
    }
  }
}

// ---- the whole sweep in ONE launch: dataflow over the 128-blocks (round 4) ---------------------------------------
// The step kernels above are a chain of 43 dependent launches per sweep at order 11 000 (22-25 us each: launch boundary,
// one compute unit's cold pull of the step's operands).  Here every 128-row block of the right-hand side has its own
// workgroup for the whole sweep.  Workgroup j works on the j-th block IN SWEEP ORDER and waits only for blocks of lower
// workgroup index: workgroups are dispatched in index order, so whatever a resident workgroup waits for is resident or
// finished — no co-residency requirement, no cooperative launch (which this runtime serialises against graph launches
// on the same stream in a way that broke the NMF example's solves).  A finished block travels through `xch` (one double
// per entry, preset to an all-ones bit pattern that no computation produces): the readers' lanes poll THEIR entry —
// the data is its own flag, one trip through the memory fabric per step instead of flag + fence + data.  Workgroup j
// subtracts L[j, k] y_k for k < j — the 128 x 128 tile was fetched into registers before the wait, the next one is on
// its way during the product — multiplies with the inverted diagonal block and publishes.  The transposed sweep mirrors
// it from the last block (tiles read with the lanes on the rows, the contiguous direction; column sums by DPP).
constexpr unsigned long long kSweepEmpty = 0xFFFFFFFFFFFFFFFFull;
struct SweepCtl { unsigned abort; unsigned pad[31]; };
// a tile whose block has only `nrows` valid rows (the last block of an order that is not a multiple of 128): rows past
// them are read from the last valid row (what lies below is padding: not numbers anyone set); their products are masked
// by the caller (forward: not accumulated; transposed: multiplied with x = 0)
__device__ inline void si_tile_load_rows(SiTile& t, const double* __restrict__ M, i64 ldm, int tid, int nrows) {
  const int ch = __builtin_amdgcn_readfirstlane(tid >> 7);
  const double* col = M + static_cast<i64>(16 * ch) * ldm;
  const int ri = tid & (SI_H - 1);
  const unsigned i = static_cast<unsigned>(ri < nrows ? ri : nrows - 1), l32 = static_cast<unsigned>(ldm);
#pragma unroll
  for (int c = 0; c < 16; ++c) t.v[c] = si_ld(col, c * l32 + i);
}
// lanes tid < 128 fetch entry tid of a published block (poll until it is no longer the preset pattern); false = gave up
__device__ inline bool sweep_fetch(const double* __restrict__ src, double* dst, unsigned* abort_word, int* s_ok, int tid) {
  if (tid == 0) *s_ok = 1;
  __syncthreads();
  if (tid < SI_H) {
    const unsigned long long* p = reinterpret_cast<const unsigned long long*>(src) + tid;
    unsigned long long v;
    unsigned spins = 0;
    while ((v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == kSweepEmpty) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 1023u) == 0u && (spins > (1u << 22) || __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
        __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *s_ok = 0;
        break;
      }
    }
    dst[tid] = __longlong_as_double(static_cast<long long>(v));
  }
  __syncthreads();
  return *s_ok != 0;
}
__device__ inline void sweep_publish(double* __restrict__ dst, double v, int tid) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst) + tid, static_cast<unsigned long long>(__double_as_longlong(v)), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
// Sixteen wavefront-wide sums at once: u[c] = this lane's term of column c.  On return lane L holds in u[t], t < 4, the
// total of column 4 (L >> 4) + t.  Halving exchanges instead of sixteen full reductions (84 instructions against 320, which
// made the transposed sweep VALU-bound: 329 us against the forward sweep's 138): v_permlane32_swap trades the upper half
// of u[t] for the lower half of u[t + 8] (one add leaves column t in lanes 0-31, column t + 8 in lanes 32-63), then
// v_permlane16_swap the same between the 16-lane rows; the four values left are summed over their row by DPP rotations.
__device__ inline void wave_colsum16(double (&u)[16]) {
  auto swap_add = [](double a, double b, auto swap) {
    const auto lo = swap(static_cast<unsigned>(__double2loint(a)), static_cast<unsigned>(__double2loint(b)));
    const auto hi = swap(static_cast<unsigned>(__double2hiint(a)), static_cast<unsigned>(__double2hiint(b)));
    return __hiloint2double(static_cast<int>(hi[0]), static_cast<int>(lo[0])) + __hiloint2double(static_cast<int>(hi[1]), static_cast<int>(lo[1]));
  };
  auto swap32 = [](unsigned a, unsigned b) { return __builtin_amdgcn_permlane32_swap(a, b, false, false); };
  auto swap16 = [](unsigned a, unsigned b) { return __builtin_amdgcn_permlane16_swap(a, b, false, false); };
#pragma unroll
  for (int t = 0; t < 8; ++t) u[t] = swap_add(u[t], u[t + 8], swap32);
#pragma unroll
  for (int t = 0; t < 4; ++t) u[t] = swap_add(u[t], u[t + 4], swap16);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    u[t] += dpp_shift_f64<0x128, 0xf>(u[t], 0.0);       // row_ror:8, 4, 2, 1: every lane of the row ends with the row's total
    u[t] += dpp_shift_f64<0x124, 0xf>(u[t], 0.0);
    u[t] += dpp_shift_f64<0x122, 0xf>(u[t], 0.0);
    u[t] += dpp_shift_f64<0x121, 0xf>(u[t], 0.0);
  }
}
// Each block row is shared by P workgroups (one compute unit draws ~35 GB/s: a row's 128 KB tiles at one per 3.9 us were
// the sweep's pace, against 1.7 us for the exchange itself — tools/micro/chain_hop.hip): member q of the row at sweep
// position j takes the tiles at sweep positions s = (j + q) mod P, + P, ...; the LAST member (q = P - 1) has the tile of the
// block published last, adds the other members' partial sums (exchanged through `pex`, same data-as-flag form) in member
// order and finishes the block.  Every wait is for a lower workgroup index.
__global__ void __launch_bounds__(SI_T) ldlt_fwd_sweep_kernel(const double* __restrict__ A, i64 ld, int n, double* __restrict__ b,
                                                              const double* __restrict__ inv, double* __restrict__ xch,
                                                              double* __restrict__ pex, SweepCtl* __restrict__ ctl, int P) {
  __shared__ double y[SI_H];
  __shared__ double acc[SI_H];
  __shared__ double part[SI_T];
  __shared__ int s_ok;
  const int tid = threadIdx.x, i = static_cast<int>(blockIdx.x) / P, q = static_cast<int>(blockIdx.x) % P, r0 = i * SI_H;
  const bool owner = q == P - 1;
  const int nrows = (n - r0 < SI_H) ? n - r0 : SI_H;                           // valid rows of this block
  if (tid < SI_H) acc[tid] = (owner && tid < nrows) ? b[r0 + tid] : 0.0;
  // two tiles in registers, not three (a third spilled 24 VGPRs to scratch): the owner's inverted diagonal block rides in
  // the look-ahead slot of its last tile
  SiTile cur, nxt;
  const double* invi = inv + static_cast<i64>(i) * (SI_H * SI_H);
  __syncthreads();                                                             // (acc: block 0 goes straight to the product below)
  int k = (i + q) % P;
  if (k < i) si_tile_load_rows(cur, A + r0 + static_cast<i64>(k) * SI_H * ld, ld, tid, nrows);
  else if (owner) si_tile_load(cur, invi, SI_H, tid);
  for (; k < i; k += P) {
    if (k + P < i) si_tile_load_rows(nxt, A + r0 + static_cast<i64>(k + P) * SI_H * ld, ld, tid, nrows);
    else if (owner) si_tile_load(nxt, invi, SI_H, tid);
    if (!sweep_fetch(xch + k * SI_H, y, &ctl->abort, &s_ok, tid)) return;
    si_tile_vec(cur, y, part, tid);
    if (tid < nrows) acc[tid] -= si_part_sum(part, tid);
    __syncthreads();
    cur = nxt;
  }
  if (!owner) {
    if (tid < SI_H) sweep_publish(pex + (static_cast<i64>(i) * P + q) * SI_H, acc[tid], tid);
    return;
  }
  for (int o = 0; o + 1 < P; ++o) {
    if (!sweep_fetch(pex + (static_cast<i64>(i) * P + o) * SI_H, y, &ctl->abort, &s_ok, tid)) return;
    if (tid < SI_H) acc[tid] += y[tid];
    __syncthreads();
  }
  si_tile_vec(cur, acc, part, tid);
  if (tid < SI_H) {
    const double v = (tid < nrows) ? si_part_sum(part, tid) : 0.0;
    sweep_publish(xch + r0, v, tid);           // (entries past n of the last block: zeros — nobody reads that block)
    if (tid < nrows) b[r0 + tid] = v;
  }
}
// L^T x = y.  Sweep position j is block i = nblk - 1 - j; its members wait for the blocks k > i (lower sweep positions), last
// first.  Tile (k, i): rows of block k (the lanes: contiguous), columns of block i — column sums over the 128 rows by
// two wavefront DPP sums each.
__global__ void __launch_bounds__(SI_T) ldlt_bwd_sweep_kernel(const double* __restrict__ A, i64 ld, int n, double* __restrict__ b,
                                                              const double* __restrict__ invT, double* __restrict__ xch,
                                                              double* __restrict__ pex, SweepCtl* __restrict__ ctl, int nblk, int P) {
  __shared__ double x[SI_H];
  __shared__ double acc[SI_H];
  __shared__ double part[SI_T];
  __shared__ double colp[2][SI_H];
  __shared__ int s_ok;
  const int tid = threadIdx.x, j = static_cast<int>(blockIdx.x) / P, q = static_cast<int>(blockIdx.x) % P, i = nblk - 1 - j, c0 = i * SI_H;
  const bool owner = q == P - 1;
  const int lane = tid & 63, half = (tid >> 6) & 1, ch = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int ncols = (n - c0 < SI_H) ? n - c0 : SI_H;
  if (tid < SI_H) acc[tid] = (owner && tid < ncols) ? b[c0 + tid] : 0.0;
  SiTile cur, nxt;
  const double* invi = invT + static_cast<i64>(i) * (SI_H * SI_H);
  __syncthreads();
  const int last_rows = n - (nblk - 1) * SI_H;                                  // valid rows of the last block
  auto load = [&](SiTile& t, int s) {                                           // tile at sweep position s: block row k = nblk - 1 - s
    const int k = nblk - 1 - s;
    si_tile_load_rows(t, A + static_cast<i64>(k) * SI_H + static_cast<i64>(c0) * ld, ld, tid, s == 0 ? last_rows : SI_H);
  };
  int s = (j + q) % P;
  if (s < j) load(cur, s);
  else if (owner) si_tile_load(cur, invi, SI_H, tid);
  for (; s < j; s += P) {
    if (s + P < j) load(nxt, s + P);
    else if (owner) si_tile_load(nxt, invi, SI_H, tid);
    if (!sweep_fetch(xch + (nblk - 1 - s) * SI_H, x, &ctl->abort, &s_ok, tid)) return;       // (entries past n of the last block were published as zeros)
    {
      const double xr = x[tid & (SI_H - 1)];
      double u[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) u[c] = cur.v[c] * xr;
      wave_colsum16(u);
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if ((lane & 15) == t) colp[half][16 * ch + 4 * (lane >> 4) + t] = u[t];
    }
    __syncthreads();
    if (tid < SI_H) acc[tid] -= colp[0][tid] + colp[1][tid];
    __syncthreads();
    cur = nxt;
  }
  if (!owner) {
    if (tid < SI_H) sweep_publish(pex + (static_cast<i64>(j) * P + q) * SI_H, acc[tid], tid);
    return;
  }
  for (int o = 0; o + 1 < P; ++o) {
    if (!sweep_fetch(pex + (static_cast<i64>(j) * P + o) * SI_H, x, &ctl->abort, &s_ok, tid)) return;
    if (tid < SI_H) acc[tid] += x[tid];
    __syncthreads();
  }
  si_tile_vec(cur, acc, part, tid);
  if (tid < SI_H) {
    const double v = (tid < ncols) ? si_part_sum(part, tid) : 0.0;
    sweep_publish(xch + c0, v, tid);
    if (tid < ncols) b[c0 + tid] = v;
  }
}

// One step of L^T x = b, blocks from the last to the first.  j0 = block whose x is final (j0 >= n: prologue, only the
// last block is solved).  Workgroup 0: columns of the previous block jp = j0 - 256: bp -= L[j0.., jp..)^T x, then
// xp = Lpp^-T bp through the two transposed inverses.  Workgroups g >= 1: 64 columns from 64 (g - 1) below jp.
// Transposed products keep the lanes on the rows (the contiguous direction): a wavefront takes a few columns, four
// 512-byte loads per column, and sums across its lanes (DPP) — a lane per column reads a cache line per lane and
// instruction, sixteen times the transactions (53 us per step measured that way, 22 for the forward step).
__global__ void __launch_bounds__(SI_T) ldlt_bwd_step_kernel(const double* __restrict__ A, i64 ld, int n, int j0,
                                                             double* __restrict__ b, const double* __restrict__ invT,
                                                             int grid_real, double* __restrict__ sink) {
  __shared__ double part[SI_T];
  __shared__ double bn[SI_B];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (static_cast<int>(blockIdx.x) >= grid_real) {
    // prefetch for the next step (block j0 - SI_B final, previous block jpn = j0 - 2 SI_B): see ldlt_fwd_step_kernel
    const int j0n = (j0 >= n) ? ((n - 1) / SI_B) * SI_B : j0 - SI_B, jpn = j0n - SI_B;
    const int g8 = (grid_real + 7) / 8 * 8, off = static_cast<int>(blockIdx.x) - g8;
    if (off < 0 || (off & 7) != 0 || jpn < 0) return;
    const int pt = tid + SI_T * (off >> 3);
    const int jbq = (n - j0n < SI_B) ? n - j0n : SI_B;
    const double* XT0n = invT + static_cast<i64>(jpn / SI_H) * (SI_H * SI_H);
    double acc = si_touch_cols(XT0n, SI_H, 2 * SI_H, SI_H, pt);                                      // both transposed inverses
    acc += si_touch_cols(A + (jpn + SI_H) + static_cast<i64>(jpn) * ld, ld, SI_H, SI_H, pt);         // coupling inside the block
    acc += si_touch_cols(A + j0n + static_cast<i64>(jpn) * ld, ld, SI_B, jbq, pt);                   // L[j0n.., jpn..)
    if (acc == 1.2345678e-301) sink[0] = acc;
    return;
  }
  const bool prologue = j0 >= n;
  const int jb = prologue ? 0 : ((n - j0 < SI_B) ? n - j0 : SI_B);
  if (blockIdx.x == 0) {
    const int jp = prologue ? ((n - 1) / SI_B) * SI_B : j0 - SI_B;
    const int jbp = (n - jp < SI_B) ? n - jp : SI_B;
    const double* XT0 = invT + static_cast<i64>(jp / SI_H) * (SI_H * SI_H);
    SiTile t0, t1;
    double cpl[16];                                   // coupling block L[jp + 128 .., jp ..): columns 8 w .. 8 w + 8, rows lane, lane + 64
    if (jbp > SI_H) {
      si_tile_load(t1, XT0 + SI_H * SI_H, SI_H, tid);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double* col = A + (jp + SI_H) + static_cast<i64>(jp + 8 * w + k) * ld;
        cpl[2 * k] = (SI_H + lane < jbp) ? col[lane] : 0.0;
        cpl[2 * k + 1] = (SI_H + 64 + lane < jbp) ? col[64 + lane] : 0.0;
      }
    }
    if (!prologue) {
      // wavefront w: columns 16 w .. 16 w + 16 of the previous block in four groups of four, rows lane + 64 q of block j0
      double xr[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) xr[q] = (64 * q + lane < jb) ? b[j0 + 64 * q + lane] : 0.0;
#pragma unroll 1
      for (int g = 0; g < 4; ++g) {
        double a[16];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double* col = A + j0 + static_cast<i64>(jp + 16 * w + 4 * g + k) * ld;
#pragma unroll
          for (int q = 0; q < 4; ++q) a[4 * k + q] = (64 * q + lane < jb) ? col[64 * q + lane] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const double sacc = wave_sum(fma(a[4 * k], xr[0], a[4 * k + 1] * xr[1]) + fma(a[4 * k + 2], xr[2], a[4 * k + 3] * xr[3]));
          if (lane == 0) part[16 * w + 4 * g + k] = sacc;
        }
      }
      si_tile_load(t0, XT0, SI_H, tid);
      __syncthreads();
      if (tid < SI_B) bn[tid] = b[jp + tid] - part[tid];
    } else {
      si_tile_load(t0, XT0, SI_H, tid);
      if (tid < SI_B) bn[tid] = (tid < jbp) ? b[jp + tid] : 0.0;
    }
    __syncthreads();
    if (jbp > SI_H) {
      si_tile_vec(t1, bn + SI_H, part, tid);
      if (tid < SI_H) {
        const double v = si_part_sum(part, tid);
        bn[SI_H + tid] = (SI_H + tid < jbp) ? v : 0.0;
        if (SI_H + tid < jbp) b[jp + SI_H + tid] = v;
      }
      __syncthreads();
      const double x0 = bn[SI_H + lane], x1 = bn[SI_H + 64 + lane];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const double sacc = wave_sum(fma(cpl[2 * k], x0, cpl[2 * k + 1] * x1));
        if (lane == 0) part[8 * w + k] = sacc;
      }
      __syncthreads();
      if (tid < SI_H) bn[tid] -= part[tid];
      __syncthreads();
    }
    si_tile_vec(t0, bn, part, tid);
    if (tid < SI_H && tid < jbp) b[jp + tid] = si_part_sum(part, tid);
  } else {
    // wavefront w: columns 4 w .. 4 w + 4 of this workgroup's 64
    const int c0 = 64 * (blockIdx.x - 1) + 4 * w;      // < j0 - 256 by the launch geometry
    double a[16], xr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double* col = A + j0 + static_cast<i64>(c0 + k) * ld;
#pragma unroll
      for (int q = 0; q < 4; ++q) a[4 * k + q] = (64 * q + lane < jb) ? col[64 * q + lane] : 0.0;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) xr[q] = (64 * q + lane < jb) ? b[j0 + 64 * q + lane] : 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const double sacc = wave_sum(fma(a[4 * k], xr[0], a[4 * k + 1] * xr[1]) + fma(a[4 * k + 2], xr[2], a[4 * k + 3] * xr[3]));
      if (lane == 0) b[c0 + k] -= sacc;
    }
  }
}

struct BlockedLdlt {
  HipExec* ex = nullptr;
  i64 n = 0, ld = 0, ldw = 0;
  double* Wp2[2] = {nullptr, nullptr};     // panel workspaces W = L D (double buffered for look-ahead)
  LdltInfo* info = nullptr;
  double* acc = nullptr;
  double* Ltop = nullptr;                  // packed operands of the current 128-column sub-panel (ldlt_top128_kernel)
  bool inv_mfma = true;                    // inverses of the diagonal 128-blocks on MFMA blocks (DNLP_LDLT_INV_MFMA=0: the 128-lane kernel)
  bool top_mfma = true;                    // the 128 x 128 top block on MFMA blocks (DNLP_LDLT_TOP_MFMA=0: the 4 x 4 tile kernel)
  bool sub128 = true;                      // panels in 128-column sub-panels (DNLP_LDLT_T128=0: the 32-column chain)
  // A full 512-column panel as diagonal block first, then ldlt_rows512_kernel (DNLP_LDLT_FUSED_ROWS=1).  OFF: measured on the
  // MI355X (round 5, tools/ldlt_fused_ab.sh) it reproduces the chain's factor bit for bit and is SLOWER at every order —
  // 11 000: 13.1 ms chain, 14.2 fused rows + ten-launch diagonal block, 15.3 with the four-workgroup diagonal block
  // (335 us per launch against 315 for the ten launches), 17.3 with a one-workgroup diagonal block (460 us: 45 MFLOP on one
  // compute unit's FP64 matrix pipes, 64 cycles per 16 x 16 x 4 on this part); ldlt_rows512_kernel 132-146 us per panel
  // (56 us of MFMA issue per wavefront).  The panel's critical path is the diagonal block, not the rows below it.
  bool fused_rows = false;
  double* Lpk = nullptr;                   // ... the packed off-diagonal blocks of that diagonal block (ldlt_pack512_kernel)
  bool diag512 = true;                     // ... the diagonal block in one launch (DNLP_LDLT_DIAG512=0: the sub-panel chain on its 512 rows)
  PanelCtl* pctl = nullptr;                // ... its flags, compared against a launch counter (never reset)
  unsigned panel_epoch = 0;
  double last_update_seconds = 0.0;
  double total_update_seconds = 0.0, total_update_flops = 0.0;   // outer (Schur) updates, timed
  i64 total_update_launches = 0;
  i64 outer_updates_full = 0;              // timed Schur updates of one complete factorisation of this order
  i64 total_bailed = 0;                    // factorisations abandoned early (wrong inertia already certain)
  std::vector<hipEvent_t> tev0, tev1;      // timing events of the Schur updates of one factorisation
  hipEvent_t evPanel = nullptr, evUpd = nullptr;
  hipStream_t s1 = nullptr;                // stream of the big trailing updates
  // in-panel overlap: the next sub-panel's diagonal block is brought up to date first (on the chain's stream) and its
  // one-workgroup factorisation runs while the rest of the in-panel update is still going on s2
  hipStream_t s2 = nullptr;
  hipEvent_t evR = nullptr, evB = nullptr, evRest = nullptr;
  bool sub_overlap = false;                // DNLP_LDLT_SUB_OVERLAP=1 enables it.  Measured (round 3, orders 1500 .. 22 000): 0-6 % SLOWER —
                                           // the one-workgroup top block then waits for a compute unit the s2 update holds
  bool time_updates = false;
  bool lookahead = true;
  bool xcd_swizzle = false;    // 8 x 8 super-tiles per XCD: cuts the W-strip re-reads ~5x but measured 1.5-3% slower (MFMA-bound), so off; DNLP_LDLT_XCD=1 enables
  bool padded = false;         // the matrix allocation has >= 128 doubles of slack behind it
  int NB = 512;                // outer panel width (K of the MFMA Schur update)
  int small_tiles_below = 384; // launches with fewer 128 x 128 tiles than this use the 64 x 64 kernel (DNLP_LDLT_SMALL_TILES)
  int small_rows_max = 5120;   // ... and only up to this many rows: above, the launch runs beside a big update of the
                               // look-ahead and the tile with the better MFMA rate wins (DNLP_LDLT_SMALL_ROWS)
  int reserve_cus = 0;         // compute units the update stream may not use (look-ahead panel kernels run there)
  int max_neg = -1;            // >= 0: give up as soon as more negative pivots than this appear
  double *Linv = nullptr, *LinvT = nullptr;   // inverted 128 x 128 diagonal blocks of the current factor (and transposes)
  bool inv_ready = false;      // ... computed by the first solve after a factorisation
  const double* inv_of = nullptr;   // ... of THIS matrix (a solve on another buffer recomputes them)
  bool solve_inv = true;       // DNLP_LDLT_SOLVE_INV=0: the substitution-based block solves

  BlockedLdlt() = default;
  BlockedLdlt(const BlockedLdlt&) = delete;
  BlockedLdlt& operator=(const BlockedLdlt&) = delete;
  ~BlockedLdlt() {
    for (hipEvent_t e : tev0) hipEventDestroy(e);
    for (hipEvent_t e : tev1) hipEventDestroy(e);
    if (evPanel) hipEventDestroy(evPanel);
    if (evUpd) hipEventDestroy(evUpd);
    for (hipEvent_t e : {evR, evB, evRest}) if (e) hipEventDestroy(e);
    if (s1) hipStreamDestroy(s1);
    if (s2) hipStreamDestroy(s2);
  }

  void init(HipExec* e, i64 n_, i64 ld_) {
    ex = e; n = n_; ld = ld_;
    ldw = (n + 7) / 8 * 8;
    // wide outer panels for large orders: the update kernel runs at the same rate for K = 512 / 768 / 1024
    // (62.3 +- 0.2 TF at n = 1e5) and half the panels means half the exposed panel heads and tails
    // (n = 1e5: 5.46 -> 5.37 s per factorisation); mid-size orders measured flat between 256 and 768
    if (n >= 20000) NB = 1024;          // (order 15 000: 512 is 3 % faster; 22 000 and 30 000: 1024 is 2-3 % faster)
    if (const char* ev = std::getenv("DNLP_LDLT_NB")) NB = std::atoi(ev);
    if (const char* ev = std::getenv("DNLP_LDLT_LOOKAHEAD")) lookahead = std::atoi(ev) != 0;
    if (const char* ev = std::getenv("DNLP_LDLT_XCD")) xcd_swizzle = std::atoi(ev) != 0;
    if (const char* ev = std::getenv("DNLP_LDLT_T128")) sub128 = std::atoi(ev) != 0;
    if (const char* ev = std::getenv("DNLP_LDLT_FUSED_ROWS")) fused_rows = std::atoi(ev) != 0;
    if (const char* ev = std::getenv("DNLP_LDLT_DIAG512")) diag512 = std::atoi(ev) != 0;
    if (const char* ev = std::getenv("DNLP_LDLT_TOP_MFMA")) top_mfma = std::atoi(ev) != 0;
    if (const char* ev = std::getenv("DNLP_LDLT_INV_MFMA")) inv_mfma = std::atoi(ev) != 0;
    if (const char* ev = std::getenv("DNLP_LDLT_SMALL_TILES")) small_tiles_below = std::atoi(ev);
    if (const char* ev = std::getenv("DNLP_LDLT_SMALL_ROWS")) small_rows_max = std::atoi(ev);
    if (const char* ev = std::getenv("DNLP_LDLT_SOLVE_INV")) solve_inv = std::atoi(ev) != 0;
    if (const char* ev = std::getenv("DNLP_LDLT_SUB_OVERLAP")) sub_overlap = std::atoi(ev) != 0;
    if (sub_overlap && sub128) {
      DNLP_HIP_CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
      DNLP_HIP_CHECK(hipEventCreateWithFlags(&evR, hipEventDisableTiming));
      DNLP_HIP_CHECK(hipEventCreateWithFlags(&evB, hipEventDisableTiming));
      DNLP_HIP_CHECK(hipEventCreateWithFlags(&evRest, hipEventDisableTiming));
    }
    if (NB < LD_nb) NB = LD_nb;
    if (NB > LD_NB_MAX) NB = LD_NB_MAX;
    NB = NB / LD_nb * LD_nb;
    Wp2[0] = ex->alloc<double>(static_cast<size_t>(ldw) * NB + 256);
    Wp2[1] = lookahead ? ex->alloc<double>(static_cast<size_t>(ldw) * NB + 256) : Wp2[0];
    info = ex->alloc<LdltInfo>(1);
    acc = ex->alloc<double>(SV_B);
    Ltop = ex->alloc<double>(4 * LD_TOP_WS);
    Lpk = ex->alloc<double>(LD_PK_WS);
    pctl = ex->alloc<PanelCtl>(1);
    DNLP_HIP_CHECK(hipMemset(pctl, 0, sizeof(PanelCtl)));
    DNLP_HIP_CHECK(hipEventCreateWithFlags(&evPanel, hipEventDisableTiming));
    DNLP_HIP_CHECK(hipEventCreateWithFlags(&evUpd, hipEventDisableTiming));
    // the trailing updates run on their own (lower-priority) stream so that the next panel's
    // latency-bound factorisation kernels overlap with the previous panel's MFMA update
    // Mid-size orders keep `reserve_cus` compute units (one per XCD for 8: mask bit i is XCD i % 8) out of the
    // update stream's CU mask: a panel kernel of the look-ahead chain otherwise waits for a CU on which BOTH
    // resident update workgroups have retired (~a tile time per launch, longer than the kernel itself).
    if (n < 16384) reserve_cus = 8;      // (order 22 000 loses 3.5 % with the mask, 11 000 gains 1 %, below 6000 neutral)
    if (const char* ev = std::getenv("DNLP_LDLT_RESERVE_CUS")) reserve_cus = std::atoi(ev);
    if (reserve_cus > 0 && lookahead) {
      int dev = 0, ncu = 0;
      DNLP_HIP_CHECK(hipGetDevice(&dev));
      DNLP_HIP_CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
      if (ncu > 2 * reserve_cus) {
        std::vector<uint32_t> mask(static_cast<size_t>((ncu + 31) / 32), 0u);
        for (int i = 0; i < ncu - reserve_cus; ++i) mask[static_cast<size_t>(i >> 5)] |= 1u << (i & 31);
        if (hipExtStreamCreateWithCUMask(&s1, static_cast<uint32_t>(mask.size()), mask.data()) != hipSuccess) {
          (void)hipGetLastError();
          s1 = nullptr;
        }
      }
    }
    if (!s1) {
      reserve_cus = 0;
      int lo_p = 0, hi_p = 0;
      DNLP_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo_p, &hi_p));
      DNLP_HIP_CHECK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, lo_p));
    }
  }

  void gemm(hipStream_t st, double* C, const double* W, const double* L, i64 ldl, int M, int Nc, int Kd, int lower) {
    if (M <= 0 || Nc <= 0 || Kd <= 0) return;
    const int ntm = (M + GM_BM - 1) / GM_BM, ntn = (Nc + GM_BN - 1) / GM_BN;
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const int vec_ok = al(W) && al(L) && (ldw % 2 == 0) && (ldl % 2 == 0);
    // operands may be read up to 127 rows past M / Nc: both allocations carry that padding
    const bool fast_ok = vec_ok && (Kd % GM_BK == 0) && padded;
    // launches that would not fill the chip with 128 x 128 tiles take four times as many 64 x 64 ones
    const i64 work_tiles = (lower && M == Nc) ? static_cast<i64>(ntm) * (ntm + 1) / 2 : static_cast<i64>(ntm) * ntn;
    if (work_tiles < small_tiles_below && M <= small_rows_max) {
      const int sm = (M + GS_B - 1) / GS_B, sn = (Nc + GS_B - 1) / GS_B;
      hipLaunchKernelGGL(gemm_nt_update_small, dim3(static_cast<unsigned>(sm) * sn), dim3(256), 0, st, C, ld, W, ldw,
                         L, ldl, M, Nc, Kd, lower, sm, vec_ok);
      DNLP_LAUNCH_CHECK();
      return;
    }
    if (fast_ok && xcd_swizzle && ntm >= 16 && ntn >= 16) {
      const unsigned nsuper = static_cast<unsigned>((ntm + 7) / 8) * static_cast<unsigned>((ntn + 7) / 8);
      hipLaunchKernelGGL(gemm_nt_update_fast, dim3(((nsuper + 7) / 8) * 8 * 64), dim3(512), 0, st, C, ld, W,
                         ldw, L, ldl, M, Nc, Kd, lower, ntm, vec_ok | 2);
    } else if (fast_ok)
      hipLaunchKernelGGL(gemm_nt_update_fast, dim3(static_cast<unsigned>(ntm) * ntn), dim3(512), 0, st, C, ld, W,
                         ldw, L, ldl, M, Nc, Kd, lower, ntm, vec_ok);
    else
      hipLaunchKernelGGL(gemm_nt_update, dim3(static_cast<unsigned>(ntm) * ntn), dim3(256), 0, st, C, ld, W, ldw,
                         L, ldl, M, Nc, Kd, lower, ntm, vec_ok);
    DNLP_LAUNCH_CHECK();
  }

  // Right-looking blocked LDL^T with one panel of look-ahead:
  //   stream s0: factor panel p | wait(update p-1) | update the NEXT panel's columns with panel p
  //   stream s1: wait(panel p)  | update everything right of the next panel with panel p
  // so the small panel kernels of p+1 run underneath the big MFMA update of p.
  bool factor(double* A, int* nneg, int* nzero) {
    hipStream_t s0 = ex->stream;
    inv_ready = false;
    DNLP_HIP_CHECK(hipMemsetAsync(info, 0, sizeof(LdltInfo), s0));      // (stream-ordered: no host round trip before the first panel)
    const double tiny = 1e-300;
    const int ni = static_cast<int>(n);
    const int npanels = (ni + NB - 1) / NB;
    if (time_updates && static_cast<int>(tev0.size()) < npanels) {
      const size_t old = tev0.size();
      tev0.resize(static_cast<size_t>(npanels));
      tev1.resize(static_cast<size_t>(npanels));
      for (size_t k = old; k < tev0.size(); ++k) {
        DNLP_HIP_CHECK(hipEventCreate(&tev0[k]));
        DNLP_HIP_CHECK(hipEventCreate(&tev1[k]));
      }
    }
    std::vector<double> upd_flops;
    bool upd_pending = false, bailed = false;
    LdltInfo cur;
    std::memset(&cur, 0, sizeof cur);
    int p = 0;
    for (int K0 = 0; K0 < ni; K0 += NB, ++p) {
      const int KB = std::min(NB, ni - K0);
      double* Wp = Wp2[p & 1];
      bool b_pending = false, rest_pending = false;
      if (fused_rows && sub128 && top_mfma && !s2 && KB == LD_P && ni - (K0 + KB) > 0) {
        // the diagonal block on its own 512 rows, then every 16 rows below in one launch (ldlt_rows512_kernel)
        const int nb = K0 + KB;
        if (diag512)
          hipLaunchKernelGGL(ldlt_diag512_kernel, dim3(LD_P / LD_T), dim3(LD_TOPM_THREADS), 0, s0, A, ld, K0, info, tiny, Ltop, Lpk, Wp, ldw, pctl,
                             ++panel_epoch);
        else {
          for (int q = 0; q < LD_P / LD_T; ++q) {
            const int j0 = K0 + LD_T * q, r0 = j0 + LD_T, rows = nb - r0;
            double* top = Ltop + static_cast<i64>(q) * LD_TOP_WS;
            hipLaunchKernelGGL(ldlt_top128_mfma_kernel, dim3(1), dim3(LD_TOPM_THREADS), 0, s0, A, ld, j0, info, tiny, top);
            if (rows > 0) {
              hipLaunchKernelGGL(ldlt_rows128_kernel, dim3((rows + 63) / 64), dim3(256), 0, s0, A, ld, j0, nb, Wp, ldw, j0 - K0, top);
              gemm(s0, A + r0 + static_cast<i64>(r0) * ld, Wp + r0 + static_cast<i64>(j0 - K0) * ldw, A + r0 + static_cast<i64>(j0) * ld, ld,
                   rows, rows, LD_T, 1);
            }
          }
          hipLaunchKernelGGL(ldlt_pack512_kernel, dim3(6 * 64), dim3(256), 0, s0, A, ld, K0, Lpk);
        }
        hipLaunchKernelGGL(ldlt_rows512_kernel, dim3((ni - nb + 63) / 64), dim3(256), 0, s0, A, ld, K0, ni, Wp, ldw, Ltop, Lpk);
        DNLP_LAUNCH_CHECK();
      } else
      for (int j0 = K0; j0 < K0 + KB; j0 += LD_nb) {
        if (sub128 && K0 + KB - j0 >= LD_T) {
          // a full 128-column sub-panel: three launches instead of twelve
          if (top_mfma)
            hipLaunchKernelGGL(ldlt_top128_mfma_kernel, dim3(1), dim3(LD_TOPM_THREADS), 0, s0, A, ld, j0, info, tiny, Ltop);
          else
            hipLaunchKernelGGL(ldlt_top128_kernel, dim3(1), dim3(LD_TOP_THREADS), 0, s0, A, ld, j0, info, tiny, Ltop);
          const int r0 = j0 + LD_T, rows = ni - r0;
          if (rows > 0) {
            if (b_pending) { DNLP_HIP_CHECK(hipStreamWaitEvent(s0, evB, 0)); b_pending = false; }    // this sub-panel's rows, brought up to date on s2
            hipLaunchKernelGGL(ldlt_rows128_kernel, dim3((rows + 63) / 64), dim3(256), 0, s0, A, ld, j0, ni, Wp, ldw,
                               j0 - K0, Ltop);
            const int nc = K0 + KB - r0;
            if (nc > 0 && s2 && rows > LD_T) {
              // In-panel update in three pieces: D = the next sub-panel's diagonal block (this stream: the next top
              // block follows at once), B = the rows below it, rest = the panel's later columns (both on s2, under the
              // next top block: one workgroup for 49 us while this update has the rest of the chip).
              const int dn = nc < LD_T ? nc : LD_T, rs = r0 + dn;
              const double* Wj = Wp + static_cast<i64>(j0 - K0) * ldw;
              DNLP_HIP_CHECK(hipEventRecord(evR, s0));
              if (rest_pending) { DNLP_HIP_CHECK(hipStreamWaitEvent(s0, evRest, 0)); rest_pending = false; }   // earlier updates of D
              gemm(s0, A + r0 + static_cast<i64>(r0) * ld, Wj + r0, A + r0 + static_cast<i64>(j0) * ld, ld, dn, dn, LD_T, 1);
              DNLP_HIP_CHECK(hipStreamWaitEvent(s2, evR, 0));
              gemm(s2, A + rs + static_cast<i64>(r0) * ld, Wj + rs, A + r0 + static_cast<i64>(j0) * ld, ld, ni - rs, dn, LD_T, 0);
              DNLP_HIP_CHECK(hipEventRecord(evB, s2));
              b_pending = true;
              if (nc > dn) {
                gemm(s2, A + rs + static_cast<i64>(rs) * ld, Wj + rs, A + rs + static_cast<i64>(j0) * ld, ld, ni - rs, nc - dn, LD_T, 1);
                DNLP_HIP_CHECK(hipEventRecord(evRest, s2));
                rest_pending = true;
              }
            } else if (nc > 0) {
              if (rest_pending) { DNLP_HIP_CHECK(hipStreamWaitEvent(s0, evRest, 0)); rest_pending = false; }
              gemm(s0, A + r0 + static_cast<i64>(r0) * ld, Wp + r0 + static_cast<i64>(j0 - K0) * ldw,
                   A + r0 + static_cast<i64>(j0) * ld, ld, rows, nc, LD_T, 1);
            }
          }
          j0 += LD_T - LD_nb;
          continue;
        }
        const int jb = std::min(LD_nb, K0 + KB - j0);
        if (b_pending) { DNLP_HIP_CHECK(hipStreamWaitEvent(s0, evB, 0)); b_pending = false; }
        if (rest_pending) { DNLP_HIP_CHECK(hipStreamWaitEvent(s0, evRest, 0)); rest_pending = false; }
        hipLaunchKernelGGL(ldlt_diag_kernel, dim3(1), dim3(64), 0, s0, A, ld, j0, jb, info, tiny);
        const int r0 = j0 + jb;
        if (r0 >= ni) continue;
        const int rows = ni - r0;
        hipLaunchKernelGGL(ldlt_trsm_kernel, dim3((rows + 255) / 256), dim3(256), 0, s0, A, ld, j0, jb, ni, Wp,
                           ldw, j0 - K0);
        const int nc = K0 + KB - r0;
        if (nc > 0)
          gemm(s0, A + r0 + static_cast<i64>(r0) * ld, Wp + r0 + static_cast<i64>(j0 - K0) * ldw,
               A + r0 + static_cast<i64>(j0) * ld, ld, rows, nc, jb, 1);
      }
      if (b_pending) DNLP_HIP_CHECK(hipStreamWaitEvent(s0, evB, 0));
      if (rest_pending) DNLP_HIP_CHECK(hipStreamWaitEvent(s0, evRest, 0));
      const int r1 = K0 + KB;
      if (max_neg >= 0 && n >= 16384) {
        // wrong inertia is known as soon as too many negative pivots have appeared: the rest
        // of the factorisation would be thrown away by the caller's regularisation loop
        // (a host round trip per panel: worth it where a factorisation takes tens of milliseconds and more — at order
        //  11 000 the 22 round trips cost 0.2-0.3 ms of a 13 ms factorisation and an abandoned attempt saves little)
        DNLP_HIP_CHECK(hipMemcpyAsync(&cur, info, sizeof cur, hipMemcpyDeviceToHost, s0));
        DNLP_HIP_CHECK(hipStreamSynchronize(s0));
        if (cur.nneg > max_neg || cur.fail) { bailed = true; break; }
      }
      if (r1 >= ni) break;
      const double* Lp = A + r1 + static_cast<i64>(K0) * ld;
      if (!lookahead) {
        if (time_updates) DNLP_HIP_CHECK(hipEventRecord(tev0[static_cast<size_t>(p)], s0));
        gemm(s0, A + r1 + static_cast<i64>(r1) * ld, Wp + r1, Lp, ld, ni - r1, ni - r1, KB, 1);
        if (time_updates) DNLP_HIP_CHECK(hipEventRecord(tev1[static_cast<size_t>(p)], s0));
        const double tr = static_cast<double>(ni - r1);
        upd_flops.push_back(tr * (tr + 1.0) * static_cast<double>(KB));
        continue;
      }
      const int r2 = std::min(r1 + NB, ni);
      DNLP_HIP_CHECK(hipEventRecord(evPanel, s0));
      // the previous big update wrote the next panel's columns too: s0 waits for it
      if (upd_pending) DNLP_HIP_CHECK(hipStreamWaitEvent(s0, evUpd, 0));
      gemm(s0, A + r1 + static_cast<i64>(r1) * ld, Wp + r1, Lp, ld, ni - r1, r2 - r1, KB, 1);
      if (r2 < ni) {
        DNLP_HIP_CHECK(hipStreamWaitEvent(s1, evPanel, 0));
        if (time_updates) DNLP_HIP_CHECK(hipEventRecord(tev0[static_cast<size_t>(p)], s1));
        gemm(s1, A + r2 + static_cast<i64>(r2) * ld, Wp + r2, A + r2 + static_cast<i64>(K0) * ld, ld, ni - r2, ni - r2, KB, 1);
        if (time_updates) DNLP_HIP_CHECK(hipEventRecord(tev1[static_cast<size_t>(p)], s1));
        DNLP_HIP_CHECK(hipEventRecord(evUpd, s1));
        upd_pending = true;
        const double tr = static_cast<double>(ni - r2);
        upd_flops.push_back(tr * (tr + 1.0) * static_cast<double>(KB));
      } else {
        upd_flops.push_back(-1.0);
      }
    }
    if (lookahead) DNLP_HIP_CHECK(hipStreamSynchronize(s1));
    if (s2) DNLP_HIP_CHECK(hipStreamSynchronize(s2));
    if (!bailed) {
      DNLP_HIP_CHECK(hipMemcpyAsync(&cur, info, sizeof cur, hipMemcpyDeviceToHost, s0));
    }
    DNLP_HIP_CHECK(hipStreamSynchronize(s0));
    float upd_ms = 0.f;
    if (time_updates) {
      for (size_t k = 0; k < upd_flops.size(); ++k) {
        if (upd_flops[k] < 0.0) continue;
        float ms = 0.f;
        DNLP_HIP_CHECK(hipEventElapsedTime(&ms, tev0[k], tev1[k]));
        upd_ms += ms;
        total_update_seconds += ms * 1e-3;
        total_update_flops += upd_flops[k];
        total_update_launches += 1;
      }
    }
    last_update_seconds = upd_ms * 1e-3;
    if (bailed) ++total_bailed;
    else { outer_updates_full = 0; for (double f : upd_flops) if (f >= 0.0) ++outer_updates_full; }
    if (bailed) { *nneg = cur.nneg + 1000000; *nzero = cur.nzero; return cur.fail == 0; }
    *nneg = cur.nneg;
    *nzero = cur.nzero;
    return cur.fail == 0;
  }

  // The two sweeps as ONE launch each (ldlt_fwd_sweep_kernel / ldlt_bwd_sweep_kernel).  false: not taken (fewer than two
  // 128-blocks, more than kSweepMaxBlocks, DNLP_LDLT_SWEEP=0, or a sweep once gave up waiting) — the step kernels run
  // instead.  A sweep that gave up restores b and reports false.
  static constexpr int kSweepMaxBlocks = 1024;      // (order 131 072; DNLP_LDLT_SWEEP_MAX_BLOCKS lowers it)
  int sweep_max_blocks = std::getenv("DNLP_LDLT_SWEEP_MAX_BLOCKS") ? std::atoi(std::getenv("DNLP_LDLT_SWEEP_MAX_BLOCKS")) : kSweepMaxBlocks;
  double* sweep_xch = nullptr;      // two exchange vectors (forward / transposed sweep), nb128 x 128 doubles each
  SweepCtl* sweep_ctl = nullptr;
  double* sweep_save = nullptr;
  bool sweep_off = std::getenv("DNLP_LDLT_SWEEP") != nullptr && std::atoi(std::getenv("DNLP_LDLT_SWEEP")) == 0;
  int sweep_split = std::getenv("DNLP_LDLT_SWEEP_SPLIT") ? std::max(1, std::min(4, std::atoi(std::getenv("DNLP_LDLT_SWEEP_SPLIT")))) : 2;
  bool sweep_solve(const double* A, double* b, int nb128) {
    if (sweep_off || nb128 < 2 || nb128 > kSweepMaxBlocks || nb128 > sweep_max_blocks) return false;
    const int ni = static_cast<int>(n), P = sweep_split;
    const size_t xn = static_cast<size_t>(nb128) * SI_H, pn = xn * static_cast<size_t>(P);
    if (!sweep_xch) {
      sweep_xch = ex->alloc<double>(2 * xn + 2 * pn);          // block exchange (forward, transposed), then the members' partial sums
      sweep_ctl = ex->alloc<SweepCtl>(1);
      sweep_save = ex->alloc<double>(static_cast<size_t>(ni));
      DNLP_HIP_CHECK(hipMemsetAsync(sweep_ctl, 0, sizeof(SweepCtl), ex->stream));
    }
    DNLP_HIP_CHECK(hipMemcpyAsync(sweep_save, b, sizeof(double) * static_cast<size_t>(ni), hipMemcpyDeviceToDevice, ex->stream));
    DNLP_HIP_CHECK(hipMemsetAsync(sweep_xch, 0xFF, sizeof(double) * (2 * xn + 2 * pn), ex->stream));          // every entry "not published yet"
    const unsigned grid = static_cast<unsigned>(nb128) * static_cast<unsigned>(P);
    hipLaunchKernelGGL(ldlt_fwd_sweep_kernel, dim3(grid), dim3(SI_T), 0, ex->stream, A, ld, ni, b, Linv, sweep_xch, sweep_xch + 2 * xn, sweep_ctl, P);
    hipLaunchKernelGGL(ldlt_diag_scale, dim3((ni + 255) / 256), dim3(256), 0, ex->stream, A, ld, ni, b);
    hipLaunchKernelGGL(ldlt_bwd_sweep_kernel, dim3(grid), dim3(SI_T), 0, ex->stream, A, ld, ni, b, LinvT, sweep_xch + xn, sweep_xch + 2 * xn + pn,
                       sweep_ctl, nb128, P);
    SweepCtl h;
    DNLP_HIP_CHECK(hipMemcpyAsync(&h, sweep_ctl, sizeof h, hipMemcpyDeviceToHost, ex->stream));
    DNLP_HIP_CHECK(hipStreamSynchronize(ex->stream));
    if (h.abort) {
      sweep_off = true;
      DNLP_HIP_CHECK(hipMemsetAsync(sweep_ctl, 0, sizeof(SweepCtl), ex->stream));
      DNLP_HIP_CHECK(hipMemcpyAsync(b, sweep_save, sizeof(double) * static_cast<size_t>(ni), hipMemcpyDeviceToDevice, ex->stream));
      return false;
    }
    DNLP_LAUNCH_CHECK();
    return true;
  }

  // L D L^T x = b on the inverted diagonal blocks: one launch per 256-column step and sweep
  bool solve_prefetch = std::getenv("DNLP_LDLT_SOLVE_PREFETCH") == nullptr || std::atoi(std::getenv("DNLP_LDLT_SOLVE_PREFETCH")) != 0;
  void solve_on_inverses(const double* A, double* b) {
    const int ni = static_cast<int>(n);
    const int nb128 = (ni + SI_H - 1) / SI_H;
    if (!Linv) {
      const size_t cnt = static_cast<size_t>(nb128 + 1) * SI_H * SI_H;
      Linv = ex->alloc<double>(cnt);
      LinvT = ex->alloc<double>(cnt);
      DNLP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ldlt_inv128_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 4096 * 8));
      DNLP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ldlt_inv128_mfma_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, SI_INVM_LDS));
    }
    if (inv_of != A) inv_ready = false;
    if (!inv_ready) {
      inv_of = A;
      if (inv_mfma)
        hipLaunchKernelGGL(ldlt_inv128_mfma_kernel, dim3(static_cast<unsigned>(nb128)), dim3(SI_INVM_THREADS), SI_INVM_LDS, ex->stream, A, ld, ni,
                           Linv, LinvT);
      else
        hipLaunchKernelGGL(ldlt_inv128_kernel, dim3(static_cast<unsigned>(nb128)), dim3(128), 3 * 4096 * 8, ex->stream, A, ld, ni,
                           Linv, LinvT);
      inv_ready = true;
    }
    if (sweep_solve(A, b, nb128)) return;
    // (+ the prefetching workgroup of the next step at the first block index past the grid that is a multiple of 8)
    auto with_pf = [&](unsigned grid) { return solve_prefetch ? (grid + 7u) / 8u * 8u + 8u * (SI_PF - 1) + 1u : grid; };
    hipLaunchKernelGGL(ldlt_fwd_step_kernel, dim3(with_pf(1)), dim3(SI_T), 0, ex->stream, A, ld, ni, -SI_B, b, Linv, 1, Linv);
    for (int j0 = 0; j0 + SI_B < ni; j0 += SI_B) {
      const int below = ni - (j0 + 2 * SI_B);
      const unsigned grid = 1u + (below > 0 ? static_cast<unsigned>((below + 63) / 64) : 0u);
      hipLaunchKernelGGL(ldlt_fwd_step_kernel, dim3(with_pf(grid)), dim3(SI_T), 0, ex->stream, A, ld, ni, j0, b, Linv, static_cast<int>(grid), Linv);
    }
    hipLaunchKernelGGL(ldlt_diag_scale, dim3((ni + 255) / 256), dim3(256), 0, ex->stream, A, ld, ni, b);
    const int last = ((ni - 1) / SI_B) * SI_B;
    hipLaunchKernelGGL(ldlt_bwd_step_kernel, dim3(with_pf(1)), dim3(SI_T), 0, ex->stream, A, ld, ni, last + SI_B >= ni ? ni : ni, b, LinvT, 1, LinvT);
    for (int j0 = last; j0 >= SI_B; j0 -= SI_B) {
      const int before = j0 - SI_B;                       // columns left of the previous block
      const unsigned grid = 1u + static_cast<unsigned>(before / 64);
      hipLaunchKernelGGL(ldlt_bwd_step_kernel, dim3(with_pf(grid)), dim3(SI_T), 0, ex->stream, A, ld, ni, j0, b, LinvT, static_cast<int>(grid), LinvT);
    }
    DNLP_LAUNCH_CHECK();
  }

  void solve(const double* A, double* b) {
    if (solve_inv && padded && ld < (static_cast<i64>(1) << 21)) { solve_on_inverses(A, b); return; }
    const int ni = static_cast<int>(n);
    for (int j0 = 0; j0 < ni; j0 += SV_B) {
      const int jb = std::min(SV_B, ni - j0);
      hipLaunchKernelGGL(ldlt_fwd_diag, dim3(1), dim3(SV_B), 0, ex->stream, A, ld, j0, jb, b);
      const int rows = ni - j0 - jb;
      if (rows > 0)
        hipLaunchKernelGGL(ldlt_fwd_update, dim3((rows + 63) / 64), dim3(256), 0, ex->stream, A, ld, j0, jb, ni, b);
    }
    hipLaunchKernelGGL(ldlt_diag_scale, dim3((ni + 255) / 256), dim3(256), 0, ex->stream, A, ld, ni, b);
    ex->zero(acc, sizeof(double) * SV_B);
    const int nblk = (ni + SV_B - 1) / SV_B;
    for (int bi = nblk - 1; bi >= 0; --bi) {
      const int j0 = bi * SV_B, jb = std::min(SV_B, ni - j0);
      const int rows = ni - j0 - jb;
      if (rows > 0) {
        const int rpc = 4096;
        hipLaunchKernelGGL(ldlt_bwd_update, dim3((jb + 3) / 4, (rows + rpc - 1) / rpc), dim3(256), 0, ex->stream, A, ld,
                           j0, jb, ni, b, acc, rpc);
      }
      hipLaunchKernelGGL(ldlt_bwd_diag, dim3(1), dim3(SV_B), 0, ex->stream, A, ld, j0, jb, b, acc);
    }
    DNLP_LAUNCH_CHECK();
  }
};

inline bool HipExec::condensed_ls(i64 N, i64 m, i64 nnzJ, const i32* jr, const i32* jc, const double* jv,
                                  const double* fixmask, const double* Dd, const double* rhs_y, double* y) {
  const i64 Npad = (N + 15) / 16 * 16;
  if (m < 2 || (m & 1) || static_cast<double>(m) * static_cast<double>(Npad) * 8.0 > 2.0e9) return false;
  if (!cls_.ldlt || cls_.N != N || cls_.m != m) {
    cls_ = CondensedLs();
    cls_.N = N; cls_.m = m; cls_.Npad = Npad;
    cls_.lds = (m + 7) / 8 * 8;
    cls_.Jd = alloc<double>(static_cast<size_t>(m) * static_cast<size_t>(Npad) + 256);      // J, column-major m x Npad
    cls_.S = alloc<double>(static_cast<size_t>(cls_.lds) * static_cast<size_t>(m) + 256);
    cls_.ldlt = new BlockedLdlt();
    { BlockedLdlt* owned = cls_.ldlt; at_exit_.push_back([owned] { delete owned; }); }
    cls_.ldlt->init(this, m, cls_.lds);
    cls_.ldlt->padded = true;
  }
  double *Jd = cls_.Jd, *S = cls_.S;
  const i64 lds = cls_.lds, mm = m;
  zero(Jd, sizeof(double) * (static_cast<size_t>(m) * static_cast<size_t>(Npad) + 256));
  map(nnzJ, [=] DNLP_HD(i64 p) { if (fixmask[jc[p]] == 0.0) Jd[jr[p] + static_cast<i64>(jc[p]) * mm] = jv[p]; });
  BlockedLdlt& bl = *cls_.ldlt;
  // S = -D - J J^T on the lower tiles: W = L = J (leading dimension m), K = Npad.  Few tiles and a long K: the k range is
  // cut into slices that fill the chip, each into its own copy, the copies added in slice order.
  const int sm = static_cast<int>((m + GS_B - 1) / GS_B);
  const i64 tiles = static_cast<i64>(sm) * (sm + 1) / 2, scount = static_cast<i64>(lds) * m;
  int slices = static_cast<int>(std::min<i64>(16, std::max<i64>(1, 1024 / tiles)));
  while (slices > 1 && (Npad / slices < 256 || static_cast<double>(slices) * static_cast<double>(scount) * 8.0 > 5.0e8)) --slices;
  if (slices > 1 && (m & 1) == 0) {
    if (cls_.parts_cap < static_cast<size_t>(slices) * static_cast<size_t>(scount)) {
      cls_.parts_cap = static_cast<size_t>(slices) * static_cast<size_t>(scount);
      cls_.parts = alloc<double>(cls_.parts_cap + 256);
    }
    double* parts = cls_.parts;
    zero(parts, sizeof(double) * static_cast<size_t>(slices) * static_cast<size_t>(scount));
    const int kslice = static_cast<int>(((Npad + slices - 1) / slices + GM_BK - 1) / GM_BK * GM_BK);
    hipLaunchKernelGGL(gemm_nt_update_small_splitk, dim3(static_cast<unsigned>(sm) * sm, static_cast<unsigned>(slices)), dim3(256), 0, stream, parts, lds,
                       scount, Jd, mm, Jd, mm, static_cast<int>(m), static_cast<int>(m), static_cast<int>(Npad), kslice, 1, sm, 1);
    DNLP_LAUNCH_CHECK();
    const int ns = slices;
    map(scount, [=] DNLP_HD(i64 e) {
      const i64 i = e % lds, j = e / lds;
      double v = (i == j && i < mm) ? -Dd[i] : 0.0;
      for (int q = 0; q < ns; ++q) v += parts[static_cast<i64>(q) * scount + e];
      S[e] = v;
    });
  } else {
    zero(S, sizeof(double) * (static_cast<size_t>(lds) * static_cast<size_t>(m) + 256));
    map(m, [=] DNLP_HD(i64 i) { S[i + i * lds] = -Dd[i]; });
    const i64 keep_ldw = bl.ldw;
    bl.ldw = m;
    bl.gemm(stream, S, Jd, Jd, m, static_cast<int>(m), static_cast<int>(m), static_cast<int>(Npad), 1);
    bl.ldw = keep_ldw;
  }
  int nneg = 0, nzero = 0;
  if (!bl.factor(S, &nneg, &nzero) || nzero > 0 || nneg != static_cast<int>(m)) return false;
  map(m, [=] DNLP_HD(i64 i) { y[i] = rhs_y[i]; });
  bl.solve(S, y);
  return true;
}

// T (r x r, lower triangle) -= Pl Pw^T with Pl, Pw of r x cols (all with leading dimension ldt): the dense-tail update
// of a level of the static-pattern sparse factorisation (sparse_plan.h panels) on the 64 x 64-tile FP64 MFMA kernel
inline void HipExec::sparse_tail_gemm(double* T, i64 ldt, const double* Pl, const double* Pw, int r, int cols) {
  if (r <= 0 || cols <= 0) return;
  const int sm = (r + GS_B - 1) / GS_B;
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  const int vec_ok = al(Pl) && al(Pw) && (ldt % 2 == 0);
  hipLaunchKernelGGL(gemm_nt_update_small, dim3(static_cast<unsigned>(sm) * sm), dim3(256), 0, stream, T, ldt, Pl, ldt, Pw, ldt, r, r,
                     cols, 1, sm, vec_ok);
  DNLP_LAUNCH_CHECK();
}

inline void HipExec::ldlt_prepare(LdltWork& w, i64 n, i64 ld, bool pivoted) {
  if (pivoted) {
    if (n > BK_NMAX) throw std::runtime_error("Bunch-Kaufman path: order above BK_NMAX");
    if (w.st && w.bk_n == n) return;                 // (prepared before for this order: the workspace is kept)
    w.bk_n = n;
    w.st = alloc<BkState>(1);
    w.bk_perm = alloc<i32>(static_cast<size_t>(n));
    w.bk_dtype = alloc<i32>(static_cast<size_t>(n));
    w.bk_w = alloc<double>(static_cast<size_t>((n + 7) / 8 * 8) * 16);
    w.bk_swaps = alloc<BkPanelSwaps>(1);
    if (const char* ev = std::getenv("DNLP_BK_PANELS")) w.bk_panels = std::atoi(ev) != 0;
  } else {
    // (a handle demoted from paired / optimistic mode prepares again: the workspace of the same order is kept)
    if (w.blocked && w.blocked->n == n && w.blocked->ld == ld) return;
    w.blocked = new BlockedLdlt();
    { BlockedLdlt* owned = w.blocked; at_exit_.push_back([owned] { delete owned; }); }
    w.blocked->init(this, n, ld);
  }
}
inline bool HipExec::ldlt_factor(LdltWork& w, double* A, i64 n, i64 ld, i32* ipiv, bool pivoted, int* nneg, int* nzero) {
  if (pivoted) return bk_factor(w, A, n, ld, ipiv, nneg, nzero);
  w.blocked->max_neg = w.expect_neg;
  w.blocked->time_updates = w.time_updates;
  w.blocked->padded = w.padded;
  return w.blocked->factor(A, nneg, nzero);
}
inline void HipExec::ldlt_stats(LdltWork& w, double* out3) {   // out3: room for 5 values
  out3[0] = out3[1] = out3[2] = out3[3] = out3[4] = 0.0;
  if (w.blocked) {
    out3[3] = static_cast<double>(w.blocked->outer_updates_full);
    out3[4] = static_cast<double>(w.blocked->total_bailed);
    out3[0] = w.blocked->total_update_seconds;
    out3[1] = w.blocked->total_update_flops;
    out3[2] = static_cast<double>(w.blocked->total_update_launches);
  }
}
inline void HipExec::ldlt_solve(LdltWork& w, const double* A, i64 n, i64 ld, const i32* ipiv, bool pivoted, double* b) {
  if (pivoted) {
    (void)ipiv;
    hipLaunchKernelGGL(bk_solve_kernel, dim3(1), dim3(BK_T), 0, stream, A, static_cast<int>(n), ld, w.bk_perm, w.bk_dtype, b);
  } else {
    w.blocked->solve(A, b);
  }
}

// ---- seeded dense symmetric test matrix (BASELINE config C4) --------------------------------
__device__ inline uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ inline double u01(uint64_t h) { return static_cast<double>(h >> 11) * (1.0 / 9007199254740992.0); }

__global__ void __launch_bounds__(256) gen_vec_kernel(double* v, i64 n, uint64_t seed) {
  const i64 i = static_cast<i64>(blockIdx.x) * 256 + threadIdx.x;
  if (i < n) v[i] = 2.0 * u01(splitmix64(seed ^ (0xA5A5A5A5ull + static_cast<uint64_t>(i) * 0x100000001B3ull))) - 1.0;
}
// A[i,j] = noise(min(i,j), max(i,j)) in [-1,1) + spike * v_i v_j
__global__ void __launch_bounds__(256) gen_sym_kernel(double* A, i64 n, i64 ld, uint64_t seed, double spike,
                                                      const double* v) {
  const i64 r = static_cast<i64>(blockIdx.x) * 256 + threadIdx.x;
  if (r >= n) return;
  for (i64 c = blockIdx.y; c < n; c += gridDim.y) {
    const uint64_t lo = static_cast<uint64_t>(r < c ? r : c), hi = static_cast<uint64_t>(r < c ? c : r);
    const double noise = 2.0 * u01(splitmix64(seed + lo * 0x9E3779B97F4A7C15ull + splitmix64(hi))) - 1.0;
    A[r + c * ld] = noise + spike * v[r] * v[c];
  }
}

}  // namespace dnlp
