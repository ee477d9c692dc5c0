// 128 x 128 diagonal block of a sub-panel with the block work on FP64 MFMA (included by ldlt_blocked.h).
//
// ldlt_top128_kernel walks 32 tile columns with two workgroup barriers each and takes its rank-4 updates on
// the vector ALU: ~54 us for one workgroup, 92 times in a row on the panel chain of an order-11 000 matrix.
// Here the block is 8 x 8 blocks of 16 x 16 and a step is a block column:
//   * wavefront w owns block row w; a block sits TRANSPOSED in the accumulator layout of
//     v_mfma_f64_16x16x4_f64 (register r of lane (lq, lr) = element (row lr, column lq + 4 r)), so a block is
//     at once the B operand of a product from the left — the same trick as ldlt_rows128_kernel;
//   * the diagonal block is factored by ITS wavefront alone, lane = row with the row in registers, pivots and
//     multipliers by v_readlane (the ldlt_diag_kernel scheme, no LDS inside, no barrier).  Sixteen more lanes
//     carry the rows of the identity through the same elimination: what they hold at the end is
//     inv(L_kk)^T, the operand of the solves below — the inverse costs no extra step on the chain;
//   * blocks below: W^T = inv(L_kk) A_ik^T (4 MFMAs), L = W D^-1; the negated L blocks go to LDS in operand
//     layout (one 512-B row per register: what a lane wrote is what the same lane of another wavefront reads);
//   * trailing blocks: A_ij^T += (-L_jk) W_ik^T (4 MFMAs per block), the block that holds the next diagonal
//     first.
// Two LDS-only barriers per block column (16 in all instead of 64), the chain between them is the sixteen
// pivots of the diagonal block.  Leaves the same packed operand copy behind as ldlt_top128_kernel.
#pragma once

namespace dnlp {

constexpr int LD_TOPM_THREADS = 64 * LD_TB;

__device__ inline void ldlt_barrier_lds_only() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int PRB>
__device__ __forceinline__ void ldlt_top128_mfma_body(double* __restrict__ A, i64 ld, int j0, LdltInfo* info, double tiny,
                                                      double* __restrict__ Ltop) {
  __shared__ __attribute__((aligned(16))) double negL[LD_TB][256];   // -L_jk of the current block column
  __shared__ __attribute__((aligned(16))) double invS[256];          // inv(L_kk), operand layout
  __shared__ double dinvS[16];
  __shared__ double stage[16][17];                                   // diagonal block between the two layouts
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lq = lane >> 4, lr = lane & 15;
  // T[p] = block (w, k + p) at step k; finished block columns rotate out at the front
  mfma_d4 T[LD_TB];
  {
    const double* src = A + (j0 + 16 * w + lr) + static_cast<i64>(j0 + lq) * ld;
#pragma unroll
    for (int p = 0; p < LD_TB; ++p)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        T[p][r] = (p < w || (p == w && lq + 4 * r <= lr)) ? src[static_cast<i64>(16 * p + 4 * r) * ld] : 0.0;
  }
  int nneg = 0, nzero = 0, fail = 0;
#pragma unroll 1
  for (int k = 0; k < LD_TB; ++k) {
    if (w == k) {
#pragma unroll
      for (int r = 0; r < 4; ++r) stage[lr][lq + 4 * r] = T[0][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
      // lanes 0..15: rows of the block; lanes 16..31: rows of the identity; the rest idle on zeros
      double E[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const double v = stage[lr][c];
        E[c] = lane < 16 ? (c <= lane ? v : 0.0) : ((lane < 32 && c == lane - 16) ? 1.0 : 0.0);
      }
      double dis[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        if constexpr (PRB & 1) { dis[c] = E[c]; continue; }
        double d = ldlt_bcast(E[c], c);
        double di = ldlt_rcp(d);                  // starts on the raw pivot; a repair redoes it
        if (!(fabs(d) > tiny)) {
          if (!(d == d)) { fail = 1; d = 1.0; }
          else { nzero += 1; d = (d < 0.0 ? -tiny : tiny); if (d == 0.0) d = 1e-300; }
          di = ldlt_rcp(d);
          if (lane == c) E[c] = d;
        }
        nneg += d < 0.0 ? 1 : 0;
        dis[c] = di;
        const double l = E[c] * di;
#pragma unroll
        for (int j = c + 1; j < 16; ++j) E[j] -= l * ldlt_bcast(E[c], j);
      }
      double mydi = 0.0;
#pragma unroll
      for (int c = 0; c < 16; ++c) mydi = lane == c ? dis[c] : mydi;
      if (lane < 16) {
        dinvS[lane] = mydi;
        Ltop[LD_TOP_DINV + 16 * k + lane] = mydi;
        double* dst = A + (j0 + 16 * k + lane) + static_cast<i64>(j0 + 16 * k) * ld;
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if (c <= lane) dst[static_cast<i64>(c) * ld] = c == lane ? E[c] : E[c] * dis[c];
      } else if (lane < 32) {
        const int j = lane - 16;
#pragma unroll
        for (int c = 0; c < 16; c += 2) {
          *reinterpret_cast<double2*>(&invS[j * 16 + c]) = double2{E[c], E[c + 1]};
          *reinterpret_cast<double2*>(&Ltop[LD_TOP_INV + k * 256 + j * 16 + c]) = double2{E[c], E[c + 1]};
        }
      }
    }
    ldlt_barrier_lds_only();
    mfma_d4 W = {0.0, 0.0, 0.0, 0.0}, nL = {0.0, 0.0, 0.0, 0.0};
    if (w > k) {
#pragma unroll
      for (int s = 0; s < 4; ++s) W = __builtin_amdgcn_mfma_f64_16x16x4f64(invS[64 * s + lane], T[0][s], W, 0, 0, 0);
      double* dst = A + (j0 + 16 * w + lr) + static_cast<i64>(j0 + 16 * k + lq) * ld;
      double* pk = Ltop + LD_TOP_NEG + (w * (w - 1) / 2 + k) * 256 + lane;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double l = W[r] * dinvS[lq + 4 * r];
        nL[r] = -l;
        negL[w][64 * r + lane] = -l;
        if constexpr ((PRB & 4) == 0) {
        dst[static_cast<i64>(4 * r) * ld] = l;
        pk[64 * r] = -l;
        }
      }
    }
    ldlt_barrier_lds_only();
    if ((PRB & 2) == 0 && w > k) {
      // own diagonal block first: the next step's chain starts from it
#pragma unroll
      for (int p = 1; p < LD_TB; ++p)
        if (p == w - k) {
#pragma unroll
          for (int s = 0; s < 4; ++s) T[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(nL[s], W[s], T[p], 0, 0, 0);
        }
#pragma unroll
      for (int p = 1; p < LD_TB; ++p)
        if (p < w - k) {
#pragma unroll
          for (int s = 0; s < 4; ++s)
            T[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(negL[k + p][64 * s + lane], W[s], T[p], 0, 0, 0);
        }
    }
    if constexpr ((PRB & 8) == 0) {
#pragma unroll
    for (int p = 0; p + 1 < LD_TB; ++p) T[p] = T[p + 1];
    }
  }
  if (lane == 0) {
    if (nneg) atomicAdd(&info->nneg, nneg);
    if (nzero) atomicAdd(&info->nzero, nzero);
    if (fail) atomicExch(&info->fail, 1);
  }
}

__global__ void __launch_bounds__(LD_TOPM_THREADS) ldlt_top128_mfma_kernel(double* __restrict__ A, i64 ld, int j0,
                                                                          LdltInfo* info, double tiny,
                                                                          double* __restrict__ Ltop) {
  ldlt_top128_mfma_body<0>(A, ld, j0, info, tiny, Ltop);
}
template <int PRB>
__global__ void __launch_bounds__(LD_TOPM_THREADS) ldlt_top128_mfma_probe(double* __restrict__ A, i64 ld, int j0,
                                                                         LdltInfo* info, double tiny,
                                                                         double* __restrict__ Ltop) {
  ldlt_top128_mfma_body<PRB>(A, ld, j0, info, tiny, Ltop);
}

}  // namespace dnlp
