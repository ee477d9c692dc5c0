// Inverses of the unit-lower 128 x 128 diagonal blocks of a factor on FP64 MFMA (included by ldlt_blocked.h;
// replaces ldlt_inv128_kernel, which DNLP_LDLT_INV_MFMA=0 restores).
//
// The inverses are what the triangular solves multiply with (ldlt_fwd_step_kernel / ldlt_bwd_step_kernel); they are
// built once per factorisation that is solved with.  ldlt_inv128_kernel does a block with 128 lanes — two 64-step
// substitutions with a column per lane, then two 64 x 64 x 64 products on the vector ALU — in ~100 us, whatever the
// order: for the dense tails of the sparse plan and the paired dense KKT systems of a few hundred rows that is as long
// as their factorisation.  Here a block is 8 x 8 blocks of 16 x 16 and one workgroup of eight wavefronts:
//   * wavefront w inverts the diagonal block L_ww: Y starts as the identity in the accumulator layout of
//     v_mfma_f64_16x16x4_f64 and takes the fifteen elimination steps Y[m][:] -= L[m][c] Y[c][:] as rank-1 MFMAs — row c
//     of Y is register c / 4 of the lanes with lane / 16 = c % 4, which is where the MFMA reads k-slot c % 4 of B, and
//     the A operand is column c of L loaded straight from the factor (the scheme of ldlt_top128_mfma_kernel's
//     diagonal blocks, without the pivots).  The result goes to LDS row-major, rows padded to 17;
//   * wavefront j then builds block column j of X = L^-1 top down: X_ij = -inv(L_ii) sum_{k=j}^{i-1} L_ik X_kj.  A
//     finished X_kj sits in the accumulators, i.e. it IS the B operand of the next product; the A operands are
//     L_ik read from the factor (lane (lq, lr): element (lr, 4 s + lq)) and inv(L_ii) read transposed from LDS.
// Longest chain: block column 0, 35 block products of four MFMAs.  Rows and columns past n are identity.
#pragma once

namespace dnlp {

constexpr int SI_INVM_THREADS = 512;
constexpr int SI_INVM_LDS = (36 * 4 * 64 + 8 * 16 * 17) * 8;      // operand copy of the 36 lower blocks + the eight inverses

__global__ void __launch_bounds__(SI_INVM_THREADS) ldlt_inv128_mfma_kernel(const double* __restrict__ A, i64 ld, int n,
                                                                          double* __restrict__ inv,
                                                                          double* __restrict__ invT) {
  extern __shared__ __align__(16) double sim_lds[];
  double* Lop = sim_lds;                          // [block (i, k), k <= i][slab s][lane]: the MFMA A operand of L_ik
  double* dinv = sim_lds + 36 * 4 * 64;           // [block][16 x 17]: inv(L_bb) row-major, rows padded
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lq = lane >> 4, lr = lane & 15;
  const int j0 = blockIdx.x * SI_H;
  // The whole lower triangle goes to LDS first, every load of the workgroup in flight at once: operand reads inside
  // the chains below are then LDS reads (fetched from the factor where they are used, each of the 35 block products of
  // the longest chain waited for its own round trip to memory: 34 us per block instead of 9).
  // Operand layout: lane (lq, lr), slab s of block (bi, bk) = L[16 bi + lr][16 bk + 4 s + lq]; zero on and above the
  // diagonal and past the matrix.
  {
    double v[18];
#pragma unroll
    for (int t = 0; t < 18; ++t) {
      const int item = w + 8 * t;                 // 144 (block, slab) items, eighteen per wavefront
      const int blk = item >> 2, sl = item & 3;
      int bi = 0;
      while ((bi + 1) * (bi + 2) / 2 <= blk) ++bi;        // blk = bi (bi + 1) / 2 + bk
      const int bk = blk - bi * (bi + 1) / 2;
      const int r = 16 * bi + lr, c = 16 * bk + 4 * sl + lq;
      v[t] = (r > c && j0 + r < n) ? A[(j0 + r) + static_cast<i64>(j0 + c) * ld] : 0.0;
    }
#pragma unroll
    for (int t = 0; t < 18; ++t) Lop[(w + 8 * t) * 64 + lane] = v[t];
  }
  __syncthreads();
  auto lop = [&](int bi, int bk, int s) -> double { return Lop[((bi * (bi + 1) / 2 + bk) * 4 + s) * 64 + lane]; };
  mfma_d4 X[8];
  {
    double av[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) av[s] = lop(w, w, s);
    mfma_d4 Y;
#pragma unroll
    for (int r = 0; r < 4; ++r) Y[r] = (lq + 4 * r == lr) ? 1.0 : 0.0;
#pragma unroll
    for (int c = 0; c < 15; ++c) {
      const int cq = c & 3, cr = c >> 2;
      const bool sel = lq == cq;
      const double a = sel ? -av[cr] : 0.0;        // -L[m][c], m > c (zero above the diagonal by construction)
      const double yb = sel ? Y[cr] : 0.0;         // row c of the inverse so far
      Y = __builtin_amdgcn_mfma_f64_16x16x4f64(a, yb, Y, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) dinv[w * 272 + (lq + 4 * r) * 17 + lr] = Y[r];
#pragma unroll
    for (int i = 0; i < 8; ++i) X[i] = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i == w) X[i] = Y;
  }
  __syncthreads();
#pragma unroll
  for (int i = 1; i < 8; ++i) {
    if (i > w) {
      mfma_d4 T = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k = 0; k < i; ++k)
        if (k >= w) {
#pragma unroll
          for (int s = 0; s < 4; ++s) T = __builtin_amdgcn_mfma_f64_16x16x4f64(lop(i, k, s), X[k][s], T, 0, 0, 0);
        }
      mfma_d4 R = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < 4; ++s)
        R = __builtin_amdgcn_mfma_f64_16x16x4f64(-dinv[i * 272 + lr * 17 + 4 * s + lq], T[s], R, 0, 0, 0);
      X[i] = R;
    }
  }
  // block (i, w) of X in the accumulator layout: register r of lane (lq, lr) = X[16 i + lq + 4 r][16 w + lr]
  double* out = inv + static_cast<i64>(blockIdx.x) * (SI_H * SI_H);      // X column-major: r + 128 c
  double* outT = invT + static_cast<i64>(blockIdx.x) * (SI_H * SI_H);    // X^T column-major: c + 128 r
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * i + lq + 4 * r, col = 16 * w + lr;
      const double v = X[i][r];                    // zero blocks above the diagonal were never touched
      out[row + SI_H * col] = v;
      outT[col + SI_H * row] = v;
    }
}

}  // namespace dnlp
