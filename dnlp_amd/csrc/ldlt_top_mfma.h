// 128 x 128 diagonal block of a sub-panel with ALL of its arithmetic on FP64 MFMA (included by ldlt_blocked.h).
//
// ldlt_top128_kernel walks 32 tile columns with two workgroup barriers each and takes its rank-4 updates on
// the vector ALU: ~44 us for one workgroup alone, 92 times in a row on the panel chain of an order-11 000 matrix.
// Here the block is 8 x 8 blocks of 16 x 16 and a step is a block column:
//   * wavefront w owns block row w; a block sits TRANSPOSED in the accumulator layout of
//     v_mfma_f64_16x16x4_f64 (register r of lane (lq, lr) = element (row lr, column lq + 4 r)), so a block is
//     at once the B operand of a product from the left — the same trick as ldlt_rows128_kernel;
//   * the diagonal block is factored by ITS wavefront alone and never leaves the accumulators: in that layout
//     row c of the Schur complement — w_n = S[n][c], n > c — sits in register c / 4 of the sixteen lanes with
//     lq = c % 4, which is where the MFMA takes k-slot c % 4 of BOTH operands from.  The rank-1 update of a
//     pivot is therefore ONE v_mfma_f64_16x16x4_f64 whose operands are that register masked to those lanes
//     (B) and the same times -1/d (A): no cross-lane traffic but the two v_readlane of the pivot, ~25
//     instructions per column where the lane-per-row scheme of ldlt_diag_kernel issues ~60 (a one-wavefront
//     chain is bound by instruction issue, not by latency: measured 1.94 us per 16 columns that way).  A
//     second accumulator block starts as the identity and takes the same updates: at the end it is
//     inv(L_kk), the operand of the solves below, at one more MFMA per column that runs under the next
//     pivot's reciprocal;
//   * blocks below: W^T = inv(L_kk) A_ik^T (4 MFMAs), L = W D^-1; the negated L blocks go to LDS in operand
//     layout (one 512-B row per register: what a lane wrote is what the same lane of another wavefront reads);
//   * trailing blocks: A_ij^T += (-L_jk) W_ik^T (4 MFMAs per block); a wavefront updates its own diagonal block
//     from registers before the barrier, so the wavefront of the next diagonal goes straight on.
// Two LDS-only barriers per block column (16 in all instead of 64); stores to the matrix and to the packed copy
// are issued after the barrier that publishes the inverse, off the chain.  Leaves the same packed operand copy
// behind as ldlt_top128_kernel.
#pragma once

namespace dnlp {

constexpr int LD_TOPM_THREADS = 64 * LD_TB;

__device__ inline void ldlt_barrier_lds_only() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// The factorisation proper, on the block row T every wavefront already holds (T[q] = block (w, w - q) transposed, zero
// above the diagonal of T[0] and for q > w): shared by the one-block kernel below and by ldlt_diag512_kernel.
__device__ __forceinline__ void ldlt_top128_body(mfma_d4 (&T)[LD_TB], double* __restrict__ A, i64 ld, int j0, double tiny,
                                                 double* __restrict__ Ltop, int& nneg, int& nzero, int& fail,
                                                 double (*negL)[256], double* invS, double* dinvS) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lq = lane >> 4, lr = lane & 15;
  // high words above this belong to magnitudes above `tiny` whatever the low word is
  const unsigned tiny_hi = (static_cast<unsigned>(__double2hiint(tiny)) & 0x7fffffffu) + 1u;
#pragma unroll 1
  for (int k = 0; k < LD_TB; ++k) {
    if (w == k) {
      mfma_d4 D = T[0], Y;
#pragma unroll
      for (int r = 0; r < 4; ++r) Y[r] = (lq + 4 * r == lr) ? 1.0 : 0.0;
      double mydi = 0.0;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        const int cq = c & 3, cr = c >> 2, src = cq * 16 + c;      // pivot c: register cr of lane src
        double d = ldlt_bcast(D[cr], src);
        const bool sel = lq == cq;
        const double b = (c < 15 && sel && lr > c) ? D[cr] : 0.0;       // w_n, n > c, in k-slot cq
        // 1/d = r0 (1 + e + e^2), e = 1 - d r0 (|e| <= 4.6e-8: the cube is below double precision) — and the
        // multipliers are formed from b r0 beside it, so the chain pivot -> update operand is four dependent
        // operations (reciprocal, e | b r0, e + e^2, operand) instead of six
        double r0 = __builtin_amdgcn_rcp(d);
        double e = fma(-d, r0, 1.0), tn = -(b * r0);
        double pe = fma(e, e, e);
        double a = fma(tn, pe, tn);                                      // -l_m, m > c
        double di = fma(r0, pe, r0);
        asm volatile("" : "+v"(a), "+v"(di));    // (computed before the pivot test's branch, not once on either side of it)
        // the pivot is uniform (two scalar registers): the common case is told apart on the scalar unit from the high
        // word alone — magnitude well above `tiny` and not NaN / infinity — and only the rest takes the exact test
        const unsigned dh = static_cast<unsigned>(__double2hiint(d)) & 0x7fffffffu;
        if (__builtin_expect(dh - tiny_hi >= 0x7ff00000u - tiny_hi, 0)) {
          if (!(fabs(d) > tiny)) {
            if (!(d == d)) { fail = 1; d = 1.0; }
            else { nzero += 1; d = (d < 0.0 ? -tiny : tiny); if (d == 0.0) d = 1e-300; }
            di = ldlt_rcp(d);
            a = -(b * di);
            if (lane == src) D[cr] = d;
          }
        }
        nneg += d < 0.0 ? 1 : 0;
        mydi = lane == c ? di : mydi;
        if (c < 15) {
          const double yb = sel ? Y[cr] : 0.0;                // row c of the inverse so far
          // (the inverse's update directly behind the block's: issued ahead of the next pivot's reciprocal it
          //  measured 1.5 us slower — FP64 vector instructions wait for the FP64 matrix pipe)
          D = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, D, 0, 0, 0);
          Y = __builtin_amdgcn_mfma_f64_16x16x4f64(a, yb, Y, 0, 0, 0);
        }
      }
      T[0] = D;
#pragma unroll
      for (int r = 0; r < 4; ++r) invS[(lq + 4 * r) * 17 + lr] = Y[r];
      if (lane < 16) dinvS[lane] = mydi;
    }
    ldlt_barrier_lds_only();
    double inva[4], dinv4[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      inva[s] = invS[lr * 17 + lq + 4 * s];
      dinv4[s] = dinvS[lq + 4 * s];
    }
    mfma_d4 W = {0.0, 0.0, 0.0, 0.0};
    if (w > k) {
#pragma unroll
      for (int q = 1; q < LD_TB; ++q)
        if (q == w - k) {
#pragma unroll
          for (int s = 0; s < 4; ++s) W = __builtin_amdgcn_mfma_f64_16x16x4f64(inva[s], T[q][s], W, 0, 0, 0);
        }
      mfma_d4 nL;
#pragma unroll
      for (int r = 0; r < 4; ++r) nL[r] = -(W[r] * dinv4[r]);
      // own diagonal block from registers: the next step's chain starts from it
#pragma unroll
      for (int s = 0; s < 4; ++s) T[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(nL[s], W[s], T[0], 0, 0, 0);
      double* dst = A + (j0 + 16 * w + lr) + static_cast<i64>(j0 + 16 * k + lq) * ld;
      double* pk = Ltop + LD_TOP_NEG + (w * (w - 1) / 2 + k) * 256 + lane;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        negL[w][64 * r + lane] = nL[r];
        dst[static_cast<i64>(4 * r) * ld] = -nL[r];
        pk[64 * r] = nL[r];
      }
    } else if (w == k) {
      double* dst = A + (j0 + 16 * k + lr) + static_cast<i64>(j0 + 16 * k + lq) * ld;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int col = lq + 4 * r;
        if (lr >= col) dst[static_cast<i64>(4 * r) * ld] = lr == col ? T[0][r] : T[0][r] * dinv4[r];
      }
      if (lane < 16) Ltop[LD_TOP_DINV + 16 * k + lane] = dinvS[lane];
    }
    if (w == LD_TB - 1) {
#pragma unroll
      for (int s = 0; s < 4; ++s) Ltop[LD_TOP_INV + k * 256 + 64 * s + lane] = inva[s];
    }
    ldlt_barrier_lds_only();
    if (w > k) {
#pragma unroll
      for (int q = 1; q < LD_TB - 1; ++q)
        if (q < w - k) {
#pragma unroll
          for (int s = 0; s < 4; ++s)
            T[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(negL[w - q][64 * s + lane], W[s], T[q], 0, 0, 0);
        }
    }
  }
}

__device__ __forceinline__ void ldlt_top128_load(mfma_d4 (&T)[LD_TB], const double* __restrict__ A, i64 ld, int j0) {
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lq = lane >> 4, lr = lane & 15;
  // T[q] = block (w, w - q): q = 0 the wavefront's own diagonal block, q = w - k the block of column k
  const double* src = A + (j0 + 16 * w + lr) + static_cast<i64>(j0 + 16 * w + lq) * ld;
#pragma unroll
  for (int q = 0; q < LD_TB; ++q)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      T[q][r] = ((q > 0 && q <= w) || (q == 0 && lq + 4 * r <= lr)) ? src[static_cast<i64>(4 * r - 16 * q) * ld] : 0.0;
}

__global__ void __launch_bounds__(LD_TOPM_THREADS) ldlt_top128_mfma_kernel(double* __restrict__ A, i64 ld, int j0,
                                                                          LdltInfo* info, double tiny,
                                                                          double* __restrict__ Ltop) {
  __shared__ __attribute__((aligned(16))) double negL[LD_TB][256];   // -L_jk of the current block column
  __shared__ double invS[16 * 17];                                   // inv(L_kk) row-major, rows padded
  __shared__ double dinvS[16];
  mfma_d4 T[LD_TB];
  ldlt_top128_load(T, A, ld, j0);
  int nneg = 0, nzero = 0, fail = 0;
  ldlt_top128_body(T, A, ld, j0, tiny, Ltop, nneg, nzero, fail, negL, invS, dinvS);
  if ((threadIdx.x & 63) == 0) {
    if (nneg) atomicAdd(&info->nneg, nneg);
    if (nzero) atomicAdd(&info->nzero, nzero);
    if (fail) atomicExch(&info->fail, 1);
  }
}

}  // namespace dnlp
