// Bunch-Kaufman LDL^T pieces shared by the host-driven path (exec_hip.h: one launch per panel) and the in-kernel path
// (exec_block.h: a batch instance's own workgroup): state, block-wide argmax, and the panel factorisation as a device
// function over PT lanes.
#pragma once
#include <hip/hip_runtime.h>

#include "exec.h"
#include "wave_ops.h"

namespace dnlp {

// ---- Bunch-Kaufman LDL^T (DSYTF2 semantics, lower) ----------------------------------------
struct BkState {
  int k, kstep, kp, pending;      // pending: previous step's columns still need scaling
  int nneg, nzero, fail, pad;
  double d11, d22, d21;           // 1x1: d11 = pivot ; 2x2: the DSYTF2 multipliers
};


// block-wide argmax of v >= 0 (v = -1: the lane has no candidate); the smallest index wins ties, as IDAMAX
template <int NT = 1024>
__device__ inline void bk_argmax(double v, int idx, double* sv, int* si, double& outv, int& outi) {
  const bool has = v >= 0.0;
  const double m = wave_all_max(has ? v : 0.0);
  const int cand = wave_all_min((has && v == m) ? idx : 0x7fffffff);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) { sv[wid] = cand == 0x7fffffff ? -1.0 : m; si[wid] = cand; }
  __syncthreads();
  // sixteen wavefront results: a four-level tree in every lane (a serial scan is fifteen dependent
  // compare / select rounds of ~100 cycles each)
  double tv[NT / 64];
  int tix[NT / 64];
#pragma unroll
  for (int w = 0; w < NT / 64; ++w) { tv[w] = sv[w]; tix[w] = si[w]; }
#pragma unroll
  for (int span = NT / 128; span >= 1; span >>= 1)
#pragma unroll
    for (int w = 0; w < span; ++w) {
      const bool take = tv[w + span] > tv[w] || (tv[w + span] == tv[w] && tix[w + span] < tix[w]);
      tv[w] = take ? tv[w + span] : tv[w];
      tix[w] = take ? tix[w + span] : tix[w];
    }
  outv = tv[0];
  outi = tix[0];
  __syncthreads();
}

// One workgroup: finish the previous step (scale its columns, advance k), then search the
// pivot for the new k, apply the symmetric interchange and publish the update multipliers.

// ---- panel-blocked Bunch-Kaufman (the LASYF idea, one workgroup per panel) --------------------------
// The unblocked pair above costs two launches and a whole-matrix rank-1 read-modify-write per column.
// Here a panel of NBP columns is factored LEFT-looking by ONE workgroup: a lane owns ROWS rows and keeps
// their entries of W = L D for the panel's finished columns in registers; a column is brought up to date
// with a ROWS x j product against the pivot row's multipliers (recomputed from that row's W and the pivot
// blocks, broadcast through LDS); the pivot search and the interchanges are the DSYTF2 ones (same choices
// as bk_pivot_kernel), and the trailing matrix is touched once per panel by bk_panel_update_kernel:
// A22 -= W21 L21^T, rank NBP.  A 2x2 pivot that would start in the panel's last column ends the panel one
// column early.  What keeps the per-column chain short (the trailing matrix was last written by the
// chip-wide update, so a first touch from this one workgroup costs an Infinity-Cache / HBM round trip of
// ~3 000 cycles):
//   * the panel's columns are touched once at the start, all loads in flight together, so the per-column
//     loads hit this XCD's L2;
//   * an interchange swaps the rows of the PANEL's finished columns only; the rows of the columns of earlier
//     panels are swapped by the update kernel that follows (recorded in st->swaps), off this chain;
//   * the barrier that ends a column waits for LDS only: the multipliers it stored are not read again here.
struct BkPanelSwaps { int count; int pad; int rows[2 * 16]; };

__device__ inline void bk_barrier_lds_only() {
  // workgroup barrier that orders LDS traffic only: release / acquire fences restricted to the local address
  // space (the compiler may not move LDS accesses across them; they lower to s_waitcnt lgkmcnt(0), so
  // outstanding global stores keep draining — no vmcnt wait, no hard-coded waitcnt immediate)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// (512 lanes x ROWS rows: 1024 lanes leave 128 VGPRs per lane, and W alone is 64 of them -- the kernel spilled)
// 1 / d without the IEEE division sequence: v_rcp_f64 (4.6e-8) and two Newton steps (1.1e-16)
__device__ inline double bk_rcp(double d) {
  double r = __builtin_amdgcn_rcp(d);
  r = fma(r, fma(-d, r, 1.0), r);
  r = fma(r, fma(-d, r, 1.0), r);
  return r;
}
constexpr int BK_PT = 512;
// (the body is a device function over PT lanes: the host-driven path launches it as bk_panel_kernel with PT = 512,
//  a batch instance of KKT order 257 .. 2048 runs it inside its own workgroup with PT = 256 — exec_block.h)
template <int ROWS, int NBP, int PT>
__device__ inline void bk_panel_body(double* A, int n, i64 ld, int* ipiv, BkState* st, double* Wg, i64 ldw, BkPanelSwaps* swaps) {
  constexpr int BK_PT = PT;
  __shared__ double sv[BK_PT / 64];
  __shared__ int si[BK_PT / 64];
  __shared__ __attribute__((aligned(16))) double vb[2][NBP];   // multipliers L[row, panel columns] of the pivot row / the candidate row
  __shared__ double xr[2][NBP + 2];        // W row of the pivot row / the candidate row, zero on both sides
  __shared__ double xw[2][NBP + 2];        // interchange: W rows and the (w, c) entries of rows kk / kp
  __shared__ double dinf[NBP][3];          // 1x1: {1/d} ; first column of a 2x2: {d11, d22, d21}
  __shared__ int dk[NBP];                  // 0: 1x1, 1 / 2: first / second column of a 2x2
  __shared__ int swp[2 * NBP];
  __shared__ double draw[NBP][2];          // diagonal entries as stored: {d} or {a_kk, off} / {c_ii}
  __shared__ int piv_out[NBP];
  __shared__ int nswp;
  __shared__ double s_akk, s_cii, s_off;
  const int tid = threadIdx.x;
  const double alpha = 0.6403882032022076;   // (1 + sqrt(17)) / 8
  const int k0 = st->k;
  if (k0 >= n || st->fail) {
    if (tid == 0) { swaps->count = 0; st->pending = 0; }
    return;
  }
  if (tid == 0) nswp = 0;
  double W[ROWS][NBP];
  {
    // first touch of the panel's columns: NBP independent loads per row in flight together
    double pf = 0.0;
#pragma unroll
    for (int s = 0; s < ROWS; ++s) {
      const int r = tid + s * BK_PT;
#pragma unroll
      for (int i = 0; i < NBP; ++i) {
        W[s][i] = 0.0;
        if (r >= k0 && r < n && k0 + i < n) pf += A[r + static_cast<i64>(k0 + i) * ld];
      }
    }
    if (pf == 1.2345678e-301) Wg[0] = pf;    // (keeps the loads)
  }
  int nneg = 0, nzero = 0, fail = 0;
  int j = 0;
  // multipliers of `row` for the panel columns done so far, by the WAVEFRONT of the lane that owns the row: the
  // owner puts its W row into LDS, lane i of the same wavefront turns column i's entry into the multiplier (its
  // pivot data, the three forms 1x1 / first / second column of a 2x2, 0 for columns not done yet) -- two LDS round
  // trips in all, where one lane walking the columns paid one dependent round trip and a branch per column
  auto publish_lrow = [&](int row_in, int slot) {
    const int row = __builtin_amdgcn_readfirstlane(row_in);
    const int owner = row % BK_PT, srow = row / BK_PT;
    if ((tid >> 6) == (owner >> 6)) {
      const int lane = tid & 63;
      if (tid == owner) {
#pragma unroll
        for (int s = 0; s < ROWS; ++s)
          if (s == srow) {
#pragma unroll
            for (int i = 0; i < NBP; ++i) xr[slot][i + 1] = W[s][i];
          }
        xr[slot][0] = 0.0;
        xr[slot][NBP + 1] = 0.0;
      }
      // one wavefront's LDS operations complete in order; the fences keep the compiler from moving the
      // xr stores below / the dk, dinf, xr loads above this point (lgkmcnt(0) only, no vmcnt wait)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
      if (lane < NBP) {
        const int d = dk[lane];
        const double e0 = dinf[lane][0], e2 = dinf[lane][2];
        const double mine = xr[slot][lane + 1], up = xr[slot][lane + 2], dn = xr[slot][lane];
        const double other = d == 1 ? up : dn;
        const double two = e2 * (e0 * mine - other);
        const double val = d == 0 ? mine * e0 : two;
        vb[slot][lane] = lane < j ? val : 0.0;
      }
    }
  };
  __syncthreads();
  while (j < NBP && k0 + j < n) {
    const int k = k0 + j;
    double a[ROWS], c[ROWS];
#pragma unroll
    for (int s = 0; s < ROWS; ++s) {
      const int r = tid + s * BK_PT;
      a[s] = (r >= k && r < n) ? A[r + static_cast<i64>(k) * ld] : 0.0;
      c[s] = 0.0;
    }
    publish_lrow(k, 0);
    __syncthreads();
    // four multipliers per LDS round trip (columns not done yet hold 0 in vb and in W)
#pragma unroll
    for (int h = 0; h < NBP; h += 4)
      if (h < j) {
        const double2 v01 = *reinterpret_cast<const double2*>(&vb[0][h]);
        const double2 v23 = *reinterpret_cast<const double2*>(&vb[0][h + 2]);
#pragma unroll
        for (int s = 0; s < ROWS; ++s) {
          a[s] -= W[s][h] * v01.x;
          a[s] -= W[s][h + 1] * v01.y;
          a[s] -= W[s][h + 2] * v23.x;
          a[s] -= W[s][h + 3] * v23.y;
        }
      }
    double v = -1.0;
    int idx = n;
#pragma unroll
    for (int s = 0; s < ROWS; ++s) {
      const int r = tid + s * BK_PT;
      if (r == k) s_akk = a[s];
      if (r > k && r < n) {
        const double av = fabs(a[s]);
        if (av > v || (av == v && r < idx)) { v = av; idx = r; }
      }
    }
    double colmax;
    int imax;
    bk_argmax<BK_PT>(v, idx, sv, si, colmax, imax);
    if (colmax < 0.0) { colmax = 0.0; imax = k; }
    const double akk = s_akk, absakk = fabs(akk);
    int kstep = 1, kp = k;
    bool zero_piv = false, use_c = false;
    const bool bad = !(absakk == absakk) || !(colmax == colmax);
    if (bad) { fail = 1; break; }
    if (fmax(absakk, colmax) == 0.0) {
      zero_piv = true;
    } else if (absakk < alpha * colmax) {
      // column imax of the trailing matrix, brought up to date the same way
#pragma unroll
      for (int s = 0; s < ROWS; ++s) {
        const int r = tid + s * BK_PT;
        if (r < k || r >= n) c[s] = 0.0;
        else if (r < imax) c[s] = A[imax + static_cast<i64>(r) * ld];
        else c[s] = A[r + static_cast<i64>(imax) * ld];
      }
      publish_lrow(imax, 1);
      __syncthreads();
      double rv = 0.0;
#pragma unroll
      for (int h = 0; h < NBP; h += 4)
        if (h < j) {
          const double2 v01 = *reinterpret_cast<const double2*>(&vb[1][h]);
          const double2 v23 = *reinterpret_cast<const double2*>(&vb[1][h + 2]);
#pragma unroll
          for (int s = 0; s < ROWS; ++s) {
            c[s] -= W[s][h] * v01.x;
            c[s] -= W[s][h + 1] * v01.y;
            c[s] -= W[s][h + 2] * v23.x;
            c[s] -= W[s][h + 3] * v23.y;
          }
        }
#pragma unroll
      for (int s = 0; s < ROWS; ++s) {
        const int r = tid + s * BK_PT;
        if (r == imax) s_cii = c[s];
        else if (r >= k && r < n) rv = fmax(rv, fabs(c[s]));
      }
      double rowmax;
      int dummy;
      bk_argmax<BK_PT>(rv, tid, sv, si, rowmax, dummy);
      const double aii = fabs(s_cii);
      if (absakk >= alpha * colmax * (colmax / rowmax)) kp = k;
      else if (aii >= alpha * rowmax) { kp = imax; use_c = true; }
      else { kp = imax; kstep = 2; }
    }
    if (kstep == 2 && j == NBP - 1) break;       // no room for the second column: the next panel starts here
    const int kk = k + kstep - 1;
    const bool swapped = !zero_piv && kp != kk;
#pragma unroll
    for (int s = 0; s < ROWS; ++s)
      if (kstep == 2 && tid + s * BK_PT == kp) s_off = a[s];      // a[kp, k]: the off-diagonal of the 2x2 block
    if (swapped) {
      // symmetric interchange of rows / columns kk and kp: trailing matrix, the panel's finished columns,
      // registers; the columns of earlier panels follow in the update kernel
      for (int i = kp + 1 + tid; i < n; i += BK_PT) {
        const double t = A[i + static_cast<i64>(kk) * ld];
        A[i + static_cast<i64>(kk) * ld] = A[i + static_cast<i64>(kp) * ld];
        A[i + static_cast<i64>(kp) * ld] = t;
      }
      for (int q = kk + 1 + tid; q < kp; q += BK_PT) {
        const double t = A[q + static_cast<i64>(kk) * ld];
        A[q + static_cast<i64>(kk) * ld] = A[kp + static_cast<i64>(q) * ld];
        A[kp + static_cast<i64>(q) * ld] = t;
      }
      if (tid == 0) {
        const double t = A[kk + static_cast<i64>(kk) * ld];
        A[kk + static_cast<i64>(kk) * ld] = A[kp + static_cast<i64>(kp) * ld];
        A[kp + static_cast<i64>(kp) * ld] = t;
        swp[2 * nswp] = kk;
        swp[2 * nswp + 1] = kp;
        nswp += 1;
      }
#pragma unroll
      for (int s = 0; s < ROWS; ++s) {
        const int r = tid + s * BK_PT;
        if (r == kk || r == kp) {
          const int slot = r == kk ? 0 : 1;
#pragma unroll
          for (int i = 0; i < NBP; ++i) xw[slot][i] = W[s][i];
          xw[slot][NBP] = a[s];
          xw[slot][NBP + 1] = c[s];
        }
      }
      bk_barrier_lds_only();                   // the swap's global stores drain under the work below
#pragma unroll
      for (int s = 0; s < ROWS; ++s) {
        const int r = tid + s * BK_PT;
        if (r == kk || r == kp) {
          const int slot = r == kk ? 1 : 0;
#pragma unroll
          for (int i = 0; i < NBP; ++i) W[s][i] = xw[slot][i];
          a[s] = xw[slot][NBP];
          c[s] = xw[slot][NBP + 1];
        }
      }
    }
    bk_barrier_lds_only();
    if (kstep == 1) {
      double d = zero_piv ? 1e-20 : (use_c ? s_cii : akk);
      if (zero_piv) nzero += 1;
      else {
        if (d < 0.0) nneg += 1;
        if (fabs(d) < 1e-300) nzero += 1;
      }
      const double inv = bk_rcp(d);
#pragma unroll
      for (int s = 0; s < ROWS; ++s) {
        const int r = tid + s * BK_PT;
        const double pv = use_c ? c[s] : a[s];
#pragma unroll
        for (int i = 0; i < NBP; ++i)
          if (i == j) W[s][i] = (r > k && r < n) ? pv : 0.0;
      }
      if (tid == 0) {
        draw[j][0] = d;
        piv_out[j] = kp + 1;
        dk[j] = 0;
        dinf[j][0] = inv;
      }
    } else {
      const double off = s_off, cii = s_cii;
      const double offinv = bk_rcp(off);
      const double d11 = cii * offinv, d22 = akk * offinv;
      const double d21 = bk_rcp(d11 * d22 - 1.0) * offinv;
      nneg += 1;
#pragma unroll
      for (int s = 0; s < ROWS; ++s) {
        const int r = tid + s * BK_PT;
        const bool below = r > k + 1 && r < n;
#pragma unroll
        for (int i = 0; i < NBP; ++i) {
          if (i == j) W[s][i] = below ? a[s] : 0.0;
          if (i == j + 1) W[s][i] = below ? c[s] : 0.0;
        }
      }
      if (tid == 0) {
        draw[j][0] = akk; draw[j][1] = off;
        draw[j + 1][0] = cii;
        piv_out[j] = -(kp + 1);
        piv_out[j + 1] = -(kp + 1);
        dk[j] = 1;
        dk[j + 1] = 2;
        dinf[j][0] = d11; dinf[j][1] = d22; dinf[j][2] = d21;
        dinf[j + 1][0] = d22; dinf[j + 1][1] = d11; dinf[j + 1][2] = d21;   // the second column's form of the same block
      }
    }
    j += kstep;
    if (swapped) __syncthreads();              // the next column's loads may touch interchanged entries
    else bk_barrier_lds_only();
  }
  const int kend = k0 + j;
  // the panel's columns go to memory once, here: multipliers from W and the pivot blocks (the stores of a
  // column step would sit in front of the next step's loads: one in-order vmcnt on this ISA)
#pragma unroll
  for (int s = 0; s < ROWS; ++s) {
    const int r = tid + s * BK_PT;
    if (r < n) {
#pragma unroll
      for (int i = 0; i < NBP; ++i)
        if (i < j) {
          const int kc = k0 + i;
          if (dk[i] == 0) {
            if (r > kc) A[r + static_cast<i64>(kc) * ld] = W[s][i] * dinf[i][0];
          } else if (dk[i] == 1 && i + 1 < NBP) {
            if (r > kc + 1) {
              const double d11 = dinf[i][0], d22 = dinf[i][1], d21 = dinf[i][2];
              A[r + static_cast<i64>(kc) * ld] = d21 * (d11 * W[s][i] - W[s][i + 1]);
              A[r + static_cast<i64>(kc + 1) * ld] = d21 * (d22 * W[s][i + 1] - W[s][i]);
            }
          }
        }
    }
    if (r >= kend && r < n) {
#pragma unroll
      for (int i = 0; i < NBP; ++i) Wg[r + static_cast<i64>(i) * ldw] = (i < j) ? W[s][i] : 0.0;
    }
  }
  if (tid < j) {
    const int kc = k0 + tid;
    ipiv[kc] = piv_out[tid];
    A[kc + static_cast<i64>(kc) * ld] = draw[tid][0];
    if (dk[tid] == 1) A[kc + 1 + static_cast<i64>(kc) * ld] = draw[tid][1];
  }
  if (tid == 0) {
    st->kp = k0;
    st->kstep = j;
    st->k = fail ? n : kend;
    st->pending = j > 0 ? 1 : 0;
    st->nneg += nneg;
    st->nzero += nzero;
    if (fail) st->fail = 1;
    swaps->count = nswp;
    for (int q = 0; q < 2 * nswp; ++q) swaps->rows[q] = swp[q];
  }
}


}  // namespace dnlp
