// Wavefront-wide reductions on DPP (row shifts inside the four 16-lane rows, then row_bcast:15 / row_bcast:31
// across them) with the result broadcast to every lane through an SGPR (v_readlane of lane 63).
//
// The __shfl_xor butterfly these replace compiles to ds_bpermute_b32: two per double and step, twelve per
// reduction, each a dependent ~100-cycle trip through the LDS crossbar (~1 500 cycles per reduction; the
// in-kernel interior-point loop did ~70 of them per iteration).  A DPP step is two v_mov_b32_dpp and the
// operation: ~200 cycles for the whole reduction.  All 64 lanes must be active at the call.
//
// Free of standard-library includes: the text also travels inside the library (wave_ops_src.inc) into the per-template
// kernels of wave_codegen.h.
#pragma once
#include <hip/hip_runtime.h>

namespace dnlp {

// barrier of ONE wavefront: its LDS stores are visible to its own lanes afterwards (no workgroup barrier)
__device__ inline void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// lane l's value of v (l uniform), to every lane
__device__ inline double readlane_d(double v, int l) {
  l = __builtin_amdgcn_readfirstlane(l);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}

// lane i <- lane (i - shift) of its row, or `keep` where there is no such lane / the row is masked off
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_shift_f64(double v, double keep) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(keep), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(keep), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ inline double wave_lane63(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// row_shr:1,2,4,8 = inclusive scan of a row (lane 15 holds the row's total), row_bcast:15 adds it into the
// next row (rows 1, 3), row_bcast:31 adds lane 31 into rows 2, 3: lane 63 holds the wavefront's total
__device__ inline double wave_all_sum(double v) {
  v += dpp_shift_f64<0x111, 0xf>(v, 0.0);
  v += dpp_shift_f64<0x112, 0xf>(v, 0.0);
  v += dpp_shift_f64<0x114, 0xf>(v, 0.0);
  v += dpp_shift_f64<0x118, 0xf>(v, 0.0);
  v += dpp_shift_f64<0x142, 0xa>(v, 0.0);
  v += dpp_shift_f64<0x143, 0xc>(v, 0.0);
  return wave_lane63(v);
}
__device__ inline double wave_all_max(double v) {
  v = fmax(v, dpp_shift_f64<0x111, 0xf>(v, v));
  v = fmax(v, dpp_shift_f64<0x112, 0xf>(v, v));
  v = fmax(v, dpp_shift_f64<0x114, 0xf>(v, v));
  v = fmax(v, dpp_shift_f64<0x118, 0xf>(v, v));
  v = fmax(v, dpp_shift_f64<0x142, 0xa>(v, v));
  v = fmax(v, dpp_shift_f64<0x143, 0xc>(v, v));
  return wave_lane63(v);
}
// N reductions side by side, step by step: one reduction is a chain of six dependent DPP steps (each two v_mov_b32_dpp, the
// operation and the wait states between them: ~180 cycles of which a wavefront alone on its SIMD spends most waiting);
// N independent chains interleaved fill each other's wait states.  Every value goes through the same tree as in
// wave_all_sum / wave_all_max: the same bits.
template <int N>
__device__ inline void wave_all_sum_n(double (&v)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += dpp_shift_f64<0x111, 0xf>(v[k], 0.0);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += dpp_shift_f64<0x112, 0xf>(v[k], 0.0);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += dpp_shift_f64<0x114, 0xf>(v[k], 0.0);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += dpp_shift_f64<0x118, 0xf>(v[k], 0.0);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += dpp_shift_f64<0x142, 0xa>(v[k], 0.0);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] += dpp_shift_f64<0x143, 0xc>(v[k], 0.0);
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = wave_lane63(v[k]);
}
template <int N>
__device__ inline void wave_all_max_n(double (&v)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = fmax(v[k], dpp_shift_f64<0x111, 0xf>(v[k], v[k]));
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = fmax(v[k], dpp_shift_f64<0x112, 0xf>(v[k], v[k]));
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = fmax(v[k], dpp_shift_f64<0x114, 0xf>(v[k], v[k]));
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = fmax(v[k], dpp_shift_f64<0x118, 0xf>(v[k], v[k]));
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = fmax(v[k], dpp_shift_f64<0x142, 0xa>(v[k], v[k]));
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = fmax(v[k], dpp_shift_f64<0x143, 0xc>(v[k], v[k]));
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = wave_lane63(v[k]);
}
template <int CTRL, int ROW_MASK>
__device__ inline int dpp_min_i32(int v) {
  return min(v, __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xf, false));
}
__device__ inline int wave_all_min(int v) {
  v = dpp_min_i32<0x111, 0xf>(v);
  v = dpp_min_i32<0x112, 0xf>(v);
  v = dpp_min_i32<0x114, 0xf>(v);
  v = dpp_min_i32<0x118, 0xf>(v);
  v = dpp_min_i32<0x142, 0xa>(v);
  v = dpp_min_i32<0x143, 0xc>(v);
  return __builtin_amdgcn_readlane(v, 63);
}
// argmax of v over the wavefront, the smallest index among equal maxima (IDAMAX); every lane gets both
__device__ inline void wave_all_argmax(double& v, int& idx) {
  const double m = wave_all_max(v);
  idx = wave_all_min(v == m ? idx : 0x7fffffff);
  v = m;
}

}  // namespace dnlp
