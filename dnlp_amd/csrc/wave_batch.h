// Template-specialised batch solver, part 3: the kernel.  One wavefront per instance, NW wavefronts per workgroup; the
// plan block (wave_plan.h) is staged once per workgroup, every wavefront claims instance after instance from the launch's
// queue and runs wave_ipm.h on it.  Where the instance's vectors and the plan live is a compile-time choice per launch:
//   <state in LDS, plan in LDS>    small templates (localization: 15 KB of plan + 4 x 33 KB of state per compute unit)
//   <state in LDS, plan global>    the plan does not fit beside one instance's state (read through L1 / L2)
//   <state global, plan global>    large templates (power flow: 655 KB of state per instance)
// The host side (batch.h BatchRunner::solve_wave) picks the richest form that fits 160 KB.
#pragma once
#include <type_traits>

#include "exec_block.h"
#include "wave_args.h"
#include "wave_ipm.h"
#include "wave_ops.h"

namespace dnlp {

// (TWO_PER_SIMD only tells the instantiations apart: the functions of wave_ipm.h called from a launch with more than four
//  wavefronts per workgroup have 256 registers and spill to scratch memory; those of a launch with up to four have 512
//  and spill to the accumulation registers — compiled once for both, the tighter budget would apply to all: measured 81.8
//  -> 84.8 us per iteration for the four-wavefront form)
template <bool STATE_LDS, bool PLAN_LDS, bool TWO_PER_SIMD = false>
struct WaveLanesT {
  typedef typename std::conditional<STATE_LDS, WLdsD, WGlbD>::type D;
  // (a plan left in global memory beside a state in LDS is read from its 16-bit copy: a state that fits LDS has no table
  //  entry beyond 16 bits, and half the bytes is half the L1 / L2 traffic of the index loads)
  typedef typename std::conditional<PLAN_LDS, WLdsI, typename std::conditional<STATE_LDS, WGlbI16, WGlbI>::type>::type I;
  static constexpr int lanes = 64;
  static constexpr bool hoist = false;        // (wave_ipm.h quality(): which form of the elementwise loops)
  __device__ static int lane() { return static_cast<int>(threadIdx.x & 63u); }
  __device__ static void sync() { wave_sync(); }
  __device__ static double sum(double v) { return wave_all_sum(v); }
  __device__ static double vmax(double v) { return wave_all_max(v); }
  // several reductions side by side (wave_ops.h: the chains fill each other's wait states; the same bits as one by one)
  template <int N> __device__ static void sum_n(double (&v)[N]) { wave_all_sum_n<N>(v); }
  template <int N> __device__ static void vmax_n(double (&v)[N]) { wave_all_max_n<N>(v); }
  __device__ static double now() { return now_sec(); }
  // per-level bounds (measured on MI355X: keeping the table one entry per lane in a register and reading it back with
  // v_readlane made the substitutions 17 % SLOWER than these plain uniform LDS loads — 46.1 -> 53.8 k cycles per iteration)
  __device__ static int tab_load(I*, int) { return 0; }
  __device__ static int tab_at(I* tab, int, int idx, int) { return static_cast<int>(tab[idx]); }
  // a value every lane holds alike, as a scalar register
  __device__ static int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
  // the dense tail keeps one row per lane (wave_ipm.h kTailSlots = 1): the entry of row `row` comes through v_readlane
  template <int SL> __device__ static double row_get(const double (&a)[SL], int row) { return readlane_d(a[0], row); }
};

template <int NW, bool STATE_LDS, bool PLAN_LDS>
__global__ void __launch_bounds__(64 * NW) wave_batch_kernel(WaveArgs a) {
  extern __shared__ __align__(16) char w_lds[];
  using P = WaveLanesT<STATE_LDS, PLAN_LDS, (NW > 4)>;
  using W = WaveIpm<P>;
  using WState = typename W::WState;
  using WD = typename P::D;
  using WI = typename P::I;
  __shared__ __align__(16) char s_state[NW][(sizeof(WState) + 15) & ~static_cast<size_t>(15)];
  __shared__ int s_inst[NW];
  const int wave = static_cast<int>(threadIdx.x >> 6), lane = static_cast<int>(threadIdx.x & 63u);
  char* pool = w_lds;
  WI* blk;
  if constexpr (PLAN_LDS) {
    // (16-bit copy: the host takes this form only for plans whose entries all fit — BatchRunner::wave_prepare)
    DNLP_WLDS int16_t* dst = (DNLP_WLDS int16_t*)pool;
    for (int k = static_cast<int>(threadIdx.x); k < a.blk_ints; k += 64 * NW) dst[k] = static_cast<int16_t>(a.blk[k]);
    blk = (WI*)dst;
    pool += (static_cast<size_t>(a.blk_ints) * 2 + 15) & ~static_cast<size_t>(15);
    __syncthreads();                   // the only workgroup barrier of the kernel
  } else if constexpr (STATE_LDS) {
    blk = (WI*)a.blk16;
  } else {
    blk = (WI*)a.blk;
  }
  WD* base;
  if constexpr (STATE_LDS) base = (WD*)(pool + static_cast<size_t>(wave) * static_cast<size_t>(a.state_doubles) * 8);
  else base = (WD*)(a.state + (static_cast<size_t>(blockIdx.x) * NW + wave) * static_cast<size_t>(a.state_doubles));
  typename W::WS* S = (typename W::WS*)s_state[wave];
  W::layout(S, reinterpret_cast<const WaveHdr*>(a.blk), blk, base);
  wave_sync();
  const int N = S->N, m = S->m;
  while (true) {
    if (lane == 0) {
      const int k = atomicAdd(a.next, 1);
      s_inst[wave] = (k < a.batch && a.order) ? a.order[k] : k;
    }
    wave_sync();
    const int inst = s_inst[wave];
    wave_sync();
    if (inst >= a.batch) break;
    // (the generic spaces hand out zero-filled vectors; nothing below relies on it, but a stale NaN must not travel
    //  from one instance into the next through an entry that a masked pass skips)
    for (i64 k = lane; k < a.state_doubles; k += 64) base[k] = 0.0;
    S->row = (WG*)(a.rows + static_cast<i64>(inst) * a.row_doubles);
    S->park = a.park + (static_cast<i64>(blockIdx.x) * NW + wave) * a.park_doubles;
    S->ws_g = a.ws_g ? a.ws_g + static_cast<i64>(inst) * m : nullptr;
    S->ws_l = a.ws_l ? a.ws_l + static_cast<i64>(inst) * N : nullptr;
    S->ws_u = a.ws_u ? a.ws_u + static_cast<i64>(inst) * N : nullptr;
    S->fallback_max_n = a.fallback_max_n;
    S->opt = a.opt;
    S->factorizations = 0;
#ifdef DNLP_WAVE_PROF
    for (int k = 0; k < kWaveProfSlots; ++k) S->prof[k] = 0ull;
#endif
    wave_sync();
    const int st = W::solve(S);
    const bool have = S->initialized && st != kWaveNeedsGeneric;
    const double sf = have ? S->sf : 1.0;
    {
      const WD *xx = S->x, *yy = S->y, *sg = S->sg, *zl = S->zL, *zu = S->zU;
      double* xo = a.x_out + static_cast<i64>(inst) * N;
      for (int j = lane; j < N; j += 64) {
        xo[j] = have ? xx[j] : 0.0;
        if (a.zl_out) a.zl_out[static_cast<i64>(inst) * N + j] = have ? zl[j] / sf : 0.0;
        if (a.zu_out) a.zu_out[static_cast<i64>(inst) * N + j] = have ? zu[j] / sf : 0.0;
      }
      if (a.multg_out)
        for (int i = lane; i < m; i += 64) a.multg_out[static_cast<i64>(inst) * m + i] = have ? yy[i] * sg[i] / sf : 0.0;
    }
    if (lane == 0) {
#ifdef DNLP_WAVE_PROF
      if (a.prof) { for (int k = 0; k < kWaveProfSlots; ++k) atomicAdd(a.prof + k, S->prof[k]); atomicAdd(a.prof + kWaveProfSlots, static_cast<unsigned long long>(S->iter)); }
#endif
      a.status_out[inst] = st;
      a.iters_out[inst] = S->iter;
      a.obj_out[inst] = have ? S->f / sf : 0.0;
      if (a.nfact_out) a.nfact_out[inst] = S->factorizations;
      if (a.times_out) {
        double* to = a.times_out + 4 * static_cast<i64>(inst);
        to[0] = S->wall; to[1] = 0.0; to[2] = 0.0; to[3] = 0.0;
      }
    }
    wave_sync();
  }
}

}  // namespace dnlp
