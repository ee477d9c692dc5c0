// Template-specialised batch solver, part 8: ONE WORKGROUP per instance, for templates whose state exceeds a compute
// unit's LDS (power flow: 617 KB, path planning: 377 KB of vectors — KKT order ~1 700).  Compiled per template at run time
// like wave_spec_kernel.h (this text is never included by the library's own translation unit; it travels as
// wave_wg_kernel_src.inc), from the same algorithm text and the same generated phases, over a lane policy of
// wspec::kNW x 64 lanes:
//   * the instance's vectors live in the workgroup's slab of global memory (L2 / Infinity Cache) — except the hottest, which
//     the host maps into the compute unit's LDS as far as they fit (wspec::kLds*: the factor's values and the arrays every
//     linear solve runs on; vectors are reached through generic pointers, the compiler types an access whose place is a
//     literal) — the plan prefix and the generated phases' work tables in global memory (read-only, shared by every workgroup);
//   * every wavefront keeps its OWN state record in LDS and runs the scalar control flow redundantly — the wavefronts
//     agree on every decision because every reduction (sum / vmax: DPP inside a wavefront, a fixed-order sum of the
//     wavefronts' partials through LDS) and the clock hand all of them the same value;
//   * a phase of the generated LDL^T / residual / CSR code is one task per lane across ALL wavefronts (wave_gen.h with
//     256 lanes per slot) between two workgroup barriers; the wide forms and the dense tail run on the first wavefront.
// The generic batch kernel (batch.h) gives such an instance four wavefronts as well, but interprets: 1.8 ms per interior-point
// iteration of power flow on MI355X.  Reference role: cvxpy/problems/problem.py:1256-1269 -> ipopt_nlpif.py:140-170 for
// examples/nlp_examples/power_flow.ipynb, path_planning.ipynb.
#pragma once

namespace dnlp {

constexpr int kWaveWgRecBytes = (static_cast<int>(sizeof(WStateT<double, WGlbI>)) + 15) & ~15;

struct __attribute__((aligned(16))) WaveWgRec { char rec[kWaveWgRecBytes]; };

__shared__ WaveWgRec g_wg_rec[wspec::kNW];                 // a state record per wavefront
__shared__ WGlbD* g_wg_vbase;                              // the workgroup's slab: the instance's vectors (wspec::v_* offsets)
__shared__ __attribute__((aligned(16))) double g_wg_lds[wspec::kLdsDoubles > 0 ? wspec::kLdsDoubles : 2];      // ... the ranges [kLds0a, kLds0b) and [kLds1a, kLds1b) of them
__shared__ WGlbI* g_wg_plan;                               // the plan block (32-bit, global memory)
__shared__ DNLP_WGLB const unsigned* g_wg_gen;             // work tables of the generated phases
__shared__ double g_wg_red[2][wspec::kNW];                 // the wavefronts' partials of a reduction, two sets used in turn (one barrier per reduction)
__shared__ double g_wg_redn[2][4][wspec::kNW];             // ... of up to four reductions side by side (sum_n / vmax_n)
__shared__ unsigned g_wg_turn[wspec::kNW];                 // ... which set a wavefront's next reduction writes (every wavefront counts alike)
__shared__ __attribute__((aligned(16))) unsigned g_wg_stage[wspec::kStageWords > 0 ? wspec::kStageWords : 4];      // the table the next narrow phase reads (wave_gen.h: staging)
__shared__ __attribute__((aligned(16))) double g_wg_wwin[wspec::kWwin > 0 ? wspec::kWwin : 2];      // a level's unscaled rows / its products (wave_gen.h: windows)
__shared__ __attribute__((aligned(16))) double g_wg_swin[wspec::kSwin > 0 ? wspec::kSwin : 2];
__shared__ double g_wg_clock;
__shared__ int g_wg_inst;

struct WaveLanesWG {
  typedef double D;                  // (generic: a vector is in the slab or in LDS)
  typedef WGlbI I;
  typedef DNLP_WGLB const unsigned* G;
  static constexpr int lanes = 64 * wspec::kNW;
  static constexpr bool hoist = true;         // (vectors in global memory: wave_ipm.h quality())
  static constexpr bool lds_generic = true;   // (vectors are generic pointers into LDS or the slab: wave_ipm.h ldl_solve types the solve's right-hand sides)
  __device__ static int lane() { return static_cast<int>(threadIdx.x); }
  __device__ static void sync() { __syncthreads(); }
  // every wavefront gets the same bits: its own DPP total, then the wavefronts' totals added in wavefront order
  // (the partials go to one of two sets in turn: a wavefront that has passed the barrier of reduction k may already write
  //  the set of reduction k + 1 while a slower one still reads set k — and nobody writes set k again before everybody has
  //  passed barrier k + 1)
  __device__ static unsigned turn() {
    const unsigned w = threadIdx.x >> 6;
    const unsigned t = g_wg_turn[w];
    if ((threadIdx.x & 63u) == 0u) g_wg_turn[w] = t ^ 1u;
    return t & 1u;
  }
  __device__ static double sum(double v) {
    const double t = wave_all_sum(v);
    const unsigned s = turn();
    if ((threadIdx.x & 63u) == 0u) g_wg_red[s][threadIdx.x >> 6] = t;
    __syncthreads();
    double r = g_wg_red[s][0];
#pragma unroll
    for (int k = 1; k < wspec::kNW; ++k) r += g_wg_red[s][k];
    return r;
  }
  __device__ static double vmax(double v) {
    const double t = wave_all_max(v);
    const unsigned s = turn();
    if ((threadIdx.x & 63u) == 0u) g_wg_red[s][threadIdx.x >> 6] = t;
    __syncthreads();
    double r = g_wg_red[s][0];
#pragma unroll
    for (int k = 1; k < wspec::kNW; ++k) r = fmax(r, g_wg_red[s][k]);
    return r;
  }
  // several reductions with ONE barrier (the DPP chains side by side: wave_ops.h)
  template <int N> __device__ static void sum_n(double (&v)[N]) {
    wave_all_sum_n<N>(v);
    const unsigned s = turn();
    if ((threadIdx.x & 63u) == 0u) {
#pragma unroll
      for (int q = 0; q < N; ++q) g_wg_redn[s][q][threadIdx.x >> 6] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < N; ++q) {
      double r = g_wg_redn[s][q][0];
#pragma unroll
      for (int k = 1; k < wspec::kNW; ++k) r += g_wg_redn[s][q][k];
      v[q] = r;
    }
  }
  template <int N> __device__ static void vmax_n(double (&v)[N]) {
    wave_all_max_n<N>(v);
    const unsigned s = turn();
    if ((threadIdx.x & 63u) == 0u) {
#pragma unroll
      for (int q = 0; q < N; ++q) g_wg_redn[s][q][threadIdx.x >> 6] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < N; ++q) {
      double r = g_wg_redn[s][q][0];
#pragma unroll
      for (int k = 1; k < wspec::kNW; ++k) r = fmax(r, g_wg_redn[s][q][k]);
      v[q] = r;
    }
  }
  // one clock for the workgroup (a time limit must stop every wavefront in the same iteration)
  __device__ static double now() {
    if (threadIdx.x == 0) g_wg_clock = now_sec();
    __syncthreads();
    const double t = g_wg_clock;
    __syncthreads();
    return t;
  }
  __device__ static int tab_load(I*, int) { return 0; }
  __device__ static int tab_at(I* tab, int, int idx, int) { return static_cast<int>(tab[idx]); }
  __device__ static int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
  template <int SL> __device__ static double row_get(const double (&a)[SL], int row) { return readlane_d(a[0], row); }
  template <class WS> __device__ static D* vec(WS*, int off) {
    if (off >= wspec::kLds0a && off < wspec::kLds0b) return (D*)(g_wg_lds + (off - wspec::kLds0a));
    if (off >= wspec::kLds1a && off < wspec::kLds1b) return (D*)(g_wg_lds + (wspec::kLds0b - wspec::kLds0a) + (off - wspec::kLds1a));
    return (D*)(g_wg_vbase + off);
  }
  // ... with its address space in the type (a literal place: wave_gen_rt.h tvec)
  template <int OFF> __device__ static auto vec_typed() {
    if constexpr (OFF >= wspec::kLds0a && OFF < wspec::kLds0b) return (DNLP_WLDS double*)(g_wg_lds + (OFF - wspec::kLds0a));
    else if constexpr (OFF >= wspec::kLds1a && OFF < wspec::kLds1b) return (DNLP_WLDS double*)(g_wg_lds + (wspec::kLds0b - wspec::kLds0a) + (OFF - wspec::kLds1a));
    else return (WGlbD*)(g_wg_vbase + OFF);
  }
  __device__ static I* tab(int off) { return g_wg_plan + off; }
  __device__ static DNLP_WLDS double* wwin() { return (DNLP_WLDS double*)g_wg_wwin; }
  __device__ static DNLP_WLDS double* swin() { return (DNLP_WLDS double*)g_wg_swin; }
  __device__ static DNLP_WLDS unsigned* stage() { return (DNLP_WLDS unsigned*)g_wg_stage; }      // (wave_gen_rt.h WG_STAGE_PUT / WG_SG)
  __device__ static G gtab() { return g_wg_gen; }
};

}  // namespace dnlp

// (kWgBound: the register budget — a bound of 512 threads holds the kernel and its functions to 256 registers per lane, so that
//  two workgroups of four wavefronts share a compute unit: these instances wait for memory, and a second workgroup fills the wait)
//
// The body is a function of its own, not inlined into the entry point, which only stages the arguments in LDS and calls it.
// Reason: the compiler of ROCm 7.0 (the one a process that imported PyTorch compiles with: torch ships its own libhiprtc /
// libamd_comgr) gives the functions of this module half the budget as accumulation registers (128 + 128) but let an entry
// point with a body of its own take 140 ordinary ones on top: a 268-register kernel under a bound of 256, whose first launch
// aborts the queue (INVALID_ISA).  An entry point that only copies and calls has next to nothing to allocate.
__shared__ __attribute__((aligned(16))) unsigned g_wg_args[sizeof(dnlp::WaveArgs) / 4];      // (raw words: WaveArgs has member initialisers)

__device__ __attribute__((noinline)) static void dnlp_wave_wg_main() {
  using namespace dnlp;
  const DNLP_WLDS WaveArgs& a = *(const DNLP_WLDS WaveArgs*)g_wg_args;
  using P = WaveLanesWG;
  using W = WaveIpm<P>;
  using WD = typename P::D;
  const int wave = static_cast<int>(threadIdx.x >> 6), tid = static_cast<int>(threadIdx.x);
  WGlbD* base = (WGlbD*)(a.state + static_cast<size_t>(blockIdx.x) * static_cast<size_t>(wspec::kStateDoubles));
  if (tid == 0) { g_wg_vbase = base; g_wg_plan = (WGlbI*)a.blk; g_wg_gen = (DNLP_WGLB const unsigned*)a.gen; }
  if ((tid & 63) == 0) g_wg_turn[wave] = 0u;
  __syncthreads();
  typename W::WS* S = (typename W::WS*)g_wg_rec[wave].rec;
  constexpr int N = wspec::k_N, m = wspec::k_m;
  while (true) {
    if (tid == 0) {
      const int k = atomicAdd(a.next, 1);
      g_wg_inst = (k < a.batch && a.order) ? a.order[k] : k;
    }
    __syncthreads();
    const int inst = g_wg_inst;
    __syncthreads();
    if (inst >= a.batch) break;
    for (int k = tid; k < wspec::kStateDoubles; k += P::lanes) base[k] = 0.0;
    for (int k = tid; k < wspec::kLdsDoubles; k += P::lanes) g_wg_lds[k] = 0.0;
    S->row = (WG*)(a.rows + static_cast<i64>(inst) * a.row_doubles);
    S->park = a.park + static_cast<i64>(blockIdx.x) * a.park_doubles;
    S->ws_g = a.ws_g ? a.ws_g + static_cast<i64>(inst) * m : nullptr;
    S->ws_l = a.ws_l ? a.ws_l + static_cast<i64>(inst) * N : nullptr;
    S->ws_u = a.ws_u ? a.ws_u + static_cast<i64>(inst) * N : nullptr;
    S->fallback_max_n = a.fallback_max_n;
    S->opt = a.opt;
    S->factorizations = 0;
#ifdef DNLP_WAVE_PROF
    for (int k = 0; k < kWaveProfSlots; ++k) S->prof[k] = 0ull;
#endif
    __syncthreads();
    const int st = W::solve(S);
    const bool have = S->initialized && st != kWaveNeedsGeneric;
    const double sf = have ? S->sf : 1.0;
    {
      const WD *xx = WV(x), *yy = WV(y), *sg = WV(sg), *zl = WV(zL), *zu = WV(zU);
      double* xo = a.x_out + static_cast<i64>(inst) * N;
      for (int j = tid; j < N; j += P::lanes) {
        xo[j] = have ? xx[j] : 0.0;
        if (a.zl_out) a.zl_out[static_cast<i64>(inst) * N + j] = have ? zl[j] / sf : 0.0;
        if (a.zu_out) a.zu_out[static_cast<i64>(inst) * N + j] = have ? zu[j] / sf : 0.0;
      }
      if (a.multg_out)
        for (int i = tid; i < m; i += P::lanes) a.multg_out[static_cast<i64>(inst) * m + i] = have ? yy[i] * sg[i] / sf : 0.0;
    }
    if (tid == 0) {
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
    __syncthreads();
  }
}

extern "C" __global__ void __launch_bounds__(wspec::kWgBound) dnlp_wave_wg_kernel(dnlp::WaveArgs a) {
  static_assert(sizeof(dnlp::WaveArgs) % 4 == 0, "copied word by word");
  for (unsigned k = threadIdx.x; k < sizeof(dnlp::WaveArgs) / 4; k += blockDim.x) g_wg_args[k] = ((const unsigned*)&a)[k];
  __syncthreads();
  dnlp_wave_wg_main();
}
