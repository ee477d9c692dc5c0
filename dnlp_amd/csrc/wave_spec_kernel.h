// Template-specialised batch solver, part 4: the kernel that wave_codegen.h compiles PER TEMPLATE at run time (hiprtc).
// This text is never included by the library's own translation unit: it travels inside the library
// (wave_spec_kernel_src.inc) and is appended to the generated constants (namespace wspec), wave_ops.h and wave_ipm.h.
//
// What the per-template kernel has over wave_batch.h's <NW, state in LDS, plan in LDS> instantiations: every size, table
// offset and vector place of the template is a literal (wave_ipm.h WK / WT / WV), the plan and the wavefronts' shares are
// static LDS arrays at fixed addresses — no pointer loads from the state record, loops over N, m <= 64 entries are one
// predicated trip, ds_read / ds_write carry their array's offset as the immediate.  Role in the reference: the serial
// re-solve loop cvxpy/problems/problem.py:1256-1269 -> ipopt_nlpif.py:140-170, as in wave_batch.h.
#pragma once

namespace dnlp {

constexpr int kWaveSpecRecBytes = (static_cast<int>(sizeof(WStateT<WLdsD, WLdsI>)) + 15) & ~15;
static_assert(kWaveSpecRecBytes <= wspec::kRecBytesMax, "the host sized the launch for a smaller state record");

struct __attribute__((aligned(16))) WaveSpecShare {
  char rec[kWaveSpecRecBytes];                  // the wavefront's state record (WStateT)
  double vec[wspec::kStateDoubles];             // its vectors (wave_ipm.h layout: wspec::v_* are offsets into this array)
};

// the part of the plan block the kernel still reads (WaveHdr::keep_gen ints: the level machinery's tables are replaced by
// generated phases), narrowed to 16 bits and staged once per workgroup, and the wavefronts' shares: static LDS, fixed addresses.
// wspec::kTabGlobal: the tables stay in GLOBAL memory (the block's 16-bit copy, the work tables as uploaded) — the form of a
// template whose shares leave no room for them (circle packing n = 10: two shares of 74 KB; its tables are 44 KB): a table
// word then comes through L1 / L2, and a compute unit holds twice the instances.
__shared__ __attribute__((aligned(16))) int16_t g_wspec_plan[wspec::kTabGlobal ? 8 : ((wspec::kPlanInts + 7) & ~7)];
__shared__ __attribute__((aligned(16))) unsigned g_wspec_gen[wspec::kTabGlobal ? 4 : ((wspec::kGenWords + 3) & ~3)];      // work tables of the generated phases (wave_gen.h)
__shared__ WGlbI16* g_wspec_plan_g;
__shared__ DNLP_WGLB const unsigned* g_wspec_gen_g;
__shared__ WaveSpecShare g_wspec_share[wspec::kNW];
__shared__ int g_wspec_inst[wspec::kNW];

template <bool B, class X, class Y> struct WSel { typedef X type; };
template <class X, class Y> struct WSel<false, X, Y> { typedef Y type; };

struct WaveLanesSpec {
  typedef WLdsD D;
  typedef typename WSel<wspec::kTabGlobal, WGlbI16, WLdsI>::type I;
  typedef typename WSel<wspec::kTabGlobal, DNLP_WGLB const unsigned*, DNLP_WLDS const unsigned*>::type G;
  static constexpr int lanes = 64;
  static constexpr bool hoist = false;
  static constexpr bool lds_generic = false;      // (vectors are LDS-typed pointers: wave_ipm.h ldl_solve)
  __device__ static int lane() { return static_cast<int>(threadIdx.x & 63u); }
  __device__ static void sync() { wave_sync(); }
  __device__ static double sum(double v) { return wave_all_sum(v); }
  __device__ static double vmax(double v) { return wave_all_max(v); }
  // several reductions side by side (wave_ops.h: the chains fill each other's wait states; the same bits as one by one)
  template <int N> __device__ static void sum_n(double (&v)[N]) { wave_all_sum_n<N>(v); }
  template <int N> __device__ static void vmax_n(double (&v)[N]) { wave_all_max_n<N>(v); }
  __device__ static double now() { return now_sec(); }
  __device__ static int tab_load(I*, int) { return 0; }
  __device__ static int tab_at(I* tab, int, int idx, int) { return static_cast<int>(tab[idx]); }
  __device__ static int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
  template <int SL> __device__ static double row_get(const double (&a)[SL], int row) { return readlane_d(a[0], row); }
  // a vector of the instance: the wavefront's share (found from its state record) + a literal offset
  template <class WS> __device__ static D* vec(WS* S, int off) {
    return (D*)((DNLP_WLDS char*)S + kWaveSpecRecBytes) + off;
  }
  // a table of the plan block
  __device__ static I* tab(int off) {
    if constexpr (wspec::kTabGlobal) return (I*)g_wspec_plan_g + off;
    else return (I*)g_wspec_plan + off;
  }
  __device__ static G gtab() {
    if constexpr (wspec::kTabGlobal) return (G)g_wspec_gen_g;
    else return (G)g_wspec_gen;
  }
};

}  // namespace dnlp

extern "C" __global__ void __launch_bounds__(64 * wspec::kNW) dnlp_wave_spec_kernel(dnlp::WaveArgs a) {
  using namespace dnlp;
  using P = WaveLanesSpec;
  using W = WaveIpm<P>;
  using WD = typename P::D;
  const int wave = static_cast<int>(threadIdx.x >> 6), lane = static_cast<int>(threadIdx.x & 63u);
  if constexpr (wspec::kTabGlobal) {
    if (threadIdx.x == 0) { g_wspec_plan_g = (WGlbI16*)a.blk16; g_wspec_gen_g = (DNLP_WGLB const unsigned*)a.gen; }
  } else {
    for (int k = static_cast<int>(threadIdx.x); k < wspec::kPlanInts; k += static_cast<int>(blockDim.x)) g_wspec_plan[k] = static_cast<int16_t>(a.blk[k]);
    for (int k = static_cast<int>(threadIdx.x); k < wspec::kGenWords; k += static_cast<int>(blockDim.x)) g_wspec_gen[k] = a.gen[k];
  }
  __syncthreads();                     // the only workgroup barrier of the kernel
  typename W::WS* S = (typename W::WS*)g_wspec_share[wave].rec;
  WD* base = (WD*)g_wspec_share[wave].vec;
  constexpr int N = wspec::k_N, m = wspec::k_m;
  while (true) {
    if (lane == 0) {
      const int k = atomicAdd(a.next, 1);
      g_wspec_inst[wave] = (k < a.batch && a.order) ? a.order[k] : k;
    }
    wave_sync();
    const int inst = g_wspec_inst[wave];
    wave_sync();
    if (inst >= a.batch) break;
    for (int k = lane; k < wspec::kStateDoubles; k += 64) base[k] = 0.0;
    S->row = (WG*)(a.rows + static_cast<i64>(inst) * a.row_doubles);
    S->park = a.park + (static_cast<i64>(blockIdx.x) * wspec::kNW + wave) * a.park_doubles;
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
      const WD *xx = WV(x), *yy = WV(y), *sg = WV(sg), *zl = WV(zL), *zu = WV(zU);
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
