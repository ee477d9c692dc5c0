// Batched solve: B instances of one tape structure, one workgroup per instance, the whole
// interior-point loop inside one kernel (exec_block.h).  BASELINE config C5 / SURVEY.md §8e.
//
// Instance data layout handed over by the host language (dnlp_amd/batch.py), `stride` doubles
// per instance, in this order:
//   c0(1) c(N+Z) b(m) Jc(nnzJ) G_val Mg_val Mw_val MJ_val MH_val seg_param(nseg) seg_param2(nseg)
//   x0(N) lb(N) ub(N) cl(m) cu(m)
// Everything else (index arrays, segment table, sparse quad_form constants) is shared and comes
// from the tape the problem handle was created with.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <new>
#include <vector>

#include <type_traits>

#include "exec_block.h"
#include "exec_hip.h"
#include "ipm_core.h"
#include "kkt_dense.h"
#include "wave_batch.h"
#include "wave_codegen.h"

namespace dnlp {

struct BatchLayout {
  i64 c0 = 0, c = 0, b = 0, Jc = 0, G = 0, Mg = 0, Mw = 0, MJ = 0, MH = 0, fp = 0, fp2 = 0, x0 = 0, lb = 0, ub = 0,
      cl = 0, cu = 0, total = 0;
};

struct BatchArgs {
  TapeView base;
  BatchLayout lay;
  double* slabs = nullptr;       // batch x lay.total
  int batch = 0;
  char* ws = nullptr;            // gridDim.x x ws_per_block
  size_t ws_per_block = 0;
  unsigned lds_bytes = 0;        // dynamic LDS pool (0: everything in global memory)
  int lds_mode = 0;              // what the pool holds: 0 nothing, 1 KKT matrix, 2 vectors, 3 both
  unsigned lds_stage_bytes = 0;  // front of the pool: staging arrays of the dense wavefront solves
  unsigned stage_global_bytes = 0;   // ... or, for sparse instances with a dense fallback, the tail of the block's global slab
  unsigned plan_stage_bytes = 0; // then: LDS copies of index arrays that every instance reads (0: all read from global), by
  int plan_stage_factor = 0;     // this mask: 1 the sparse plan's solve-phase arrays, 2 the tape's product indices (CooIdx),
                                 // 4 the plan's factor-phase / assembly arrays (update triples, value positions)
  i64 plan_rows = 0;             // struct rows of the plan (length of sidx / sblk)
  IpmOptions opt;
  double *x_out = nullptr, *obj_out = nullptr, *multg_out = nullptr, *zl_out = nullptr, *zu_out = nullptr;
  int *status_out = nullptr, *iters_out = nullptr, *nfact_out = nullptr;
  double* times_out = nullptr;   // batch x 4: wall, t_eval, t_factor, t_solve (seconds, device clock)
  SparsePlan sp;                 // static-pattern sparse KKT plan (shared by all instances)
  int use_sparse = 0;
  i64 fallback_max_n = 0;        // sparse instances up to this order may switch to in-kernel Bunch-Kaufman
  const double *ws_g = nullptr, *ws_l = nullptr, *ws_u = nullptr;   // per-instance warm-start multipliers (batch-major) or null
  int* next = nullptr;           // work queue head: instances are claimed dynamically (iteration counts vary 10x)
  const int* order = nullptr;    // queue position -> instance (longest expected first), or null for 0, 1, 2, ...
};

// The solver objects of an instance (exec space, model, KKT, interior-point state) live in LDS,
// one private copy per wavefront: every wavefront runs the control flow redundantly, and a field
// read is an LDS access (~0.1 us) instead of a scratch-memory round trip through L2 (~0.5 us)
// on the critical path of each of the several hundred maps / reductions of an iteration.
// (measured in round 4 and not kept: the same kernel held to 256 registers per lane, amdgpu_waves_per_eu(2, 2), eight
//  instances per compute unit: an iteration takes 0.27 ms instead of 0.155 by the device clock and a launch of 8192 /
//  65 536 instances 59 / 300 ms instead of 60 / 333 — two wavefronts on a SIMD share the memory pipeline that one
//  already keeps busy: the kernel is bound by the NUMBER of its memory operations, not by their latency)
template <int NT>
__global__ void __launch_bounds__(NT) batch_solve_kernel(BatchArgs a) {
  extern __shared__ __align__(64) char lds_dyn[];
  using EX = BlockExecT<NT>;
  using KktT = DenseKkt<EX>;
  using ModelT = Model<EX>;
  using IpmT = Ipm<EX, KktT>;
  constexpr int NW = NT / 64;
  struct alignas(16) Objs {
    alignas(16) char ex[sizeof(EX)];
    alignas(16) char md[sizeof(ModelT)];
    alignas(16) char kkt[sizeof(KktT)];
    alignas(16) char ipm[sizeof(IpmT)];
  };
  // (exec.h DNLP_THIS_IN_LDS: the member functions of these objects ASSUME that they live in LDS — EX::objects_in_lds;
  //  an instance of BlockExecT / Model / DenseKkt / Ipm over it placed anywhere else would be undefined behaviour)
  __shared__ Objs s_objs[NW];
  __shared__ double s_red[8];
  __shared__ int s_redi[8];
  // staging of the dense single-wavefront solves: carved from the front of the dynamic pool, and
  // only when the dense path can be taken (sparse instances without a fallback do not pay for it)
  double* s_vec = reinterpret_cast<double*>(lds_dyn);
  int* s_piv = reinterpret_cast<int*>(lds_dyn + sizeof(double) * EX::kWaveSolveMax);
  const unsigned stage = a.lds_stage_bytes;
  // The triangular solves walk small index arrays of the plan (block nodes, struct offsets, row ->
  // block, row -> node) in chains of dependent loads, ~11 level phases per solve and ~4 solves per
  // iteration: from L2 that chain is the largest single cost of an iteration.  The arrays are the same
  // for every instance, so the workgroup copies them into LDS once and every instance it solves
  // reads them through a patched view of the plan.
  SparsePlan spl = a.sp;
  CooIdx cjr = a.base.jac_by_row, cjc = a.base.jac_by_col, chs = a.base.hess_sym;
  const unsigned pstage = a.use_sparse ? a.plan_stage_bytes : 0u;
  if (pstage) {
    char* p = lds_dyn + stage;
    auto put = [&](auto*& field, i64 count) {
      using T = typename std::remove_pointer<typename std::remove_reference<decltype(field)>::type>::type;
      T* dst = reinterpret_cast<T*>(p);
      for (i64 i = threadIdx.x; i < count; i += NT) dst[i] = field[i];
      field = dst;
      p += (static_cast<size_t>(count) * sizeof(T) + 7) & ~static_cast<size_t>(7);
    };
    if (a.plan_stage_factor & 1) {
      put(spl.bnode, 2 * spl.nblk);
      put(spl.soff, spl.nblk + 1);
      put(spl.loff, spl.nblk);
      put(spl.doff, spl.nblk);
      put(spl.lev_off, spl.nlev + 1);
      put(spl.sblk, a.plan_rows);
      put(spl.sidx, a.plan_rows);
      put(spl.lev_f, spl.nlev + 1);
      put(spl.fnode, spl.nfwd);
      put(spl.foff, spl.nfwd + 1);
      put(spl.fa, a.plan_rows);
      put(spl.fu0, a.plan_rows);
      put(spl.fu1, a.plan_rows);
    }
    if (a.plan_stage_factor & 2) {
      auto putc = [&](CooIdx& c) {
        if (!c.ptr) return;
        put(c.ptr, c.nout + 1); put(c.ent, c.total); put(c.src, c.total); put(c.heavy, c.nheavy);
      };
      putc(cjr); putc(cjc); putc(chs);
    }
    if (a.plan_stage_factor & 4) {
      put(spl.lev_g, spl.nlev + 1);
      put(spl.gdst, spl.ngrp);
      put(spl.goff, spl.ngrp + 1);
      put(spl.tau, spl.ntrip);
      put(spl.tav, spl.ntrip);
      put(spl.hpos, a.base.nnzH);
      put(spl.jpos, a.base.nnzJ);
      put(spl.dpos, spl.n);
    }
    __syncthreads();
  }
  __shared__ int s_inst;
  Objs& o = s_objs[threadIdx.x >> 6];
  while (true) {
    if (threadIdx.x == 0) {
      const int k = atomicAdd(a.next, 1);
      s_inst = (k < a.batch && a.order) ? a.order[k] : k;
    }
    __syncthreads();
    const int inst = s_inst;
    __syncthreads();
    if (inst >= a.batch) break;
    char* slab = a.ws + static_cast<size_t>(blockIdx.x) * a.ws_per_block;
    const size_t slab_cap = a.ws_per_block - a.stage_global_bytes;
    double* g_vec = a.stage_global_bytes ? reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(slab + slab_cap) + 63) & ~static_cast<uintptr_t>(63)) : nullptr;
    int* g_piv = g_vec ? reinterpret_cast<int*>(g_vec + EX::kWaveSolveMax) : nullptr;
    EX* ex = new (o.ex) EX(slab, slab_cap, lds_dyn + stage + pstage,
                           a.lds_bytes - stage - pstage, s_red, s_redi, stage ? s_vec : g_vec, stage ? s_piv : g_piv);
    ex->lds_mode = a.lds_mode;
    double* sl = a.slabs + static_cast<i64>(inst) * a.lay.total;
    TapeView t = a.base;
    t.jac_by_row = cjr; t.jac_by_col = cjc; t.hess_sym = chs;
    t.c0 = sl[a.lay.c0];
    t.c = sl + a.lay.c; t.b = sl + a.lay.b; t.Jc = sl + a.lay.Jc;
    t.G.val = sl + a.lay.G; t.Mg.val = sl + a.lay.Mg; t.Mw.val = sl + a.lay.Mw; t.MJ.val = sl + a.lay.MJ;
    t.MH.val = sl + a.lay.MH; t.flat_p = sl + a.lay.fp; t.flat_p2 = sl + a.lay.fp2;
    t.d_x0 = sl + a.lay.x0; t.d_lb = sl + a.lay.lb; t.d_ub = sl + a.lay.ub; t.d_cl = sl + a.lay.cl; t.d_cu = sl + a.lay.cu;
    ModelT* md = new (o.md) ModelT();
    md->init_view(ex, t);
    KktT* kkt = new (o.kkt) KktT();
    if (a.use_sparse) {
      kkt->init_sparse(ex, t.N, t.m, spl);
      kkt->pivot_max_n = static_cast<i64>(1) << 40;
      kkt->fallback_max_n = a.fallback_max_n;
    } else {
      kkt->pivot_max_n = static_cast<i64>(1) << 40;   // always the pivoted (Bunch-Kaufman) factorisation
      kkt->init(ex, t.N, t.m);
    }
    IpmT* ipm = new (o.ipm) IpmT(ex, md, kkt);
    ipm->opt = a.opt;
    if (a.ws_g) {
      ipm->ws_mult_g = a.ws_g + static_cast<i64>(inst) * t.m;
      ipm->ws_mult_xL = a.ws_l + static_cast<i64>(inst) * t.N;
      ipm->ws_mult_xU = a.ws_u + static_cast<i64>(inst) * t.N;
    }
    ipm->allocate();
    int st = Internal_Error;
    if (!ex->overflow) {
      st = ipm->solve(t.d_x0);
      if (ipm->initialized)
        ipm->extract_exec(a.x_out + static_cast<i64>(inst) * t.N, a.multg_out ? a.multg_out + static_cast<i64>(inst) * t.m : nullptr,
                          a.zl_out ? a.zl_out + static_cast<i64>(inst) * t.N : nullptr,
                          a.zu_out ? a.zu_out + static_cast<i64>(inst) * t.N : nullptr, nullptr);
    }
    if (threadIdx.x == 0) {
      a.status_out[inst] = st;
      a.iters_out[inst] = ipm->iter;
      a.obj_out[inst] = ipm->initialized ? ipm->objective_unscaled() : 0.0;
      if (a.nfact_out) a.nfact_out[inst] = ipm->stats.factorizations;
      if (a.times_out) {
        double* to = a.times_out + 4 * static_cast<i64>(inst);
        to[0] = ipm->stats.wall; to[1] = ipm->stats.t_eval; to[2] = ipm->stats.t_factor; to[3] = ipm->stats.t_solve;
      }
    }
    __syncthreads();
  }
}

// PACKED form for small sparse instances (round 4): a workgroup of NW wavefronts, one instance per wavefront, the
// wavefronts independent of each other after the staging.  The index arrays that every instance reads (sparse plan,
// product indices: ~10 KB for localization) exist ONCE per compute unit instead of once per instance, which leaves
// room for EVERY vector of the NW instances in LDS (localization: 4 x 37 KB + 10 KB of 160 KB) — no vector in the
// global slab, and because that is guaranteed (BlockExecT<64, true>::alloc refuses anything else before
// freeze_vectors()), the solver's element accesses are compiled as LDS instructions (exec.h VecP, vectors_in_lds)
// instead of flat ones.  The host takes this kernel when the sizes allow it (BatchRunner::solve_impl) and falls back
// to batch_solve_kernel when an instance reports that its vectors did not fit.
struct PackedObjSizes { size_t ex, md, kkt, ipm; };
template <int NW>
__global__ void __launch_bounds__(64 * NW) batch_solve_packed_kernel(BatchArgs a, unsigned wave_lds_bytes) {
  extern __shared__ __align__(64) char lds_dyn[];
  using EX = BlockExecT<64, true>;
  using KktT = DenseKkt<EX>;
  using ModelT = Model<EX>;
  using IpmT = Ipm<EX, KktT>;
  struct alignas(16) Objs {
    alignas(16) char ex[sizeof(EX)];
    alignas(16) char md[sizeof(ModelT)];
    alignas(16) char kkt[sizeof(KktT)];
    alignas(16) char ipm[sizeof(IpmT)];
  };
  __shared__ Objs s_objs[NW];
  __shared__ double s_red[NW][8];
  __shared__ int s_redi[NW][8];
  __shared__ int s_inst[NW];
  const int wave = static_cast<int>(threadIdx.x >> 6), lane = static_cast<int>(threadIdx.x & 63u);
  SparsePlan spl = a.sp;
  CooIdx cjr = a.base.jac_by_row, cjc = a.base.jac_by_col, chs = a.base.hess_sym;
  const unsigned pstage = a.plan_stage_bytes;
  {
    // every array the instances share, once per workgroup (the same `put` order as batch_solve_kernel's tiers 1 | 2 | 4)
    char* p = lds_dyn;
    auto put = [&](auto*& field, i64 count) {
      using T = typename std::remove_pointer<typename std::remove_reference<decltype(field)>::type>::type;
      T* dst = reinterpret_cast<T*>(p);
      for (i64 i = threadIdx.x; i < count; i += 64 * NW) dst[i] = field[i];
      field = dst;
      p += (static_cast<size_t>(count) * sizeof(T) + 7) & ~static_cast<size_t>(7);
    };
    put(spl.bnode, 2 * spl.nblk); put(spl.soff, spl.nblk + 1); put(spl.loff, spl.nblk); put(spl.doff, spl.nblk);
    put(spl.lev_off, spl.nlev + 1); put(spl.sblk, a.plan_rows); put(spl.sidx, a.plan_rows);
    put(spl.lev_f, spl.nlev + 1); put(spl.fnode, spl.nfwd); put(spl.foff, spl.nfwd + 1);
    put(spl.fa, a.plan_rows); put(spl.fu0, a.plan_rows); put(spl.fu1, a.plan_rows);
    auto putc = [&](CooIdx& c) { if (c.ptr) { put(c.ptr, c.nout + 1); put(c.ent, c.total); put(c.src, c.total); put(c.heavy, c.nheavy); } };
    putc(cjr); putc(cjc); putc(chs);
    put(spl.lev_g, spl.nlev + 1); put(spl.gdst, spl.ngrp); put(spl.goff, spl.ngrp + 1);
    put(spl.tau, spl.ntrip); put(spl.tav, spl.ntrip);
    put(spl.hpos, a.base.nnzH); put(spl.jpos, a.base.nnzJ); put(spl.dpos, spl.n);
    __syncthreads();                 // the only workgroup barrier of the kernel
  }
  Objs& o = s_objs[wave];
  char* my_lds = lds_dyn + pstage + static_cast<size_t>(wave) * wave_lds_bytes;
  char* slab = a.ws + (static_cast<size_t>(blockIdx.x) * NW + wave) * a.ws_per_block;
  const size_t slab_cap = a.ws_per_block - a.stage_global_bytes;
  double* g_vec = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(slab + slab_cap) + 63) & ~static_cast<uintptr_t>(63));
  int* g_piv = reinterpret_cast<int*>(g_vec + EX::kWaveSolveMax);
  while (true) {
    if (lane == 0) {
      const int k = atomicAdd(a.next, 1);
      s_inst[wave] = (k < a.batch && a.order) ? a.order[k] : k;
    }
    wave_sync();
    const int inst = s_inst[wave];
    wave_sync();
    if (inst >= a.batch) break;
    EX* ex = new (o.ex) EX(slab, slab_cap, my_lds, wave_lds_bytes, s_red[wave], s_redi[wave], g_vec, g_piv);
    ex->lds_mode = 3;
    double* sl = a.slabs + static_cast<i64>(inst) * a.lay.total;
    TapeView t = a.base;
    t.jac_by_row = cjr; t.jac_by_col = cjc; t.hess_sym = chs;
    t.c0 = sl[a.lay.c0];
    t.c = sl + a.lay.c; t.b = sl + a.lay.b; t.Jc = sl + a.lay.Jc;
    t.G.val = sl + a.lay.G; t.Mg.val = sl + a.lay.Mg; t.Mw.val = sl + a.lay.Mw; t.MJ.val = sl + a.lay.MJ;
    t.MH.val = sl + a.lay.MH; t.flat_p = sl + a.lay.fp; t.flat_p2 = sl + a.lay.fp2;
    t.d_x0 = sl + a.lay.x0; t.d_lb = sl + a.lay.lb; t.d_ub = sl + a.lay.ub; t.d_cl = sl + a.lay.cl; t.d_cu = sl + a.lay.cu;
    ModelT* md = new (o.md) ModelT();
    md->init_view(ex, t);
    KktT* kkt = new (o.kkt) KktT();
    kkt->init_sparse(ex, t.N, t.m, spl);
    kkt->pivot_max_n = static_cast<i64>(1) << 40;
    kkt->fallback_max_n = a.fallback_max_n;
    IpmT* ipm = new (o.ipm) IpmT(ex, md, kkt);
    ipm->opt = a.opt;
    if (a.ws_g) {
      ipm->ws_mult_g = a.ws_g + static_cast<i64>(inst) * t.m;
      ipm->ws_mult_xL = a.ws_l + static_cast<i64>(inst) * t.N;
      ipm->ws_mult_xU = a.ws_u + static_cast<i64>(inst) * t.N;
    }
    ipm->allocate();
    ex->freeze_vectors();            // (what a dense fallback allocates later — its matrix, pivots, scratch — may live in the slab)
    int st = Internal_Error;
    if (!ex->overflow) {
      st = ipm->solve(t.d_x0);
      if (ipm->initialized)
        ipm->extract_exec(a.x_out + static_cast<i64>(inst) * t.N, a.multg_out ? a.multg_out + static_cast<i64>(inst) * t.m : nullptr,
                          a.zl_out ? a.zl_out + static_cast<i64>(inst) * t.N : nullptr,
                          a.zu_out ? a.zu_out + static_cast<i64>(inst) * t.N : nullptr, nullptr);
    } else {
      st = -198;                     // the vectors did not fit the wavefront's LDS share: the host takes the other kernel
    }
    if (lane == 0) {
      a.status_out[inst] = st;
      a.iters_out[inst] = ex->overflow ? 0 : ipm->iter;
      a.obj_out[inst] = (!ex->overflow && ipm->initialized) ? ipm->objective_unscaled() : 0.0;
      if (a.nfact_out) a.nfact_out[inst] = ipm->stats.factorizations;
      if (a.times_out) {
        double* to = a.times_out + 4 * static_cast<i64>(inst);
        to[0] = ipm->stats.wall; to[1] = ipm->stats.t_eval; to[2] = ipm->stats.t_factor; to[3] = ipm->stats.t_solve;
      }
    }
    wave_sync();
  }
}

// Instance data from parameter rows, on the device: slab[b][o] = d0[o] + sum_k D[o][k] (theta[b][k] - theta0[k]) with
// D in CSR over the slab's own layout (the per-segment parameters already gathered to per-flat-row ones).
// A lane owns one slab entry of one instance; the instance's parameter row (tens of doubles) is read by
// every lane of the workgroup through the scalar / L1 path.  Rows without entries (infinite bounds,
// constants) copy d0.  Writes are coalesced along o.
__global__ void __launch_bounds__(256) batch_affine_expand_kernel(double* __restrict__ slabs, i64 total, int batch, int P,
                                                                  const double* __restrict__ theta, const double* __restrict__ theta0,
                                                                  const double* __restrict__ d0, const i64* __restrict__ indptr,
                                                                  const int* __restrict__ indices, const double* __restrict__ vals) {
  const i64 o = static_cast<i64>(blockIdx.x) * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (o >= total || b >= batch) return;
  const double* th = theta + static_cast<i64>(b) * P;
  double v = d0[o];
  for (i64 e = indptr[o]; e < indptr[o + 1]; ++e) {
    const int k = indices[e];
    v += vals[e] * (th[k] - theta0[k]);
  }
  slabs[static_cast<i64>(b) * total + o] = v;
}

// Host side: tables uploaded once per problem handle, slabs per call.
// Result rows {index in the launch, objective, status, iterations, x*} of a launch, packed on the device: what a rank
// hands to the path's one collective (dnlp_amd/batch.py gather_rows) without a trip through host memory.
__global__ void __launch_bounds__(256) batch_pack_rows_kernel(double* __restrict__ rows, int width, int batch, int N, const double* __restrict__ x,
                                                              const double* __restrict__ obj, const int* __restrict__ status,
                                                              const int* __restrict__ iters) {
  const i64 total = static_cast<i64>(batch) * width;
  for (i64 e = static_cast<i64>(blockIdx.x) * 256 + threadIdx.x; e < total; e += static_cast<i64>(gridDim.x) * 256) {
    const int i = static_cast<int>(e / width), c = static_cast<int>(e % width);
    rows[e] = c == 0 ? static_cast<double>(i) : c == 1 ? obj[i] : c == 2 ? static_cast<double>(status[i]) : c == 3 ? static_cast<double>(iters[i])
                                                                                                                  : x[static_cast<i64>(i) * N + (c - 4)];
  }
}

struct BatchRunner {
  HipExec* ex = nullptr;
  hipStream_t stream = nullptr;   // this runner's launches and copies (the handle's stream; a batch-stream slot has its own)
  bool own_stream = false;
  Tape<HipExec>* tape = nullptr;
  SegHost* d_segs = nullptr;
  i64* d_red = nullptr;
  SparseConst* d_sparse = nullptr;
  BatchLayout lay;
  i64 in_stride = 0;
  int last_grid = 0, last_threads = 0, last_lds_mode = 0, last_per_cu = 0;   // launch plan of the last solve
  bool last_order_lpt = false;   // the last solve took its instances longest-first (a re-solve of the same rows)
  bool last_packed = false;      // the last solve ran batch_solve_packed_kernel
  bool packed_disabled = false;  // an instance's vectors did not fit the packed kernel's LDS share once
  bool have_sparse = false, force_sparse = false;
  // warm-start multipliers for the NEXT solve (consumed by it)
  std::vector<double> h_ws_g, h_ws_l, h_ws_u;
  int ws_batch = 0;
  void set_warm_start(int batch, const double* mg, const double* mxl, const double* mxu) {
    const size_t m = static_cast<size_t>(tape->m), N = static_cast<size_t>(tape->N), B = static_cast<size_t>(batch);
    h_ws_g.assign(mg, mg + B * m); h_ws_l.assign(mxl, mxl + B * N); h_ws_u.assign(mxu, mxu + B * N);
    ws_batch = batch;
  }
  SparsePlan dev_plan;
  i64 plan_rows = 0;
  void set_sparse_plan(const SparsePlanHost& hp) { dev_plan = hp.upload(ex); have_sparse = true; plan_rows = static_cast<i64>(hp.sidx.size()); host_plan = &hp; }
  // ---- the wavefront solver of small sparse templates (wave_plan.h / wave_ipm.h / wave_batch.h) ----
  const SparsePlanHost* host_plan = nullptr;      // (owned by the problem handle, as the tape is)
  std::vector<i32> wave_blk;                      // the template's plan block; empty: the template takes the generic kernel
  i32* d_wave_blk = nullptr;
  int16_t* d_wave_blk16 = nullptr;                // the same block narrowed to 16 bits (forms with the state in LDS and the plan in global memory)
  bool wave_checked = false;
  bool wave_fits16 = false;                       // every table entry of the block fits 16 bits: it can be staged in LDS
  std::string wave_why;                           // why not, when not
  int last_wave = 0;                              // the last solve: 0 generic kernel, else 100 * wavefronts per workgroup + 10 * state in LDS + plan in LDS
  int last_wave_refused = 0;                      // instances of the last solve that the wavefront solver handed to the generic kernel
  // the per-template kernel (wave_codegen.h): compiled by hiprtc at the template's first large launch, cached on disk
  RtcKernel wave_spec;
  int wave_spec_nw = 0;                           // wavefronts per workgroup it was compiled for (its static LDS holds that many shares)
  unsigned* d_wave_gen = nullptr;                 // work tables of its generated LDL^T phases (wave_gen.h)
  int wave_gen_words = 0;
  bool wave_spec_prof = false;
  bool wave_spec_tables_global = false;           // its tables stay in global memory (more wavefronts per compute unit)
  bool last_wave_spec = false;                    // the last solve ran it
  double wave_spec_compile_seconds = 0.0;
  // DNLP_WAVE_SPEC: 0 never, 1 for every launch of a template whose state and plan fit LDS, unset: launches of at least
  // kWaveSpecMinBatch instances (a compilation costs seconds once per template and machine; a 1024-instance launch 10 ms)
  static constexpr int kWaveSpecMinBatch = 1024;
  bool wave_spec_prepare(int batch) {
    const char* e = std::getenv("DNLP_WAVE_SPEC");
    const int mode = e ? std::atoi(e) : -1;
    if (mode == 0) return false;
    if (mode < 0 && batch < kWaveSpecMinBatch && !wave_spec.ok) return false;
    if (wave_spec.tried) return wave_spec.ok;
    const WaveHdr& h = *reinterpret_cast<const WaveHdr*>(wave_blk.data());
    wave_spec.tried = true;
    if (wave_gen_refusal(h)[0]) return false;
    // (the kernel stages the first keep_gen ints of the block, narrowed to 16 bits, and the generated phases' work tables)
    bool fits = true;
    for (size_t k = sizeof(WaveHdr) / 4; k < static_cast<size_t>(h.keep_gen) && fits; ++k) fits = wave_blk[k] >= -32768 && wave_blk[k] <= 32767;
    if (!fits) return false;
    const double t0 = now_sec();
    const WaveGen gen = wave_generate(wave_blk);
    // (DNLP_WAVE_SPEC_PROF: the kernel is compiled with the cycle counters of wave_ipm.h W_P0 / W_P1 and the host prints them)
    wave_spec_prof = std::getenv("DNLP_WAVE_SPEC_PROF") != nullptr;
    // tables in LDS, or — when the shares leave no room for them — in global memory with more wavefronts per compute unit
    // (a table word through L1 / L2 costs an iteration ~20 %: worth it from 1.2 x the wavefronts on)
    const int fit_l = wave_spec_max_waves(h, gen.G.size(), wave_spec_prof, false);
    const int fit_g = (wave_fits16 && d_wave_blk16) ? wave_spec_max_waves(h, gen.G.size(), wave_spec_prof, true) : 0;
    bool tabg = 5 * fit_g > 6 * fit_l;
    if (const char* tg = std::getenv("DNLP_WAVE_SPEC_TABLES")) tabg = std::atoi(tg) == 1 ? fit_g > 0 : false;      // (1 global, 0 LDS)
    const int fit = tabg ? fit_g : fit_l;
    if (fit < 1) return false;
    wave_spec_tables_global = tabg;
    const std::string src = wave_spec_source(wave_blk, fit, gen, wave_spec_prof, tabg);
    if (!wave_spec.load(src, "dnlp_wave_spec_kernel")) {
      std::fprintf(stderr, "[dnlp] per-template batch kernel not available (the library's own kernel is used): %s\n", wave_spec.log.substr(0, 2000).c_str());
      return false;
    }
    if (!wave_spec.fits(fit)) {
      wave_spec.ok = false;
      rtc_cache_drop(src);
      std::fprintf(stderr, "[dnlp] per-template batch kernel: its register allocation does not fit %d wavefronts per workgroup (the library's own kernel is used)\n", fit);
      return false;
    }
    wave_gen_words = static_cast<int>(gen.G.size());
    DNLP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_wave_gen), (gen.G.size() + 4) * sizeof(unsigned)));
    if (!gen.G.empty()) DNLP_HIP_CHECK(hipMemcpy(d_wave_gen, gen.G.data(), gen.G.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    wave_spec_nw = fit;
    wave_spec_compile_seconds = now_sec() - t0;
    if (std::getenv("DNLP_BATCH_DEBUG")) std::fprintf(stderr, "[batch] per-template kernel: %d wavefronts per workgroup, tables in %s, %.2f s to compile / load\n", fit, tabg ? "global memory" : "LDS", wave_spec_compile_seconds);
    return true;
  }
  // the WORKGROUP-per-instance kernel of templates whose state exceeds LDS (wave_wg_kernel.h; same switches as the
  // per-template kernel above: DNLP_WAVE_SPEC 0 / 1 / unset, DNLP_WAVE_WG_WAVES = wavefronts per workgroup, default 4)
  RtcKernel wave_wg;
  int wave_wg_nw = 0, wave_wg_per_cu = 1, wave_wg_gen_words = 0;
  unsigned* d_wave_wg_gen = nullptr;
  std::vector<i32> wave_wg_blk;                   // its plan block (no dense tail in registers)
  i32* d_wave_wg_blk = nullptr;
  bool wave_wg_prof = false;
  // the workgroup kernel's own plan block: WITHOUT the dense tail in registers (one-wavefront code; the chain's few levels run as
  // narrow generated phases instead — power flow: tail_forward alone was 59 k of 1 590 k cycles per iteration).  Built where
  // wave_prepare() builds the library's block — on the caller's thread: build_wave_plan reads the tape's index arrays back
  // through the handle's stream, which the worker threads of a batch stream's slots must not share.
  bool wave_wg_plan() {
    const WaveHdr& h = *reinterpret_cast<const WaveHdr*>(wave_blk.data());
    WaveLayoutIn l;
    l.c0 = lay.c0; l.c = lay.c; l.b = lay.b; l.Jc = lay.Jc; l.G = lay.G; l.Mg = lay.Mg; l.Mw = lay.Mw; l.MJ = lay.MJ; l.MH = lay.MH;
    l.fp = lay.fp; l.fp2 = lay.fp2; l.x0 = lay.x0; l.lb = lay.lb; l.ub = lay.ub; l.cl = lay.cl; l.cu = lay.cu; l.total = lay.total;
    const bool tail = std::getenv("DNLP_WAVE_WG_TAIL") && std::atoi(std::getenv("DNLP_WAVE_WG_TAIL")) == 1;
    wave_wg_blk = build_wave_plan(ex, *tape, *host_plan, l, tail);
    if (reinterpret_cast<const WaveHdr*>(wave_wg_blk.data())->state_doubles != h.state_doubles) { wave_wg_blk.assign(1, 0); return false; }
    DNLP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_wave_wg_blk), wave_wg_blk.size() * sizeof(i32)));
    DNLP_HIP_CHECK(hipMemcpy(d_wave_wg_blk, wave_wg_blk.data(), wave_wg_blk.size() * sizeof(i32), hipMemcpyHostToDevice));
    return true;
  }
  bool wave_wg_prepare(int batch) {
    const char* e = std::getenv("DNLP_WAVE_SPEC");
    const int mode = e ? std::atoi(e) : -1;
    if (mode == 0) return false;
    if (mode < 0 && batch < kWaveSpecMinBatch && !wave_wg.ok) return false;
    if (wave_wg.tried) return wave_wg.ok;
    wave_wg.tried = true;
    const WaveHdr& h = *reinterpret_cast<const WaveHdr*>(wave_blk.data());
    if (wave_gen_refusal(h)[0]) return false;
    // A template whose state EXCEEDS LDS: eight wavefronts per instance, the compute unit to itself.  A template whose state
    // fits but leaves room for only one or two wavefronts per compute unit in the one-wavefront kernels (circle packing n = 10:
    // 74 KB per instance) can take this kernel too — TWO wavefronts per instance, FOUR workgroups per compute unit with a
    // quarter of the LDS each (hot arrays there, the rest in the slab): eight wavefronts per unit instead of two, 173 us per
    // iteration instead of 138 — 102 k problems/s at 8192 instances against 71 k (solve_wave takes it from 6 instances per
    // compute unit on; at 1024 the one-wavefront kernel is as fast: 39 k).
    int sl_form = 0;
    { int nw0 = 0, pl0 = 0; wave_form(nw0, sl_form, pl0); }
    int nwg = std::getenv("DNLP_WAVE_WG_WAVES") ? std::atoi(std::getenv("DNLP_WAVE_WG_WAVES")) : (sl_form ? 2 : 8);
    if (nwg < 1 || nwg > 8) nwg = sl_form ? 2 : 8;
    int share = std::getenv("DNLP_WAVE_WG_SHARE") ? std::atoi(std::getenv("DNLP_WAVE_WG_SHARE")) : (sl_form ? 4 : 1);
    if (share < 1 || share > 16) share = 1;
    const double t0 = now_sec();
    if (wave_wg_blk.empty() && !wave_wg_plan()) return false;
    if (wave_wg_blk.size() < sizeof(WaveHdr) / 4) return false;
    const bool lds_vec_gen = !(std::getenv("DNLP_WAVE_WG_LDS") && std::atoi(std::getenv("DNLP_WAVE_WG_LDS")) == 0);
    const WaveGen gen = wave_wg_generate(wave_wg_blk, nwg, lds_vec_gen, share);
    wave_wg_prof = std::getenv("DNLP_WAVE_SPEC_PROF") != nullptr;
    // (DNLP_WAVE_WG_BOUND: threads the register budget is sized for — 512 with four wavefronts: two workgroups per compute unit)
    const int bound = std::getenv("DNLP_WAVE_WG_BOUND") ? std::atoi(std::getenv("DNLP_WAVE_WG_BOUND")) : 512;
    const bool lds_vec = !(std::getenv("DNLP_WAVE_WG_LDS") && std::atoi(std::getenv("DNLP_WAVE_WG_LDS")) == 0);
    const std::string src = wave_wg_source(wave_wg_blk, nwg, gen, wave_wg_prof, bound, lds_vec, share);
    if (!wave_wg.load(src, "dnlp_wave_wg_kernel")) {
      std::fprintf(stderr, "[dnlp] workgroup-per-instance batch kernel not available (the library's own kernel is used): %s\n", wave_wg.log.substr(0, 2000).c_str());
      return false;
    }
    if (!wave_wg.fits(nwg)) {
      wave_wg.ok = false;
      rtc_cache_drop(src);
      std::fprintf(stderr, "[dnlp] workgroup-per-instance batch kernel: its register allocation does not fit %d wavefronts per workgroup (the library's own kernel is used)\n", nwg);
      return false;
    }
    wave_wg_gen_words = static_cast<int>(gen.G.size());
    DNLP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_wave_wg_gen), (gen.G.size() + 4) * sizeof(unsigned)));
    if (!gen.G.empty()) DNLP_HIP_CHECK(hipMemcpy(d_wave_wg_gen, gen.G.data(), gen.G.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    wave_wg_nw = nwg;
    int occ = 1;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&occ, wave_wg.fn, 64 * nwg, 0) != hipSuccess || occ < 1) occ = 1;
    wave_wg_per_cu = occ;
    if (const char* pc = std::getenv("DNLP_WAVE_WG_PER_CU")) { const int v = std::atoi(pc); if (v >= 1 && v <= 8) wave_wg_per_cu = v; }
    if (std::getenv("DNLP_BATCH_DEBUG"))
      std::fprintf(stderr, "[batch] workgroup-per-instance kernel: %d wavefronts per instance, %d workgroups per compute unit, %d table words, %.2f s to generate / compile / load\n",
                   nwg, wave_wg_per_cu, wave_wg_gen_words, now_sec() - t0);
    return true;
  }
  bool wave_prepare() {
    if (!wave_checked) {
      wave_checked = true;
      wave_why = wave_plan_refusal(*tape, have_sparse ? host_plan : nullptr);
      if (wave_why.empty()) {
        WaveLayoutIn l;
        l.c0 = lay.c0; l.c = lay.c; l.b = lay.b; l.Jc = lay.Jc; l.G = lay.G; l.Mg = lay.Mg; l.Mw = lay.Mw; l.MJ = lay.MJ; l.MH = lay.MH;
        l.fp = lay.fp; l.fp2 = lay.fp2; l.x0 = lay.x0; l.lb = lay.lb; l.ub = lay.ub; l.cl = lay.cl; l.cu = lay.cu; l.total = lay.total;
        wave_blk = build_wave_plan(ex, *tape, *host_plan, l);
        wave_fits16 = true;
        for (size_t k = sizeof(WaveHdr) / 4; k < wave_blk.size() && wave_fits16; ++k) wave_fits16 = wave_blk[k] >= -32768 && wave_blk[k] <= 32767;
        DNLP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_wave_blk), wave_blk.size() * sizeof(i32)));
        DNLP_HIP_CHECK(hipMemcpy(d_wave_blk, wave_blk.data(), wave_blk.size() * sizeof(i32), hipMemcpyHostToDevice));
        if (wave_fits16) {
          std::vector<int16_t> narrow16(wave_blk.size());
          for (size_t k = 0; k < wave_blk.size(); ++k) narrow16[k] = static_cast<int16_t>(wave_blk[k]);
          DNLP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&d_wave_blk16), narrow16.size() * sizeof(int16_t)));
          DNLP_HIP_CHECK(hipMemcpy(d_wave_blk16, narrow16.data(), narrow16.size() * sizeof(int16_t), hipMemcpyHostToDevice));
        }
        // a template whose state exceeds LDS goes through the workgroup-per-instance kernel: its plan block now, too
        // (... and one whose state fits only once or twice per compute unit: wave_wg_prepare's second form)
        { int nw = 0, sl = 0, pl = 0; wave_form(nw, sl, pl); if ((!sl || nw <= 2) && !wave_gen_refusal(*reinterpret_cast<const WaveHdr*>(wave_blk.data()))[0]) wave_wg_plan(); }
      }
    }
    return !wave_blk.empty();
  }
  // device buffers kept across calls (grow-only): a call is then one H2D copy, one launch and the
  // result copies — no allocation on the steady-state path
  struct Buf { void* p = nullptr; size_t cap = 0; };
  Buf bufs[24];
  int nbuf_used = 0;
  int ncu = 0;
  std::vector<double> slab;
  // Iteration counts of the previous solve of the SAME ROWS (prev_key): a re-solve of a parametrised batch (warm or
  // cold) takes its instances longest-first, so the instances that need 150+
  // iterations start at once instead of being the tail of the launch (mean 31, max 184-198 at 8192 localization
  // instances: a third of the launch was the tail).  A heuristic only — any order gives the same results.
  std::vector<int> prev_iters;
  uint64_t prev_key = 0;        // hash of the input rows prev_iters belongs to: only a RE-SOLVE OF THE SAME ROWS inherits the order
  static uint64_t rows_hash(const double* rows, size_t count) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ static_cast<uint64_t>(count);
    const uint64_t* w = reinterpret_cast<const uint64_t*>(rows);
    for (size_t i = 0; i < count; ++i) { h ^= w[i]; h *= 0x100000001B3ull; h ^= h >> 29; }
    return h ? h : 1;
  }

  // affine parameter -> instance-data map (dnlp_batch_set_affine_map), resident on the device, in slab layout
  int aff_P = -1;
  std::vector<double> h_aff_d0, h_aff_theta0, h_aff_vals;      // (host copy of what set_affine_map was given: a batch-stream slot replays it)
  std::vector<i64> h_aff_indptr;
  std::vector<int> h_aff_indices;
  double *aff_d0 = nullptr, *aff_theta0 = nullptr, *aff_val = nullptr, *aff_theta = nullptr;
  i64* aff_indptr = nullptr;
  int* aff_idx = nullptr;
  size_t aff_theta_cap = 0;
  void free_affine() {
    for (void* q : {static_cast<void*>(aff_d0), static_cast<void*>(aff_theta0), static_cast<void*>(aff_val),
                    static_cast<void*>(aff_indptr), static_cast<void*>(aff_idx), static_cast<void*>(aff_theta)})
      if (q) hipFree(q);
    aff_d0 = aff_theta0 = aff_val = aff_theta = nullptr; aff_indptr = nullptr; aff_idx = nullptr; aff_P = -1; aff_theta_cap = 0;
  }
  // d0: in_stride doubles (the base instance, BATCH_DATA_KEYS order); (indptr, indices, vals): CSR of the
  // in_stride x P sensitivity; theta0: P.  Re-expressed over the slab layout and uploaded once.
  void set_affine_map(int P, const double* d0, const double* theta0, const i64* indptr, const int* indices, const double* vals) {
    const Tape<HipExec>& t = *tape;
    DNLP_HIP_CHECK(hipSetDevice(ex->device));
    {
      const i64 rows = in_stride, nz = indptr[rows];
      std::vector<double> a(d0, d0 + rows), b(theta0, theta0 + P), v(vals, vals + nz);
      std::vector<i64> ip(indptr, indptr + rows + 1);
      std::vector<int> ix(indices, indices + nz);
      h_aff_d0.swap(a); h_aff_theta0.swap(b); h_aff_vals.swap(v); h_aff_indptr.swap(ip); h_aff_indices.swap(ix);
    }
    free_affine();
    const i64 head = 1 + (t.N + t.Z) + t.m + t.nnzJ + t.G.nnz + t.Mg.nnz + t.Mw.nnz + t.MJ.nnz + t.MH.nnz;
    auto src_row = [&](i64 o) -> i64 {
      if (o < head) return o;
      if (o < lay.fp2) return head + t.h_flat_seg[static_cast<size_t>(o - lay.fp)];
      if (o < lay.x0) return head + t.nseg + t.h_flat_seg[static_cast<size_t>(o - lay.fp2)];
      return head + 2 * t.nseg + (o - lay.x0);
    };
    std::vector<double> h_d0(static_cast<size_t>(lay.total)), h_val;
    std::vector<i64> h_ptr(static_cast<size_t>(lay.total) + 1, 0);
    std::vector<int> h_idx;
    for (i64 o = 0; o < lay.total; ++o) {
      const i64 j = src_row(o);
      h_d0[static_cast<size_t>(o)] = d0[j];
      for (i64 e = indptr[j]; e < indptr[j + 1]; ++e) {
        if (indices[e] < 0 || indices[e] >= P) throw std::runtime_error("batch affine map: column index out of range");
        h_idx.push_back(indices[e]); h_val.push_back(vals[e]);
      }
      h_ptr[static_cast<size_t>(o) + 1] = static_cast<i64>(h_idx.size());
    }
    auto up = [&](auto** dst, const auto* src, size_t n) {
      using T = std::remove_pointer_t<std::remove_pointer_t<decltype(dst)>>;
      DNLP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(dst), (n ? n : 1) * sizeof(T)));
      if (n) DNLP_HIP_CHECK(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    };
    up(&aff_d0, h_d0.data(), h_d0.size());
    up(&aff_theta0, theta0, static_cast<size_t>(P));
    up(&aff_indptr, h_ptr.data(), h_ptr.size());
    up(&aff_idx, h_idx.data(), h_idx.size());
    up(&aff_val, h_val.data(), h_val.size());
    aff_P = P;
  }

  ~BatchRunner() {
    free_affine();
    for (Buf& b : bufs) if (b.p) hipFree(b.p);
    if (d_segs) hipFree(d_segs);
    if (d_red) hipFree(d_red);
    if (d_sparse) hipFree(d_sparse);
    if (d_wave_blk) hipFree(d_wave_blk);
    if (d_wave_blk16) hipFree(d_wave_blk16);
    if (d_wave_gen) hipFree(d_wave_gen);
    if (d_wave_wg_gen) hipFree(d_wave_wg_gen);
    if (d_wave_wg_blk) hipFree(d_wave_wg_blk);
    if (d_rows) hipFree(d_rows);
    if (own_stream && stream) hipStreamDestroy(stream);
  }
  void release() { nbuf_used = 0; }
  // result rows of the last launch on the device (dnlp_batch_result_rows): {index, objective, status, iterations, x*}
  double* d_rows = nullptr;
  size_t d_rows_cap = 0;
  int rows_batch = 0, rows_width = 0;
  bool rows_wanted = false, rows_nested = false;
  void pack_rows(int batch, const double* x, const double* obj, const int* status, const int* iters) {
    if (!rows_wanted || rows_nested) return;
    const int width = 4 + static_cast<int>(tape->N);
    const size_t need = static_cast<size_t>(batch) * static_cast<size_t>(width) * sizeof(double);
    if (need > d_rows_cap) {
      if (d_rows) DNLP_HIP_CHECK(hipFree(d_rows));
      d_rows = nullptr;
      DNLP_HIP_CHECK(hipMalloc(&d_rows, need));
      d_rows_cap = need;
    }
    const unsigned grid = static_cast<unsigned>(std::min<i64>((static_cast<i64>(batch) * width + 255) / 256, 1024));
    hipLaunchKernelGGL(batch_pack_rows_kernel, dim3(grid), dim3(256), 0, stream, d_rows, width, batch, static_cast<int>(tape->N), x, obj, status, iters);
    DNLP_LAUNCH_CHECK();
    DNLP_HIP_CHECK(hipStreamSynchronize(stream));
    rows_batch = batch; rows_width = width;
  }
  // (an instance the wavefront solver refused was re-solved and merged on the host: its row follows)
  void patch_row(int k, double obj, int status, int iters, const double* x) {
    if (!rows_wanted || rows_nested || !d_rows || k >= rows_batch) return;
    std::vector<double> row(static_cast<size_t>(rows_width));
    row[0] = k; row[1] = obj; row[2] = status; row[3] = iters;
    std::copy(x, x + tape->N, row.begin() + 4);
    DNLP_HIP_CHECK(hipMemcpy(d_rows + static_cast<size_t>(k) * rows_width, row.data(), row.size() * sizeof(double), hipMemcpyHostToDevice));
  }
  template <class T> T* dalloc(size_t n) {
    Buf& b = bufs[nbuf_used++];
    const size_t bytes = (n ? n : 1) * sizeof(T);
    if (bytes > b.cap) {
      if (b.p) DNLP_HIP_CHECK(hipFree(b.p));
      b.p = nullptr;
      DNLP_HIP_CHECK(hipMalloc(&b.p, bytes));
      b.cap = bytes;
    }
    return static_cast<T*>(b.p);
  }

  // bytes of the LDS copies of the shared index arrays (batch_solve_kernel's `put` order and padding)
  struct IdxBytes { size_t solve = 0, coo = 0, factor = 0; };
  IdxBytes index_bytes() const {
    auto pad8 = [](size_t b) { return (b + 7) & ~static_cast<size_t>(7); };
    const Tape<HipExec>& t = *tape;
    const size_t nb = static_cast<size_t>(dev_plan.nblk), nl = static_cast<size_t>(dev_plan.nlev), nr = static_cast<size_t>(plan_rows);
    const size_t nf = static_cast<size_t>(dev_plan.nfwd), ng = static_cast<size_t>(dev_plan.ngrp), nt = static_cast<size_t>(dev_plan.ntrip);
    IdxBytes b;
    b.solve = pad8(8 * nb) + pad8(4 * (nb + 1)) + 2 * pad8(4 * nb) + pad8(4 * (nl + 1)) + 2 * pad8(4 * nr) +
              pad8(4 * (nl + 1)) + pad8(4 * nf) + pad8(4 * (nf + 1)) + 3 * pad8(4 * nr);
    for (const CooIdx* c : {&t.jac_by_row, &t.jac_by_col, &t.hess_sym})
      if (c->ptr) b.coo += pad8(4 * static_cast<size_t>(c->nout + 1)) + 2 * pad8(4 * static_cast<size_t>(c->total)) + pad8(4 * static_cast<size_t>(c->nheavy));
    b.factor = pad8(4 * (nl + 1)) + pad8(4 * ng) + pad8(4 * (ng + 1)) + 2 * pad8(4 * nt) + pad8(4 * static_cast<size_t>(t.nnzH)) +
               pad8(4 * static_cast<size_t>(t.nnzJ)) + pad8(4 * static_cast<size_t>(dev_plan.n));
    return b;
  }

  // bytes of every vector one instance allocates (BlockExecT::alloc: 64-byte granules), in the order of
  // Model::init_view, DenseKkt::init_sparse, Ipm::allocate
  size_t packed_vector_bytes() const {
    const Tape<HipExec>& t = *tape;
    const size_t N = static_cast<size_t>(t.N), m = static_cast<size_t>(t.m), Z = static_cast<size_t>(t.Z);
    size_t total = 0;
    auto al = [&](size_t n) { total += ((n ? n : 1) * 8 + 15) & ~static_cast<size_t>(15); };      // (BlockExecT<64, true>::alloc: 16-byte granules)
    al(N + Z); al(static_cast<size_t>(t.nd)); al(static_cast<size_t>(t.nh)); al(Z); al(1 + m); al(static_cast<size_t>(t.nnzH)); al(N);
    al(static_cast<size_t>(t.nblk));
    al(static_cast<size_t>(dev_plan.nvals)); al(static_cast<size_t>(dev_plan.nvals + 3 * dev_plan.nblk + 8));
    for (int k = 0; k < 14; ++k) al(N);                       // x zL zU xL xU grad dx dzL dzU xt Sx rx tN fixmask
    for (int k = 0; k < 22; ++k) al(m);                       // s y vL vU sL sU eqmask g sg ds dy dvL dvU st gt Dd Ss rs rp tM csoc zeroM
    al(static_cast<size_t>(t.nnzJ));
    for (int k = 0; k < 4; ++k) al(N + m);                    // rhs sol res cor
    for (int k = 0; k < 2; ++k) { for (int q = 0; q < 3; ++q) al(N); for (int q = 0; q < 4; ++q) al(m); }     // aff, cen
    return total;
  }

  void init(HipExec* e, Tape<HipExec>* t, bool private_stream = false) {
    ex = e; tape = t;
    stream = e->stream;
    if (private_stream) {
      DNLP_HIP_CHECK(hipSetDevice(e->device));
      DNLP_HIP_CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
      own_stream = true;
    }
    if (t->nblk > 0 || t->ndense > 0)
      throw std::runtime_error("batched solve: tapes with dense quad_form blocks are solved one at a time (dnlp_solve)");
    auto up = [&](auto** dst, const auto* src, size_t n) {
      using T = std::remove_pointer_t<std::remove_pointer_t<decltype(dst)>>;
      DNLP_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(dst), (n ? n : 1) * sizeof(T)));
      if (n) DNLP_HIP_CHECK(hipMemcpy(*dst, src, n * sizeof(T), hipMemcpyHostToDevice));
    };
    up(&d_segs, t->h_segs.data(), t->h_segs.size());
    up(&d_red, t->h_red_segs.data(), t->h_red_segs.size());
    up(&d_sparse, t->h_sparse.data(), t->h_sparse.size());
    i64 o = 0;
    auto take = [&](i64& f, i64 n) { f = o; o += n; };
    take(lay.c0, 1); take(lay.c, t->N + t->Z); take(lay.b, t->m); take(lay.Jc, t->nnzJ);
    take(lay.G, t->G.nnz); take(lay.Mg, t->Mg.nnz); take(lay.Mw, t->Mw.nnz); take(lay.MJ, t->MJ.nnz); take(lay.MH, t->MH.nnz);
    take(lay.fp, t->nflat); take(lay.fp2, t->nflat);
    take(lay.x0, t->N); take(lay.lb, t->N); take(lay.ub, t->N); take(lay.cl, t->m); take(lay.cu, t->m);
    lay.total = o;
    in_stride = 1 + (t->N + t->Z) + t->m + t->nnzJ + t->G.nnz + t->Mg.nnz + t->Mw.nnz + t->MJ.nnz + t->MH.nnz +
                2 * t->nseg + 3 * t->N + 2 * t->m;
  }

  // data: batch x stride (host).  Outputs: host arrays (mult_g / zl / zu may be null).
  void solve(int batch, const double* data, i64 stride, const IpmOptions& opt, double* x_out, double* obj_out,
             double* multg_out, double* zl_out, double* zu_out, int* status_out, int* iters_out, int* nfact_out,
             double* seconds, double* times_out = nullptr) {
    if (stride != in_stride) throw std::runtime_error("batched solve: instance stride does not match the tape");
    solve_impl(batch, data, nullptr, opt, x_out, obj_out, multg_out, zl_out, zu_out, status_out, iters_out, nfact_out, seconds,
               times_out);
  }
  // theta: batch x P parameter rows (host); the instance data is generated on the device from the map of
  // set_affine_map: the call moves batch x P doubles in and the results out.
  void solve_theta(int batch, const double* theta, int P, const IpmOptions& opt, double* x_out, double* obj_out,
                   double* multg_out, double* zl_out, double* zu_out, int* status_out, int* iters_out, int* nfact_out,
                   double* seconds, double* times_out = nullptr) {
    if (aff_P < 0) throw std::runtime_error("batched solve: no affine parameter map set (dnlp_batch_set_affine_map)");
    if (P != aff_P) throw std::runtime_error("batched solve: parameter rows do not match the affine map");
    solve_impl(batch, nullptr, theta, opt, x_out, obj_out, multg_out, zl_out, zu_out, status_out, iters_out, nfact_out, seconds,
               times_out);
  }

  void solve_impl(int batch, const double* data, const double* theta, const IpmOptions& opt, double* x_out, double* obj_out,
                  double* multg_out, double* zl_out, double* zu_out, int* status_out, int* iters_out, int* nfact_out,
                  double* seconds, double* times_out, bool allow_wave = true) {
    const i64 stride = in_stride;
    if (batch < 0) throw std::runtime_error("batch solve: negative instance count");
    if (batch == 0) {                 // an empty launch (an empty shard of a sharded batch) is a launch of nothing
      if (seconds) *seconds = 0.0;
      ws_batch = 0;
      last_grid = 0; last_wave = 0; last_wave_refused = 0;
      if (!rows_nested) { rows_batch = 0; rows_width = 4 + static_cast<int>(tape->N); }
      return;
    }
    const bool dbg = std::getenv("DNLP_BATCH_DEBUG") != nullptr;
    const double tdbg0 = now_sec();
    auto mark = [&](const char* what) { if (dbg) std::fprintf(stderr, "[batch] %-22s %.4f s\n", what, now_sec() - tdbg0); };
    const Tape<HipExec>& t = *tape;
    DNLP_HIP_CHECK(hipSetDevice(ex->device));
    release();
    BatchArgs a;
    a.base = t;           // slice: the view with the shared exec-space index arrays
    a.base.segs = d_segs; a.base.red_segs = d_red; a.base.sparse = d_sparse;
    a.base.dense_ptr = nullptr; a.base.dense_ld = nullptr; a.base.blocks = nullptr;
    a.lay = lay;
    a.batch = batch;
    a.opt = opt;
    a.slabs = dalloc<double>(static_cast<size_t>(batch) * static_cast<size_t>(lay.total));
    if (theta) {
      const size_t nth = static_cast<size_t>(batch) * static_cast<size_t>(aff_P);
      if (nth > aff_theta_cap) {
        if (aff_theta) DNLP_HIP_CHECK(hipFree(aff_theta));
        aff_theta = nullptr;
        DNLP_HIP_CHECK(hipMalloc(&aff_theta, (nth ? nth : 1) * sizeof(double)));
        aff_theta_cap = nth;
      }
      if (nth) DNLP_HIP_CHECK(hipMemcpyAsync(aff_theta, theta, nth * sizeof(double), hipMemcpyHostToDevice, stream));
      const dim3 grid(static_cast<unsigned>((lay.total + 255) / 256), static_cast<unsigned>(batch));
      hipLaunchKernelGGL(batch_affine_expand_kernel, grid, dim3(256), 0, stream, a.slabs, lay.total, batch, aff_P, aff_theta,
                         aff_theta0, aff_d0, aff_indptr, aff_idx, aff_val);
      DNLP_LAUNCH_CHECK();
      mark("slab generated on device");
    } else {
      // host gather: the per-segment parameters become per-flat-row parameters
      slab.resize(static_cast<size_t>(batch) * static_cast<size_t>(lay.total));
      const i64 head = 1 + (t.N + t.Z) + t.m + t.nnzJ + t.G.nnz + t.Mg.nnz + t.Mw.nnz + t.MJ.nnz + t.MH.nnz;
      const i64 tail = 3 * t.N + 2 * t.m;
      for (int k = 0; k < batch; ++k) {
        const double* src = data + static_cast<i64>(k) * stride;
        double* dst = slab.data() + static_cast<i64>(k) * lay.total;
        std::copy(src, src + head, dst);
        const double *sp = src + head, *sp2 = sp + t.nseg;
        for (i64 f = 0; f < t.nflat; ++f) {
          dst[lay.fp + f] = sp[t.h_flat_seg[static_cast<size_t>(f)]];
          dst[lay.fp2 + f] = sp2[t.h_flat_seg[static_cast<size_t>(f)]];
        }
        std::copy(sp2 + t.nseg, sp2 + t.nseg + tail, dst + lay.x0);
      }
      mark("slab built");
      DNLP_HIP_CHECK(hipMemcpy(a.slabs, slab.data(), slab.size() * sizeof(double), hipMemcpyHostToDevice));
      mark("slab uploaded");
    }
    last_wave = 0;
    last_wave_refused = 0;
    // (templates whose state does not fit LDS keep the generic kernel, which gives such an instance four wavefronts —
    //  measured at 1024 instances: power flow 5.3 k problems/s against 3.9 k through the one-wavefront global-memory form,
    //  path planning 4.5 against 4.8; DNLP_BATCH_WAVE=2 takes the wavefront solver for them too, 0 never)
    const int wave_env = std::getenv("DNLP_BATCH_WAVE") ? std::atoi(std::getenv("DNLP_BATCH_WAVE")) : 1;
    bool take_wave = allow_wave && wave_env != 0 && wave_prepare();
    if (take_wave && wave_env != 2) { int nw = 0, sl = 0, pl = 0; wave_form(nw, sl, pl); take_wave = sl != 0 || wave_wg_prepare(batch); }
    if (take_wave) {
      solve_wave(a, batch, data, theta, opt, x_out, obj_out, multg_out, zl_out, zu_out, status_out, iters_out, nfact_out, seconds, times_out);
      return;
    }
    const i64 n = t.N + t.m, ld = (n + 7) / 8 * 8;
    // KKT matrix in LDS when it fits beside the static reduction scratch (160 KB per workgroup)
    a.use_sparse = have_sparse ? 1 : 0;
    a.sp = dev_plan;
    // working set of one instance: the "matrix" (dense KKT, or the sparse factor when it is large)
    // and the vectors (model + interior-point state; counts follow Model::init_view / Ipm::allocate)
    const size_t sparse_vals = have_sparse ? static_cast<size_t>(2 * dev_plan.nvals + 3 * dev_plan.nblk + 32) : 0;
    const bool sparse_big = sparse_vals * 8 >= 16384;
    const size_t kbytes = have_sparse ? (sparse_big ? ((sparse_vals * 8 + 63) & ~static_cast<size_t>(63)) : 0)
                                      : (((static_cast<size_t>(ld) * n + 256) * 8 + 63) & ~static_cast<size_t>(63));
    const size_t vdoubles = static_cast<size_t>(26 * t.N + 36 * t.m + 2 * t.Z + t.nd + t.nh + t.nnzH + t.nnzJ + 2 * n + 64) +
                            (n > 256 && n <= 2048 ? static_cast<size_t>(18 * n + 128) : 0) +      // (panel-blocked factorisation: W, permutation, pivot types)
                            (have_sparse && !sparse_big ? sparse_vals : 0);
    const size_t vbytes = ((vdoubles * 8 * 21 / 20) + 96 * 64 + 255) & ~static_cast<size_t>(255);
    // LDS reservation for the vectors: tight (the allocator spills to the global slab, which always
    // has room for the whole working set, so an underestimate costs latency on a few arrays only)
    const size_t vdoubles_lds = vdoubles - static_cast<size_t>(have_sparse ? 2 * n : 0);
    size_t vbytes_lds = (vdoubles_lds * 8 + 80 * 32 + 63) & ~static_cast<size_t>(63);
    if (const char* e = std::getenv("DNLP_BATCH_VLDS_KB")) {       // experiment: cap the vectors' LDS share (the rest spills to the global slab)
      const size_t cap_b = static_cast<size_t>(std::atoi(e)) * 1024;
      if (cap_b >= 4096 && cap_b < vbytes_lds) vbytes_lds = cap_b & ~static_cast<size_t>(63);
    }
    // lanes per instance: one wavefront up to order 256 (factorisation and solves are
    // single-wavefront there, vectors are at most a few hundred long), four above
    // sparse instances have no dense matrix to factor, so one wavefront suffices at any order;
    // when the batch does not even fill the CUs, large instances get four wavefronts each instead
    // (a one-off solve of order ~2000 is then ~3x faster)
    // Sparse instances of order > 512 get four wavefronts each while the batch is at most four instances per compute unit
    // (1024 on the MI355X; DNLP_BATCH_WIDE_MAX): such a launch is a few rounds of its slowest instances (power flow: 104
    // iterations against a mean of 17.7; one to three instances resident per compute unit), so the ~3x faster instance wins:
    // 1024 power-flow instances 0.424 -> 0.341 s, path planning 0.306 -> 0.241 s; at 4096 instances the one-wavefront form
    // is as fast or faster (0.59 / 0.63 s against 0.59 / 0.73 s).  The kernel — and with it the order of the sums, the
    // last bits of a result — therefore depends on the launch's size for these templates: results repeat bit for bit for
    // the same launch plan, not across shard sizes.
    if (this->ncu == 0) {
      int v = 0;
      DNLP_HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ex->device));
      this->ncu = v;
    }
    const int wide_max = std::getenv("DNLP_BATCH_WIDE_MAX") ? std::atoi(std::getenv("DNLP_BATCH_WIDE_MAX")) : 4 * this->ncu;
    const bool wave = n <= 256 || (have_sparse && !(n > 512 && batch <= wide_max));
    const void* kern = wave ? reinterpret_cast<const void*>(batch_solve_kernel<64>)
                            : reinterpret_cast<const void*>(batch_solve_kernel<256>);
    const int nthreads = wave ? 64 : 256;
    hipFuncAttributes fa;
    DNLP_HIP_CHECK(hipFuncGetAttributes(&fa, kern));
    const size_t lds_max = 160 * 1024 - fa.sharedSizeBytes - 512;
    // LDS plan.  Measured on MI355X (profiles/r01_c5_batch_lds_modes.json): throughput follows the
    // number of instances resident per CU first (one wavefront each, at most one per SIMD with
    // this kernel's register budget), the per-iteration latency second (both in LDS < matrix in
    // LDS < vectors in LDS < nothing).  So: the richest plan among those with the most instances
    // per CU.  DNLP_BATCH_LDS=0..3 overrides, for experiments.
    // staging for the dense single-wavefront solves is needed whenever the dense path can run
    const bool dense_possible = !have_sparse || (n <= 512 && !force_sparse);
    // (the staging arrays of the dense wavefront solves live in LDS only where the dense path is THE path; a sparse
    //  instance that may fall back to it — rarely, and with the lazy fallback only in the retry rungs — gets them at the
    //  end of its global slab instead: 6 KB of LDS per instance that the vectors and the plan's index arrays use)
    const size_t stage_bytes = !have_sparse ? static_cast<size_t>(BlockExecT<64>::kWaveSolveMax) * 12 : 0;
    const size_t stage_global = (have_sparse && dense_possible) ? static_cast<size_t>(BlockExecT<64>::kWaveSolveMax) * 12 + 64 : 0;
    const int slots_max = wave ? 4 : 1;
    // One-wavefront sparse instances whose vectors exceed LDS by far (AC power flow: 413 KB) still take the share of a
    // four-per-CU plan: the allocator hands out LDS in allocation order and spills the rest to the global slab
    // (1024 power-flow instances 2.24 -> 2.34 k problems/s; path planning unchanged; DNLP_BATCH_PART_LDS=0 disables).
    if (wave && have_sparse && !(std::getenv("DNLP_BATCH_PART_LDS") && std::atoi(std::getenv("DNLP_BATCH_PART_LDS")) == 0)) {
      const size_t room = (160 * 1024) / static_cast<size_t>(slots_max);
      const size_t fixed = stage_bytes + fa.sharedSizeBytes + 256 + 512;
      if (room > fixed + 4096 && 3 * (room - fixed) < 2 * vbytes_lds) {      // (below two thirds: the trim further down handles the rest)
        const size_t cap = (room - fixed) & ~static_cast<size_t>(63);
        if (vbytes_lds > cap) vbytes_lds = cap;
      }
    }
    auto slots = [&](int md_) {
      const size_t dyn = stage_bytes + (md_ & 1 ? kbytes : 0) + (md_ & 2 ? vbytes_lds : 0);
      if (dyn > lds_max) return 0;
      const size_t per = dyn + fa.sharedSizeBytes + 256;
      return static_cast<int>(std::min<size_t>(slots_max, (160 * 1024) / per));
    };
    // measured per-iteration latency relative to "nothing in LDS" (profiles/r01_c5_batch_lds_modes.json
    // and the sparse re-measurement): throughput ~ resident instances / latency
    const double lat[4] = {1.0, 0.65, 0.53, 0.5};
    int mode = 0;
    double best = slots(0) / lat[0];
    for (int cand : {1, 2, 3}) {
      if (have_sparse && !sparse_big && (cand & 1)) continue;      // no separate "matrix" to place
      const double sc = slots(cand) / lat[cand];
      if (sc > best * 1.02) { best = sc; mode = cand; }
    }
    // A batch that mode 3 (factor AND vectors in LDS, fewer instances per CU) gets through in two rounds is made of
    // iteration latency, not of residency: circle packing n = 10, 1024 instances: 2 per CU with both in LDS 16.9 k
    // problems/s, 4 per CU with the vectors only 14.7 k (8192 instances keep the 4-per-CU plan: 30.5 k/s)
    if (wave && have_sparse && sparse_big && mode != 3 && slots(3) >= 1) {
      if (this->ncu == 0) {
        int v = 0;
        DNLP_HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ex->device));
        this->ncu = v;
      }
      if (static_cast<i64>(batch) <= 2 * static_cast<i64>(this->ncu) * slots(3)) mode = 3;
    }
    if (const char* e = std::getenv("DNLP_BATCH_LDS")) {
      const int want = std::atoi(e);
      if (want >= 0 && want <= 3 && slots(want) > 0) mode = want;
    }
    // One more resident instance for the price of the last third of the vectors: the allocator hands out LDS
    // in allocation order (iterate, directions, residuals first) and spills the tail to the global slab.
    // Localization, 65 536 instances: 3 per CU with all 41 KB of vectors in LDS 209.8 k problems/s, 4 per CU with
    // 28 KB 220.0 k, 4 per CU with 24 KB 168.7 k (the hot arrays start to spill) -- so never below two thirds.
    if ((mode & 2) && wave && !std::getenv("DNLP_BATCH_VLDS_KB")) {
      const int s_now = slots(mode);
      if (s_now >= 1 && s_now < slots_max) {
        // (+ room for the LDS copy of the plan's index arrays.  While the 6 KB staging area of the dense fallback still sat
        //  in LDS this was a choice — plan arrays OR 2.6 KB more of the vectors, the first better for short queues, the
        //  second for long ones: 65 536 localization instances 246 against 268 k problems/s — with the staging area in the
        //  global slab both fit: 290 k/s; tools/micro/batch_lds_share_sweep.sh)
        size_t idx_reserve = 3072;
        if (have_sparse) {
          const IdxBytes ib = index_bytes();
          int want = 7;
          if (const char* e = std::getenv("DNLP_BATCH_IDX_LDS")) want = std::atoi(e) & 7;
          idx_reserve = (want & 1 ? ib.solve : 0) + (want & 2 ? ib.coo : 0) + (want & 4 ? ib.factor : 0) + 128;
        }
        const size_t fixed = stage_bytes + (mode & 1 ? kbytes : 0) + fa.sharedSizeBytes + 256 + idx_reserve;
        const size_t room = (160 * 1024) / static_cast<size_t>(s_now + 1);
        if (room > fixed) {
          const size_t want = (room - fixed) & ~static_cast<size_t>(63);
          if (want < vbytes_lds && 3 * want >= 2 * vbytes_lds) vbytes_lds = want;
        }
      }
    }
    a.lds_mode = mode;
    a.lds_stage_bytes = static_cast<unsigned>((stage_bytes + 63) & ~static_cast<size_t>(63));
    a.lds_bytes = static_cast<unsigned>(a.lds_stage_bytes + (mode & 1 ? kbytes : 0) + (mode & 2 ? vbytes_lds : 0));
    // LDS copies of the index arrays every instance reads — when they do not cost a resident instance
    a.plan_stage_bytes = 0;
    a.plan_stage_factor = 0;
    a.plan_rows = plan_rows;
    if (have_sparse && !std::getenv("DNLP_BATCH_NO_PLAN_LDS")) {
      const IdxBytes ib = index_bytes();
      int want = 7;
      if (const char* e = std::getenv("DNLP_BATCH_IDX_LDS")) want = std::atoi(e) & 7;
      const size_t per0 = a.lds_bytes + fa.sharedSizeBytes + 256;
      const size_t cap = 160 * 1024;
      const size_t s0 = std::min<size_t>(slots_max, cap / per0);
      // most valuable first: the solves' arrays (~10 level phases per solve, ~10 solves per iteration), the product
      // indices (~15 products per iteration), the factorisation's arrays (1.3 factorisations per iteration)
      size_t add = 0;
      const size_t part[3] = {ib.solve, ib.coo, ib.factor};
      for (int k = 0; k < 3; ++k) {
        if (!(want & (1 << k))) continue;
        const size_t per = per0 + add + part[k] + 64;
        if (per <= cap && std::min<size_t>(slots_max, cap / per) == s0) { add += part[k]; a.plan_stage_factor |= 1 << k; }
      }
      if (a.plan_stage_factor) {
        a.plan_stage_bytes = static_cast<unsigned>((add + 63) & ~static_cast<size_t>(63));
        a.lds_bytes += a.plan_stage_bytes;
      }
    }
    // room for the dense matrix of an instance that falls back from the sparse path (global memory)
    const bool fb = have_sparse && n <= 512 && !force_sparse;
    a.fallback_max_n = fb ? 512 : 0;
    const size_t fbbytes = fb ? (((static_cast<size_t>(ld) * n + 256) * 8 + 2 * static_cast<size_t>(n) * 8 + 4096 + 63) & ~static_cast<size_t>(63)) : 0;
    a.ws_per_block = 256 + vbytes + (mode & 1 ? 0 : kbytes) + fbbytes + stage_global;
    a.stage_global_bytes = static_cast<unsigned>(stage_global);
    DNLP_HIP_CHECK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(a.lds_bytes)));
    int per_cu = 1, ncu = 256;
    if (wave) DNLP_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, batch_solve_kernel<64>, nthreads, a.lds_bytes));
    else DNLP_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, batch_solve_kernel<256>, nthreads, a.lds_bytes));
    if (this->ncu == 0) {
      int v = 0;
      DNLP_HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ex->device));
      this->ncu = v;
    }
    ncu = this->ncu;
    if (per_cu < 1) per_cu = 1;
    if (per_cu > 8) per_cu = 8;
    if (const char* e = std::getenv("DNLP_BATCH_PER_CU")) { const int w = std::atoi(e); if (w >= 1 && w <= 8) per_cu = w; }
    int grid = std::min(batch, ncu * per_cu);
    // Packed form (batch_solve_packed_kernel): four instances per workgroup share one LDS copy of the index arrays and
    // keep ALL their vectors in LDS — taken when that fits a compute unit and the regular plan holds no more instances
    // per compute unit than four (DNLP_BATCH_PACKED=0 disables; an instance whose vectors did not fit reports -198 and
    // the handle goes back to the regular kernel for good)
    constexpr int kPackedWaves = 4;
    unsigned pk_wave_bytes = 0;
    bool packed = false;
    if (wave && have_sparse && !sparse_big && !packed_disabled && per_cu <= kPackedWaves &&
        !(std::getenv("DNLP_BATCH_PACKED") && std::atoi(std::getenv("DNLP_BATCH_PACKED")) == 0)) {
      const IdxBytes ib = index_bytes();
      const size_t idx = (ib.solve + ib.coo + ib.factor + 63) & ~static_cast<size_t>(63);
      const size_t wv = (packed_vector_bytes() + 64 + 15) & ~static_cast<size_t>(15);
      // (a single vector of 16 KB or more is never placed in LDS by the packed kernel's allocator: not a packed template)
      const size_t widest = 8 * static_cast<size_t>(std::max<i64>(std::max<i64>(t.N + t.Z, t.N + t.m), std::max<i64>(dev_plan.nvals + 3 * dev_plan.nblk + 8, std::max<i64>(t.nnzJ, t.nnzH))));
      hipFuncAttributes fp;
      DNLP_HIP_CHECK(hipFuncGetAttributes(&fp, reinterpret_cast<const void*>(batch_solve_packed_kernel<kPackedWaves>)));
      if (widest < 16384 && idx + kPackedWaves * wv + fp.sharedSizeBytes + 512 <= 160 * 1024) {
        packed = true;
        pk_wave_bytes = static_cast<unsigned>(wv);
        a.plan_stage_bytes = static_cast<unsigned>(idx);
        a.plan_stage_factor = 7;
        a.lds_stage_bytes = 0;
        a.lds_bytes = static_cast<unsigned>(idx + kPackedWaves * wv);
        a.stage_global_bytes = static_cast<unsigned>(static_cast<size_t>(BlockExecT<64>::kWaveSolveMax) * 12 + 64);
        a.ws_per_block = 256 + fbbytes + a.stage_global_bytes + 64;       // per WAVEFRONT here: only a dense fallback uses it
        DNLP_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(batch_solve_packed_kernel<kPackedWaves>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(a.lds_bytes)));
        grid = std::min((batch + kPackedWaves - 1) / kPackedWaves, ncu);
        per_cu = kPackedWaves;
        mode = 3;
      }
    }
    last_grid = grid; last_threads = packed ? 64 * kPackedWaves : nthreads; last_lds_mode = mode; last_per_cu = per_cu; last_packed = packed;
    if (dbg) std::fprintf(stderr, "[batch] plan: %s threads %d lds_mode %d static LDS %zu B dynamic %u B (index arrays %u B, per wavefront %u B) -> %d per CU, grid %d\n",
                          packed ? "PACKED" : "regular", last_threads, mode, static_cast<size_t>(fa.sharedSizeBytes), a.lds_bytes, a.plan_stage_bytes,
                          pk_wave_bytes, per_cu, grid);
    a.ws = dalloc<char>(static_cast<size_t>(grid) * (packed ? kPackedWaves : 1) * a.ws_per_block);
    a.x_out = dalloc<double>(static_cast<size_t>(batch) * t.N);
    a.obj_out = dalloc<double>(static_cast<size_t>(batch));
    a.multg_out = multg_out ? dalloc<double>(static_cast<size_t>(batch) * t.m) : nullptr;
    a.zl_out = zl_out ? dalloc<double>(static_cast<size_t>(batch) * t.N) : nullptr;
    a.zu_out = zu_out ? dalloc<double>(static_cast<size_t>(batch) * t.N) : nullptr;
    a.status_out = dalloc<int>(static_cast<size_t>(batch));
    a.iters_out = dalloc<int>(static_cast<size_t>(batch));
    a.nfact_out = dalloc<int>(static_cast<size_t>(batch));
    a.times_out = times_out ? dalloc<double>(4 * static_cast<size_t>(batch)) : nullptr;
    if (ws_batch == batch && opt.warm_start) {
      double* g = dalloc<double>(h_ws_g.size());
      double* l = dalloc<double>(h_ws_l.size());
      double* u = dalloc<double>(h_ws_u.size());
      if (!h_ws_g.empty()) DNLP_HIP_CHECK(hipMemcpy(g, h_ws_g.data(), h_ws_g.size() * 8, hipMemcpyHostToDevice));
      DNLP_HIP_CHECK(hipMemcpy(l, h_ws_l.data(), h_ws_l.size() * 8, hipMemcpyHostToDevice));
      DNLP_HIP_CHECK(hipMemcpy(u, h_ws_u.data(), h_ws_u.size() * 8, hipMemcpyHostToDevice));
      a.ws_g = g; a.ws_l = l; a.ws_u = u;
    }
    ws_batch = 0;
    a.next = dalloc<int>(1);
    DNLP_HIP_CHECK(hipMemsetAsync(a.next, 0, sizeof(int), stream));
    // longest-first only for a re-solve of the SAME rows (warm starts, repeated what-if solves): a different batch of
    // equal size must not inherit an order derived from unrelated iteration counts, and takes its instances first-come
    const uint64_t key = theta ? rows_hash(theta, static_cast<size_t>(batch) * static_cast<size_t>(aff_P))
                               : rows_hash(data, static_cast<size_t>(batch) * static_cast<size_t>(stride));
    last_order_lpt = false;
    if (static_cast<int>(prev_iters.size()) == batch && key == prev_key && batch > grid && !std::getenv("DNLP_BATCH_FIFO")) {
      last_order_lpt = true;
      std::vector<int> ord(static_cast<size_t>(batch));
      for (int k = 0; k < batch; ++k) ord[static_cast<size_t>(k)] = k;
      std::stable_sort(ord.begin(), ord.end(), [&](int p, int q) { return prev_iters[static_cast<size_t>(p)] > prev_iters[static_cast<size_t>(q)]; });
      int* d_ord = dalloc<int>(static_cast<size_t>(batch));
      DNLP_HIP_CHECK(hipMemcpyAsync(d_ord, ord.data(), sizeof(int) * static_cast<size_t>(batch), hipMemcpyHostToDevice, stream));
      DNLP_HIP_CHECK(hipStreamSynchronize(stream));     // `ord` is a local
      a.order = d_ord;
    }
    mark("plan + buffers");
    hipEvent_t e0, e1;
    DNLP_HIP_CHECK(hipEventCreate(&e0));
    DNLP_HIP_CHECK(hipEventCreate(&e1));
    DNLP_HIP_CHECK(hipEventRecord(e0, stream));
    if (packed) hipLaunchKernelGGL(batch_solve_packed_kernel<kPackedWaves>, dim3(static_cast<unsigned>(grid)), dim3(64 * kPackedWaves), a.lds_bytes,
                                   stream, a, pk_wave_bytes);
    else if (wave) hipLaunchKernelGGL(batch_solve_kernel<64>, dim3(static_cast<unsigned>(grid)), dim3(64), a.lds_bytes, stream, a);
    else hipLaunchKernelGGL(batch_solve_kernel<256>, dim3(static_cast<unsigned>(grid)), dim3(256), a.lds_bytes, stream, a);
    DNLP_LAUNCH_CHECK();
    DNLP_HIP_CHECK(hipEventRecord(e1, stream));
    DNLP_HIP_CHECK(hipStreamSynchronize(stream));
    float ms = 0.f;
    DNLP_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    mark("kernel done");
    if (seconds) *seconds = 1e-3 * ms;
    auto down = [&](void* h, const void* d, size_t bytes) { if (h && bytes) DNLP_HIP_CHECK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost)); };
    down(x_out, a.x_out, sizeof(double) * static_cast<size_t>(batch) * t.N);
    down(obj_out, a.obj_out, sizeof(double) * batch);
    down(multg_out, a.multg_out, sizeof(double) * static_cast<size_t>(batch) * t.m);
    down(zl_out, a.zl_out, sizeof(double) * static_cast<size_t>(batch) * t.N);
    down(zu_out, a.zu_out, sizeof(double) * static_cast<size_t>(batch) * t.N);
    down(status_out, a.status_out, sizeof(int) * batch);
    down(iters_out, a.iters_out, sizeof(int) * batch);
    down(nfact_out, a.nfact_out, sizeof(int) * batch);
    down(times_out, a.times_out, sizeof(double) * 4 * batch);
    if (packed) {
      std::vector<int> st_local;
      const int* st_host = status_out;
      if (!st_host) { st_local.resize(static_cast<size_t>(batch)); down(st_local.data(), a.status_out, sizeof(int) * batch); st_host = st_local.data(); }
      bool refused = false;
      for (int k = 0; k < batch && !refused; ++k) refused = st_host[k] == -198;
      if (refused) {
        // the estimate of the vectors' bytes was short for this tape: the regular kernel, from now on
        packed_disabled = true;
        if (a.ws_g) ws_batch = batch;      // (a warm-started batch stays warm in the retry: h_ws_* still hold its multipliers)
        release();
        solve_impl(batch, data, theta, opt, x_out, obj_out, multg_out, zl_out, zu_out, status_out, iters_out, nfact_out, seconds, times_out, allow_wave);
        return;
      }
    }
    if (iters_out) { prev_iters.assign(iters_out, iters_out + batch); prev_key = key; }
    pack_rows(batch, a.x_out, a.obj_out, a.status_out, a.iters_out);
    release();
    mark("results copied");
  }
  // The same launch through the wavefront solver (the caller has generated a.slabs).  Instances it refuses
  // (kWaveNeedsGeneric: a structurally singular static pivot sequence, the generic kernel's Bunch-Kaufman switch) are
  // solved by the generic kernel in a second, small launch and their results merged.
  // launch form of this template: wavefronts per workgroup, state in LDS, plan in LDS — the richest that fits 160 KB
  // (state and plan in LDS: as many wavefronts per workgroup — per compute unit — as fit, up to eight.  Measured on the
  //  MI355X, circle packing n = 4 at 16 384 instances: four wavefronts per compute unit 382 k problems/s, six 518 k, and an
  //  instance's own wall time only 5 % longer — a second wavefront on a SIMD fills the cycles the first one waits for LDS)
  int wf_nw = 0, wf_sl = 0, wf_pl = 0;      // the form of this template, found once (hipFuncGetAttributes per candidate)
  int wf_nw_glb = 0;                         // wavefronts per workgroup that fit with the plan in global memory (up to four)
  int wf_per_cu[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // ... and the occupancy of <nw, LDS, LDS> / of the fallback form at [0]
  size_t wave_static_full(int nw) {
    switch (nw) {
      case 8: return wave_static_lds<8, true, true>();
      case 7: return wave_static_lds<7, true, true>();
      case 6: return wave_static_lds<6, true, true>();
      case 5: return wave_static_lds<5, true, true>();
      case 4: return wave_static_lds<4, true, true>();
      case 3: return wave_static_lds<3, true, true>();
      case 2: return wave_static_lds<2, true, true>();
      default: return wave_static_lds<1, true, true>();
    }
  }
  void wave_form(int& nw, int& sl, int& pl) {
    if (wf_nw > 0 && !std::getenv("DNLP_WAVE_FORM")) { nw = wf_nw; sl = wf_sl; pl = wf_pl; return; }
    const WaveHdr& h = *reinterpret_cast<const WaveHdr*>(wave_blk.data());
    const size_t plan_b = wave_fits16 ? ((static_cast<size_t>(h.total) * 2 + 15) & ~static_cast<size_t>(15)) : (static_cast<size_t>(1) << 30);
    const size_t state_b = static_cast<size_t>(h.state_doubles) * 8;
    const size_t cap = 160 * 1024 - 256;
    nw = 0;
    for (int k = 8; k >= 1 && nw == 0; --k)
      if (plan_b + static_cast<size_t>(k) * state_b + wave_static_full(k) <= cap) { nw = k; sl = 1; pl = 1; }
    // with the plan left in global memory (read through L1 / L2: measured +23 % per iteration on circle packing n = 10) more
    // wavefronts may fit — worth it when a launch has the instances for them (solve_wave decides per launch)
    wf_nw_glb = 0;
    for (int k = 4; k >= 1 && wf_nw_glb == 0; --k) {
      const size_t st = k == 4 ? wave_static_lds<4, true, false>() : k == 3 ? wave_static_lds<3, true, false>() : k == 2 ? wave_static_lds<2, true, false>()
                                                                                                                  : wave_static_lds<1, true, false>();
      if (wave_fits16 && static_cast<size_t>(k) * state_b + st <= cap) wf_nw_glb = k;      // (state in LDS: the 16-bit copy of the plan)
    }
    if (nw > 0) {}
    else if (wf_nw_glb > 0) { nw = wf_nw_glb; sl = 1; pl = 0; }
    else { nw = 4; sl = 0; pl = 0; }
    if (const char* e = std::getenv("DNLP_WAVE_FORM")) {        // experiments: "811" .. "111", "210", "110", "400"
      const int f = std::atoi(e);
      if (f > 0) {
        nw = f / 100; sl = (f / 10) % 10; pl = f % 10;
        if ((pl || sl) && !wave_fits16) throw std::runtime_error("wavefront solver: this plan does not fit 16-bit tables");
      }
    } else { wf_nw = nw; wf_sl = sl; wf_pl = pl; }
  }
  template <int NW, bool SL, bool PL>
  void launch_wave(const WaveArgs& w, int grid, unsigned lds, hipStream_t stream) {
    const void* k = reinterpret_cast<const void*>(wave_batch_kernel<NW, SL, PL>);
    DNLP_HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    hipLaunchKernelGGL((wave_batch_kernel<NW, SL, PL>), dim3(static_cast<unsigned>(grid)), dim3(64 * NW), lds, stream, w);
  }
  template <int NW, bool SL, bool PL>
  int wave_occupancy(unsigned lds) {
    const void* k = reinterpret_cast<const void*>(wave_batch_kernel<NW, SL, PL>);
    DNLP_HIP_CHECK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    int v = 1;
    DNLP_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, wave_batch_kernel<NW, SL, PL>, 64 * NW, lds));
    return v < 1 ? 1 : v;
  }
  template <int NW, bool SL, bool PL>
  size_t wave_static_lds() {
    hipFuncAttributes fa;
    DNLP_HIP_CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(wave_batch_kernel<NW, SL, PL>)));
    return fa.sharedSizeBytes;
  }
  void solve_wave(const BatchArgs& a, int batch, const double* data, const double* theta, const IpmOptions& opt, double* x_out, double* obj_out,
                  double* multg_out, double* zl_out, double* zu_out, int* status_out, int* iters_out, int* nfact_out,
                  double* seconds, double* times_out) {
    const Tape<HipExec>& t = *tape;
    const WaveHdr& h = *reinterpret_cast<const WaveHdr*>(wave_blk.data());
    if (this->ncu == 0) {
      int v = 0;
      DNLP_HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, ex->device));
      this->ncu = v;
    }
    int nw = 0, sl = 0, pl = 0;
    wave_form(nw, sl, pl);
    const size_t plan_b = wave_fits16 ? ((static_cast<size_t>(h.total) * 2 + 15) & ~static_cast<size_t>(15)) : 0;
    const size_t state_b = static_cast<size_t>(h.state_doubles) * 8;
    WaveArgs w;
    w.blk = d_wave_blk; w.blk16 = d_wave_blk16; w.blk_ints = h.total;
    w.gen = d_wave_gen; w.gen_words = wave_gen_words;
    w.rows = a.slabs; w.row_doubles = lay.total;
    w.batch = batch;
    w.state_doubles = h.state_doubles;
    w.opt = opt;
    const i64 n = t.N + t.m;
    w.fallback_max_n = (n <= 512 && !force_sparse) ? 512 : 0;
    // a launch that does not fill the chip spreads out: no more wavefronts per compute unit than instances per compute unit
    // (1024 instances on 256 units: four each, whatever would fit — a lone wavefront on its SIMD is the fastest instance)
    bool spec = false;
    if (sl && !std::getenv("DNLP_WAVE_FORM")) {
      const int want = std::max(1, (batch + this->ncu - 1) / this->ncu);
      if (pl && want > nw && wf_nw_glb > nw) { nw = std::min(want, wf_nw_glb); pl = 0; }      // more instances than fit beside the plan
      else nw = std::min(nw, want);
      // the per-template kernel when the template has one and it holds (nearly) as many wavefronts per compute unit as the
      // library's own form would run (it stages less: the level machinery's tables are generated code): its static LDS (plan prefix +
      // work tables + wave_spec_nw shares) fills the compute unit, a launch puts as many wavefronts into a workgroup as it
      // has instances per compute unit
      const bool spec_forced = std::getenv("DNLP_WAVE_SPEC") && std::atoi(std::getenv("DNLP_WAVE_SPEC")) == 1;
      // (a wavefront of the per-template kernel gets through an iteration ~1.45 x faster than one of the library's: it is
      //  taken as long as it holds at least three quarters of the wavefronts per compute unit the library's form would run)
      if (wave_spec_prepare(batch) && (spec_forced || 4 * std::min(wave_spec_nw, want) >= 3 * nw)) {
        spec = true;
        nw = std::min(wave_spec_nw, want); pl = wave_spec_tables_global ? 0 : 1;
        w.gen = d_wave_gen; w.gen_words = wave_gen_words;
      }
    }
    // a template whose state exceeds LDS: a workgroup per instance through the generated phases when the kernel is there
    // ... and a template whose state fits, but only once or twice per compute unit: from six instances per compute unit on (measured
    // on circle packing n = 10: 1280 instances 47.2 k problems/s through the one-wavefront kernel / 46.4 k through this one, 1536: 50.2 / 55.1 k)
    // (wave_wg_prepare: two wavefronts per instance, four workgroups per unit).  DNLP_WAVE_WG_SMALL=0 never, =1 at any size.
    bool wg_small = false;
    if (sl && !std::getenv("DNLP_WAVE_FORM")) {
      const int per_unit = spec ? wave_spec_nw : nw;                    // (wavefronts = instances per compute unit of the kernel above)
      const char* ws = std::getenv("DNLP_WAVE_WG_SMALL");
      const int wsm = ws ? std::atoi(ws) : -1;
      wg_small = wsm == 2 || (wsm != 0 && per_unit <= 2 && (wsm == 1 || batch >= 6 * this->ncu));      // (2: whatever the template — experiments)
    }
    const bool wg = (!sl || wg_small) && !std::getenv("DNLP_WAVE_FORM") && wave_wg_prepare(batch);
    if (wg) spec = false;
    if (wg) { sl = 0; pl = 0; }      // (the kernel has its own static LDS; the form reported is 100 x wavefronts per instance)
    if (wg) { nw = wave_wg_nw; w.gen = d_wave_wg_gen; w.gen_words = wave_wg_gen_words; w.blk = d_wave_wg_blk; w.blk_ints = static_cast<int>(wave_wg_blk.size()); }
    int per_cu = 1;
    const unsigned lds = static_cast<unsigned>((pl ? plan_b : 0) + (sl ? static_cast<size_t>(nw) * state_b : 0));
    const int form = 100 * nw + 10 * sl + pl;
    int cached_none = 0;
    int& cached = (sl && pl) ? wf_per_cu[nw] : cached_none;
    if (spec) per_cu = 1;
    else if (wg) per_cu = wave_wg_per_cu;
    else if (cached > 0 && !std::getenv("DNLP_WAVE_FORM")) per_cu = cached;
    else switch (form) {
      case 811: per_cu = wave_occupancy<8, true, true>(lds); break;
      case 711: per_cu = wave_occupancy<7, true, true>(lds); break;
      case 611: per_cu = wave_occupancy<6, true, true>(lds); break;
      case 511: per_cu = wave_occupancy<5, true, true>(lds); break;
      case 411: per_cu = wave_occupancy<4, true, true>(lds); break;
      case 311: per_cu = wave_occupancy<3, true, true>(lds); break;
      case 211: per_cu = wave_occupancy<2, true, true>(lds); break;
      case 111: per_cu = wave_occupancy<1, true, true>(lds); break;
      case 410: per_cu = wave_occupancy<4, true, false>(lds); break;
      case 310: per_cu = wave_occupancy<3, true, false>(lds); break;
      case 210: per_cu = wave_occupancy<2, true, false>(lds); break;
      case 110: per_cu = wave_occupancy<1, true, false>(lds); break;
      case 400: per_cu = wave_occupancy<4, false, false>(lds); break;
      default: throw std::runtime_error("wavefront solver: no such launch form");
    }
    if (!std::getenv("DNLP_WAVE_FORM") && !spec && !wg) cached = per_cu;
    if (const char* e = std::getenv("DNLP_WAVE_PER_CU")) { const int v = std::atoi(e); if (v >= 1 && v <= 16) per_cu = v; }
    int grid = wg ? std::min(batch, ncu * per_cu) : std::min((batch + nw - 1) / nw, ncu * per_cu);
    if (grid < 1) grid = 1;
    if (wg) w.state = dalloc<double>(static_cast<size_t>(grid) * static_cast<size_t>(h.state_doubles));
    else if (!sl) w.state = dalloc<double>(static_cast<size_t>(grid) * static_cast<size_t>(nw) * static_cast<size_t>(h.state_doubles));
    w.park_doubles = wave_park_doubles(t.N, t.m);
    w.park = dalloc<double>(static_cast<size_t>(grid) * static_cast<size_t>(wg ? 1 : spec ? wave_spec_nw : nw) * static_cast<size_t>(w.park_doubles));
    w.x_out = dalloc<double>(static_cast<size_t>(batch) * t.N);
    w.obj_out = dalloc<double>(static_cast<size_t>(batch));
    w.multg_out = multg_out ? dalloc<double>(static_cast<size_t>(batch) * t.m) : nullptr;
    w.zl_out = zl_out ? dalloc<double>(static_cast<size_t>(batch) * t.N) : nullptr;
    w.zu_out = zu_out ? dalloc<double>(static_cast<size_t>(batch) * t.N) : nullptr;
    w.status_out = dalloc<int>(static_cast<size_t>(batch));
    w.iters_out = dalloc<int>(static_cast<size_t>(batch));
    w.nfact_out = dalloc<int>(static_cast<size_t>(batch));
    w.times_out = times_out ? dalloc<double>(4 * static_cast<size_t>(batch)) : nullptr;
    const bool warm = ws_batch == batch && opt.warm_start;
    std::vector<double> keep_g, keep_l, keep_u;
    if (warm) {
      double* g = dalloc<double>(h_ws_g.size());
      double* l = dalloc<double>(h_ws_l.size());
      double* u = dalloc<double>(h_ws_u.size());
      if (!h_ws_g.empty()) DNLP_HIP_CHECK(hipMemcpy(g, h_ws_g.data(), h_ws_g.size() * 8, hipMemcpyHostToDevice));
      DNLP_HIP_CHECK(hipMemcpy(l, h_ws_l.data(), h_ws_l.size() * 8, hipMemcpyHostToDevice));
      DNLP_HIP_CHECK(hipMemcpy(u, h_ws_u.data(), h_ws_u.size() * 8, hipMemcpyHostToDevice));
      w.ws_g = g; w.ws_l = l; w.ws_u = u;
      keep_g = h_ws_g; keep_l = h_ws_l; keep_u = h_ws_u;
    }
    ws_batch = 0;
    w.next = dalloc<int>(1);
    DNLP_HIP_CHECK(hipMemsetAsync(w.next, 0, sizeof(int), stream));
#ifdef DNLP_WAVE_PROF
    const bool want_prof = true;
#else
    const bool want_prof = (spec && wave_spec_prof) || (wg && wave_wg_prof);
#endif
    if (want_prof) {
      w.prof = dalloc<unsigned long long>(kWaveProfSlots + 1);
      DNLP_HIP_CHECK(hipMemsetAsync(w.prof, 0, sizeof(unsigned long long) * (kWaveProfSlots + 1), stream));
    }
    const uint64_t key = theta ? rows_hash(theta, static_cast<size_t>(batch) * static_cast<size_t>(aff_P))
                               : rows_hash(data, static_cast<size_t>(batch) * static_cast<size_t>(in_stride));
    last_order_lpt = false;
    if (static_cast<int>(prev_iters.size()) == batch && key == prev_key && batch > grid * (wg ? 1 : nw) && !std::getenv("DNLP_BATCH_FIFO")) {
      last_order_lpt = true;
      std::vector<int> ord(static_cast<size_t>(batch));
      for (int k = 0; k < batch; ++k) ord[static_cast<size_t>(k)] = k;
      std::stable_sort(ord.begin(), ord.end(), [&](int p, int q) { return prev_iters[static_cast<size_t>(p)] > prev_iters[static_cast<size_t>(q)]; });
      int* d_ord = dalloc<int>(static_cast<size_t>(batch));
      DNLP_HIP_CHECK(hipMemcpyAsync(d_ord, ord.data(), sizeof(int) * static_cast<size_t>(batch), hipMemcpyHostToDevice, stream));
      DNLP_HIP_CHECK(hipStreamSynchronize(stream));
      w.order = d_ord;
    }
    last_grid = grid; last_threads = 64 * nw; last_lds_mode = (wg ? 0 : 2 * sl + pl) + ((spec || wg) ? 4 : 0) + (wg ? 8 : 0) /* + 8: the workgroup-per-instance kernel */; last_per_cu = wg ? per_cu : nw * per_cu; last_packed = false;
    last_wave = 100 * nw + 10 * sl + pl;
    last_wave_spec = spec;
    if (std::getenv("DNLP_BATCH_DEBUG"))
      std::fprintf(stderr, "[batch] wavefront solver: %d wavefronts per workgroup, state %s (%zu B per instance), plan %s (%zu B), dynamic LDS %u B, grid %d\n",
                   nw, sl ? "in LDS" : "in global memory", state_b, pl ? "in LDS" : "in global memory", plan_b, lds, grid);
    const bool dbg = std::getenv("DNLP_BATCH_DEBUG") != nullptr;
    const double tdbg0 = now_sec();
    auto mark = [&](const char* what) { if (dbg) std::fprintf(stderr, "[batch] %-22s %.4f s\n", what, now_sec() - tdbg0); };
    hipEvent_t e0, e1;
    DNLP_HIP_CHECK(hipEventCreate(&e0));
    DNLP_HIP_CHECK(hipEventCreate(&e1));
    DNLP_HIP_CHECK(hipEventRecord(e0, stream));
    if (wg) {
      void* kargs[] = {&w};
      DNLP_HIP_CHECK(hipModuleLaunchKernel(wave_wg.fn, static_cast<unsigned>(grid), 1, 1, static_cast<unsigned>(64 * nw), 1, 1, 0, stream, kargs, nullptr));
    } else if (spec) {
      void* kargs[] = {&w};
      DNLP_HIP_CHECK(hipModuleLaunchKernel(wave_spec.fn, static_cast<unsigned>(grid), 1, 1, static_cast<unsigned>(64 * nw), 1, 1, 0, stream, kargs, nullptr));
    } else switch (form) {
      case 811: launch_wave<8, true, true>(w, grid, lds, stream); break;
      case 711: launch_wave<7, true, true>(w, grid, lds, stream); break;
      case 611: launch_wave<6, true, true>(w, grid, lds, stream); break;
      case 511: launch_wave<5, true, true>(w, grid, lds, stream); break;
      case 411: launch_wave<4, true, true>(w, grid, lds, stream); break;
      case 311: launch_wave<3, true, true>(w, grid, lds, stream); break;
      case 211: launch_wave<2, true, true>(w, grid, lds, stream); break;
      case 111: launch_wave<1, true, true>(w, grid, lds, stream); break;
      case 410: launch_wave<4, true, false>(w, grid, lds, stream); break;
      case 310: launch_wave<3, true, false>(w, grid, lds, stream); break;
      case 210: launch_wave<2, true, false>(w, grid, lds, stream); break;
      case 110: launch_wave<1, true, false>(w, grid, lds, stream); break;
      case 400: launch_wave<4, false, false>(w, grid, lds, stream); break;
      default: throw std::runtime_error("wavefront solver: no such launch form");
    }
    DNLP_LAUNCH_CHECK();
    DNLP_HIP_CHECK(hipEventRecord(e1, stream));
    DNLP_HIP_CHECK(hipStreamSynchronize(stream));
    float ms = 0.f;
    DNLP_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    double total_sec = 1e-3 * ms;
    mark("wave kernel done");
    auto down = [&](void* hp, const void* d, size_t bytes) { if (hp && bytes) DNLP_HIP_CHECK(hipMemcpy(hp, d, bytes, hipMemcpyDeviceToHost)); };
    std::vector<int> st_local;
    int* st_host = status_out;
    if (!st_host) { st_local.resize(static_cast<size_t>(batch)); st_host = st_local.data(); }
    down(x_out, w.x_out, sizeof(double) * static_cast<size_t>(batch) * t.N);
    down(obj_out, w.obj_out, sizeof(double) * batch);
    down(multg_out, w.multg_out, sizeof(double) * static_cast<size_t>(batch) * t.m);
    down(zl_out, w.zl_out, sizeof(double) * static_cast<size_t>(batch) * t.N);
    down(zu_out, w.zu_out, sizeof(double) * static_cast<size_t>(batch) * t.N);
    down(st_host, w.status_out, sizeof(int) * batch);
    down(iters_out, w.iters_out, sizeof(int) * batch);
    down(nfact_out, w.nfact_out, sizeof(int) * batch);
    down(times_out, w.times_out, sizeof(double) * 4 * batch);
    if (w.prof) {
      unsigned long long pr[kWaveProfSlots + 1];
      DNLP_HIP_CHECK(hipMemcpy(pr, w.prof, sizeof pr, hipMemcpyDeviceToHost));
      const double it = static_cast<double>(pr[kWaveProfSlots] ? pr[kWaveProfSlots] : 1);
      static const char* nm[kWaveProfSlots] = {"solve (all)", "begin", "error", "eval_hessian", "barrier_terms", "assemble", "ldl_factor", "ldl_solve",
                                               "coo products", "residual combine", "quality()", "direction loops", "max_steps+measures", "trial point + f,g",
                                               "accept_trial", "(sweep)", "(spmv)", "mu oracle prologue", "[update_mu]", "[factor_with_inertia]", "[quality_function_mu]",
                                               "[solve_refined]", "kkt_solve copy", "[line search]", "(tail_factor)", "(tail_forward)", "(tail_backward)", ""};
      std::fprintf(stderr, "[wave profile] %llu iterations; cycles per iteration by phase (s_memtime ticks):\n", pr[kWaveProfSlots]);
      for (int k = 0; k < kWaveProfSlots; ++k) if (pr[k]) std::fprintf(stderr, "[wave profile]   %-22s %10.0f\n", nm[k], static_cast<double>(pr[k]) / it);
    }
    mark("wave results copied");
    pack_rows(batch, w.x_out, w.obj_out, w.status_out, w.iters_out);
    release();
    // the refused instances, through the generic kernel
    std::vector<int> refused;
    // (DNLP_WAVE_REFUSE_EVERY=k, tests: every k-th instance is treated as refused — the merge path without waiting for a
    //  structurally singular pivot sequence to come along)
    const int refuse_every = std::getenv("DNLP_WAVE_REFUSE_EVERY") ? std::atoi(std::getenv("DNLP_WAVE_REFUSE_EVERY")) : 0;
    for (int k = 0; k < batch; ++k) if (st_host[k] == kWaveNeedsGeneric || (refuse_every > 0 && k % refuse_every == 0)) refused.push_back(k);
    last_wave_refused = static_cast<int>(refused.size());
    if (!refused.empty()) {
      const int nb = static_cast<int>(refused.size());
      const i64 width = theta ? aff_P : in_stride;
      const double* src = theta ? theta : data;
      std::vector<double> sub(static_cast<size_t>(nb) * static_cast<size_t>(width));
      for (int q = 0; q < nb; ++q) std::copy(src + static_cast<i64>(refused[q]) * width, src + static_cast<i64>(refused[q] + 1) * width, sub.begin() + static_cast<i64>(q) * width);
      const size_t N = static_cast<size_t>(t.N), m = static_cast<size_t>(t.m);
      std::vector<double> sx(nb * N), sobj(nb), smg(multg_out ? nb * m : 0), szl(zl_out ? nb * N : 0), szu(zu_out ? nb * N : 0), stm(times_out ? 4 * nb : 0);
      std::vector<int> sst(nb), sit(nb), snf(nb);
      if (warm) {
        std::vector<double> g2(nb * m), l2(nb * N), u2(nb * N);
        for (int q = 0; q < nb; ++q) {
          std::copy(keep_g.begin() + refused[q] * m, keep_g.begin() + (refused[q] + 1) * m, g2.begin() + q * m);
          std::copy(keep_l.begin() + refused[q] * N, keep_l.begin() + (refused[q] + 1) * N, l2.begin() + q * N);
          std::copy(keep_u.begin() + refused[q] * N, keep_u.begin() + (refused[q] + 1) * N, u2.begin() + q * N);
        }
        set_warm_start(nb, g2.data(), l2.data(), u2.data());
      }
      const int wave_form = last_wave;
      const std::vector<int> keep_iters = prev_iters;
      const uint64_t keep_key = prev_key;
      double sec2 = 0.0;
      const bool nested_before = rows_nested;
      rows_nested = true;
      try {
        solve_impl(nb, theta ? nullptr : sub.data(), theta ? sub.data() : nullptr, opt, sx.data(), sobj.data(), multg_out ? smg.data() : nullptr,
                   zl_out ? szl.data() : nullptr, zu_out ? szu.data() : nullptr, sst.data(), sit.data(), snf.data(), &sec2, times_out ? stm.data() : nullptr, false);
      } catch (...) { rows_nested = nested_before; throw; }
      rows_nested = nested_before;
      prev_iters = keep_iters; prev_key = keep_key;
      total_sec += sec2;
      for (int q = 0; q < nb; ++q) {
        const int k = refused[q];
        if (x_out) std::copy(sx.begin() + q * N, sx.begin() + (q + 1) * N, x_out + static_cast<size_t>(k) * N);
        if (obj_out) obj_out[k] = sobj[q];
        if (multg_out) std::copy(smg.begin() + q * m, smg.begin() + (q + 1) * m, multg_out + static_cast<size_t>(k) * m);
        if (zl_out) std::copy(szl.begin() + q * N, szl.begin() + (q + 1) * N, zl_out + static_cast<size_t>(k) * N);
        if (zu_out) std::copy(szu.begin() + q * N, szu.begin() + (q + 1) * N, zu_out + static_cast<size_t>(k) * N);
        st_host[k] = sst[q];
        if (iters_out) iters_out[k] = sit[q];
        if (nfact_out) nfact_out[k] = snf[q];
        if (times_out) std::copy(stm.begin() + 4 * q, stm.begin() + 4 * q + 4, times_out + 4 * static_cast<size_t>(k));
        patch_row(k, sobj[q], sst[q], sit[q], sx.data() + q * N);
      }
      last_wave = wave_form;
      last_wave_refused = nb;
    }
    if (seconds) *seconds = total_sec;
    if (iters_out) { prev_iters.assign(iters_out, iters_out + batch); prev_key = key; }
  }
};

}  // namespace dnlp
