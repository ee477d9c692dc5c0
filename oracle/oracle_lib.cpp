// TEST INFRASTRUCTURE — host build of the solver core behind the same C surface as the
// product library, exported with the prefix orc_ (liboracle_dnlp.so).  See host_exec.h.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include "host_exec.h"
#include "../dnlp_amd/csrc/capi_impl.h"

DNLP_DEFINE_CAPI(orc_, dnlp::HostExec, orc_problem)

// CPU-baseline switch (bench.py): route the pivoted dense factorisation through LAPACK DSYTRF/DSYTRS
// of the given shared library; returns the BLAS thread count (>= 1) or 0 when the library or its
// symbols are missing.  threads > 0 sets the BLAS thread count first.
extern "C" int orc_use_lapack(const char* path, int threads) {
  dnlp::HostLapack& L = dnlp::HostLapack::get();
  if (!path || !*path) { L.sytrf = nullptr; L.sytrs = nullptr; return 0; }
  if (!L.load(path)) return 0;
  if (threads > 0 && L.set_threads) L.set_threads(threads);
  return L.get_threads ? L.get_threads() : 1;
}

// Seconds of one DSYTRF of a seeded symmetric indefinite matrix of order n with `threads` BLAS
// threads (bench.py picks the thread count that factors fastest on this box: OpenBLAS's DSYTRF does
// not scale to every core of a large host).  Negative when no LAPACK is loaded.
extern "C" double orc_lapack_probe(int n, int threads) {
  dnlp::HostLapack& L = dnlp::HostLapack::get();
  if (!L.sytrf) return -1.0;
  if (threads > 0 && L.set_threads) L.set_threads(threads);
  std::vector<double> A(static_cast<size_t>(n) * n);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (int j = 0; j < n; ++j)
    for (int i = j; i < n; ++i) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      A[static_cast<size_t>(i) + static_cast<size_t>(j) * n] = static_cast<double>(s >> 11) / 9007199254740992.0 - 0.5;
    }
  std::vector<int> ipiv(static_cast<size_t>(n));
  int info = 0, lwork = -1;
  double wq = 0.0;
  L.sytrf("L", &n, A.data(), &n, ipiv.data(), &wq, &lwork, &info);
  lwork = static_cast<int>(wq) > n ? static_cast<int>(wq) : n;
  std::vector<double> work(static_cast<size_t>(lwork));
  const double t0 = dnlp::now_sec();
  L.sytrf("L", &n, A.data(), &n, ipiv.data(), work.data(), &lwork, &info);
  return dnlp::now_sec() - t0;
}

// CPU-baseline switch: unpivoted dense factorisations of order >= 512 take the blocked LDL^T whose trailing update is
// DGEMM (the device's algorithm on the host's BLAS); needs orc_use_lapack first.  Returns 1 when the BLAS has DGEMM / DTRSM.
extern "C" int orc_use_blocked_ldlt(int on) {
  dnlp::HostLapack& L = dnlp::HostLapack::get();
  L.blocked_unpivoted = on != 0 && L.gemm && L.trsm;
  return L.blocked_unpivoted ? 1 : 0;
}

// GFLOP/s of one square DGEMM of order n with `threads` BLAS threads: the rate the blocked factorisation can approach.
extern "C" double orc_dgemm_probe(int n, int threads) {
  dnlp::HostLapack& L = dnlp::HostLapack::get();
  if (!L.gemm) return -1.0;
  if (threads > 0 && L.set_threads) L.set_threads(threads);
  std::vector<double> A(static_cast<size_t>(n) * n, 0.5), B(static_cast<size_t>(n) * n, 0.25), C(static_cast<size_t>(n) * n, 0.0);
  const double one = 1.0, zero = 0.0;
  L.gemm("N", "T", &n, &n, &n, &one, A.data(), &n, B.data(), &n, &zero, C.data(), &n);      // warm the threads
  const double t0 = dnlp::now_sec();
  L.gemm("N", "T", &n, &n, &n, &one, A.data(), &n, B.data(), &n, &zero, C.data(), &n);
  return 2.0 * n * n * n / (dnlp::now_sec() - t0) / 1e9;
}

// Seconds of the four phases of the last blocked unpivoted LDL^T (diagonal blocks, DTRSM, W copy, DGEMM); zeros unless
// DNLP_HOST_LDLT_TIMING is set.
extern "C" void orc_blocked_ldlt_phases(double* out4) {
  dnlp::HostLapack& L = dnlp::HostLapack::get();
  for (int k = 0; k < 4; ++k) out4[k] = L.last_phases[k];
}

// ---- wavefront batch solver (dnlp_amd/csrc/wave_ipm.h) on ONE host lane ------------------------------------------------
// The CPU pin of the template-specialised batch kernel: the restated interior-point loop runs over the same plan block
// and the same instance rows as on the device, with a one-lane policy whose serial sums are the host build's — next to
// (which = 1) the generic algorithm text (ipm_core.h over HostExec) set up the way the generic batch kernel sets it up
// (batch.h: one shared tape view patched with the instance's data, the template's static-pattern plan).
#define DNLP_WAVE_FILTER_CAP 1024      // (HostControlled::kFilterCap: the host build it is compared with)
#include "../dnlp_amd/csrc/wave_ipm.h"

namespace {
struct HostLane {
  typedef double D;
  typedef const dnlp::i32 I;
  static constexpr int lanes = 1;
  static constexpr bool hoist = false;
  static int lane() { return 0; }
  static void sync() {}
  static double sum(double v) { return v; }
  static double vmax(double v) { return v; }
  template <int N> static void sum_n(double (&)[N]) {}
  template <int N> static void vmax_n(double (&)[N]) {}
  static double now() { return dnlp::now_sec(); }
  static int tab_load(const dnlp::i32*, int) { return 0; }
  static int tab_at(const dnlp::i32* tab, int, int idx, int) { return tab[idx]; }
  static int uni(int v) { return v; }
  template <int SL> static double row_get(const double (&a)[SL], int row) { return a[row]; }      // one lane owns every tail row
};
}  // namespace

// data: batch x stride rows in BATCH_DATA_KEYS order (dnlp_amd/batch.py).  Outputs batch-major; mult_* may be null.
extern "C" int orc_wave_solve_batch(orc_problem* vp, int batch, const double* data, int64_t stride, int which, double* x, double* obj,
                                    int* status, int* iters, int* nfact, double* mult_g, double* mult_x_L, double* mult_x_U) {
  using namespace dnlp;
  orc_problem_t* p = vp;
  DNLP_TRY(
    p->plan_linear_solver();
    const Tape<HostExec>& t = *p->model.owner;
    if (!p->use_sparse) { tls_error() = "no sparse plan for this tape"; return -1; }
    const char* why = wave_plan_refusal(t, &p->sparse_plan);
    if (which == 0 && why[0]) { tls_error() = why; return -2; }
    const WaveLayoutIn lay = wave_layout_of(t);
    const i64 head = 1 + (t.N + t.Z) + t.m + t.nnzJ + t.G.nnz + t.Mg.nnz + t.Mw.nnz + t.MJ.nnz + t.MH.nnz;
    const i64 tail = 3 * t.N + 2 * t.m;
    if (stride != head + 2 * t.nseg + tail) { tls_error() = "instance stride does not match the tape"; return -3; }
    std::vector<double> row(static_cast<size_t>(lay.total));
    std::vector<i32> blk;
    std::vector<double> state;
    if (which == 0) {
      blk = build_wave_plan(&p->ex, t, p->sparse_plan, lay);
      state.assign(static_cast<size_t>(reinterpret_cast<const WaveHdr*>(blk.data())->state_doubles) + 8, 0.0);
      if (std::getenv("DNLP_WAVE_DEBUG")) {
        const WaveHdr* hh = reinterpret_cast<const WaveHdr*>(blk.data());
        std::fprintf(stderr, "[wave] plan block %d ints (%.1f KB), state %d doubles (%.1f KB), WState %zu B; units %d, nvals %d, blocks %d, levels %d, triples %d; dense tail of order %d from level %d\n",
                     hh->total, hh->total * 4 / 1024.0, hh->state_doubles, hh->state_doubles * 8 / 1024.0, sizeof(WaveIpm<HostLane>::WState), hh->nunits, hh->sp_nvals, hh->sp_nblk, hh->sp_nlev, hh->sp_ntrip, hh->tail_T, hh->tail_L);
        std::fprintf(stderr, "[wave] N %d m %d Z %d nd %d nh %d nnzJ %d nnzH %d scr %d\n", hh->N, hh->m, hh->Z, hh->nd, hh->nh, hh->nnzJ, hh->nnzH, hh->scr_doubles);
      }
    }
    const bool fb = (t.N + t.m) <= 512 && p->linear_solver != 2;
    for (int k = 0; k < batch; ++k) {
      const double* src = data + static_cast<i64>(k) * stride;
      std::copy(src, src + head, row.begin());
      const double *sp = src + head, *sp2 = sp + t.nseg;
      for (i64 f = 0; f < t.nflat; ++f) {
        row[static_cast<size_t>(lay.fp + f)] = sp[t.h_flat_seg[static_cast<size_t>(f)]];
        row[static_cast<size_t>(lay.fp2 + f)] = sp2[t.h_flat_seg[static_cast<size_t>(f)]];
      }
      std::copy(sp2 + t.nseg, sp2 + t.nseg + tail, row.begin() + lay.x0);
      if (which == 0) {
        WaveIpm<HostLane>::WState S;
        std::fill(state.begin(), state.end(), 0.0);
        if (WaveIpm<HostLane>::layout(&S, reinterpret_cast<const WaveHdr*>(blk.data()), blk.data(), state.data()) !=
            reinterpret_cast<const WaveHdr*>(blk.data())->state_doubles) throw std::runtime_error("wave layout and wave_state_doubles disagree");
        std::vector<double> park(static_cast<size_t>(wave_park_doubles(t.N, t.m)) + 8, 0.0);
        S.row = row.data();
        S.park = park.data();
        S.ws_g = S.ws_l = S.ws_u = nullptr;
        S.fallback_max_n = fb ? 512 : 0;
        S.opt = p->opt;
        const int st = WaveIpm<HostLane>::solve(&S);
        status[k] = st; iters[k] = S.iter; if (nfact) nfact[k] = S.factorizations;
        obj[k] = S.initialized ? S.f / S.sf : 0.0;
        for (i64 j = 0; j < t.N; ++j) {
          x[static_cast<i64>(k) * t.N + j] = S.x[j];
          if (mult_x_L) mult_x_L[static_cast<i64>(k) * t.N + j] = S.zL[j] / S.sf;
          if (mult_x_U) mult_x_U[static_cast<i64>(k) * t.N + j] = S.zU[j] / S.sf;
        }
        if (mult_g) for (i64 i = 0; i < t.m; ++i) mult_g[static_cast<i64>(k) * t.m + i] = S.y[i] * S.sg[i] / S.sf;
      } else {
        HostExec ex;
        TapeView v = t;
        v.c0 = row[static_cast<size_t>(lay.c0)];
        v.c = row.data() + lay.c; v.b = row.data() + lay.b; v.Jc = row.data() + lay.Jc;
        v.G.val = row.data() + lay.G; v.Mg.val = row.data() + lay.Mg; v.Mw.val = row.data() + lay.Mw; v.MJ.val = row.data() + lay.MJ;
        v.MH.val = row.data() + lay.MH;
        // (the host evaluator reads per-SEGMENT parameters through flat_p / flat_p2 indexed by flat row, as the kernel does)
        v.flat_p = row.data() + lay.fp; v.flat_p2 = row.data() + lay.fp2;
        v.d_x0 = row.data() + lay.x0; v.d_lb = row.data() + lay.lb; v.d_ub = row.data() + lay.ub; v.d_cl = row.data() + lay.cl; v.d_cu = row.data() + lay.cu;
        Model<HostExec> md;
        md.init_view(&ex, v);
        DenseKkt<HostExec> kkt;
        SparsePlan pl = p->sparse_plan.upload(&ex);
        kkt.init_sparse(&ex, v.N, v.m, pl);
        kkt.pivot_max_n = static_cast<i64>(1) << 40;
        kkt.fallback_max_n = fb ? 512 : 0;
        Ipm<HostExec, DenseKkt<HostExec>> ipm(&ex, &md, &kkt);
        ipm.opt = p->opt;
        ipm.allocate();
        const int st = ipm.solve(v.d_x0);
        status[k] = st; iters[k] = ipm.iter; if (nfact) nfact[k] = ipm.stats.factorizations;
        obj[k] = ipm.initialized ? ipm.objective_unscaled() : 0.0;
        if (ipm.initialized)
          ipm.extract(x + static_cast<i64>(k) * t.N, nullptr, mult_g ? mult_g + static_cast<i64>(k) * t.m : nullptr,
                      mult_x_L ? mult_x_L + static_cast<i64>(k) * t.N : nullptr, mult_x_U ? mult_x_U + static_cast<i64>(k) * t.N : nullptr, nullptr);
      }
    }
    return 0;)
}

// The text of a template's per-template kernel (dnlp_amd/csrc/wave_codegen.h), as the product library would hand it to
// hiprtc: tests compile it for gfx950 without a GPU.  Returns the length of the text (0-terminated copy in buf when it fits),
// a negative code when the template is not the wavefront solver's.
#include "../dnlp_amd/csrc/wave_codegen.h"
extern "C" long long orc_wave_spec_source(orc_problem* vp, int nw, char* buf, long long cap) {
  using namespace dnlp;
  orc_problem_t* p = vp;
  DNLP_TRY(
    p->plan_linear_solver();
    const Tape<HostExec>& t = *p->model.owner;
    if (!p->use_sparse) { tls_error() = "no sparse plan for this tape"; return -1; }
    const char* why = wave_plan_refusal(t, &p->sparse_plan);
    if (why[0]) { tls_error() = why; return -2; }
    const std::vector<i32> blk = build_wave_plan(&p->ex, t, p->sparse_plan, wave_layout_of(t));
    const WaveHdr& h = *reinterpret_cast<const WaveHdr*>(blk.data());
    if (wave_gen_refusal(h)[0]) { tls_error() = wave_gen_refusal(h); return -4; }
    const WaveGen gen = wave_generate(blk);
    const int fit = wave_spec_max_waves(h, gen.G.size());
    if (fit < 1) { tls_error() = "the template's state does not fit a compute unit's LDS"; return -3; }
    const std::string src = wave_spec_source(blk, nw > 0 ? std::min(nw, fit) : fit, gen);
    if (buf && cap > static_cast<long long>(src.size())) { std::memcpy(buf, src.data(), src.size()); buf[src.size()] = 0; }
    return static_cast<long long>(src.size());)
}

// The text of a template's workgroup-per-instance kernel (wave_codegen.h wave_wg_source: templates whose state exceeds LDS).
extern "C" long long orc_wave_wg_source(orc_problem* vp, int nwg, char* buf, long long cap) {
  using namespace dnlp;
  orc_problem_t* p = vp;
  DNLP_TRY(
    p->plan_linear_solver();
    const Tape<HostExec>& t = *p->model.owner;
    if (!p->use_sparse) { tls_error() = "no sparse plan for this tape"; return -1; }
    const char* why = wave_plan_refusal(t, &p->sparse_plan);
    if (why[0]) { tls_error() = why; return -2; }
    const std::vector<i32> blk = build_wave_plan(&p->ex, t, p->sparse_plan, wave_layout_of(t), false);      // (no register tail: batch.h wave_wg_prepare)
    const WaveHdr& h = *reinterpret_cast<const WaveHdr*>(blk.data());
    if (wave_gen_refusal(h)[0]) { tls_error() = wave_gen_refusal(h); return -4; }
    if (nwg < 1 || nwg > 8) { tls_error() = "1 .. 8 wavefronts per workgroup"; return -5; }
    const WaveGen gen = wave_wg_generate(blk, nwg);
    const std::string src = wave_wg_source(blk, nwg, gen);
    if (buf && cap > static_cast<long long>(src.size())) { std::memcpy(buf, src.data(), src.size()); buf[src.size()] = 0; }
    return static_cast<long long>(src.size());)
}

// ---- the per-template straight-line LDL^T phases (dnlp_amd/csrc/wave_gen.h) on the host ---------------------------------
// orc_wave_gen_host_source: a self-contained translation unit for g++ — the template's constants (namespace wspec), its
// plan block and work tables as arrays, wave_ipm.h compiled with -DDNLP_WAVE_SPEC -DDNLP_WAVE_GEN over a one-thread lane
// policy (the generated phases play 64 lanes one after the other), and a driver `wgen_host_solve`.  tests/test_wave_gen_cpu.py
// builds it and compares its bits with orc_wave_solve_batch (the interpreted text on one host lane).
#include "../dnlp_amd/csrc/wave_gen.h"
extern "C" long long orc_wave_gen_host_source(orc_problem* vp, int lanes, char* buf, long long cap) {
  using namespace dnlp;
  orc_problem_t* p = vp;
  DNLP_TRY(
    p->plan_linear_solver();
    const Tape<HostExec>& t = *p->model.owner;
    if (!p->use_sparse) { tls_error() = "no sparse plan for this tape"; return -1; }
    const char* why = wave_plan_refusal(t, &p->sparse_plan);
    if (why[0]) { tls_error() = why; return -2; }
    const std::vector<i32> blk = build_wave_plan(&p->ex, t, p->sparse_plan, wave_layout_of(t));
    // (more than 64 lanes: the workgroup kernel's forms — staging and the factorisation's windows are in the text; on the host
    //  their macros name the arrays themselves)
    const WaveGen gen = wave_generate(blk, lanes > 0 ? lanes : 64, lanes > 64 ? 2048 : 0, lanes > 64 ? 2048 : 0);
    std::string s;
    s += "#include \"ipm_core.h\"\n#include \"wave_plan.h\"\n";
    s += wave_spec_constants(blk, 1);
    s += "#define DNLP_WAVE_SPEC 1\n#define DNLP_WAVE_GEN 1\n#define DNLP_WAVE_FILTER_CAP 1024\n#include \"wave_ipm.h\"\n#include \"wave_gen_rt.h\"\n";
    s += lanes > 64 ? "#define WGEN_HOST_HOIST true\n" : "#define WGEN_HOST_HOIST false\n";
    char b[64];
    s += "static const int32_t k_blk[] = {";
    for (size_t k = 0; k < blk.size(); ++k) { std::snprintf(b, sizeof b, "%s%d", k ? "," : "", blk[k]); s += b; if ((k & 31) == 31) s += "\n"; }
    s += "};\nstatic const uint32_t k_gen[] = {";
    for (size_t k = 0; k < gen.G.size(); ++k) { std::snprintf(b, sizeof b, "%s%uu", k ? "," : "", gen.G[k]); s += b; if ((k & 31) == 31) s += "\n"; }
    if (gen.G.empty()) s += "0u";
    s += "};\n";
    s += R"WGH(
namespace {
double* g_vec = nullptr;
struct HostSpecLane {
  typedef double D;
  typedef const dnlp::i32 I;
  typedef const uint32_t* G;
  static constexpr int lanes = 1;
  static constexpr bool hoist = WGEN_HOST_HOIST;      // (the elementwise loops' form of the workgroup kernel when its lanes are asked for)
  static int lane() { return 0; }
  static void sync() {}
  static double sum(double v) { return v; }
  static double vmax(double v) { return v; }
  template <int N> static void sum_n(double (&)[N]) {}
  template <int N> static void vmax_n(double (&)[N]) {}
  static double now() { return dnlp::now_sec(); }
  static int tab_load(const dnlp::i32*, int) { return 0; }
  static int tab_at(const dnlp::i32* tab, int, int idx, int) { return tab[idx]; }
  static int uni(int v) { return v; }
  template <int SL> static double row_get(const double (&a)[SL], int row) { return a[row]; }
  template <class WS> static D* vec(WS*, int off) { return g_vec + off; }
  static I* tab(int off) { return k_blk + off; }
  static G gtab() { return k_gen; }
};
}  // namespace
)WGH";
    s += gen.code;
    s += R"WGH(
// rows: batch x wspec::k_l_total expanded instance rows (orc_wave_expand_rows); opt: the oracle handle's IpmOptions bytes
extern "C" int wgen_host_solve(int batch, const double* rows, long long row_doubles, const void* opt_bytes, long long opt_size, int fallback_512,
                               double* x, double* obj, int* status, int* iters, int* nfact, double* mult_g, double* zl, double* zu) {
  using namespace dnlp;
  typedef HostSpecLane P;
  typedef WaveIpm<P> W;
  if (opt_size != static_cast<long long>(sizeof(IpmOptions))) return -1;
  std::vector<double> state(static_cast<size_t>(wspec::kStateDoubles) + 64, 0.0);
  std::vector<double> park(static_cast<size_t>(wave_park_doubles(wspec::k_N, wspec::k_m)) + 8, 0.0);
  g_vec = state.data();
  for (int k = 0; k < batch; ++k) {
    W::WState st;
    W::WState* S = &st;
    std::fill(state.begin(), state.end(), 0.0);
    S->row = rows + static_cast<long long>(k) * row_doubles;
    S->park = park.data();
    S->ws_g = S->ws_l = S->ws_u = nullptr;
    S->fallback_max_n = fallback_512 ? 512 : 0;
    std::memcpy(&S->opt, opt_bytes, sizeof(IpmOptions));
    S->factorizations = 0;
    const int rc = W::solve(S);
    status[k] = rc; iters[k] = S->iter; nfact[k] = S->factorizations;
    obj[k] = S->initialized ? S->f / S->sf : 0.0;
    const double *xx = WV(x), *a = WV(zL), *b = WV(zU), *yy = WV(y), *sg = WV(sg);
    for (int j = 0; j < wspec::k_N; ++j) {
      x[static_cast<long long>(k) * wspec::k_N + j] = xx[j];
      zl[static_cast<long long>(k) * wspec::k_N + j] = a[j] / S->sf;
      zu[static_cast<long long>(k) * wspec::k_N + j] = b[j] / S->sf;
    }
    for (int i = 0; i < wspec::k_m; ++i) mult_g[static_cast<long long>(k) * wspec::k_m + i] = yy[i] * sg[i] / S->sf;
  }
  return 0;
}
)WGH";
    if (buf && cap > static_cast<long long>(s.size())) { std::memcpy(buf, s.data(), s.size()); buf[s.size()] = 0; }
    return static_cast<long long>(s.size());)
}

extern "C" long long orc_sizeof_ipm_options() { return static_cast<long long>(sizeof(dnlp::IpmOptions)); }

// the instance rows as the wavefront solver reads them (batch.h layout: per-segment parameters expanded per flat row), and the
// handle's options as bytes — inputs of wgen_host_solve.  Returns the doubles per row (out may be null to ask for it).
extern "C" long long orc_wave_expand_rows(orc_problem* vp, int batch, const double* data, int64_t stride, double* out, void* opt_out, long long opt_cap,
                                          int* fallback_512) {
  using namespace dnlp;
  orc_problem_t* p = vp;
  DNLP_TRY(
    p->plan_linear_solver();
    const Tape<HostExec>& t = *p->model.owner;
    const WaveLayoutIn lay = wave_layout_of(t);
    if (opt_out && opt_cap >= static_cast<long long>(sizeof(IpmOptions))) std::memcpy(opt_out, &p->opt, sizeof(IpmOptions));
    if (fallback_512) *fallback_512 = ((t.N + t.m) <= 512 && p->linear_solver != 2) ? 1 : 0;
    if (!out) return lay.total;
    const i64 head = 1 + (t.N + t.Z) + t.m + t.nnzJ + t.G.nnz + t.Mg.nnz + t.Mw.nnz + t.MJ.nnz + t.MH.nnz;
    const i64 tail = 3 * t.N + 2 * t.m;
    if (stride != head + 2 * t.nseg + tail) { tls_error() = "instance stride does not match the tape"; return -3; }
    for (int k = 0; k < batch; ++k) {
      const double* src = data + static_cast<i64>(k) * stride;
      double* row = out + static_cast<i64>(k) * lay.total;
      std::copy(src, src + head, row);
      const double *sp = src + head, *sp2 = sp + t.nseg;
      for (i64 f = 0; f < t.nflat; ++f) {
        row[lay.fp + f] = sp[t.h_flat_seg[static_cast<size_t>(f)]];
        row[lay.fp2 + f] = sp2[t.h_flat_seg[static_cast<size_t>(f)]];
      }
      std::copy(sp2 + t.nseg, sp2 + t.nseg + tail, row + lay.x0);
    }
    return lay.total;)
}

// plan statistics per level (tools / tests: what the wavefront solver's level phases are made of)
extern "C" int orc_wave_plan_levels(orc_problem* vp, int32_t* out, int cap) {
  using namespace dnlp;
  orc_problem_t* p = vp;
  DNLP_TRY(
    p->plan_linear_solver();
    if (!p->use_sparse) return -1;
    const SparsePlanHost& sp = p->sparse_plan;
    const int nlev = static_cast<int>(sp.lev_off.size()) - 1;
    int w = 0;
    for (int l = 0; l < nlev && w + 8 <= cap; ++l) {
      const int b0 = sp.lev_off[l], b1 = sp.lev_off[l + 1];
      int maxs = 0, two = 0;
      for (int k = b0; k < b1; ++k) { maxs = std::max(maxs, sp.soff[k + 1] - sp.soff[k]); two += sp.bnode[2 * k + 1] >= 0; }
      const int g0 = sp.lev_g[l], g1 = sp.lev_g[l + 1];
      int maxg = 0;
      for (int g = g0; g < g1; ++g) maxg = std::max(maxg, sp.goff[g + 1] - sp.goff[g]);
      const int h0 = sp.lev_f[l], h1 = sp.lev_f[l + 1];
      int maxf = 0;
      for (int h = h0; h < h1; ++h) maxf = std::max(maxf, sp.foff[h + 1] - sp.foff[h]);
      out[w++] = b1 - b0; out[w++] = two; out[w++] = sp.soff[b1] - sp.soff[b0]; out[w++] = maxs;
      out[w++] = g1 - g0; out[w++] = sp.goff[g1] - sp.goff[g0]; out[w++] = maxg; out[w++] = (h1 - h0) * 1000 + maxf;
    }
    return nlev;)
}
