// libdnlp_hip.so — the C ABI of include/dnlp_hip.h instantiated over the HIP execution space.
// The public header is included FIRST: every definition below must agree with its prototype (a
// drifted signature is a compile error — conflicting types for a C-linkage function).
#include "../../include/dnlp_hip.h"

#include "ldlt_blocked.h"
#include "capi_impl.h"
#include "batch.h"
#include "lower_maps.h"
#include "linform.h"

// struct dnlp_problem (the opaque handle of the header) IS the problem object over the HIP space
DNLP_DEFINE_CAPI(dnlp_, dnlp::HipExec, dnlp_problem)

using namespace dnlp;

extern "C" {

int dnlp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// Batched solve (BASELINE C5): `batch` instances sharing the structure of p's tape; see batch.h
// for the per-instance data layout.  One kernel launch, one workgroup per instance.
int dnlp_solve_batch_timed(dnlp_problem* vp, int batch, const double* data, int64_t stride, double* x, double* obj, double* mult_g,
                           double* mult_x_L, double* mult_x_U, int* status, int* iters, int* factorizations, double* seconds,
                           double* times);
static BatchRunner* batch_runner(dnlp_problem_t* p) {
  if (!p->batch_state) {
    auto* r = new BatchRunner();
    std::shared_ptr<void> hold(r, [](void* q) { delete static_cast<BatchRunner*>(q); });
    r->init(&p->ex, p->model.owner);
    p->plan_linear_solver();
    if (p->use_sparse) {
      if (p->sparse_plan.panels_dropped) {
        // the handle's plan was built for the dense tail of the host-driven loop: its panel blocks carry no tail x tail
        // triples.  The in-kernel solver needs the whole update program: a second analysis, without the tail.
        if (!p->batch_plan_full) {
          p->batch_plan_full.reset(new dnlp::SparsePlanHost());
          dnlp::build_sparse_plan(*p->model.owner, *p->batch_plan_full, p->opt.bound_relax_factor > 0.0, p->plan_jabs.empty() ? nullptr : &p->plan_jabs, false);
        }
        r->set_sparse_plan(*p->batch_plan_full);
      } else {
        r->set_sparse_plan(p->sparse_plan);
      }
      r->force_sparse = p->linear_solver == 2;
    }
    p->batch_state = hold;
  }
  return static_cast<BatchRunner*>(p->batch_state.get());
}
// The in-kernel solver has no quasi-Newton mode.  The reference hands hessian_approximation to IPOPT for every solve
// (ipopt_nlpif.py:153-168); a batch launch that would silently run the exact Hessian instead is refused with IPOPT's
// Invalid_Option (-12) — the front-end runs such starts one at a time through the host-driven loop (problem.py).
static int batch_rejects_limited_memory() {
  dnlp::tls_error() = "hessian_approximation=limited-memory is not available inside a batch launch (the in-kernel solver "
                      "evaluates the exact Hessian): solve the instances one at a time, or use hessian_approximation=exact";
  return -12;
}
int64_t dnlp_batch_stride(dnlp_problem* vp) {
  dnlp_problem_t* p = vp;
  DNLP_TRY(return batch_runner(p)->in_stride;)
}
/* per-instance multipliers (batch-major) for the next dnlp_solve_batch with warm_start_init_point=yes */
int dnlp_batch_warm_start(dnlp_problem* vp, int batch, const double* mult_g, const double* mult_x_L, const double* mult_x_U) {
  dnlp_problem_t* p = vp;
  DNLP_TRY(batch_runner(p)->set_warm_start(batch, mult_g, mult_x_L, mult_x_U); return 0;)
}
int dnlp_solve_batch(dnlp_problem* vp, int batch, const double* data, int64_t stride, double* x, double* obj, double* mult_g,
                     double* mult_x_L, double* mult_x_U, int* status, int* iters, int* factorizations, double* seconds) {
  return dnlp_solve_batch_timed(vp, batch, data, stride, x, obj, mult_g, mult_x_L, mult_x_U, status, iters, factorizations,
                                seconds, nullptr);
}
/* same, plus per-instance device-clock phase times: times[4*i..] = wall, t_eval, t_factor, t_solve */
int dnlp_solve_batch_timed(dnlp_problem* vp, int batch, const double* data, int64_t stride, double* x, double* obj, double* mult_g,
                           double* mult_x_L, double* mult_x_U, int* status, int* iters, int* factorizations, double* seconds,
                           double* times) {
  dnlp_problem_t* p = vp;
  DNLP_TRY(
    BatchRunner& r = *batch_runner(p);
    p->ex.sync();
    if (p->opt.hessian_approximation == 1) return batch_rejects_limited_memory();
    r.solve(batch, data, stride, p->opt, x, obj, mult_g, mult_x_L, mult_x_U, status, iters, factorizations, seconds, times);
    return 0;)
}

/* Parametrised batches whose instance data is affine in the parameters (dnlp_amd.batch.ParametricBatch): the map is
   handed over once — d0 = the base instance's data row (dnlp_batch_stride doubles), theta0 = its P parameter values,
   (indptr, indices, vals) = CSR of the stride x P sensitivity — and stays on the device. */
int dnlp_batch_set_affine_map(dnlp_problem* vp, int n_params, const double* d0, const double* theta0, const int64_t* indptr,
                              const int32_t* indices, const double* vals) {
  dnlp_problem_t* p = vp;
  DNLP_TRY(batch_runner(p)->set_affine_map(n_params, d0, theta0, reinterpret_cast<const dnlp::i64*>(indptr), indices, vals); return 0;)
}
/* dnlp_solve_batch_timed with the instances given as parameter rows (batch x n_params): the instance data is generated
   on the device, so a call moves the parameter rows in and the results out. */
int dnlp_solve_batch_theta(dnlp_problem* vp, int batch, const double* theta, int n_params, double* x, double* obj, double* mult_g,
                           double* mult_x_L, double* mult_x_U, int* status, int* iters, int* factorizations, double* seconds,
                           double* times) {
  dnlp_problem_t* p = vp;
  DNLP_TRY(
    BatchRunner& r = *batch_runner(p);
    p->ex.sync();
    if (p->opt.hessian_approximation == 1) return batch_rejects_limited_memory();
    r.solve_theta(batch, theta, n_params, p->opt, x, obj, mult_g, mult_x_L, mult_x_U, status, iters, factorizations, seconds, times);
    return 0;)
}

// ---- batches in flight (include/dnlp_hip.h dnlp_batch_stream_*) ----------------------------------------------------------
}  // extern "C"
#include <condition_variable>
#include <map>
#include <mutex>
#include <set>
#include <thread>
// A stream = nslots runners over the same tape and plan, each with ONE worker thread for its lifetime (round 5 started a
// thread per submission and always took the round-robin slot).  A submission goes to whichever slot is idle — the caller
// waits only when all are busy, and then for the first to finish; a result stays on record until it is waited for (a ticket
// can be waited for once: afterwards, and for a ticket that was never handed out, the call says which of the two it is).
struct dnlp_batch_stream {
  dnlp_problem_t* p = nullptr;
  struct Job {
    int ticket = -1, batch = 0, n_params = 0;
    const double* theta = nullptr;
    double *x = nullptr, *obj = nullptr, *mult_g = nullptr, *mult_x_L = nullptr, *mult_x_U = nullptr;
    int *status = nullptr, *iters = nullptr, *factorizations = nullptr;
    IpmOptions opt;
  };
  struct Result { int rc = 0; double seconds = 0.0; std::string err; };
  struct Slot {
    BatchRunner runner;
    std::thread th;
    std::condition_variable cv;        // work for this slot (or stop)
    bool busy = false, has_job = false;
    Job job;
  };
  std::mutex mu;
  std::condition_variable cv_done;     // a slot became idle / a result was recorded
  std::vector<std::unique_ptr<Slot>> slots;
  std::map<int, Result> done;          // finished, not yet waited for
  std::set<int> inflight;
  int next_ticket = 0;
  bool stop = false;

  void worker(Slot* sl) {
    std::unique_lock<std::mutex> lk(mu);
    while (true) {
      sl->cv.wait(lk, [&] { return sl->has_job || stop; });
      if (!sl->has_job) return;          // (stop with nothing queued)
      const Job j = sl->job;
      sl->has_job = false;
      lk.unlock();
      Result r;
      try {
        sl->runner.solve_theta(j.batch, j.theta, j.n_params, j.opt, j.x, j.obj, j.mult_g, j.mult_x_L, j.mult_x_U, j.status, j.iters, j.factorizations, &r.seconds);
      } catch (const std::exception& e) { r.rc = -199; r.err = e.what(); }
      catch (...) { r.rc = -199; r.err = "unknown exception"; }
      lk.lock();
      done[j.ticket] = r;
      inflight.erase(j.ticket);
      sl->busy = false;
      cv_done.notify_all();
    }
  }
};
extern "C" {
dnlp_batch_stream* dnlp_batch_stream_create(dnlp_problem* vp, int nslots) {
  dnlp_problem_t* p = vp;
  try {
    if (nslots < 1 || nslots > 16) throw std::runtime_error("dnlp_batch_stream_create: 1 .. 16 slots");
    BatchRunner& main = *batch_runner(p);
    if (main.aff_P < 0) throw std::runtime_error("dnlp_batch_stream_create: set the affine parameter map first (dnlp_batch_set_affine_map)");
    p->ex.sync();
    std::unique_ptr<dnlp_batch_stream> s(new dnlp_batch_stream());
    s->p = p;
    for (int k = 0; k < nslots; ++k) {
      // a slot = a runner of its own over the SAME tape and plan: own stream, own device buffers, own copy of the map
      std::unique_ptr<dnlp_batch_stream::Slot> sl(new dnlp_batch_stream::Slot());
      sl->runner.init(&p->ex, p->model.owner, true);
      if (main.have_sparse) { sl->runner.set_sparse_plan(*main.host_plan); sl->runner.force_sparse = main.force_sparse; }
      sl->runner.set_affine_map(main.aff_P, main.h_aff_d0.data(), main.h_aff_theta0.data(), main.h_aff_indptr.data(), main.h_aff_indices.data(),
                                main.h_aff_vals.data());
      sl->runner.wave_prepare();            // (reads the tape's index arrays back through the handle's stream: here, on the caller's thread)
      s->slots.push_back(std::move(sl));
    }
    p->ex.sync();
    dnlp_batch_stream* raw = s.get();
    for (auto& sl : s->slots) { dnlp_batch_stream::Slot* q = sl.get(); q->th = std::thread([raw, q] { raw->worker(q); }); }
    return s.release();
  } catch (const std::exception& e) { dnlp::tls_error() = e.what(); return nullptr; }
}
int dnlp_batch_stream_submit(dnlp_batch_stream* s, int batch, const double* theta, int n_params, double* x, double* obj, double* mult_g,
                             double* mult_x_L, double* mult_x_U, int* status, int* iters, int* factorizations) {
  DNLP_TRY(
    if (s->p->opt.hessian_approximation == 1) return batch_rejects_limited_memory();
    std::unique_lock<std::mutex> lk(s->mu);
    dnlp_batch_stream::Slot* sl = nullptr;
    auto idle = [&] { for (auto& q : s->slots) if (!q->busy) { sl = q.get(); return true; } return false; };
    s->cv_done.wait(lk, idle);                          // every slot busy: whichever finishes first
    dnlp_batch_stream::Job& j = sl->job;
    j.ticket = s->next_ticket++;
    j.batch = batch; j.theta = theta; j.n_params = n_params;
    j.x = x; j.obj = obj; j.mult_g = mult_g; j.mult_x_L = mult_x_L; j.mult_x_U = mult_x_U;
    j.status = status; j.iters = iters; j.factorizations = factorizations;
    j.opt = s->p->opt;
    sl->busy = true; sl->has_job = true;
    s->inflight.insert(j.ticket);
    sl->cv.notify_one();
    return j.ticket;)
}
int dnlp_batch_stream_wait(dnlp_batch_stream* s, int ticket, double* kernel_seconds) {
  DNLP_TRY(
    std::unique_lock<std::mutex> lk(s->mu);
    if (ticket < 0 || ticket >= s->next_ticket) { dnlp::tls_error() = "dnlp_batch_stream_wait: no such ticket"; return -1; }
    if (!s->done.count(ticket) && !s->inflight.count(ticket)) {
      dnlp::tls_error() = "dnlp_batch_stream_wait: that ticket has been waited for already (a result is handed out once)";
      return -2;
    }
    s->cv_done.wait(lk, [&] { return s->done.count(ticket) != 0; });
    const dnlp_batch_stream::Result r = s->done[ticket];
    s->done.erase(ticket);
    if (kernel_seconds) *kernel_seconds = r.seconds;
    if (r.rc != 0) dnlp::tls_error() = r.err;
    return r.rc;)
}
void dnlp_batch_stream_destroy(dnlp_batch_stream* s) {
  if (!s) return;
  {
    std::unique_lock<std::mutex> lk(s->mu);
    s->cv_done.wait(lk, [&] { return s->inflight.empty(); });      // (what was submitted is finished: the caller's arrays are being written)
    s->stop = true;
    for (auto& sl : s->slots) sl->cv.notify_one();
  }
  for (auto& sl : s->slots) if (sl->th.joinable()) sl->th.join();
  delete s;
}

int dnlp_batch_keep_result_rows(dnlp_problem* vp, int on) {
  dnlp_problem_t* p = vp;
  DNLP_TRY(
    batch_runner(p)->rows_wanted = on != 0;
    return 0;)
}
int dnlp_batch_result_rows(dnlp_problem* vp, const double** rows, int64_t* n_rows, int64_t* width) {
  dnlp_problem_t* p = vp;
  DNLP_TRY(
    BatchRunner& r = *batch_runner(p);
    if (!r.rows_wanted) { dnlp::tls_error() = "dnlp_batch_result_rows: ask for them first (dnlp_batch_keep_result_rows)"; return -1; }
    if (r.rows_width == 0) { dnlp::tls_error() = "dnlp_batch_result_rows: no launch yet"; return -1; }
    *rows = r.rows_batch > 0 ? r.d_rows : nullptr;
    *n_rows = r.rows_batch;
    *width = r.rows_width;
    return 0;)
}

int dnlp_batch_launch_info(dnlp_problem* vp, int32_t* out8) {
  dnlp_problem_t* p = vp;
  DNLP_TRY(
    BatchRunner& r = *batch_runner(p);
    out8[0] = r.last_grid; out8[1] = r.last_threads; out8[2] = r.last_lds_mode; out8[3] = r.last_per_cu; out8[4] = r.last_packed ? 1 : 0;
    out8[5] = r.last_order_lpt ? 1 : 0; out8[6] = r.last_wave; out8[7] = r.last_wave_refused;
    return 0;)
}

// Average seconds of one fused f + grad f evaluation with x resident in HBM (HIP events around
// `reps` back-to-back evaluations on the problem's stream): the measurement behind the C2 roofline
// line (tools/run_c2.py).
int dnlp_time_fused(dnlp_problem* vp, const double* xfree, int reps, double* seconds) {
  dnlp_problem_t* p = vp;
  DNLP_TRY(
    if (!p->fused.present) { dnlp::tls_error() = "no fused objective program in this tape"; return -11; }
    const size_t nf = static_cast<size_t>(p->fused.nfree);
    double* dx = p->ex.alloc<double>(nf);
    double* dg = p->ex.alloc<double>(nf);
    p->ex.h2d(dx, xfree, 8 * nf);
    p->fused.eval(dx, dg);
    hipEvent_t e0, e1;
    DNLP_HIP_CHECK(hipEventCreate(&e0));
    DNLP_HIP_CHECK(hipEventCreate(&e1));
    DNLP_HIP_CHECK(hipEventRecord(e0, p->ex.stream));
    // the generated kernel is launched back to back (no scalar read-back between launches: kernel time,
    // not host round trips); the interpreter path keeps its evaluation call
    for (int r = 0; r < reps; ++r)
      if (!p->ex.fused_generated_launch(p->fused.progs, dx, p->fused.consts, dg, p->fused.nfree)) p->fused.eval(dx, dg);
    DNLP_HIP_CHECK(hipEventRecord(e1, p->ex.stream));
    DNLP_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    DNLP_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    hipEventDestroy(e0); hipEventDestroy(e1);
    p->ex.release(dx); p->ex.release(dg);
    *seconds = 1e-3 * ms / (reps > 0 ? reps : 1);
    return 0;)
}

// Generated-kernel self check (no GPU needed: hiprtc cross-compiles): parse the fused programs of a
// tape blob, generate the specialised f + grad f kernel with `elems_per_lane` entries per lane and
// compile it for gfx950.  0 = compiled, 1 = this objective has no generated form (reason in log),
// 2 = compiler error (text in log), -11 = the tape carries no fused program.
int dnlp_fused_codegen_check(const void* blob, size_t len, int elems_per_lane, char* src_out, size_t src_cap, char* log_out,
                             size_t log_cap) {
  DNLP_TRY(
    auto put = [](char* dst, size_t cap, const std::string& t) {
      if (!dst || !cap) return;
      const size_t n = t.size() < cap - 1 ? t.size() : cap - 1;
      std::memcpy(dst, t.data(), n);
      dst[n] = 0;
    };
    TapeBlob tb(blob, len);
    std::vector<FusedSlotProg> progs;
    i64 nconst = 0; i64 nfree = 0; double c0 = 0.0;
    if (!fused_parse_programs(tb, progs, nconst, c0, nfree)) { put(log_out, log_cap, "no fused program in this tape"); return -11; }
    const FusedCodegenInfo info = fused_codegen_plan(progs, elems_per_lane > 0 ? elems_per_lane : 4);
    if (!info.ok) { put(log_out, log_cap, info.why); return 1; }
    const std::string src = fused_codegen_eval_source(progs, info);
    put(src_out, src_cap, src);
    std::string log;
    const std::vector<char> code = rtc_compile(src, log, false);
    put(log_out, log_cap, log);
    return code.empty() ? 2 : 0;)
}

// Same self check for the device-resident L-BFGS translation unit (lbfgs_codegen.h): the four slot
// kernels around the generated element code.
int dnlp_lbfgs_codegen_check(const void* blob, size_t len, int elems_per_lane, char* src_out, size_t src_cap, char* log_out,
                             size_t log_cap) {
  DNLP_TRY(
    auto put = [](char* dst, size_t cap, const std::string& t) {
      if (!dst || !cap) return;
      const size_t n = t.size() < cap - 1 ? t.size() : cap - 1;
      std::memcpy(dst, t.data(), n);
      dst[n] = 0;
    };
    TapeBlob tb(blob, len);
    std::vector<FusedSlotProg> progs;
    i64 nconst = 0; i64 nfree = 0; double c0 = 0.0;
    if (!fused_parse_programs(tb, progs, nconst, c0, nfree)) { put(log_out, log_cap, "no fused program in this tape"); return -11; }
    const FusedCodegenInfo info = fused_codegen_plan(progs, elems_per_lane > 0 ? elems_per_lane : 4);
    if (!info.ok) { put(log_out, log_cap, info.why); return 1; }
    // with the persistent single-launch kernel when a slice per compute unit (256 of them) fits its LDS, as the
    // execution space decides (exec_hip.h: lbfgs_generated_solve)
    long long per = 0;
    int pmode = 0;
    if (nfree >= 1024 && info.hi - info.lo >= 1) {
      per = ((nfree + 255) / 256 + info.E - 1) / info.E * info.E;
      pmode = (2 * 10 + 5) * (per + 64) * 8 + 12 * 1024 <= 150 * 1024 ? 0 : 5 * (per + 64) * 8 + 12 * 1024 <= 150 * 1024 ? 1 : 2;
      if (const char* v = std::getenv("DNLP_LBFGS_PERSIST_MODE")) { const int w = std::atoi(v); if (w >= pmode && w <= 2) pmode = w; }
    }
    const std::string src = lbfgs_codegen_source(progs, info, 10, per, pmode);
    put(src_out, src_cap, src);
    std::string log;
    const std::vector<char> code = rtc_compile(src, log, false);
    put(log_out, log_cap, log);
    return code.empty() ? 2 : 0;)
}

int dnlp_rtc_compiler(char* out, size_t cap) {
  DNLP_TRY(
    const RtcChoice c = rtc_choice();
    if (out && cap) {
      const size_t n = c.identity.size() < cap - 1 ? c.identity.size() : cap - 1;
      std::memcpy(out, c.identity.data(), n);
      out[n] = 0;
    }
    return c.clang.empty() ? 0 : 1;)
}

const char* dnlp_version(void) { return "dnlp_amd 0.1.0 (gfx950)"; }

int dnlp_dev_alloc(int device, size_t bytes, void** out) {
  DNLP_TRY(DNLP_HIP_CHECK(hipSetDevice(device)); DNLP_HIP_CHECK(hipMalloc(out, bytes)); return 0;)
}
int dnlp_dev_free(int device, void* p) {
  DNLP_TRY(DNLP_HIP_CHECK(hipSetDevice(device)); DNLP_HIP_CHECK(hipFree(p)); return 0;)
}
int dnlp_dev_copy(int device, void* dst, const void* src, size_t bytes, int kind) {
  DNLP_TRY(DNLP_HIP_CHECK(hipSetDevice(device));
           DNLP_HIP_CHECK(hipMemcpy(dst, src, bytes, kind == 0 ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
           return 0;)
}

int dnlp_gen_symmetric(int device, double* A, int64_t n, int64_t ld, uint64_t seed, double spike, double* dv) {
  DNLP_TRY(
    DNLP_HIP_CHECK(hipSetDevice(device));
    double* v = dv;
    if (!v) DNLP_HIP_CHECK(hipMalloc(&v, sizeof(double) * n));
    hipLaunchKernelGGL(gen_vec_kernel, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, 0, v, n, seed);
    const i64 nrb = (n + 255) / 256;
    hipLaunchKernelGGL(gen_sym_kernel, dim3(static_cast<unsigned>(nrb), static_cast<unsigned>(n < 16384 ? n : 16384)),
                       dim3(256), 0, 0, A, n, ld, seed, spike, v);
    DNLP_HIP_CHECK(hipGetLastError());
    DNLP_HIP_CHECK(hipDeviceSynchronize());
    if (!dv) DNLP_HIP_CHECK(hipFree(v));
    return 0;)
}

int dnlp_dev_symv(int device, const double* A, int64_t n, int64_t ld, const double* x, double* y) {
  DNLP_TRY(
    HipExec ex(device);
    double* dx = ex.alloc<double>(static_cast<size_t>(n));
    double* dy = ex.alloc<double>(static_cast<size_t>(n));
    ex.h2d(dx, x, sizeof(double) * n);
    ex.gemv_sym(n, A, ld, dx, dy);
    ex.d2h(y, dy, sizeof(double) * n);
    return 0;)
}

// ---- host side of the lowering (linform.h, lower_maps.h): no device is touched -------------------------------------
struct dnlp_linform : dnlp::LinFormH {};
#define DNLP_LF_TRY(...) try { __VA_ARGS__ } catch (const std::exception& e) { dnlp::tls_error() = e.what(); return nullptr; }
static dnlp_linform* lf_wrap(dnlp::LinFormH* f) { return static_cast<dnlp_linform*>(f); }
dnlp_linform* dnlp_lf_const(int64_t ncol, int64_t n, const double* b) { DNLP_LF_TRY(return lf_wrap(dnlp::lf_const(ncol, n, b));) }
dnlp_linform* dnlp_lf_range(int64_t ncol, int64_t n, int64_t col0) { DNLP_LF_TRY(return lf_wrap(dnlp::lf_range(ncol, n, col0));) }
dnlp_linform* dnlp_lf_select(const dnlp_linform* a, const int64_t* sel, int64_t n) {
  DNLP_LF_TRY(return lf_wrap(dnlp::lf_select(*a, reinterpret_cast<const long long*>(sel), n));)
}
dnlp_linform* dnlp_lf_add(dnlp_linform* a, dnlp_linform* b) { DNLP_LF_TRY(return lf_wrap(dnlp::lf_add(*a, *b));) }
/* diag(s) a; s == NULL: -a */
dnlp_linform* dnlp_lf_scale(const dnlp_linform* a, const double* s) { DNLP_LF_TRY(return lf_wrap(dnlp::lf_scale_rows(*a, s));) }
/* S a, S a constant CSR matrix with srows rows and a's row count as columns */
dnlp_linform* dnlp_lf_apply_csr(const dnlp_linform* a, int64_t srows, const int64_t* s_ptr, const int32_t* s_idx, const double* s_val) {
  DNLP_LF_TRY(return lf_wrap(dnlp::lf_apply_csr(*a, srows, reinterpret_cast<const long long*>(s_ptr), s_idx, s_val));)
}
/* S a, S a dense constant (srows x a's row count, row-major); exact zeros of S carry no entry */
dnlp_linform* dnlp_lf_apply_dense(const dnlp_linform* a, int64_t srows, const double* S) {
  DNLP_LF_TRY(return lf_wrap(dnlp::lf_apply_dense(*a, srows, S));)
}
dnlp_linform* dnlp_lf_vstack(dnlp_linform* const* parts, int n) {
  DNLP_LF_TRY(std::vector<dnlp::LinFormH*> v(parts, parts + n); return lf_wrap(dnlp::lf_vstack(v.data(), n));)
}
void dnlp_lf_free(dnlp_linform* a) { delete a; }
/* info[0..3] = rows, columns, stored coefficients, 1 when the form is a plain selection of columns */
int dnlp_lf_info(const dnlp_linform* a, int64_t* info) {
  info[0] = a->rows; info[1] = a->ncol; info[2] = a->nnz(); info[3] = dnlp::lf_is_selection(*a) ? 1 : 0;
  return 0;
}
/* 1 and the column of every row in out[0..rows) when the form is a plain selection of columns below n_cols, else 0 */
int dnlp_lf_gather(const dnlp_linform* a, int64_t n_cols, int64_t* out) {
  if (!dnlp::lf_is_selection(*a)) return 0;
  for (long long r = 0; r < a->rows; ++r) {
    if (a->idx[static_cast<size_t>(r)] >= n_cols) return 0;
    out[r] = a->idx[static_cast<size_t>(r)];
  }
  return 1;
}
/* copies out the CSR arrays (canonical: rows sorted, duplicates summed) and the constants; any pointer may be NULL */
int dnlp_lf_export(dnlp_linform* a, int64_t* ptr, int32_t* idx, double* val, double* b) {
  dnlp::lf_canonicalize(*a);
  if (ptr) std::memcpy(ptr, a->ptr.data(), a->ptr.size() * sizeof(int64_t));
  if (idx && !a->idx.empty()) std::memcpy(idx, a->idx.data(), a->idx.size() * sizeof(int32_t));
  if (val && !a->val.empty()) std::memcpy(val, a->val.data(), a->val.size() * sizeof(double));
  if (b && !a->b.empty()) std::memcpy(b, a->b.data(), a->b.size() * sizeof(double));
  return 0;
}
/* the same arrays without a copy: pointers into the handle (canonical form), valid until dnlp_lf_free */
int dnlp_lf_view(dnlp_linform* a, const int64_t** ptr, const int32_t** idx, const double** val, const double** b) {
  dnlp::lf_canonicalize(*a);
  static_assert(sizeof(long long) == sizeof(int64_t), "row pointers are exported as int64");
  *ptr = reinterpret_cast<const int64_t*>(a->ptr.data()); *idx = a->idx.data(); *val = a->val.data(); *b = a->b.data();
  return 0;
}

/* 1 when the dense n x n matrix (row-major, leading dimension ld) is exactly symmetric, else 0 (threads; the check of
 * quad_form's constant — 800 MB at BASELINE C3 — was the largest single cost of that problem's lowering in numpy) */
int dnlp_is_symmetric(const double* P, int64_t n, int64_t ld) { return dnlp::lm_is_symmetric(P, n, ld) ? 1 : 0; }

struct dnlp_lowered { dnlp::LowerMapsOut out; };

dnlp_lowered* dnlp_lower_maps(int64_t N, int64_t Z, int64_t m, int64_t nd, int64_t nh, const int64_t* G_ptr, const int32_t* G_idx,
                              const double* G_val, const double* c, const int64_t* drow, const int64_t* dcol, const int64_t* hrow,
                              const int64_t* hcol, int n_blocks, const int64_t* block_x0, const int64_t* block_n) {
  try {
    dnlp::LowerMapsIn in;
    in.N = N; in.Z = Z; in.m = m; in.nd = nd; in.nh = nh;
    in.Gp = reinterpret_cast<const dnlp::lm_i64*>(G_ptr); in.Gi = G_idx; in.Gv = G_val; in.c = c;
    in.drow = reinterpret_cast<const dnlp::lm_i64*>(drow); in.dcol = reinterpret_cast<const dnlp::lm_i64*>(dcol);
    in.hrow = reinterpret_cast<const dnlp::lm_i64*>(hrow); in.hcol = reinterpret_cast<const dnlp::lm_i64*>(hcol);
    in.nblk = n_blocks;
    in.blk_x0 = reinterpret_cast<const dnlp::lm_i64*>(block_x0); in.blk_n = reinterpret_cast<const dnlp::lm_i64*>(block_n);
    auto* h = new dnlp_lowered();
    dnlp::lower_maps_build(in, h->out);
    return h;
  } catch (const std::exception& e) { dnlp::tls_error() = e.what(); return nullptr; }
}
void dnlp_lowered_free(dnlp_lowered* h) { delete h; }
/* sizes[0..8] = G_changed, nnz(G), nnz(Mg), nnz(Mw), nnz(MJ), nnzJ, nnz(MH), nnzH, jac_is_G */
int dnlp_lowered_sizes(const dnlp_lowered* h, int64_t* sizes) {
  const dnlp::LowerMapsOut& o = h->out;
  sizes[8] = o.jac_is_G;
  sizes[0] = o.G_changed; sizes[1] = static_cast<int64_t>(o.G.idx.size()); sizes[2] = static_cast<int64_t>(o.Mg.idx.size());
  sizes[3] = static_cast<int64_t>(o.Mw.idx.size()); sizes[4] = static_cast<int64_t>(o.MJ.idx.size());
  sizes[5] = static_cast<int64_t>(o.MJ.ptr.empty() ? 0 : o.MJ.ptr.size() - 1); sizes[6] = static_cast<int64_t>(o.MH.idx.size()); sizes[7] = static_cast<int64_t>(o.hr.size());
  return 0;
}
/* which: 0 G (only when G_changed), 1 Mg, 2 Mw, 3 MJ, 4 MH; the caller's arrays have the sizes of dnlp_lowered_sizes */
int dnlp_lowered_csr(const dnlp_lowered* h, int which, int64_t* ptr, int32_t* idx, double* val) {
  const dnlp::LowerMapsOut& o = h->out;
  const dnlp::LmCsr* M = which == 0 ? &o.G : which == 1 ? &o.Mg : which == 2 ? &o.Mw : which == 3 ? &o.MJ : which == 4 ? &o.MH : nullptr;
  if (!M) { dnlp::tls_error() = "dnlp_lowered_csr: no such map"; return -1; }
  if (!M->ptr.empty()) std::memcpy(ptr, M->ptr.data(), M->ptr.size() * sizeof(int64_t));
  if (!M->idx.empty()) { std::memcpy(idx, M->idx.data(), M->idx.size() * sizeof(int32_t)); std::memcpy(val, M->val.data(), M->val.size() * sizeof(double)); }
  return 0;
}
/* the same map without a copy: pointers into the handle, valid until dnlp_lowered_free */
int dnlp_lowered_csr_view(const dnlp_lowered* h, int which, const int64_t** ptr, const int32_t** idx, const double** val) {
  const dnlp::LowerMapsOut& o = h->out;
  const dnlp::LmCsr* M = which == 0 ? &o.G : which == 1 ? &o.Mg : which == 2 ? &o.Mw : which == 3 ? &o.MJ : which == 4 ? &o.MH : nullptr;
  if (!M) { dnlp::tls_error() = "dnlp_lowered_csr_view: no such map"; return -1; }
  *ptr = reinterpret_cast<const int64_t*>(M->ptr.data()); *idx = M->idx.data(); *val = M->val.data();
  return 0;
}
/* which: 0 Jacobian (rows, cols, Jc), 1 Hessian (rows, cols; vals ignored) */
int dnlp_lowered_pattern(const dnlp_lowered* h, int which, int32_t* rows, int32_t* cols, double* vals) {
  const dnlp::LowerMapsOut& o = h->out;
  const std::vector<dnlp::lm_i32>& r = which == 0 ? o.jr : o.hr;
  const std::vector<dnlp::lm_i32>& c = which == 0 ? o.jc : o.hc;
  if (!r.empty()) { std::memcpy(rows, r.data(), r.size() * sizeof(int32_t)); std::memcpy(cols, c.data(), c.size() * sizeof(int32_t)); }
  if (which == 0 && vals && !o.Jc.empty()) std::memcpy(vals, o.Jc.data(), o.Jc.size() * sizeof(double));
  return 0;
}
/* dense block b: *mode = 2 (contiguous run: pos[0] is its first position) or 1 (table); *count = entries of pos */
int dnlp_lowered_block(const dnlp_lowered* h, int b, int* mode, int64_t* count, int64_t* pos) {
  const dnlp::LowerMapsOut& o = h->out;
  if (b < 0 || b >= static_cast<int>(o.blk_mode.size())) { dnlp::tls_error() = "dnlp_lowered_block: no such block"; return -1; }
  *mode = o.blk_mode[static_cast<size_t>(b)];
  *count = static_cast<int64_t>(o.blk_pos[static_cast<size_t>(b)].size());
  if (pos) std::memcpy(pos, o.blk_pos[static_cast<size_t>(b)].data(), o.blk_pos[static_cast<size_t>(b)].size() * sizeof(int64_t));
  return 0;
}

int dnlp_ldlt_host(int device, double* A, int64_t n, int64_t ld, int32_t* ipiv, int pivoted, int* nneg, int* nzero,
                   const double* rhs, double* sol, double* seconds) {
  DNLP_TRY(
    HipExec ex(device);
    const i64 ldd = (n + 7) / 8 * 8;
    double* dA = ex.alloc<double>(static_cast<size_t>(ldd) * n + 256);
    i32* dp = ex.alloc<i32>(static_cast<size_t>(n));
    double* db = ex.alloc<double>(static_cast<size_t>(n));
    DNLP_HIP_CHECK(hipMemcpy2DAsync(dA, ldd * 8, A, ld * 8, n * 8, n, hipMemcpyHostToDevice, ex.stream));
    ex.sync();
    HipExec::LdltWork w;
    w.padded = true;
    ex.ldlt_prepare(w, n, ldd, pivoted != 0);
    double t0 = now_sec();
    bool ok = ex.ldlt_factor(w, dA, n, ldd, dp, pivoted != 0, nneg, nzero);
    ex.sync();
    if (seconds) *seconds = now_sec() - t0;
    if (rhs && sol) {
      ex.h2d(db, rhs, sizeof(double) * n);
      ex.ldlt_solve(w, dA, n, ldd, dp, pivoted != 0, db);
      ex.d2h(sol, db, sizeof(double) * n);
    }
    DNLP_HIP_CHECK(hipMemcpy2DAsync(A, ld * 8, dA, ldd * 8, n * 8, n, hipMemcpyDeviceToHost, ex.stream));
    if (ipiv) { ex.sync(); ex.d2h(ipiv, dp, sizeof(i32) * n); }
    ex.sync();
    return ok ? 0 : 1;)
}

int dnlp_ldlt_device(int device, double* dA, int64_t n, int64_t ld, int* nneg, int* nzero, double* seconds,
                     double* update_seconds) {
  DNLP_TRY(
    HipExec ex(device);
    BlockedLdlt bl;
    bl.init(&ex, n, ld);
    bl.padded = true;   // contract of this entry point: >= 1 KiB of slack behind the matrix
    bl.time_updates = update_seconds != nullptr;
    ex.sync();
    double t0 = now_sec();
    bool ok = bl.factor(dA, nneg, nzero);
    ex.sync();
    if (seconds) *seconds = now_sec() - t0;
    if (update_seconds) *update_seconds = bl.last_update_seconds;
    return ok ? 0 : 1;)
}

}  // extern "C"
