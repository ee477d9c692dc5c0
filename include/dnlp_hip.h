/* dnlp_hip.h — C ABI of the MI355X-native disciplined-NLP solve path (libdnlp_hip.so).
 *
 * This is the drop-in boundary for the `solve(nlp=True)` hot path of cvxgrp/DNLP.  Each entry
 * point cites the reference interface it replaces (paths relative to the reference tree).
 * All pointers are plain host pointers unless a parameter says "device"; the caller owns every
 * buffer; a non-zero int return means failure (dnlp_last_error() has the text), mirroring
 * IPOPT's "eval_* returned false".  One dnlp_problem owns one HIP stream and is not
 * thread-safe; distinct problems may be used from distinct threads / devices.
 *
 * Oracle level — lets any IPOPT-C-interface-shaped solver drive the GPU tape.  Replaces the
 * Python callback object cyipopt receives as `problem_obj`
 * (cvxpy/reductions/solvers/nlp_solvers/nlp_solver.py:181-427, handed over at
 * nlp_solvers/ipopt_nlpif.py:143-151): objective/gradient/constraints/jacobian(+structure)/
 * hessian(+structure) == IPOPT's eval_f / eval_grad_f / eval_g / eval_jac_g / eval_h.
 *
 * Solver level — replaces the dispatch `nlp.solve(x0)` into third-party IPOPT
 * (nlp_solvers/ipopt_nlpif.py:153-170): options by name, status as IPOPT's
 * ApplicationReturnStatus integer (table at ipopt_nlpif.py:31-61).
 */
#ifndef DNLP_HIP_H
#define DNLP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dnlp_problem dnlp_problem;

/* ---- library / device ------------------------------------------------------------------ */
int dnlp_device_count(void);                 /* number of visible HIP devices (0: none)     */
const char* dnlp_last_error(void);           /* text of the last failure on this thread     */
const char* dnlp_version(void);

/* ---- lifecycle ------------------------------------------------------------------------- */
/* Build a device problem from a serialised tape (dnlp_amd/tape.py).  Replaces the construction
 * of `Oracles(problem, x0, m)` (nlp_solver.py:181-203) + `cyipopt.Problem(n, m, problem_obj,
 * lb, ub, cl, cu)` (ipopt_nlpif.py:143-151).  Returns NULL on failure. */
dnlp_problem* dnlp_create(const void* tape_blob, size_t len, int device);
/* The same tape as separate named arrays (name, dtype 0 = f64 / 1 = i32 / 2 = i64, element count, pointer
 * aligned to the element size): nothing is serialised, each array is read where the caller has it and
 * copied into HBM during the call -- the arrays need not outlive it.  For tapes whose dense constant
 * blocks run to gigabytes (BASELINE config C3: 1.16 GB; the reference hands such blocks to its
 * oracles by reference too: `Constant` values held by the expression tree, atoms/quad_form.py:33-47). */
typedef struct dnlp_tape_array {
  const char* name;
  int32_t dtype;
  int32_t reserved;
  uint64_t count;
  const void* data;
} dnlp_tape_array;
dnlp_problem* dnlp_create_arrays(const dnlp_tape_array* arrays, int n_arrays, int device);
void dnlp_destroy(dnlp_problem* p);
/* Bind a device-resident dense FP64 column-major matrix (order n, leading dimension ld) as
 * constant `const_id` of the tape (quad_form matrices too large to travel through the host;
 * the reference keeps them as a NumPy Constant, atoms/quad_form.py:33-47). */
int dnlp_bind_dense(dnlp_problem* p, int const_id, const double* device_ptr, int64_t ld);

/* ---- dimensions / bounds  (data dict of nlp_solver.py:62-79) ----------------------------- */
int dnlp_dims(dnlp_problem* p, int64_t* n, int64_t* m, int64_t* nnz_jac, int64_t* nnz_hess);
int dnlp_bounds(dnlp_problem* p, double* lb, double* ub, double* cl, double* cu, double* x0);

/* ---- oracle level ------------------------------------------------------------------------ */
/* Oracles.objective  (nlp_solver.py:212-216)  == IPOPT eval_f */
int dnlp_eval_f(dnlp_problem* p, const double* x, int new_x, double* f);
/* Oracles.gradient   (nlp_solver.py:218-235)  == IPOPT eval_grad_f */
int dnlp_eval_grad_f(dnlp_problem* p, const double* x, int new_x, double* grad);
/* Oracles.constraints (nlp_solver.py:237-244) == IPOPT eval_g */
int dnlp_eval_g(dnlp_problem* p, const double* x, int new_x, double* g);
/* Oracles.jacobianstructure / jacobian (nlp_solver.py:278-335) == IPOPT eval_jac_g:
 * structure is returned when vals == NULL (0-based, row-major sorted COO). */
int dnlp_eval_jac_g(dnlp_problem* p, const double* x, int new_x, int32_t* iRow, int32_t* jCol,
                    double* vals);
/* Oracles.hessianstructure / hessian (nlp_solver.py:374-421) == IPOPT eval_h:
 * lower triangle of sigma*Hess f + sum lambda_i Hess g_i; structure when vals == NULL. */
int dnlp_eval_h(dnlp_problem* p, const double* x, int new_x, double sigma, const double* lambda,
                int new_lambda, int32_t* iRow, int32_t* jCol, double* vals);

/* ---- solver level ------------------------------------------------------------------------ */
/* nlp.add_option(name, value) (ipopt_nlpif.py:161-168); numeric values are passed as text. */
int dnlp_set_option(dnlp_problem* p, const char* key, const char* val);
/* Every option back to its default (a handle kept for the next solve of the same problem starts from
 * the defaults the reference sets on a fresh cyipopt.Problem, ipopt_nlpif.py:153-160). */
int dnlp_reset_options(dnlp_problem* p);
/* nlp.solve(x0) (ipopt_nlpif.py:170).  x_inout: start point in, solution out.  Any output
 * pointer may be NULL.  Returns the IPOPT ApplicationReturnStatus integer. */
int dnlp_solve(dnlp_problem* p, double* x_inout, double* obj, double* g, double* mult_g,
               double* mult_x_L, double* mult_x_U, int* iters);
/* Per-iteration callback: Oracles.intermediate (nlp_solver.py:423-427), which cyipopt invokes once
 * per iteration with IPOPT's intermediate_callback values (alg_mod 0 regular / 1 restoration).
 * Called at iteration 0 and after every accepted iterate of dnlp_solve / dnlp_ipm_step, on the
 * calling thread.  Return non-zero to continue, zero to stop: the solve then ends with status 5
 * (User_Requested_Stop, ipopt_nlpif.py:31-61), the iterate so far being returned.  NULL removes it.
 * The batched entry points run entirely on the device and do not call back. */
typedef int (*dnlp_intermediate_cb)(int alg_mod, int iter_count, double obj_value, double inf_pr,
                                    double inf_du, double mu, double d_norm,
                                    double regularization_size, double alpha_du, double alpha_pr,
                                    int ls_trials, void* user_data);
int dnlp_set_intermediate_cb(dnlp_problem* p, dnlp_intermediate_cb cb, void* user_data);
/* Stepwise form of the same loop (used by bench.py to time exactly K iterations):
 * begin() initialises from x0; step() performs up to `max_steps` iterations and returns 99
 * while the loop should continue, otherwise the final status; *steps_done counts only iterations
 * that were carried out (the call that merely detects convergence or a limit adds nothing);
 * finish() extracts results. */
int dnlp_ipm_begin(dnlp_problem* p, const double* x0);
int dnlp_ipm_step(dnlp_problem* p, int max_steps, int* steps_done);
int dnlp_ipm_finish(dnlp_problem* p, double* x, double* obj, double* g, double* mult_g,
                    double* mult_x_L, double* mult_x_U, int* iters);
/* Unconstrained problems whose canonical constraints only define auxiliary variables
 * (BASELINE config C2): reduced-space L-BFGS on the user's variables, built from tape f / grad f
 * evaluations and a line search only — no KKT system.  Same role as nlp.solve(x0) for such
 * problems (ipopt_nlpif.py:170).  Returns 0 converged, -1 iteration limit, 3 line search
 * failure, -11 when the tape has no reduced-space structure. */
int dnlp_solve_reduced(dnlp_problem* p, double* x_inout, double* obj, int* iters, int* evals,
                       double* gnorm);
/* How the last dnlp_solve_reduced ran: out[0] = 1 when the device-resident loop did (every decision on
 * the device, option lbfgs_device_loop=no turns it off), out[1] = its seconds (enqueue to final state
 * read-back), out[2] = line-search slots enqueued, out[3] = 1 when the fused objective was used,
 * out[4] = wall seconds of the whole call (n <= 6 values). */
int dnlp_reduced_info(dnlp_problem* p, double* out, int n);
/* Batched variant (SURVEY.md 8b "dnlp_solve_batch"; BASELINE config C5): `batch` independent
 * instances that share the structure of p's tape and differ in data — the role of the
 * reference's serial best_of / re-solve loop (problems/problem.py:1256-1269, which
 * re-canonicalises and calls nlp.solve once per instance).  ONE kernel launch: one workgroup
 * per instance runs the whole interior-point loop on the device (csrc/batch.h).
 * data: batch x stride doubles, row layout
 *   c0(1) c(N+Z) b(m) Jc(nnzJ) G_val Mg_val Mw_val MJ_val MH_val seg_param(nseg) seg_param2(nseg)
 *   x0(N) lb(N) ub(N) cl(m) cu(m)            (stride = dnlp_batch_stride(p))
 * Outputs are batch-major host arrays; mult_* may be NULL.  status[i] is the IPOPT status
 * integer of instance i.  Options are the ones set with dnlp_set_option. */
int64_t dnlp_batch_stride(dnlp_problem* p);
int dnlp_solve_batch(dnlp_problem* p, int batch, const double* data, int64_t stride, double* x,
                     double* obj, double* mult_g, double* mult_x_L, double* mult_x_U, int* status,
                     int* iters, int* factorizations, double* kernel_seconds);
/* Per-instance multipliers (batch-major: batch x m, batch x N, batch x N) consumed by the next
 * dnlp_solve_batch when the option warm_start_init_point=yes is set (the rows' x0 carry the primal
 * start). */
int dnlp_batch_warm_start(dnlp_problem* p, int batch, const double* mult_g, const double* mult_x_L,
                          const double* mult_x_U);
/* Same, plus per-instance phase times on the device clock:
 * times[4*i + 0..3] = whole solve, tape evaluations, KKT factorisations, KKT solves (seconds). */
int dnlp_solve_batch_timed(dnlp_problem* p, int batch, const double* data, int64_t stride, double* x,
                           double* obj, double* mult_g, double* mult_x_L, double* mult_x_U,
                           int* status, int* iters, int* factorizations, double* kernel_seconds,
                           double* times);
/* Parametrised batches (the reference re-canonicalises and re-solves one Problem per parameter value,
 * problems/problem.py:1256-1269; a Parameter's value reaches the tape data through the lowering).  When the
 * instance data is an affine function of the P parameter values — dnlp_amd.batch.ParametricBatch recovers and
 * checks the map — it is handed over ONCE: d0 = data row of the base instance (dnlp_batch_stride doubles),
 * theta0 = its parameter values (P), (indptr[stride + 1], indices, vals) = CSR of the stride x P sensitivity.
 * dnlp_solve_batch_theta then takes batch x P parameter rows and generates the instance data on the device.
 * batch = 0 (an empty shard) is a launch of nothing: returns 0, writes nothing; batch < 0 is an error.  Same for
 * dnlp_solve_batch / _timed and a stream's submit. */
int dnlp_batch_set_affine_map(dnlp_problem* p, int n_params, const double* d0, const double* theta0,
                              const int64_t* indptr, const int32_t* indices, const double* vals);
int dnlp_solve_batch_theta(dnlp_problem* p, int batch, const double* theta, int n_params, double* x, double* obj,
                           double* mult_g, double* mult_x_L, double* mult_x_U, int* status, int* iters,
                           int* factorizations, double* kernel_seconds, double* times);

/* A STREAM of parametrised batches with several launches in flight (the reference's counterpart is the serial loop of
 * problems/problem.py:1256-1269; dnlp_amd.batch.ParametricBatch.solve_many used Python threads and one handle per worker
 * for this).  A launch lasts as long as its slowest instance; with `slots` launches in flight — each slot has its own HIP
 * stream, device buffers and host thread inside the library — the workgroups of the next batch take the compute units
 * the tail of the previous one leaves idle.  Results are bit for bit those of dnlp_solve_batch_theta on the same rows.
 *   create   after dnlp_batch_set_affine_map; slots >= 1 (2 is what pays); options are those of `p` at each submit
 *   submit   batch x n_params parameter rows and the output arrays of dnlp_solve_batch_theta (batch-major, mult_* may be
 *            NULL); goes to whichever slot is idle and returns a ticket >= 0 at once — unless every slot is busy: then it
 *            first waits for the FIRST of them to finish — or a negative error code.  The arrays must stay valid until the
 *            ticket has been waited for.  (Every slot has one worker thread for the stream's lifetime.)
 *   wait     blocks until that submission's outputs are filled; returns its dnlp_solve_batch_theta code; *kernel_seconds
 *            (may be NULL) = its launch's device time.  A result stays on record until it is waited for, however many
 *            submissions follow; it is handed out once: a second wait for the ticket returns -2, a ticket that was never
 *            handed out -1 (dnlp_last_error says which).
 *   destroy  waits for everything in flight. */
typedef struct dnlp_batch_stream dnlp_batch_stream;
dnlp_batch_stream* dnlp_batch_stream_create(dnlp_problem* p, int slots);
int dnlp_batch_stream_submit(dnlp_batch_stream* s, int batch, const double* theta, int n_params, double* x, double* obj,
                             double* mult_g, double* mult_x_L, double* mult_x_U, int* status, int* iters, int* factorizations);
int dnlp_batch_stream_wait(dnlp_batch_stream* s, int ticket, double* kernel_seconds);
void dnlp_batch_stream_destroy(dnlp_batch_stream* s);

/* The result rows of the last dnlp_solve_batch* launch of this handle, PACKED ON THE DEVICE: n_rows x width doubles,
 * row i = {i, objective, status, iterations, x*[0 .. N)} (width = 4 + N), valid until the handle's next launch.  It is
 * what a rank contributes to the path's one exchange (the reference's counterpart is the serial loop of
 * problems/problem.py:1256-1269 appending to one Python list) without a trip through host memory: dnlp_amd.batch wraps
 * the pointer as a device tensor and hands it to the RCCL all_gather.  Off by default (one more small kernel per launch):
 * dnlp_batch_keep_result_rows(p, 1) turns it on for the launches that follow.  An empty launch has n_rows = 0, rows NULL. */
int dnlp_batch_keep_result_rows(dnlp_problem* p, int on);
int dnlp_batch_result_rows(dnlp_problem* p, const double** rows, int64_t* n_rows, int64_t* width);

/* What the LAST dnlp_solve_batch* call of this handle launched (diagnostics; the reference has no counterpart —
 * its loop of problems/problem.py:1256-1269 is serial).  out[0] grid, [1] lanes per workgroup, [2] LDS mode
 * (wavefront solver: 2 x state in LDS + plan in LDS, + 4 when the kernel was compiled for this template at run time,
 * + 8 when that was the workgroup-per-instance kernel: [1] / 64 wavefronts per instance, [3] workgroups per unit),
 * [3] instances resident per compute unit, [4] 1 = packed generic kernel, [5] 1 = longest-first order,
 * [6] wavefront solver form (0: the generic kernel; else 100 x wavefronts per workgroup + 10 x state in LDS +
 * plan in LDS; csrc/wave_batch.h), [7] instances the wavefront solver handed to the generic kernel. */
int dnlp_batch_launch_info(dnlp_problem* p, int32_t* out8);
/* f and grad f of the USER's variables from the fused element program of an unconstrained
 * elementwise-sum objective (tape arrays fz_*, dnlp_amd/fused.py; BASELINE config C2): one kernel,
 * x read once, grad accumulated once — eval_f + eval_grad_f of nlp_solver.py:212-235 on the
 * problem as written, without the auxiliary variables of dnlp2smooth.  xfree / grad hold the
 * user's variables in canonical order.  -11 when the tape carries no such program. */
int dnlp_eval_fused(dnlp_problem* p, const double* xfree, double* f, double* grad);
/* Average seconds of one fused evaluation with x resident in HBM (HIP events, `reps` evaluations). */
int dnlp_time_fused(dnlp_problem* p, const double* xfree, int reps, double* seconds);
/* The fused objective is evaluated by a kernel GENERATED from the element program at lowering time
 * (hiprtc, gfx950; option `fused_codegen=no` keeps the interpreter).  This self check parses the fused
 * programs of a tape blob, generates the kernel text (returned in src_out) and compiles it; it needs no
 * GPU.  0 = compiled, 1 = no generated form for this objective (reason in log_out), 2 = compiler error
 * (text in log_out), -11 = the tape has no fused program. */
int dnlp_fused_codegen_check(const void* tape_blob, size_t len, int elems_per_lane, char* src_out,
                             size_t src_cap, char* log_out, size_t log_cap);
/* The same self check for the device-resident L-BFGS kernels generated around that element code
 * (dnlp_solve_reduced runs them when the objective has a generated form: every line-search and
 * convergence decision is taken on the device, the host only enqueues). */
int dnlp_lbfgs_codegen_check(const void* tape_blob, size_t len, int elems_per_lane, char* src_out,
                             size_t src_cap, char* log_out, size_t log_cap);
/* Which compiler the generated kernels of this process go through (dnlp_amd/csrc/fused_rtc.h): 0 = hiprtc in
 * process — whatever libhiprtc / libamd_comgr the loader bound first: the pair PyTorch ships when torch was
 * imported before this library, else the ROCm install's — 1 = the ROCm install's clang++ run as a child
 * process ($DNLP_RTC_COMPILER=clang: the same code objects whichever hiprtc is bound).  `out` receives the
 * compiler's file, size and time: the identity the kernel cache is keyed by, so that the two never share
 * entries.  No counterpart in the reference (it compiles nothing at run time). */
int dnlp_rtc_compiler(char* out, size_t cap);
/* Dual warm start (IPOPT `warm_start_init_point`; SURVEY.md 8f-4.  The reference accepts
 * `warm_start` and ignores it, ipopt_nlpif.py:126-127): with the option
 * `warm_start_init_point=yes`, the next dnlp_solve / dnlp_ipm_begin starts from x_inout AND these
 * multipliers (user units, as dnlp_solve returns them), pushed into the interior by
 * warm_start_bound_push / _bound_frac / _mult_bound_push (default 1e-3); no least-squares
 * multiplier estimate is computed.  NULL pointers clear the stored multipliers. */
int dnlp_set_warm_start(dnlp_problem* p, const double* mult_g, const double* mult_x_L, const double* mult_x_U);
/* Linear-solver plan of the handle (decided once, from the sparsity pattern of the tape):
 * out[0] = 1 static-pattern sparse LDL^T / 0 dense, out[1] = factor values, out[2] = pivot blocks,
 * out[3] = largest block struct, out[4] = static 2x2 pivot pairs, out[5] = update triples,
 * out[6] = elimination-tree levels, out[7] = 1 when the dense path is the pivoted (Bunch-Kaufman)
 * factorisation (order <= kkt_pivot_max_n; the device space accepts at most 4096 there) (8 values).
 * `dnlp_set_option(p, "linear_solver", "dense" | "sparse")` forces a path before the first solve. */
int dnlp_kkt_info(dnlp_problem* p, int64_t* out8);
/* linear solver of the handle as it stands (it can change during a solve): -1 not chosen yet (no solve so far), 0 sparse
 * static-pattern LDL^T, 1 dense Bunch-Kaufman, 2 dense unpivoted blocked LDL^T, 3 dense unpivoted on rotated static pairs,
 * 4 started as 3 and was handed to Bunch-Kaufman (element growth / zero pivot of the static sequence) */
int dnlp_kkt_mode(dnlp_problem* p);
/* nodes of the dense tail of the sparse plan (0: none, or a dense KKT path).  A plan with a tail belongs to the host-driven
 * loop: its update program has no entries for the tail, which the in-kernel solver would need (it gets a second, full plan) */
int64_t dnlp_kkt_tail_nodes(dnlp_problem* p);
/* Statistics (n <= 24 values).  Of the last solve: stats[0..12] = iterations, factorizations, wall,
 * t_eval, t_factor, t_solve, mu, inf_pr, inf_du, compl, nlp_error, last_delta_w, objective scaling;
 * [13..15] = seconds, flops, launches of the timed outer Schur-complement updates (option
 * time_kernels=yes).  Over the life of the handle (they survive dnlp_ipm_begin): [16] iterations and
 * [17] factorizations of all dnlp_ipm_begin / dnlp_ipm_step calls, [18] begin calls, [19] seconds
 * spent in them, [20] factorizations skipped by the certified regularisation bound (last solve),
 * [21] outer Schur updates of one complete blocked factorisation, [22] blocked factorisations
 * abandoned early, [23] reserved, 0 (a batch launch with hessian_approximation=limited-memory is refused with
 * -12 Invalid_Option: the in-kernel solver has no quasi-Newton mode and does not substitute the exact Hessian). */
int dnlp_get_stats(dnlp_problem* p, double* stats, int n);
/* Iteration log of the last solve (IPOPT-style table), NUL terminated; returns bytes needed. */
size_t dnlp_get_log(dnlp_problem* p, char* buf, size_t cap);

/* ---- device utilities used by the drop-in host side ------------------------------------ */
/* Raw HBM allocation for constants that live on the device only. */
int dnlp_dev_alloc(int device, size_t bytes, void** out_device_ptr);
int dnlp_dev_free(int device, void* device_ptr);
int dnlp_dev_copy(int device, void* dst, const void* src, size_t bytes, int kind); /* 0 h2d 1 d2h */
/* Fill a device n x n column-major matrix with the seeded dense symmetric test matrix used by
 * BASELINE config C4: A = noise(i,j) + spike * v v^T; also returns nothing on
 * the host.  v is written to `device_v` (n doubles) when not NULL. */
int dnlp_gen_symmetric(int device, double* device_A, int64_t n, int64_t ld, uint64_t seed,
                       double spike, double* device_v);
/* y = A x for a device-resident symmetric matrix (host x, y): power-iteration support. */
int dnlp_dev_symv(int device, const double* device_A, int64_t n, int64_t ld, const double* x,
                  double* y);

/* ---- host side of the lowering ----------------------------------------------------------------
 * The role of cvxcore's build_matrix (cvxpy/cvxcore/src/cvxcore.cpp:161-215) and of the reference's per-callback
 * COO bookkeeping (nlp_solvers/nlp_solver.py:246-276, 337-372): from the constraint rows G over [x; z] (CSR, any
 * order, duplicates allowed), the objective coefficients c and the derivative triplets (drow, dcol: d z / d x;
 * hrow, hcol: lower oriented second derivatives) that the front-end's DAG walk emits, build the canonical G, the
 * constant CSR maps Mg, Mw, MJ, MH, the affine Jacobian part Jc and the sorted unique Jacobian / Hessian patterns
 * (+ the positions of the listed dense quad_form blocks x0, n), in C++ with a few host threads.  No device is
 * touched.  Fetch with the accessors below (sizes first), then dnlp_lowered_free. */
/* Affine forms value = A [x; z] + b (A: CSR over the problem's columns) as opaque handles: the front-end's DAG walk
 * composes them with the operations affine atoms need (the role of cvxcore's lin_ops, cvxcore/src/LinOpOperations.cpp).
 * Every function returns a NEW form (NULL on error, text in dnlp_last_error); dnlp_lf_free releases one. */
typedef struct dnlp_linform dnlp_linform;
dnlp_linform* dnlp_lf_const(int64_t ncol, int64_t n, const double* b);                    /* no coefficients, constants b */
dnlp_linform* dnlp_lf_range(int64_t ncol, int64_t n, int64_t col0);                        /* row r = column col0 + r */
dnlp_linform* dnlp_lf_select(const dnlp_linform* a, const int64_t* sel, int64_t n);       /* rows sel[0..n) of a */
dnlp_linform* dnlp_lf_add(dnlp_linform* a, dnlp_linform* b);
dnlp_linform* dnlp_lf_scale(const dnlp_linform* a, const double* s);                       /* diag(s) a; s == NULL: -a */
dnlp_linform* dnlp_lf_apply_csr(const dnlp_linform* a, int64_t srows, const int64_t* s_ptr, const int32_t* s_idx,
                                const double* s_val);                                       /* S a */
dnlp_linform* dnlp_lf_apply_dense(const dnlp_linform* a, int64_t srows, const double* S); /* S a, S dense row-major */
dnlp_linform* dnlp_lf_vstack(dnlp_linform* const* parts, int n);
void dnlp_lf_free(dnlp_linform* a);
int dnlp_lf_info(const dnlp_linform* a, int64_t* info);   /* rows, columns, stored coefficients, is-a-plain-selection */
int dnlp_lf_gather(const dnlp_linform* a, int64_t n_cols, int64_t* out);                 /* plain selection of columns < n_cols? */
int dnlp_lf_export(dnlp_linform* a, int64_t* ptr, int32_t* idx, double* val, double* b);   /* canonical CSR + constants */
/* the same arrays without a copy: pointers into the handle, valid until dnlp_lf_free (forms that differ only in their
 * constants share the matrix arrays) */
int dnlp_lf_view(dnlp_linform* a, const int64_t** ptr, const int32_t** idx, const double** val, const double** b);

/* exact symmetry of a dense row-major n x n constant (threads): 1 / 0 */
int dnlp_is_symmetric(const double* P, int64_t n, int64_t ld);

typedef struct dnlp_lowered dnlp_lowered;
dnlp_lowered* dnlp_lower_maps(int64_t N, int64_t Z, int64_t m, int64_t nd, int64_t nh, const int64_t* G_ptr,
                              const int32_t* G_idx, const double* G_val, const double* c, const int64_t* drow,
                              const int64_t* dcol, const int64_t* hrow, const int64_t* hcol, int n_blocks,
                              const int64_t* block_x0, const int64_t* block_n);
void dnlp_lowered_free(dnlp_lowered* h);
/* sizes[0..8] = G_changed, nnz(G), nnz(Mg), nnz(Mw), nnz(MJ), nnzJ, nnz(MH), nnzH, jac_is_G (1: every row is affine
 * in x alone and G was canonical — the Jacobian pattern and Jc ARE the caller's G arrays, nothing is copied) */
int dnlp_lowered_sizes(const dnlp_lowered* h, int64_t* sizes);
/* which: 0 G (only when G_changed), 1 Mg, 2 Mw, 3 MJ, 4 MH */
int dnlp_lowered_csr(const dnlp_lowered* h, int which, int64_t* ptr, int32_t* idx, double* val);
/* the same map without a copy: pointers into the handle, valid until dnlp_lowered_free */
int dnlp_lowered_csr_view(const dnlp_lowered* h, int which, const int64_t** ptr, const int32_t** idx, const double** val);
/* which: 0 Jacobian (rows, cols, Jc), 1 Hessian (rows, cols) */
int dnlp_lowered_pattern(const dnlp_lowered* h, int which, int32_t* rows, int32_t* cols, double* vals);
/* dense block b: *mode = 2 (contiguous run, pos[0] = first position) or 1 (table of *count positions) */
int dnlp_lowered_block(const dnlp_lowered* h, int b, int* mode, int64_t* count, int64_t* pos);

/* ---- factorisation kernels exposed for parity tests and benchmarks ----------------------- */
/* In-place LDL^T of a host column-major symmetric matrix (lower triangle referenced) through
 * the device path: pivoted (Bunch-Kaufman) or blocked unpivoted with the FP64-MFMA trailing
 * update.  Outputs the factored matrix, pivots and the inertia.  Pivoted: the pivot choices and ipiv are
 * DSYTF2's, but every interchange is also applied to the columns already factored, so the matrix
 * holds ONE unit-lower factor L of P A P^T = L D L^T (P = the interchanges in order), not DSYTRS's
 * interleaved form. */
int dnlp_ldlt_host(int device, double* A, int64_t n, int64_t ld, int32_t* ipiv, int pivoted,
                   int* nneg, int* nzero, const double* rhs, double* sol, double* seconds);
/* Time the blocked FP64-MFMA LDL^T on a device-resident matrix (destroys it).  The allocation
 * must extend at least 1 KiB past the last matrix element (128-row tiles over-read). */
int dnlp_ldlt_device(int device, double* device_A, int64_t n, int64_t ld, int* nneg, int* nzero,
                     double* seconds, double* update_seconds);

#ifdef __cplusplus
}
#endif
#endif /* DNLP_HIP_H */
