"""bench.py — interior-point iterations/sec on BASELINE config C4 (the metric's workload):
maximise x'Ax on the unit sphere, dense symmetric A (n = 1e5 by default, 80 GB FP64 in HBM),
through the drop-in front-end (quad_form + sum_squares -> dnlp2smooth -> device tape) and the
on-device interior-point loop (tape f/grad/g/Jac/Hess kernels, dense KKT assembly, blocked
LDL^T with the FP64-MFMA Schur-complement update, solves with refinement, filter line search).

A "step" is one interior-point iteration.  N > 1 runs one replica per GPU (the KKT
factorisation does not shard without a distributed factorisation): weak
scaling, no data-path collective, one RCCL all_gather of the per-rank results at the end.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus 8 --steps 20 --warmup 5          # starts its 8 ranks itself (child processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--workload c5 --batch 8192` is BASELINE config C5 by the same command: the batch of parametrised
paper NLPs is sharded over the N ranks (8 x 1024 at N = 8), one launch per rank, ONE RCCL
all_gather of {id, objective, status, iterations, x*}; a step is one pass over the whole batch.

With --gpus N > 1 and no WORLD_SIZE in the environment this process touches neither torch nor HIP:
it starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD process and
exits with its code (never an exec of a process that initialised the GPU).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAFFIC_PROFILE = "r06_pmc_bench_traffic.json"   # rocprofv3 --pmc passes of this command (refreshed per round)
PEAK_FP64_MFMA_TFLOPS = 78.6   # MI355X datasheet FP64 matrix peak (dense); see DESIGN.md §5


def build_problem(n, seed, device):
    import dnlp_amd as cp
    from dnlp_amd.device import symmetric_test_matrix
    A = symmetric_test_matrix(n, seed=seed, spike_eig=4.0 * np.sqrt(n), device=device)
    x = cp.Variable(n)
    rng = np.random.default_rng(seed)
    x.value = np.ones(n) / np.sqrt(n) + 0.1 * rng.standard_normal(n) / np.sqrt(n)
    prob = cp.Problem(cp.Maximize(cp.quad_form(x, cp.Constant(A.handle))), [cp.sum_squares(x) == 1])
    return prob, A


def lower(prob):
    """The reduction chain of Problem.solve(nlp=True) up to the solver call."""
    chain = prob._build_chain(None)
    data, inv = chain.apply(prob)
    return chain, data, inv


def run_steps(handle, x0, n_steps):
    """Exactly n_steps interior-point iterations that were carried out (dnlp_ipm_step counts an
    iteration only when the iterate advanced; the call that merely detects convergence adds
    nothing).  When the solve converges before the count is reached it restarts from a perturbed
    point; the restart's dnlp_ipm_begin (one tape sweep + multiplier start, no O(n^3) work) stays
    inside the timed region and is reported separately (stats[19])."""
    done = 0
    restarts = 0
    while done < n_steps:
        rc, k = handle.ipm_step(n_steps - done)
        done += k
        if rc != 99 and done < n_steps:
            restarts += 1
            rng = np.random.default_rng(1000 + restarts)
            handle.ipm_begin(x0 + 0.3 * rng.standard_normal(x0.size) / np.sqrt(x0.size))
    return restarts


def full_solve(chain, data, A, n):
    """One complete solve through the solver entry point of the front-end (HIPNLP.solve_via_data, the role of
    IPOPT.solve_via_data, ipopt_nlpif.py:104-174) on the bench's own problem, and its answer against the
    closed form: the optimum of max x'Ax on the unit sphere is lambda_max(A)."""
    t0 = time.time()
    sol = chain.solver.solve_via_data(data, True, False, dict(kkt_pivot_max_n=0))
    wall = time.time() - t0
    rng = np.random.default_rng(7)
    v = rng.standard_normal(n)
    v /= np.linalg.norm(v)
    lam = 0.0
    for _ in range(300):
        w = A.symv(v)
        lam_new = float(v @ w)
        v = w / np.linalg.norm(w)
        done = abs(lam_new - lam) <= 1e-13 * abs(lam_new)
        lam = lam_new
        if done:
            break
    value = -float(sol["obj_val"])                 # the chain flips Maximize into Minimize(-.)
    st = sol["stats"]
    return {"wall_s": wall, "iterations": int(sol["iterations"]), "status": int(sol["status"]),
            "factorizations": int(st[1]), "factor_s": float(st[4]), "value": value, "lambda_max_power_iteration": lam,
            "rel_err_vs_power_iteration": abs(value - lam) / abs(lam),
            "iters_per_s_whole_solve": int(sol["iterations"]) / wall if wall > 0 else None}


CPU_SWEEP = ((1000, 3), (2000, 3), (4000, 2), (10000, 1), (20000, 1))   # (order n, timed iterations) — BASELINE.md §3; the last order: blocked column only
CPU_DSYTRF_MAX_N = 10000      # (pivoted Bunch-Kaufman takes ~6 factorisations per iteration without the Lanczos bound: 60 s at 2e4)


def usable_cores():
    """(cores this process tree can use, logical CPUs it sees): the affinity mask capped by the cgroup's CPU quota."""
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    usable = visible
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            usable = max(1, min(visible, int(float(q) / float(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return usable, visible


def cpu_baseline(seed, device, sweep=CPU_SWEEP):
    """IPOPT-class CPU baseline on the box's host cores, same run: the host build of the same
    interior-point algorithm (oracle/, test infrastructure) on the same generator and front-end at the
    orders of BASELINE.md §3, with its dense KKT factorisation done two ways:
      * `dsytrf`:  LAPACK DSYTRF / DSYTRS (blocked Bunch-Kaufman of the OpenBLAS inside the image's scipy wheel —
        the dense counterpart of the MA27 / MUMPS factorisation IPOPT calls) at its fastest thread count;
      * `blocked`: the device's algorithm on the host's BLAS — unpivoted right-looking blocked LDL^T whose trailing
        update is DGEMM on ALL cores (DTRSM for the panel rows), with the DGEMM rate of the box beside it.
    `value` is the better of the two at the largest order.  n = 1e5 itself is out of reach of any host
    factorisation in bench time (n^3/3 = 3.3e14 flop per attempt), so the sweep and its fitted scaling law are
    reported, never an extrapolated figure as if measured."""
    import ctypes.util
    # the cores this process may actually use: the GPU boxes show 256 logical CPUs and grant 16 CPUs' worth of time
    # (cgroup cpu.max) — rounds 1-4 sized the BLAS pools from os.cpu_count(), and 64-256 threads on 16 CPUs is what made
    # this sweep erratic (DGEMM 1.1-5.1 TF/s "by thread count", n = 4000 at 26.7 GF/s, a fitted exponent of 0.96)
    threads, visible = usable_cores()
    os.environ.setdefault("OMP_NUM_THREADS", str(min(32, threads)))
    os.environ.setdefault("DNLP_HOST_LDLT_TIMING", "1")
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import (OracleProblem, blocked_ldlt_phases, dgemm_gflops, lapack_best_threads, set_blas_threads,
                                    use_blocked_ldlt, use_lapack)
    import dnlp_amd as cp
    from dnlp_amd.device import symmetric_test_matrix
    from dnlp_amd.dnlp2smooth import Dnlp2Smooth
    from dnlp_amd.nlp_solver import HIPNLP, build_nlp_data
    blas_threads = use_lapack(0)
    probe, gemm_rate, gemm_threads = None, None, None
    if blas_threads:
        # OpenBLAS's DSYTRF does not scale to every core of a large host (64 threads measured 3x slower
        # than 16 on the GPU box): take the fastest of a few thread counts, and say which
        cand = sorted({t for t in (4, 8, 16, 32, 64, threads) if t <= threads})
        blas_threads, probe = lapack_best_threads(cand)
        # DGEMM does scale: the blocked factorisation runs on every core (or on the count where DGEMM peaks)
        rates = {t: dgemm_gflops(3000, t) for t in sorted({max(1, threads // 2), threads})}
        gemm_threads = max(rates, key=rates.get)
        gemm_rate = {"gflops_by_threads": rates, "n": 3000}
    have_blocked = bool(blas_threads) and use_blocked_ldlt(True)
    blocked_probe = None
    kind = "LAPACK dsytrf/dsytrs (scipy OpenBLAS, %d threads)" % blas_threads if blas_threads else \
        "restated DSYTF2 (no LAPACK found)"
    table = []
    for n_cpu, steps in sweep:
        A = symmetric_test_matrix(n_cpu, seed=seed, spike_eig=4.0 * np.sqrt(n_cpu), device=device)
        Ah = A.to_host()
        A.free()
        x = cp.Variable(n_cpu)
        rng = np.random.default_rng(seed)
        x0 = np.ones(n_cpu) / np.sqrt(n_cpu) + 0.1 * rng.standard_normal(n_cpu) / np.sqrt(n_cpu)
        x.value = x0
        prob = cp.Problem(cp.Minimize(-cp.quad_form(x, Ah)), [cp.sum_squares(x) == 1])
        smooth, _ = Dnlp2Smooth().apply(prob)
        data, _ = build_nlp_data(smooth)
        blob = serialize(data["tape_arrays"])
        row = {"n": n_cpu}
        if not have_blocked and n_cpu > CPU_DSYTRF_MAX_N:
            continue
        for column in (("dsytrf", "blocked") if have_blocked else ("dsytrf",)):
            if column == "dsytrf" and n_cpu > CPU_DSYTRF_MAX_N:
                continue
            orc = OracleProblem(blob)
            for k, v in HIPNLP.DEFAULT_OPTIONS.items():
                orc.set_option(k, v)
            # the certified Lanczos lower bound on delta_w that the device's solve of the n = 1e5 problem uses (ipm_core.h:
            # on from order 12 000) at every order of the sweep: a timed iteration is then ~one factorisation on both sides
            # (round 4: six doomed attempts per iteration at n = 1e4 on the host, one on the device)
            orc.set_option("lanczos_min_n", 500)
            if column == "dsytrf":
                if blas_threads:
                    set_blas_threads(blas_threads)
                orc.set_option("kkt_pivot_max_n", 10 ** 9)     # pivoted (Bunch-Kaufman) at every order, as IPOPT's solvers are
            else:
                orc.set_option("kkt_pivot_max_n", 0)           # unpivoted blocked LDL^T at every order, as on the device
                if blocked_probe is None and n_cpu == CPU_DSYTRF_MAX_N:
                    # the thread count that FACTORS fastest, measured at the largest order of the sweep (the DGEMM probe
                    # alone misleads: its best count is not the best for the mix of DTRSM, small and large DGEMMs); the
                    # smaller orders above ran with the DGEMM probe's count
                    blocked_probe = {}
                    for tcand in sorted({t for t in (8, 16, 32, 64) if t <= threads} or {threads}):
                        set_blas_threads(tcand)
                        orc.ipm_begin(data["x0"])
                        orc.ipm_step(1)
                        blocked_probe[tcand] = float(orc.stats()[4]) / max(int(orc.stats()[1]), 1)
                    gemm_threads = min(blocked_probe, key=blocked_probe.get)
                set_blas_threads(gemm_threads)
            orc.ipm_begin(data["x0"])
            t0 = time.time()
            rc, k = orc.ipm_step(steps)
            dt = time.time() - t0
            st = orc.stats()
            nf = max(int(st[1]), 1)
            if column == "blocked":
                row["blocked_phase_seconds_last_factorization"] = blocked_ldlt_phases()
            row[column] = {"iterations": k, "factorizations": int(st[1]), "seconds": dt,
                           "iters_per_s": k / dt if dt > 0 else None, "s_per_factorization": float(st[4]) / nf,
                           "factorization_gflops": (n_cpu + 1) ** 3 / 3.0 / (float(st[4]) / nf) / 1e9 if st[4] > 0 else None}
            orc.close()
        table.append(row)
        del Ah, data, smooth, prob, blob
    best_col = "dsytrf"
    both = [r for r in table if "dsytrf" in r and "blocked" in r]
    if have_blocked and both and (both[-1]["blocked"]["iters_per_s"] or 0) > (both[-1]["dsytrf"]["iters_per_s"] or 0):
        best_col = "blocked"
    rows_best = [r for r in table if best_col in r]
    # scaling law of the factorisation from the TWO LARGEST orders of the reported column: the small orders of the sweep
    # are thread-start and panel overhead (n = 1000 factors at 3-10 GF/s), and a fit over all of them measures that
    ln = np.log([r["n"] for r in rows_best[-2:]])
    lt = np.log([max(r[best_col]["s_per_factorization"], 1e-12) for r in rows_best[-2:]])
    expo = float((lt[1] - lt[0]) / (ln[1] - ln[0])) if len(rows_best) >= 2 else None
    last = rows_best[-1]
    cores = (gemm_threads if best_col == "blocked" else blas_threads) or threads
    return {"value": last[best_col]["iters_per_s"], "unit": "iters/s", "cores": cores, "kind": "port",
            "n": last["n"], "factorization": best_col, "sweep": table, "factorization_time_exponent": expo,
            "host_cores": threads, "host_logical_cpus_visible": visible, "dsytrf_n3000_seconds_by_threads": probe, "dgemm": gemm_rate,
            "blocked_s_per_factorization_by_threads": blocked_probe, "blocked_threads": gemm_threads,
            "sample": "host build of the same interior-point algorithm on the same generator / front-end at n in %s with %s "
                      "timed iterations, dense KKT two ways: %s; and the unpivoted blocked LDL^T with its trailing update "
                      "through DGEMM on %s threads (the device's algorithm on the host's BLAS).  value = iters/s of the "
                      "faster one (%s) at n=%d; its seconds per factorisation scale as n^%.2f between the two largest orders (n^3 "
                      "asymptotically; the blocked column runs with the certified Lanczos bound on delta_w as the device does — one "
                      "factorisation per iteration on both sides; n=1e5 is %.3g x the n=%d flops per factorisation); libipopt on this box: %s"
                      % ([r["n"] for r in table], [r.get("dsytrf", r.get("blocked"))["iterations"] for r in table], kind, gemm_threads, best_col,
                         last["n"], expo if expo is not None else float("nan"), (1e5 / last["n"]) ** 3, last["n"],
                         ctypes.util.find_library("ipopt") or "not found")}


C5_TRAFFIC_PROFILE = "r06_pmc_wave_8192.json"  # rocprofv3 --pmc passes of `bench.py --workload c5` (per round)
PEAK_HBM_GBS = 8000.0                           # MI355X HBM3E peak (MI355X_MICROARCH.md)


def c5_template(which):
    sys.path.insert(0, os.path.join(ROOT, "tests"))     # problem definitions (paper notebooks), not the oracle
    import batch_problems as bp
    return {"localization": bp.template_localization, "circle_packing": bp.template_circle_packing,
            "circle_packing10": lambda: bp.template_circle_packing(10),
            "path_planning": bp.template_path_planning, "power_flow": bp.template_power_flow}[which]()


def c5_cpu_baseline(which, batch, first, budget_s=12.0):
    """The host build of the same interior-point algorithm on ALL host cores ("across all host cores"): one
    worker process per core, ONE shared tape + the template's static-pattern plan as on the device, whole passes over the
    same batch for ~budget_s seconds (tools/c5_cpu_allcores.py; a process tree of its own that never touches the GPU)."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "tools", "c5_cpu_allcores.py"), "--which", which, "--batch", str(batch),
           "--first", str(first), "--budget", str(budget_s)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=budget_s * 8 + 240)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    if out.returncode != 0 or not lines:
        raise RuntimeError("c5_cpu_allcores.py failed: %s" % (out.stderr.strip().splitlines() or ["no output"])[-1])
    return json.loads(lines[-1])


def bench_c5(args):
    """BASELINE config C5: `--batch` parametrised paper NLPs, contiguous shards of ceil(B/N) instances
    per rank, one batch_solve launch per rank and step, ONE all_gather of the result rows per step
    (inside the timed region).  A step = one pass over the whole batch, from parameter rows in host
    memory to gathered results on every rank."""
    rank, world, local, dist, tdev, torch = init_ranks(args)
    from dnlp_amd import _capi
    _capi.require_device(local)
    from dnlp_amd.batch import ParametricBatch, shard_bounds
    prob, params, sample, _ = c5_template(args.which)
    pb = ParametricBatch(prob, params)
    B = args.batch
    lo, hi = shard_bounds(B, rank, world)
    # Every step solves a FRESH batch (instance ids k B .. (k+1) B - 1 of the same generator), generated before the
    # clock starts: the real use case.  A re-solve of the SAME rows takes its instances longest-first from the
    # previous solve's iteration counts (csrc/batch.h prev_iters, keyed on the rows); that figure is reported
    # beside the headline as `resolve_same_batch_problems_per_s`, never as `value`.
    n_batches = args.warmup + args.steps
    all_thetas = [np.stack([sample(k * B + i) for i in range(B)]) for k in range(n_batches)]
    thetas = all_thetas[-1]

    def barrier():
        if dist is not None:
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def step(k):
        rows, info = pb.solve_sharded(all_thetas[k], device=local)
        return rows, info

    for k in range(args.warmup):
        step(k)
    barrier()
    t0 = time.time()
    ksec, gbytes, iters_all, optimal_all = 0.0, 0, 0.0, 0
    for k in range(args.warmup, n_batches):
        rows, info = step(k)
        ksec += info["kernel_seconds"]
        gbytes += info["gathered_bytes"]
        iters_all += float(rows[:, 3].sum())
        optimal_all += int(np.sum(rows[:, 2] == 0))
    barrier()
    dt = time.time() - t0
    # secondary figure: the last batch solved again (twice: the first re-solve already has the order)
    barrier()
    t1 = time.time()
    for _ in range(2):
        pb.solve_sharded(thetas, device=local)
    barrier()
    dt_resolve = (time.time() - t1) / 2.0
    # secondary figure: TWO batches in flight (ParametricBatch.solve_many) — the same fresh batches (first come)
    # through two handles on two streams: the next launch's instances take the compute units the tail of the previous
    # one leaves idle (a launch is as long as its slowest instance; evenly spread the work is ~60 % of it).  What a
    # caller with a stream of batches gets; one batch at a time stays the headline.
    dt_two, n_two = None, 0
    if world == 1 and args.steps >= 2:
        ks = list(range(args.warmup, n_batches))
        pb.solve_many([all_thetas[0], all_thetas[0]], device=local, in_flight=2)      # (the second handle's first call)
        barrier()
        t2 = time.time()
        pb.solve_many([all_thetas[k] for k in ks], device=local, in_flight=2)
        barrier()
        dt_two, n_two = time.time() - t2, len(ks)
    # secondary figure: the per-GPU shard BASELINE's config names (8 x 1024) — 1024 fresh instances per launch, all
    # resident at once: the launch is its slowest instance
    dt_1024, it_1024, launch_info, dt_1024_stream = None, None, None, None
    if world == 1 and B > 1024:
        small = [np.stack([sample((n_batches + k) * B + i) for i in range(1024)]) for k in range(4)]
        pb.solve(small[0], device=local)
        barrier()
        t3 = time.time()
        for k in range(1, 4):
            r1024 = pb.solve(small[k], device=local)
        barrier()
        dt_1024, it_1024 = (time.time() - t3) / 3.0, int(r1024.iterations.max())
        # ... and a STREAM of such shards with four launches in flight (dnlp_batch_stream_*): what a rank that is handed
        # one 1024-instance shard after another gets
        more = [np.stack([sample((n_batches + 4 + k) * B + i) for i in range(1024)]) for k in range(12)]
        pb.solve_many(more[:4], device=local, in_flight=4)
        barrier()
        t4 = time.time()
        pb.solve_many(more, device=local, in_flight=4)
        barrier()
        dt_1024_stream = (time.time() - t4) / len(more)
    launch_info = info.get("launch")                 # the HEADLINE launch's form (the last timed step of this rank)
    launch_1024 = pb.solve(thetas[: min(B, 1024)], device=local).raw.get("launch") if world == 1 else None
    assert rows.shape[0] == B and np.array_equal(rows[:, 0], np.arange(B)), "gathered rows are not the whole batch"
    gathered_ranks, backend_name = info["ranks"], info["backend"]
    if dist is not None:
        tm = torch.tensor([dt, ksec], device=tdev, dtype=torch.float64)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dt_all, ksec_all = float(tm[0].item()), float(tm[1].item())
        assert gathered_ranks == args.gpus, "the gather saw %d ranks, --gpus %d" % (gathered_ranks, args.gpus)
        if args.backend == "nccl":
            assert backend_name == "nccl", backend_name
            backend_name = "nccl (RCCL)"
    else:
        dt_all, ksec_all = dt, ksec
        r1, b1, backend_name = single_rank_collective(torch, local, rows[0, :4].tolist())
    if rank == 0:
        N, m = int(pb.arrays0["dims"][0]), int(pb.arrays0["dims"][1])
        h = pb._handle
        nnzj, nnzh = int(h.nnz_jac), int(h.nnz_hess)
        iters_total = iters_all / max(args.steps, 1)
        # per-iteration oracle bytes f 8N + grad 16N + g (8N+8m) + J (8N+8nnzJ) + H (8N+8m+8nnzH)
        bytes_iter = 8 * N + 16 * N + (8 * N + 8 * m) + (8 * N + 8 * nnzj) + (8 * N + 8 * m + 8 * nnzh)
        per_launch_s = ksec_all / args.steps
        shard = hi - lo
        alg_bytes_launch = bytes_iter * iters_total * (shard / float(B))
        traffic, traffic_src, issue = None, None, None
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", C5_TRAFFIC_PROFILE)))
            if pj.get("which") == args.which and int(pj.get("batch", 0)) == shard:
                traffic = pj["traffic_bytes_per_launch"]
                traffic_src = "profiles/" + C5_TRAFFIC_PROFILE
                # the bound that binds: how much of a wavefront's time issues instructions, and how many it issues per
                # interior-point iteration (counter passes of this command; per launch / this run's iterations per launch)
                ins = pj.get("instructions_per_launch") or {}
                per_it = {k: v / max(iters_total, 1.0) for k, v in ins.items() if v is not None}
                issue = {"wave_cycles_issuing": pj.get("wave_cycles_issuing"), "wave_cycles_waiting": pj.get("wave_cycles_waiting"),
                         "wave_cycles_waiting_for_instructions": pj.get("wave_cycles_issue_stalled"),
                         "wavefronts_per_simd": (launch_info or {}).get("per_cu", 0) / 4.0 if launch_info else None,
                         "instructions_per_iteration": per_it,
                         "instructions_per_iteration_total": sum(per_it.values()) if per_it else None,
                         "source": "profiles/" + C5_TRAFFIC_PROFILE}
        except OSError:
            pass
        except (KeyError, ValueError) as e:
            sys.stderr.write("bench: profiles/%s exists but cannot be read as a C5 counter profile (%r)\n" % (C5_TRAFFIC_PROFILE, e))
        out = {
            "metric": "problems/sec, batch of %d parametrised paper NLPs (%s), sharded over the GPUs" % (B, args.which),
            "value": B * args.steps / dt_all, "unit": "problems/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt_all / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C5: %d x %d instances of %s (N=%d, m=%d, nnzJ=%d, nnzH=%d), parameters drawn per "
                                   "instance from default_rng(instance id), a fresh batch every step; one batch_solve launch per rank, one "
                                   "all_gather of {id, obj, status, iterations, x*} per step"
                                   % (world, -(-B // world), args.which, N, m, nnzj, nnzh),
                       "batch_total": B, "shard": [lo, hi],
                       "optimal": optimal_all // max(args.steps, 1), "acceptable": int(np.sum(rows[:, 2] == 1)),
                       "instance_order": "first come (every step is a fresh batch)",
                       "resolve_same_batch_problems_per_s": B / dt_resolve if dt_resolve > 0 else None,
                       "two_batches_in_flight_problems_per_s": B * n_two / dt_two if dt_two else None,
                       "shard_of_1024_problems_per_s": 1024 / dt_1024 if dt_1024 else None,
                       "shards_of_1024_four_in_flight_problems_per_s": 1024 / dt_1024_stream if dt_1024_stream else None,
                       "shard_of_1024_slowest_instance_iterations": it_1024,
                       "kernel_form": launch_info, "kernel_form_shard_of_1024": launch_1024,
                       "ip_iterations_per_pass": iters_total,
                       "aggregate_ip_iterations_per_s": iters_total * args.steps / dt_all,
                       "problems_per_s_kernel_only": B * args.steps / ksec_all if ksec_all > 0 else None,
                       "gathered_ranks": gathered_ranks, "gathered_bytes": gbytes // max(args.steps, 1),
                       "collective_backend": backend_name},
            "roofline": {"bound": "hbm", "kernel": (("dnlp_wave_wg_kernel (per-template, a workgroup per instance)" if ((launch_info or {}).get("wave_wg") or (launch_info or {}).get("lds_mode", 3) == 0) else
                                                      "dnlp_wave_spec_kernel (per-template: generated LDL^T / residual / CSR phases)")
                                                     if (launch_info or {}).get("wave_spec") else
                                                     "wave_batch_kernel" if (launch_info or {}).get("wave_form") else "batch_solve_kernel") +
                         " (whole interior-point loop per wavefront)",
                         "achieved": alg_bytes_launch / per_launch_s / 1e9 if per_launch_s > 0 else None,
                         "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": alg_bytes_launch / per_launch_s / 1e9 / PEAK_HBM_GBS if per_launch_s > 0 else None,
                         "algorithmic_bytes_per_iteration": bytes_iter, "avg_launch_ms": 1e3 * per_launch_s,
                         "traffic": traffic, "traffic_source": traffic_src, "issue": issue,
                         "note": "one wavefront per instance runs the interior-point loop out of LDS: bound by the instructions it issues and "
                                 "its LDS round trips (`issue`), neither the HBM nor the MFMA roof is near (DESIGN.md 4d / 4e)"},
        }
        if world > 1 or args.no_cpu:
            out["cpu_baseline"] = {"value": None, "unit": "problems/s", "cores": 0, "kind": "port",
                                   "sample": "not timed (N > 1 or --no-cpu; see the N = 1 line)"}
        else:
            try:
                out["cpu_baseline"] = c5_cpu_baseline(args.which, B, (n_batches - 1) * B)
            except Exception as e:
                out["cpu_baseline"] = {"value": None, "unit": "problems/s", "cores": 1, "kind": "port",
                                       "sample": "failed: %s" % e}
        emit(out)
    pb.close()
    if dist is not None:
        dist.destroy_process_group()


def emit(line):
    """The ONE JSON line, last on stdout: whatever native libraries left in the C stdio buffers (RCCL
    prints a version banner through printf, which a pipe holds back until exit) goes out first."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.write(json.dumps(line) + "\n")
    sys.stdout.flush()


def flush_native_stdio():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def self_launch(n_ranks):
    """--gpus N > 1 without a launcher: start the N ranks as children of this process, which has not
    imported torch or touched HIP (the reference's counterpart is the serial multistart loop,
    problems/problem.py:1256-1269).  Rank 0's JSON line reaches stdout through the inherited stream;
    the exit code is the launcher's (non-zero when any rank fails)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n_ranks)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def init_ranks(args):
    """(rank, world, local device, dist module or None, tensor device, torch).  One process per GPU;
    a launch with more ranks than visible GPUs fails loudly instead of sharing devices."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench: --gpus %d but the launcher set WORLD_SIZE=%d" % (args.gpus, world))
    import torch
    dist, tdev = None, "cpu"
    if world > 1:
        import datetime
        import torch.distributed as dist
        if args.backend == "nccl":
            ndev = torch.cuda.device_count()
            if ndev < world:
                raise SystemExit("bench: %d ranks but %d GPUs visible (one process per GPU)" % (world, ndev))
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local),
                                    timeout=datetime.timedelta(seconds=600))
            tdev = "cuda"
        else:
            dist.init_process_group(args.backend)
            local = local % max(torch.cuda.device_count(), 1)     # gloo debugging runs may share a GPU
    os.environ["DNLP_DEVICE"] = str(local)
    if dist is not None:
        dist.barrier()                 # communicator creation (and RCCL's printf banner) happen here, on every rank
        flush_native_stdio()           # ... so no rank holds native stdout text back until after rank 0's JSON line
    return rank, world, local, dist, tdev, torch


def single_rank_collective(torch, local, payload):
    """N = 1: the path's one exchange still runs once over RCCL (a one-rank group: all_reduce +
    all_gather on device tensors), after the timed region — evidence that the collective the N > 1
    runs depend on initialises and executes on this box.  Reported, never required."""
    import datetime
    import socket
    try:
        import torch.distributed as dist
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=120))
        t = torch.tensor(payload, dtype=torch.float64, device="cuda")
        mx = t.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        parts = [torch.zeros_like(t)]
        dist.all_gather(parts, t)
        torch.cuda.synchronize()
        ok = bool(torch.equal(parts[0], t) and torch.equal(mx, t))
        name = dist.get_backend()
        n = dist.get_world_size()
        dist.destroy_process_group()
        if not ok:
            return 1, 0, "nccl (RCCL): one-rank all_gather returned other data"
        return n, int(t.numel() * t.element_size()), "%s (RCCL)" % name if name == "nccl" else name
    except Exception as e:            # pragma: no cover - depends on the box
        return 1, 0, "unavailable: %s" % str(e)[:200]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 3; 5 for --workload c5)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 1; 2 for --workload c5)")
    ap.add_argument("--workload", default="c4", choices=("c4", "c5"),
                    help="c4: dense n=1e5 NLP, iterations/s (the metric's config); c5: sharded batch of paper NLPs")
    ap.add_argument("--order", type=int, default=100000, help="order n of the dense NLP (BASELINE: 1e5)")
    ap.add_argument("--batch", type=int, default=8192, help="c5: instances in the whole job (BASELINE: 8 x 1024)")
    ap.add_argument("--which", default="localization",
                    help="c5 member: localization | circle_packing10 | power_flow | path_planning | circle_packing")
    ap.add_argument("--cpu-sweep", default="", help="CPU baseline orders as n:iters,... (default: 1000:3,2000:3,4000:2,10000:1)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-full-solve", action="store_true", help="c4: skip the complete solve after the timed region")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench: --gpus must be >= 1")
    # the launch decision comes before torch / HIP are touched: children, never a re-exec
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))
    if args.workload == "c5":
        args.steps = 5 if args.steps is None else args.steps
        args.warmup = 2 if args.warmup is None else args.warmup
        return bench_c5(args)
    args.steps = 3 if args.steps is None else args.steps
    args.warmup = 1 if args.warmup is None else args.warmup

    rank, world, local, dist, tdev, torch = init_ranks(args)

    from dnlp_amd import _capi
    _capi.require_device(local)
    n = args.order
    prob, A = build_problem(n, seed=rank, device=local)
    chain, data, inv = lower(prob)
    h = data["handle"]
    for k, v in chain.solver.DEFAULT_OPTIONS.items():
        h.set_option(k, v)
    h.set_option("kkt_pivot_max_n", 0)
    h.set_option("time_kernels", "yes")
    x0 = data["x0"]
    h.ipm_begin(x0)
    run_steps(h, x0, args.warmup)
    st0 = h.stats()

    def barrier():
        if dist is not None:
            dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    barrier()
    t0 = time.time()
    restarts = run_steps(h, x0, args.steps)
    barrier()
    dt = time.time() - t0
    info = h.ipm_finish()
    st1 = h.stats()
    gathered_ranks, gathered_bytes, backend_name = 1, 0, None
    if dist is not None:
        tmax = torch.tensor([dt], device=tdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt_all = float(tmax.item())
        # the one exchange of the path: gather per-rank results (objective, iterations, status)
        mine = torch.tensor([info["obj_val"], float(info["iterations"]), float(info["status"])],
                            device=tdev, dtype=torch.float64)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        # evidence for the scaling run: how many ranks the collective saw and what it moved
        gathered_ranks = dist.get_world_size()
        gathered_bytes = int(sum(g.numel() * g.element_size() for g in gathered))
        backend_name = "%s (RCCL)" % dist.get_backend() if dist.get_backend() == "nccl" else dist.get_backend()
        assert gathered_ranks == args.gpus, "the gather saw %d ranks, --gpus %d" % (gathered_ranks, args.gpus)
        if args.backend == "nccl":
            assert backend_name == "nccl (RCCL)", backend_name
    else:
        dt_all = dt
        gathered_ranks, gathered_bytes, backend_name = single_rank_collective(
            torch, local, [info["obj_val"], float(info["iterations"]), float(info["status"])])
    if rank == 0:
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes of this same
        # command (FETCH_SIZE and WRITE_SIZE cannot share a pass); the committed summary is quoted
        # only when it was taken at the same order n
        traffic, traffic_src = None, None
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_PROFILE)))
            if int(pj["n"]) == n:
                traffic = pj["avg_traffic_bytes_per_launch"]
                traffic_src = "profiles/" + TRAFFIC_PROFILE + " (" + pj["command"] + ")"
        except OSError:
            pass
        except (KeyError, ValueError) as e:
            # (round 5 shipped a file in another schema and the line silently lost its traffic figure)
            sys.stderr.write("bench: profiles/%s exists but cannot be read as a traffic profile (%r): roofline.traffic is null\n"
                             % (TRAFFIC_PROFILE, e))
        upd_s, upd_f, upd_l = st1[13] - st0[13], st1[14] - st0[14], st1[15] - st0[15]
        achieved = upd_f / upd_s / 1e12 if upd_s > 0 else None
        # counters that survive restarts (stats[16..19]); every one is a delta over the timed region
        iters_done = int(st1[16] - st0[16])
        factorizations = int(st1[17] - st0[17])
        begin_calls, begin_s = int(st1[18] - st0[18]), float(st1[19] - st0[19])
        per_fact, bailed = int(st1[21]), int(st1[22] - st0[22])
        # every complete blocked factorisation of this order issues the same number of outer Schur
        # updates; restarts add begin() factorisations, early-abandoned attempts launch fewer
        consistent = bailed > 0 or int(upd_l) == per_fact * factorizations
        if iters_done != args.steps or not consistent:
            sys.stderr.write("bench: counter mismatch: iterations %d (asked %d), update launches %d, "
                             "factorizations %d x %d outer updates, abandoned %d\n"
                             % (iters_done, args.steps, int(upd_l), factorizations, per_fact, bailed))
        assert iters_done == args.steps, "timed region must hold exactly --steps carried-out iterations"
        assert consistent, "Schur-update launches do not match the factorisation count"
        out = {
            "metric": "interior-point iters/sec, n=%d dense NLP (sphere quad_form max), 1 replica per GPU" % n,
            "value": world * iters_done / dt_all,
            "unit": "iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_all / iters_done,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C4: max x'Ax s.t. ||x||^2=1, dense symmetric A generated in HBM, "
                                   "n=%d (KKT order %d), full Jac_g / Hess_L callbacks" % (n, n + 1),
                       "kkt": "blocked unpivoted LDL^T, FP64-MFMA Schur update", "replicas": world,
                       "iterations_in_timed_region": iters_done,
                       "factorizations_in_timed_region": factorizations,
                       "outer_updates_per_factorization": per_fact,
                       "restarts_in_timed_region": restarts,
                       "restart_begin_seconds_in_timed_region": begin_s if begin_calls else 0.0,
                       "gathered_ranks": gathered_ranks, "gathered_bytes": gathered_bytes,
                       "collective_backend": backend_name},
            "roofline": {"bound": "mfma", "kernel": "gemm_nt_update_fast (Schur-complement update)",
                         "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": (achieved / PEAK_FP64_MFMA_TFLOPS) if achieved else None,
                         "launches": int(upd_l), "avg_launch_ms": 1e3 * upd_s / upd_l if upd_l else None,
                         "avg_launch_gflop": upd_f / upd_l / 1e9 if upd_l else None,
                         "whole_iteration_tflops": (n + 1.0) ** 3 / 3.0 * factorizations / dt / 1e12,
                         "traffic": traffic, "traffic_unit": "bytes per launch (FETCH x2-corrected + WRITE)",
                         "traffic_source": traffic_src},
        }
        if not args.no_full_solve:
            # the second half of BASELINE's metric ("solve wall-clock"): ONE complete solve(nlp=True) of the same
            # problem from the same start, after the timed region, checked against lambda_max(A) by power iteration
            # with the device's symmetric product (reference entry point: ipopt_nlpif.py:170, nlp.solve(x0))
            try:
                out["config"]["full_solve"] = full_solve(chain, data, A, n)
            except Exception as e:
                out["config"]["full_solve"] = {"wall_s": None, "error": str(e)[:300]}
        if world > 1:
            # the CPU baseline is a property of the host, not of N: it is timed in the N = 1 run only
            out["cpu_baseline"] = {"value": None, "unit": "iters/s", "cores": 0, "kind": "port",
                                   "sample": "not timed at N > 1 (see the N = 1 line)"}
        elif not args.no_cpu:
            try:
                sweep = tuple(tuple(int(v) for v in kv.split(":")) for kv in args.cpu_sweep.split(",")) \
                    if args.cpu_sweep else CPU_SWEEP
                out["cpu_baseline"] = cpu_baseline(0, local, sweep)
            except Exception as e:   # the baseline is reported, never required for the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "iters/s", "cores": 1, "kind": "port",
                                       "sample": "failed: %s" % e}
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
