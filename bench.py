"""bench.py — interior-point iterations/sec on BASELINE config C4 (the metric's workload):
maximise x'Ax on the unit sphere, dense symmetric A (n = 1e5 by default, 80 GB FP64 in HBM),
through the drop-in front-end (quad_form + sum_squares -> dnlp2smooth -> device tape) and the
on-device interior-point loop (tape f/grad/g/Jac/Hess kernels, dense KKT assembly, blocked
LDL^T with the FP64-MFMA Schur-complement update, solves with refinement, filter line search).

A "step" is one interior-point iteration.  N > 1 runs one replica per GPU (the KKT
factorisation does not shard without a distributed factorisation): weak
scaling, no data-path collective, one RCCL all_gather of the per-rank results at the end.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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
    """Exactly n_steps interior-point iterations; restarts from a perturbed point if the
    solve converges before the count is reached (counted iterations are all real ones)."""
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


def cpu_baseline(n_cpu, seed, device, steps):
    """The CPU port (oracle/, host instantiation of the same algorithm) on a bounded sample of
    the same workload: same generator and front-end at order n_cpu, `steps` iterations."""
    # the scalar OpenMP LDL^T of the port scales to a few dozen threads, not to every core of
    # the box: pin the thread count before libgomp starts and report exactly that number
    threads = min(os.cpu_count() or 1, 32)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    import dnlp_amd as cp
    from dnlp_amd.device import symmetric_test_matrix
    from dnlp_amd.dnlp2smooth import Dnlp2Smooth
    from dnlp_amd.nlp_solver import build_nlp_data
    A = symmetric_test_matrix(n_cpu, seed=seed, spike_eig=4.0 * np.sqrt(n_cpu), device=device)
    Ah = A.to_host()
    A.free()
    x = cp.Variable(n_cpu)
    rng = np.random.default_rng(seed)
    x.value = np.ones(n_cpu) / np.sqrt(n_cpu) + 0.1 * rng.standard_normal(n_cpu) / np.sqrt(n_cpu)
    prob = cp.Problem(cp.Minimize(-cp.quad_form(x, Ah)), [cp.sum_squares(x) == 1])
    smooth, _ = Dnlp2Smooth().apply(prob)
    data, _ = build_nlp_data(smooth)
    orc = OracleProblem(serialize(data["tape_arrays"]))
    orc.set_option("kkt_pivot_max_n", 0)      # same unpivoted LDL^T path as the GPU run
    orc.ipm_begin(data["x0"])
    t0 = time.time()
    rc, k = orc.ipm_step(steps)
    dt = time.time() - t0
    import ctypes.util
    cores = threads
    return {"value": k / dt, "unit": "iters/s", "cores": cores, "kind": "port",
            "sample": "same generator/front-end at n=%d (dense KKT order %d), %d iterations of the "
                      "host build of the same algorithm (OpenMP LDL^T on %d threads); work per "
                      "iteration scales as n^3/3, so n=1e5 is %.3g x this sample per iteration; "
                      "libipopt on this box: %s" % (n_cpu, n_cpu + 1, k, cores, (1e5 / n_cpu) ** 3,
                                                    ctypes.util.find_library("ipopt") or "not found")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--order", type=int, default=100000, help="order n of the dense NLP (BASELINE: 1e5)")
    ap.add_argument("--cpu-n", type=int, default=2400)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    ndev = max(torch.cuda.device_count(), 1)
    local = local % ndev            # one process per GPU; wraps only in single-GPU debugging runs
    os.environ["DNLP_DEVICE"] = str(local)
    dist = None
    tdev = "cpu"
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            tdev = "cuda"
        else:
            dist.init_process_group(args.backend)

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
    st0 = h.ipm_finish()["stats"].copy()

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
    st1 = info["stats"]
    if dist is not None:
        tmax = torch.tensor([dt], device=tdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt_all = float(tmax.item())
        # the one exchange of the path: gather per-rank results (objective, iterations, status)
        mine = torch.tensor([info["obj_val"], float(info["iterations"]), float(info["status"])],
                            device=tdev, dtype=torch.float64)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
    else:
        dt_all = dt
    if rank == 0:
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes of this same
        # command (FETCH_SIZE and WRITE_SIZE cannot share a pass); the committed summary is quoted
        # only when it was taken at the same order n
        traffic, traffic_src = None, None
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_bench_traffic.json")))
            if int(pj["n"]) == n:
                traffic = pj["avg_traffic_bytes_per_launch"]
                traffic_src = "profiles/r01_pmc_bench_traffic.json (" + pj["command"] + ")"
        except (OSError, KeyError, ValueError):
            pass
        upd_s, upd_f, upd_l = st1[13] - st0[13], st1[14] - st0[14], st1[15] - st0[15]
        achieved = upd_f / upd_s / 1e12 if upd_s > 0 else None
        out = {
            "metric": "interior-point iters/sec, n=%d dense NLP (sphere quad_form max), 1 replica per GPU" % n,
            "value": world * args.steps / dt_all,
            "unit": "iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_all / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "C4: max x'Ax s.t. ||x||^2=1, dense symmetric A generated in HBM, "
                                   "n=%d (KKT order %d), full Jac_g / Hess_L callbacks" % (n, n + 1),
                       "kkt": "blocked unpivoted LDL^T, FP64-MFMA Schur update", "replicas": world,
                       "restarts_in_timed_region": restarts,
                       "factorizations_in_timed_region": int(st1[1] - st0[1])},
            "roofline": {"bound": "mfma", "kernel": "gemm_nt_update (Schur-complement update)",
                         "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": (achieved / PEAK_FP64_MFMA_TFLOPS) if achieved else None,
                         "launches": int(upd_l), "avg_launch_ms": 1e3 * upd_s / upd_l if upd_l else None,
                         "avg_launch_gflop": upd_f / upd_l / 1e9 if upd_l else None,
                         "traffic": traffic, "traffic_unit": "bytes per launch (FETCH x2-corrected + WRITE)",
                         "traffic_source": traffic_src},
        }
        if not args.no_cpu:
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_n, 0, local, 3)
            except Exception as e:   # the baseline is reported, never required for the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "iters/s", "cores": 1, "kind": "port",
                                       "sample": "failed: %s" % e}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
