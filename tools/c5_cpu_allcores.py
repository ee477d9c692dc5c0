"""CPU baseline of BASELINE config C5 on ALL host cores (bench.py's cpu_baseline leg; BASELINE.md: "the host build across
all host cores"): the host build of the generic interior-point algorithm text (ipm_core.h over the test oracle's host
space) solves the batch's instances with ONE shared tape and the template's static-pattern plan — exactly what the
device's batch kernels do — in a pool of worker processes, one per core, each taking chunks of instances until the
time budget is used up.  No GPU is touched in this process tree.  Prints one JSON line.

    python tools/c5_cpu_allcores.py --which localization --batch 8192 --budget 12 [--workers N] [--first 0]"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

# one thread per worker process, BEFORE numpy loads its BLAS: a GPU box shows 256 logical CPUs and grants 16 — every worker's
# BLAS / OpenMP pool of 256 spinning threads is charged to that quota, and the cgroup then throttles the whole tree (measured
# there: the same power-flow sample at 868, 8, 39 and 411 problems/s in four runs in a row, 1 000-2 000 thread-seconds
# throttled per ten seconds of wall time)
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ[_v] = "1"

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

_state = {}


def _init(which, first, batch):
    import batch_problems as bp
    from dnlp_amd.batch import ParametricBatch
    from wave_oracle import HostBatch
    tmpl = {"localization": bp.template_localization, "circle_packing": bp.template_circle_packing,
            "circle_packing10": lambda: bp.template_circle_packing(10),
            "path_planning": bp.template_path_planning, "power_flow": bp.template_power_flow}[which]
    prob, params, sample, _ = tmpl()
    _state["hb"] = HostBatch(ParametricBatch(prob, params))
    # (the instances' parameter rows and data rows are generated before the clock starts, as on the device side)
    _state["mat"] = np.ascontiguousarray(_state["hb"].pb.data(np.stack([sample(first + i) for i in range(batch)])))


def _chunk(args):
    lo, hi, deadline = args
    if time.time() > deadline:
        return 0, 0, 0, 0.0
    t0 = time.time()
    r = _state["hb"].solve_rows(_state["mat"][lo:hi], 1)
    return hi - lo, int(r["iters"].sum()), int((r["status"] == 0).sum()), time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="localization")
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--budget", type=float, default=12.0)
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--chunk", type=int, default=16)
    a = ap.parse_args()
    # the cores this process tree may actually use: the affinity mask, capped by the cgroup's CPU quota (the GPU boxes
    # show 256 logical CPUs and grant 16 CPUs' worth of time: 32 workers there are throttled to half the rate of 16)
    avail, quota_note = len(os.sched_getaffinity(0)), ""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cap = max(1, int(float(q) / float(per) + 0.5))
            if cap < avail:
                quota_note = "; %d logical CPUs visible, cgroup cpu.max grants %d" % (avail, cap)
                avail = cap
    except (OSError, ValueError):
        pass
    workers = a.workers or avail
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    ctx = mp.get_context("fork")
    with ctx.Pool(workers, initializer=_init, initargs=(a.which, a.first, a.batch)) as pool:
        pool.map(_chunk, [(0, 1, time.time() + 1e9)] * workers)          # every worker has lowered the template
        t0 = time.time()
        deadline = t0 + a.budget
        passes, done, iters, optimal, busy = 0, 0, 0, 0, 0.0
        while time.time() < deadline:                                    # whole passes over the batch until the budget is used
            jobs = [(lo, min(lo + a.chunk, a.batch), deadline) for lo in range(0, a.batch, a.chunk)]
            for n, it, ok, sec in pool.imap_unordered(_chunk, jobs):
                done += n; iters += it; optimal += ok; busy += sec
            passes += 1
        dt = time.time() - t0
    print(json.dumps({"value": done / dt, "unit": "problems/s", "cores": workers, "kind": "port", "iters_per_s": iters / dt,
                      "problems": done, "optimal": optimal, "seconds": dt, "per_core_problems_per_s": done / busy if busy else None,
                      "sample": "host build of the generic interior-point algorithm (ipm_core.h, sparse static-pattern LDL^T), ONE shared "
                                "tape + the template's plan as on the device, %d worker processes (one per usable host core) over the "
                                "batch's instances for %.1f s: %d problems%s" % (workers, dt, done, quota_note)}))


if __name__ == "__main__":
    main()
