"""Batched launch on dense problems of several KKT orders (n variables under n/2 dense equality rows, a double well per
coordinate): seconds per in-kernel factorisation / solve by the device clock, against the serial loop's wall per solve.
python tools/dense_batch_orders.py [batch]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dnlp_amd as cp  # noqa: E402
from dnlp_amd.batch import _device_handle, instance_data  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for nv in (200, 400, 600, 900, 1300):
    rng = np.random.default_rng(5)
    m = nv // 2
    A = rng.standard_normal((m, nv))
    xs = rng.uniform(-1.0, 1.0, nv)
    x = cp.Variable(nv, name="x")
    x.sample_bounds = [-2.0, 2.0]
    prob = cp.Problem(cp.Minimize(cp.sum(cp.power(x, 4)) - 3.0 * cp.sum(cp.square(x))), [A @ x == A @ xs])
    chain = prob._build_chain(None)
    np.random.seed(1)
    rows = []
    for run in range(B):
        prob.set_random_NLP_initial_point(run)
        data, inv = chain.apply(prob, make_handle=False)
        rows.append(instance_data(data["tape_arrays"]))
    h = _device_handle(data["tape_arrays"], data["tape"], None, {"print_level": 0})
    raw = h.solve_batch(np.stack(rows), want_duals=True)
    ph = raw["phase_seconds"]
    nf = raw["factorizations"].sum()
    h.close()
    t = time.time()
    prob.set_random_NLP_initial_point(0)
    prob.solve(nlp=True)
    serial = time.time() - t
    print(json.dumps({"kkt_order": nv + m, "batch": B, "kernel_s": round(float(raw["kernel_seconds"]), 4),
                      "iterations_mean": float(raw["iterations"].mean()), "statuses": sorted(set(int(v) for v in raw["status"])),
                      "ms_per_factorization": round(1e3 * float(ph[:, 2].sum() / nf), 3),
                      "ms_per_iteration_solves": round(1e3 * float(ph[:, 3].sum() / raw["iterations"].sum()), 3),
                      "serial_one_solve_wall_s": round(serial, 3), "serial_iterations": int(prob.solver_stats.num_iters)}), flush=True)
