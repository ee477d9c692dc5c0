"""BASELINE config C2 measurement: tape f / grad f evaluation of the Rosenbrock chain in the
reference-canonical form (N = 4n-3 variables) for an n-sweep.  Run under
rocprofv3 --kernel-trace --stats to get per-kernel durations; this script reports wall time per
oracle call (which includes the host<->device copies of x and grad) and the algorithmic bytes."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import warnings  # noqa: E402

import dnlp_amd as cp  # noqa: E402
from dnlp_amd import _capi  # noqa: E402
from dnlp_amd.dnlp2smooth import Dnlp2Smooth  # noqa: E402
from dnlp_amd.nlp_solver import build_nlp_data  # noqa: E402
from dnlp_amd.tape import serialize  # noqa: E402
from problem_zoo import rosenbrock_chain  # noqa: E402

warnings.simplefilter("ignore")
out = []
for n in [int(a) for a in sys.argv[1:]] or [100000, 1000000]:
    t0 = time.time()
    prob = rosenbrock_chain(cp, n)
    smooth, _ = Dnlp2Smooth().apply(prob)
    data, _ = build_nlp_data(smooth)
    blob = serialize(data["tape_arrays"])
    t_lower = time.time() - t0
    h = _capi.DeviceProblem(blob, data["tape"], device=0)
    N, m = h.n, h.m
    x = np.random.default_rng(0).standard_normal(N)
    h.eval_f(x); h.eval_grad_f(x); h.eval_g(x); h.eval_jac_g(x)
    reps = 10
    t0 = time.time()
    for _ in range(reps):
        h.eval_f(x)
    tf = (time.time() - t0) / reps
    t0 = time.time()
    for _ in range(reps):
        h.eval_grad_f(x)
    tg = (time.time() - t0) / reps
    t0 = time.time()
    for _ in range(reps):
        h.eval_jac_g(x)
    tj = (time.time() - t0) / reps
    lam = np.ones(m)
    t0 = time.time()
    for _ in range(reps):
        h.eval_h(x, lam, 1.0)
    th = (time.time() - t0) / reps
    rec = {"n": n, "N": N, "m": m, "nnzJ": h.nnz_jac, "nnzH": h.nnz_hess, "lower_sec": t_lower,
           "wall_f_ms": 1e3 * tf, "wall_grad_ms": 1e3 * tg, "wall_jac_ms": 1e3 * tj, "wall_hess_ms": 1e3 * th,
           "alg_bytes_f": 8 * N, "alg_bytes_grad": 16 * N, "alg_bytes_jac": 8 * N + 8 * h.nnz_jac,
           "alg_bytes_hess": 8 * N + 8 * m + 8 * h.nnz_hess}
    print(json.dumps(rec), flush=True)
    out.append(rec)
    h.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "tape_sweep.json"), "w"), indent=1)
