"""BASELINE config C2 on the MI355X: unconstrained Rosenbrock chain, reduced-space L-BFGS
(tape f / grad f evaluation + line search only).  Reports solve time, evaluations and the
algorithmic-bandwidth figure 16 n bytes per f + grad f evaluation for an
n-sweep; writes gpurun_out/c2.json."""
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dnlp_amd as cp  # noqa: E402
from problem_zoo import rosenbrock_chain  # noqa: E402

warnings.simplefilter("ignore")
out = []
for n in [int(a) for a in sys.argv[1:]] or [100000, 1000000, 4000000]:
    p = rosenbrock_chain(cp, n)
    t0 = time.time()
    chain = p._build_chain(None)
    data, inv = chain.apply(p)
    t_lower = time.time() - t0
    t0 = time.time()
    info = chain.solver.solve_via_data(data, True, False, {"algorithm": "lbfgs"})
    t_solve = time.time() - t0
    p.unpack_results(info, chain, inv)
    x = p.variables()[0].value
    rec = {"n": n, "N_canonical": len(data["x0"]), "status": p.status, "iterations": info["iterations"],
           "evaluations": info["evaluations"], "f": info["obj_val"], "max_abs_x_minus_1": float(np.max(np.abs(x - 1))),
           "lower_sec": t_lower, "solve_sec": t_solve, "ms_per_f_grad_eval": 1e3 * t_solve / max(info["evaluations"], 1),
           "alg_GBps_16n": 16.0 * n * info["evaluations"] / t_solve / 1e9,
           "iters_per_sec": info["iterations"] / t_solve}
    print(json.dumps(rec), flush=True)
    out.append(rec)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "c2.json"), "w"), indent=1)
