"""BASELINE config C2 in the reference's CANONICAL form (what it hands to IPOPT: N = 4n - 3
variables, m = 3n - 3 equalities, dnlp2smooth.py:42-111) through the interior-point loop with the
static-pattern sparse KKT factorisation (csrc/sparse_plan.h).  The KKT order is 7n - 6: 699 994
at n = 1e5 — a dense factorisation is impossible, the sparse one has 2.0e6 factor values."""
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
for n in [int(a) for a in sys.argv[1:]] or [1000, 10000]:
    p = rosenbrock_chain(cp, n)
    t0 = time.time()
    chain = p._build_chain(None)
    data, inv = chain.apply(p)
    t_lower = time.time() - t0
    info_k = data["handle"].kkt_info()
    t0 = time.time()
    info = chain.solver.solve_via_data(data, True, False, {"device_loop": "host"})
    t_solve = time.time() - t0
    p.unpack_results(info, chain, inv)
    x = p.variables()[0].value
    rec = {"n": n, "kkt_order": len(data["x0"]) + len(data["cl"]), "plan": info_k, "status": p.status,
           "iterations": info["iterations"], "factorizations": int(info["stats"][1]), "f": info["obj_val"],
           "max_abs_x_minus_1": float(np.max(np.abs(x - 1))), "lower_sec": t_lower, "solve_sec": t_solve,
           "factor_sec": float(info["stats"][4])}
    print(json.dumps(rec), flush=True)
    out.append(rec)
    data["handle"].close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "c2_ipm.json"), "w"), indent=1)
