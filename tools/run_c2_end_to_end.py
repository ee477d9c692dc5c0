"""BASELINE config C2 end to end through the front-end: Problem.solve(nlp=True, algorithm="lbfgs") on the
Rosenbrock chain, first call (lowering + upload + kernel generation + solve) and the second call of the same
Problem object (cached handle).  python tools/run_c2_end_to_end.py [n]"""
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
out = []
for rep in range(3):                       # rep 0 pays the one-off costs of the process (hiprtc, code-object cache)
    p = rosenbrock_chain(cp, n)
    t0 = time.time()
    p.solve(nlp=True, algorithm="lbfgs", tol=1e-9)
    t_first = time.time() - t0
    err = float(np.max(np.abs(p.variables()[0].value - 1.0)))
    p.variables()[0].value = None                       # the same default start as the first call
    t0 = time.time()
    p.solve(nlp=True, algorithm="lbfgs", tol=1e-9)
    t_second = time.time() - t0
    d = p._nlp_cache["data"]
    out.append({"n": n, "rep": rep, "first_call_sec": t_first, "second_call_sec": t_second, "status": p.status,
                "max_abs_x_minus_1": err, "tape_N": int(d["tape"].N), "tape_m": int(d["tape"].m),
                "path": p._nlp_cache["sig"][0] if isinstance(p._nlp_cache["sig"][0], str) else "canonical"})
    print(json.dumps(out[-1]), flush=True)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "c2_end_to_end.json"), "w"), indent=1)
