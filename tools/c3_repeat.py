"""BASELINE C3 solved three times through one handle: first-call cost (allocation, code-object load) against the
repeat, and the kernel-time share (stats)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dnlp_amd as cp  # noqa: E402

n, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 1000)
rng = np.random.default_rng(0)
Gm = rng.standard_normal((n, n))
Q = Gm.T @ Gm / n + np.eye(n)
c = rng.standard_normal(n)
A = rng.standard_normal((m, n))
b = A @ rng.standard_normal(n)
x = cp.Variable(n)
prob = cp.Problem(cp.Minimize(0.5 * cp.quad_form(x, Q) + c @ x), [A @ x == b])
chain = prob._build_chain(None)
t0 = time.time()
data, inv = chain.apply(prob)
out = {"lower_sec": time.time() - t0, "solves": []}
for rep in range(3):
    t0 = time.time()
    info = chain.solver.solve_via_data(data, True, False, {"time_kernels": "yes"})
    st = info["stats"]
    out["solves"].append({"wall_sec": time.time() - t0, "status": int(info["status"]), "iters": int(info["iterations"]),
                          "t_factor": float(st[4]), "stats_head": [float(v) for v in st[:12]]})
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "c3_repeat.json"), "w"), indent=1)
