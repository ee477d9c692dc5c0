"""BASELINE config C3 at full size on the MI355X: dense equality-constrained QP, n=1e4, m=1e3
(SURVEY.md §8d).  One Newton step is exact; reports factorisation time and TFLOP/s."""
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
xh = rng.standard_normal(n)
b = A @ xh
x = cp.Variable(n)
prob = cp.Problem(cp.Minimize(0.5 * cp.quad_form(x, Q) + c @ x), [A @ x == b])
t0 = time.time()
chain = prob._build_chain(None)
data, inv = chain.apply(prob)
t_lower = time.time() - t0
t0 = time.time()
info = chain.solver.solve_via_data(data, True, False, {"time_kernels": "yes"})
t_solve = time.time() - t0
K = np.block([[Q, A.T], [A, np.zeros((m, m))]])
t0 = time.time()
sol = np.linalg.solve(K, np.concatenate([-c, b]))
t_lapack = time.time() - t0
st = info["stats"]
out = {"n": n, "m": m, "status": info["status"], "iters": info["iterations"], "lower_sec": t_lower,
       "solve_sec": t_solve, "t_factor": st[4], "factorizations": st[1],
       "kkt_flops": (n + m) ** 3 / 3.0, "factor_TFLOPs": st[1] * (n + m) ** 3 / 3.0 / st[4] / 1e12,
       "mfma_update_TFLOPs": st[14] / st[13] / 1e12 if st[13] > 0 else None,
       "x_err": float(np.max(np.abs(info["x"] - sol[:n]))), "dual_err": float(np.max(np.abs(info["mult_g"] - sol[n:]))),
       "host_lapack_solve_sec": t_lapack}
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "c3_n%d.json" % n), "w"), indent=1)
