"""Full solve of BASELINE config C4 at order n on the MI355X, checked against the analytic
answer: the optimal value of  max x'Ax s.t. ||x|| = 1  is lambda_max(A), obtained here by
power iteration with the device symmetric product.  Writes gpurun_out/full_solve_n<N>.json."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import dnlp_amd as cp  # noqa: E402
from dnlp_amd.device import symmetric_test_matrix  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
opts = {}
for a in sys.argv[2:]:
    k, v = a.split("=")
    opts[k] = v
A = symmetric_test_matrix(n, seed=0, spike_eig=4.0 * np.sqrt(n))
x = cp.Variable(n)
rng = np.random.default_rng(0)
x.value = np.ones(n) / np.sqrt(n) + 0.1 * rng.standard_normal(n) / np.sqrt(n)
prob = cp.Problem(cp.Maximize(cp.quad_form(x, cp.Constant(A.handle))), [cp.sum_squares(x) == 1])
t0 = time.time()
chain = prob._build_chain(None)
data, inv = chain.apply(prob)
t_lower = time.time() - t0
h = data["handle"]
t0 = time.time()
sol = chain.solver.solve_via_data(data, True, False, dict(kkt_pivot_max_n=0, **opts))
t_solve = time.time() - t0
log = h.log()
prob.unpack_results(sol, chain, inv)
# power iteration started from the solution (a few steps certify it) and from a random vector
v = rng.standard_normal(n)
v /= np.linalg.norm(v)
lam = 0.0
for it in range(300):
    w = A.symv(v)
    lam_new = float(v @ w)
    v = w / np.linalg.norm(w)
    if abs(lam_new - lam) <= 1e-13 * abs(lam_new):
        lam = lam_new
        break
    lam = lam_new
xs = x.value / np.linalg.norm(x.value)
res = np.linalg.norm(A.symv(xs) - prob.value * xs)
out = {"n": n, "status": prob.status, "value": prob.value, "lambda_max_power_iter": lam,
       "rel_err": abs(prob.value - lam) / abs(lam), "eig_residual": res, "iters": sol["iterations"],
       "solve_sec": t_solve, "lower_sec": t_lower, "iters_per_sec": sol["iterations"] / t_solve,
       "stats": [float(s) for s in sol["stats"]], "power_iters": it + 1}
print(json.dumps(out))
print(log)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "full_solve_n%d.json" % n), "w"), indent=1)
open(os.path.join(ROOT, "gpurun_out", "full_solve_n%d.log" % n), "w").write(log)
