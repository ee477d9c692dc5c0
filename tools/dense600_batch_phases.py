"""Phase split of a batched launch on the dense order-600 problem of tools/dense600_best_of.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dnlp_amd as cp  # noqa: E402
from dnlp_amd.batch import _device_handle, instance_data  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rng = np.random.default_rng(5)
n, m = 400, 200
A = rng.standard_normal((m, n))
xs = rng.uniform(-1.0, 1.0, n)
x = cp.Variable(n, name="x")
x.sample_bounds = [-2.0, 2.0]
prob = cp.Problem(cp.Minimize(cp.sum(cp.power(x, 4)) - 3.0 * cp.sum(cp.square(x))), [A @ x == A @ xs])
chain = prob._build_chain(None)
np.random.seed(1)
rows = []
for run in range(B):
    prob.set_random_NLP_initial_point(run)
    data, inv = chain.apply(prob, make_handle=False)
    rows.append(instance_data(data["tape_arrays"]))
opts = {"print_level": 0}
for k, v in [a.split("=") for a in sys.argv[2:]]:
    opts[k] = v
h = _device_handle(data["tape_arrays"], data["tape"], None, opts)
print("kkt info", {k: getattr(h, k)() for k in ("kkt_info",) if hasattr(h, k)})
raw = h.solve_batch(np.stack(rows), want_duals=True)
ph = raw["phase_seconds"]
print("kernel_seconds", raw["kernel_seconds"], "iters", raw["iterations"], "fact", raw.get("factorizations"), "status", raw["status"])
print("phase seconds per instance [wall, tape, factorisation, solves]:", np.round(ph, 3))
