"""best_of on a dense KKT of order 600 (400 variables, 200 dense equality rows): batched launch against the serial loop.
python tools/dense600_best_of.py [best_of]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dnlp_amd as cp  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def build():
    rng = np.random.default_rng(5)
    n, m = 400, 200
    A = rng.standard_normal((m, n))
    xs = rng.uniform(-1.0, 1.0, n)
    x = cp.Variable(n, name="x")
    x.sample_bounds = [-2.0, 2.0]
    obj = cp.sum(cp.power(x, 4)) - 3.0 * cp.sum(cp.square(x))
    return cp.Problem(cp.Minimize(obj), [A @ x == A @ xs]), x


for batch in (True, False):
    prob, x = build()
    np.random.seed(1)
    t = time.time()
    prob.solve(nlp=True, best_of=B, batch=batch)
    dt = time.time() - t
    objs = np.array(prob.solver_stats.extra_stats["all_objs_from_best_of"])
    print("batch" if batch else "serial", "wall %.2f s" % dt, "value", prob.value, "objs", np.round(objs[:8], 6), "iters", prob.solver_stats.num_iters, flush=True)
