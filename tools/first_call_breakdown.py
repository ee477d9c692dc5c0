"""Where the first call of a solve goes before the first kernel: canonical form (dnlp2smooth), lowering (DAG walk over
C++ affine forms + dnlp_lower_maps), tape serialisation / upload (dnlp_create*), symbolic sparse plan (kkt_info), and
the solve itself — for BASELINE C2 (canonical form), C3 and the notebook examples.  One JSON line per workload."""
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
from dnlp_amd import lowering, nlp_solver  # noqa: E402
from dnlp_amd.dnlp2smooth import Dnlp2Smooth  # noqa: E402
from paper_examples import PAPER, PAPER_LARGE  # noqa: E402
from problem_zoo import rosenbrock_chain  # noqa: E402

warnings.simplefilter("ignore")


def c3(cp_):
    n, m = 10000, 1000
    rng = np.random.default_rng(0)
    Gm = rng.standard_normal((n, n))
    Q = Gm.T @ Gm / n + np.eye(n)
    x = cp_.Variable(n)
    A = rng.standard_normal((m, n))
    return cp_.Problem(cp_.Minimize(0.5 * cp_.quad_form(x, Q) + rng.standard_normal(n) @ x), [A @ x == A @ rng.standard_normal(n)])


WORK = {"c2_canonical_n1e5": lambda c: rosenbrock_chain(c, 100000), "c3": c3}
for k in ("nb_nmf", "nb_nmf_small", "nb_path_planning", "nb_power_flow", "nb_portfolio_construction", "hs071"):
    src = dict(PAPER)
    src.update(PAPER_LARGE)
    try:
        from problem_zoo import ZOO
        src.update(ZOO)
    except Exception:
        pass
    if k in src:
        WORK[k] = src[k]

marks = {}
orig_lower = lowering.lower_problem


def timed_lower(*a, **kw):
    t = time.time()
    r = orig_lower(*a, **kw)
    marks["lower_problem"] = time.time() - t
    return r


lowering.lower_problem = timed_lower
nlp_solver.lower_problem = timed_lower
out = []
# what every process pays once, whatever it solves first (library load, HIP context, first kernel-module load): a toy problem
# first, reported on its own line, so that the workloads' lines are THEIR first calls
_t = time.time()
_x = cp.Variable(3)
_p = cp.Problem(cp.Minimize(cp.sum(cp.exp(_x)) + cp.sum_squares(_x)), [cp.sum(_x) == 1])
_p.solve(nlp=True)
print(json.dumps({"workload": "process_init (first solve of a three-variable problem)", "seconds": time.time() - _t}), flush=True)
for name in [a for a in sys.argv[1:] if not a.startswith('-')] or list(WORK):
    prob = WORK[name](cp)
    marks.clear()
    t0 = time.time()
    chain = prob._build_chain(None)
    t1 = time.time()
    if os.environ.get("DNLP_PROFILE_APPLY"):
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        data, inv = chain.apply(prob)
        pr.disable()
        pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(12)
    else:
        data, inv = chain.apply(prob)
    t2 = time.time()
    info_k = data["handle"].kkt_info()
    t3 = time.time()
    opts = {"device_loop": "host"} if name.startswith("c2") else {}
    info = chain.solver.solve_via_data(data, True, False, opts)
    t4 = time.time()
    rec = {"workload": name, "N": len(data["x0"]), "m": len(data["cl"]), "chain_sec": t1 - t0, "apply_sec": t2 - t1,
           "lower_problem_sec": marks.get("lower_problem"), "plan_sec": t3 - t2, "first_solve_sec": t4 - t3,
           "status": int(info["status"]), "iterations": int(info["iterations"])}
    t5 = time.time()
    info = chain.solver.solve_via_data(data, True, False, opts)
    rec["second_solve_sec"] = time.time() - t5
    print(json.dumps(rec), flush=True)
    out.append(rec)
    data["handle"].close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "first_call_breakdown.json"), "w"), indent=1)
