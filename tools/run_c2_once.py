"""One C2 solve (Rosenbrock chain, reduced L-BFGS) for profiling: python tools/run_c2_once.py [n] [reps] [key=value ...]"""
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dnlp_amd as cp  # noqa: E402
from problem_zoo import rosenbrock_chain  # noqa: E402

warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
opts = {"algorithm": "lbfgs"}
opts.update(dict(kv.split("=") for kv in sys.argv[3:]))
p = rosenbrock_chain(cp, n)
chain = p._build_chain(None)
data, inv = chain.apply(p)
for rep in range(reps):
    t0 = time.time()
    info = chain.solver.solve_via_data(data, True, False, dict(opts))
    dt = time.time() - t0
    print(json.dumps({"n": n, "rep": rep, "solve_sec": dt, "iterations": info["iterations"], "evaluations": info["evaluations"],
                      "device_loop": info.get("device_loop"), "device_loop_sec": info.get("device_loop_seconds"),
                      "slots": info.get("device_loop_slots"), "status": info["status"]}))
