"""BASELINE C2 at its stated size (n = 1e5 by default): the device-resident L-BFGS alone, repeated on a live handle —
seconds per solve, slots, iterations, evaluations — for the variants selected by environment variables
(DNLP_LBFGS_GRAPH, DNLP_LBFGS_E).  Prints one JSON line per size."""
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
for n in [int(a) for a in sys.argv[1:]] or [100000]:
    p = rosenbrock_chain(cp, n)
    chain = p._build_chain(None)
    data, inv = chain.apply(p)
    rows = []
    for rep in range(6):
        t0 = time.time()
        info = chain.solver.solve_via_data(data, True, False, {"algorithm": "lbfgs"})
        wall = time.time() - t0
        rows.append({"wall_ms": 1e3 * wall, "device_loop_ms": 1e3 * info.get("device_loop_seconds", 0.0),
                     "slots": info.get("device_loop_slots"), "iterations": info["iterations"],
                     "evaluations": info["evaluations"], "status": info["status"], "f": info["obj_val"]})
    best = min(rows[1:], key=lambda r: r["device_loop_ms"])
    print(json.dumps({"n": n, "graph": os.environ.get("DNLP_LBFGS_GRAPH", "1"), "E": os.environ.get("DNLP_LBFGS_E", "auto"),
                      "best": best, "us_per_slot": 1e3 * best["device_loop_ms"] / max(best["slots"], 1),
                      "all_device_loop_ms": [round(r["device_loop_ms"], 3) for r in rows]}))
