"""A/B of the in-kernel (one workgroup, batch of one) and the host-driven interior-point loop on the paper
examples that take the sparse KKT path: python tools/device_loop_ab.py [example ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dnlp_amd as cp  # noqa: E402
from paper_examples import PAPER, PAPER_LARGE  # noqa: E402

ALL = dict(PAPER)
ALL.update(PAPER_LARGE)
names = sys.argv[1:] or ["nb_nmf_small", "nb_path_planning", "nb_power_flow", "nb_circle_packing", "nb_localization"]
for name in names:
    for mode in ("device", "host"):
        best = None
        for rep in range(2):
            prob = ALL[name](cp)
            chain = prob._build_chain(None)
            data, inv = chain.apply(prob)
            info_k = data["handle"].kkt_info()
            t0 = time.time()
            info = chain.solver.solve_via_data(data, True, False, {"device_loop": "yes" if mode == "device" else "no"})
            dt = time.time() - t0
            best = dt if best is None else min(best, dt)
        print(json.dumps({"example": name, "mode": mode, "solve_sec": best, "iters": int(info["iterations"]), "status": int(info["status"]),
                          "order": len(data["x0"]) + len(data["cl"]), "plan": {k: info_k.get(k) for k in ("sparse", "update_triples", "levels", "factor_values")}}))
