"""Localization batches of B fresh instances through ParametricBatch.solve_many with 1 .. 4 launches in flight:
problems/s over 12 batches.  python tools/c5_in_flight.py [B]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import batch_problems as bp  # noqa: E402
from dnlp_amd.batch import ParametricBatch  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nb = 12
prob, params, sample, _ = bp.template_localization()
pb = ParametricBatch(prob, params)
batches = [np.stack([sample(k * B + i) for i in range(B)]) for k in range(nb)]
out = {"which": "localization", "batch": B, "batches": nb, "problems_per_s_by_launches_in_flight": {}}
for w in (1, 2, 3, 4):
    pb.solve_many(batches[:w], in_flight=w)                    # handles of the workers
    t = time.time()
    res = pb.solve_many(batches, in_flight=w)
    dt = time.time() - t
    out["problems_per_s_by_launches_in_flight"][str(w)] = round(B * nb / dt, 1)
    out["optimal"] = int(sum(int(np.sum(r.status == 0)) for r in res))
print(json.dumps(out))
