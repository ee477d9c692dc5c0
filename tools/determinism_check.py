"""Run-to-run reproducibility of a solve: the same canonical problem solved five times through one handle and
through fresh handles; prints iteration counts, statuses and whether objective and x are bitwise equal.
python tools/determinism_check.py [example ...]   (names of tests/paper_examples.py)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dnlp_amd as cp  # noqa: E402
from paper_examples import PAPER, PAPER_LARGE  # noqa: E402

ALL = dict(PAPER)
ALL.update(PAPER_LARGE)
names = sys.argv[1:] or ["nb_power_flow", "nb_phase_retrieval", "nb_localization", "nb_nmf_small"]
for name in names:
    runs = []
    for fresh in range(2):
        prob = ALL[name](cp)
        chain = prob._build_chain(None)
        data, inv = chain.apply(prob)
        for rep in range(3):
            info = chain.solver.solve_via_data(dict(data), True, False, {})
            runs.append((int(info["iterations"]), int(info["status"]), float(info["obj_val"]), np.array(info["x"])))
    its = [r[0] for r in runs]
    same_obj = all(r[2] == runs[0][2] for r in runs)
    same_x = all(np.array_equal(r[3], runs[0][3]) for r in runs)
    spread = max(abs(r[2] - runs[0][2]) for r in runs) / max(abs(runs[0][2]), 1e-300)
    print(f"{name:24s} iterations {its} status {[r[1] for r in runs]} objective bitwise equal {same_obj} "
          f"(relative spread {spread:.1e}) x bitwise equal {same_x}")
