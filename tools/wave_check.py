"""Wavefront batch solver (csrc/wave_*.h) against the generic batch kernel (csrc/batch.h) on the device: the same fresh
batch through both, instance by instance — status, iteration count, objective, x — and the kernel time of each.

    python tools/wave_check.py --which localization,circle_packing --batch 1024 --reps 3
Writes one JSON line per template (and gpurun_out/wave_check.jsonl)."""
import argparse
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

ap = argparse.ArgumentParser()
ap.add_argument("--which", default="localization")
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--skip-generic", action="store_true")
ap.add_argument("--out", default="wave_check.jsonl")
args = ap.parse_args()

TMPL = {"localization": bp.template_localization, "circle_packing": bp.template_circle_packing,
        "circle_packing10": lambda: bp.template_circle_packing(10),
        "path_planning": bp.template_path_planning, "power_flow": bp.template_power_flow}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
out = open(os.path.join(ROOT, "gpurun_out", args.out), "a")
for which in args.which.split(","):
    prob, params, sample, _ = TMPL[which]()
    pb = ParametricBatch(prob, params)
    B = args.batch
    res = {}
    for mode in (["wave"] if args.skip_generic else ["wave", "generic"]):
        os.environ["DNLP_BATCH_WAVE"] = "1" if mode == "wave" else "0"
        best = None
        for rep in range(args.reps):
            thetas = np.stack([sample(rep * B + i) for i in range(B)])         # a fresh batch every repetition
            t0 = time.time()
            r = pb.solve(thetas)
            wall = time.time() - t0
            if rep == 0:
                first = r
            if best is None or r.kernel_seconds < best[0]:
                best = (r.kernel_seconds, wall, r)
        res[mode] = (first, best)
    w0, wb = res["wave"]
    row = {"problem": which, "batch": B, "N": int(pb.arrays0["dims"][0]), "m": int(pb.arrays0["dims"][1]),
           "wave_launch": w0.raw.get("launch"),
           "wave_kernel_ms_best": 1e3 * wb[0], "wave_problems_per_s_kernel": B / wb[0], "wave_problems_per_s_wall": B / wb[1],
           "wave_iters_mean": float(w0.iterations.mean()), "wave_iters_max": int(w0.iterations.max()),
           "wave_us_per_iteration_if_spread": 1e6 * wb[0] / float(wb[2].iterations.sum()) * min(B, 1024),
           "wave_instance_wall_ms_mean": float(1e3 * wb[2].raw["phase_seconds"][:, 0].mean()),
           "wave_instance_us_per_iter": float(1e6 * wb[2].raw["phase_seconds"][:, 0].sum() / wb[2].iterations.sum()),
           "wave_optimal": int(np.sum(w0.status == 0)), "wave_status_hist": {int(k): int(v) for k, v in zip(*np.unique(w0.status, return_counts=True))}}
    if "generic" in res:
        g0, gb = res["generic"]
        same = (w0.status == g0.status)
        rel = np.abs(w0.obj_val - g0.obj_val) / np.maximum(1.0, np.abs(g0.obj_val))
        both = (w0.status == 0) & (g0.status == 0)
        row.update({"generic_launch": g0.raw.get("launch"), "generic_kernel_ms_best": 1e3 * gb[0],
                    "generic_problems_per_s_kernel": B / gb[0],
                    "generic_instance_us_per_iter": float(1e6 * gb[2].raw["phase_seconds"][:, 0].sum() / gb[2].iterations.sum()),
                    "generic_optimal": int(np.sum(g0.status == 0)),
                    "same_status": int(same.sum()), "same_iterations": int(np.sum(w0.iterations == g0.iterations)),
                    "max_rel_obj_diff_both_optimal": float(rel[both].max()) if both.any() else None,
                    "max_abs_x_diff_both_optimal": float(np.abs(w0.x[both] - g0.x[both]).max()) if both.any() else None,
                    "speedup_kernel": gb[0] / wb[0]})
    print(json.dumps(row), flush=True)
    out.write(json.dumps(row) + "\n")
    out.flush()
    pb.close()
