"""BASELINE config C5 through the single-launch batch path (csrc/batch.h): B parametrised
localization / circle-packing instances, one workgroup per instance, the whole interior-point
loop on the device.  Prints problems/s and checks a sample of instances against the CPU oracle."""
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
from dnlp_amd.batch import ParametricBatch, arrays_with_data  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--check", type=int, default=16)
ap.add_argument("--which", default="localization,circle_packing")
ap.add_argument("--reps", type=int, default=2)
args = ap.parse_args()

rows = []
for which in args.which.split(","):
    tmpl = {"localization": bp.template_localization, "circle_packing": bp.template_circle_packing}[which]
    prob, params, sample, var = tmpl()
    t0 = time.time()
    pb = ParametricBatch(prob, params)
    t_template = time.time() - t0
    thetas = np.stack([sample(i) for i in range(args.batch)])
    t0 = time.time()
    mat = pb.data(thetas)
    t_data = time.time() - t0
    best = None
    for rep in range(args.reps):
        t0 = time.time()
        res = pb.solve(thetas)
        wall = time.time() - t0
        if best is None or wall < best[0]:
            best = (wall, res)
    wall, res = best
    row = {"problem": which, "batch": args.batch, "N": int(pb.arrays0["dims"][0]), "m": int(pb.arrays0["dims"][1]),
           "affine_template": bool(pb.affine), "template_sec": t_template, "data_sec": t_data,
           "kernel_sec": res.kernel_seconds, "wall_sec": wall,
           "problems_per_sec_kernel": args.batch / res.kernel_seconds,
           "problems_per_sec_wall": args.batch / wall,
           "optimal": int(np.sum(res.status == 0)), "acceptable": int(np.sum(res.status == 1)),
           "iters_mean": float(res.iterations.mean()), "iters_max": int(res.iterations.max()),
           "iters_per_sec_kernel": float(res.iterations.sum() / res.kernel_seconds),
           "factorizations_mean": float(res.factorizations.mean()),
           "phase_ms_per_iter": dict(zip(("wall", "eval", "factor", "solve"),
                                         (1e3 * res.raw["phase_seconds"].sum(axis=0) / res.iterations.sum()).tolist()))}
    if args.check:
        from dnlp_amd.nlp_solver import HIPNLP
        from dnlp_amd.tape import serialize
        from oracle.oracle_capi import OracleProblem
        worst_obj, worst_x, same_iters = 0.0, 0.0, 0
        t0 = time.time()
        for i in range(min(args.check, args.batch)):
            orc = OracleProblem(serialize(arrays_with_data(pb.arrays0, mat[i])))
            for k, v in HIPNLP.DEFAULT_OPTIONS.items():
                orc.set_option(k, v)
            oi = orc.solve(mat[i][-0:][pb.d0.size - 0:] if False else arrays_with_data(pb.arrays0, mat[i])["x0"])
            worst_obj = max(worst_obj, abs(oi["obj_val"] - res.raw["obj_val"][i]) / max(1.0, abs(oi["obj_val"])))
            worst_x = max(worst_x, float(np.max(np.abs(oi["x"] - res.x[i]))))
            same_iters += int(oi["iterations"] == res.iterations[i])
        row.update({"checked": min(args.check, args.batch), "max_rel_obj_diff_vs_oracle": worst_obj,
                    "max_abs_x_diff_vs_oracle": worst_x, "same_iteration_count": same_iters,
                    "oracle_sec_per_problem": (time.time() - t0) / min(args.check, args.batch)})
    rows.append(row)
    print(json.dumps(row))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "c5_batch.json"), "w"), indent=1)
