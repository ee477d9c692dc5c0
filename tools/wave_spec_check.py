"""The per-template batch kernel (csrc/wave_codegen.h: compiled by hiprtc at run time) against the library's own wavefront
kernel (csrc/wave_batch.h) on the device: the same fresh batch through both — bits of the results, kernel time, an
instance's own microseconds per iteration.

    python tools/wave_spec_check.py --which localization,circle_packing --batch 1024,8192 --reps 3
Writes one JSON line per template and batch size (and gpurun_out/wave_spec_check.jsonl)."""
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
ap.add_argument("--batch", default="1024")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--out", default="wave_spec_check.jsonl")
args = ap.parse_args()

TMPL = {"localization": bp.template_localization, "circle_packing": bp.template_circle_packing,
        "circle_packing10": lambda: bp.template_circle_packing(10)}
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
out = open(os.path.join(ROOT, "gpurun_out", args.out), "a")
for which in args.which.split(","):
    prob, params, sample, _ = TMPL[which]()
    pb = ParametricBatch(prob, params)
    for B in [int(b) for b in args.batch.split(",")]:
        res = {}
        for mode in ("spec", "own"):
            os.environ["DNLP_WAVE_SPEC"] = "1" if mode == "spec" else "0"
            best, first, t_first = None, None, None
            for rep in range(args.reps):
                thetas = np.stack([sample(rep * B + i) for i in range(B)])         # a fresh batch every repetition
                t0 = time.time()
                r = pb.solve(thetas, want_duals=True)
                wall = time.time() - t0
                if rep == 0:
                    first, t_first = r, wall
                if best is None or r.kernel_seconds < best[0]:
                    best = (r.kernel_seconds, wall, r)
            res[mode] = (first, best, t_first)
        s0, sb, st = res["spec"]
        o0, ob, ot = res["own"]
        bits = all(np.array_equal(getattr(s0, k), getattr(o0, k)) for k in ("status", "iterations", "obj_val", "x")) and \
            np.array_equal(s0.raw["mult_g"], o0.raw["mult_g"])
        row = {"problem": which, "batch": B, "spec_launch": s0.raw.get("launch"), "own_launch": o0.raw.get("launch"),
               "first_call_s_spec": st, "first_call_s_own": ot,
               "spec_kernel_ms_best": 1e3 * sb[0], "own_kernel_ms_best": 1e3 * ob[0],
               "spec_problems_per_s_kernel": B / sb[0], "own_problems_per_s_kernel": B / ob[0],
               "spec_problems_per_s_wall": B / sb[1], "own_problems_per_s_wall": B / ob[1],
               "spec_instance_us_per_iter": float(1e6 * sb[2].raw["phase_seconds"][:, 0].sum() / sb[2].iterations.sum()),
               "own_instance_us_per_iter": float(1e6 * ob[2].raw["phase_seconds"][:, 0].sum() / ob[2].iterations.sum()),
               "iters_mean": float(s0.iterations.mean()), "iters_max": int(s0.iterations.max()),
               "identical_bits": bool(bits), "same_status": int(np.sum(s0.status == o0.status)),
               "same_iterations": int(np.sum(s0.iterations == o0.iterations)), "optimal_spec": int(np.sum(s0.status == 0)),
               "optimal_own": int(np.sum(o0.status == 0)), "speedup_kernel": ob[0] / sb[0]}
        print(json.dumps(row), flush=True)
        out.write(json.dumps(row) + "\n")
        out.flush()
    pb.close()
