"""One line per example of gpurun_out/paper_examples.json (written by tools/run_paper_examples.py)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for r in json.load(open(os.path.join(ROOT, "gpurun_out", "paper_examples.json"))):
    print("  %-26s it %3d nf %3d solve %.4f without timers %.4f factor %.4f" % (
        r["example"], r["iters"], r["factorizations"], r["solve_sec"], r["solve_sec_without_timers"], r["factor_sec"]))
