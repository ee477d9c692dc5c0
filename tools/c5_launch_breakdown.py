"""Where the wall time of ONE batch launch goes (C5): `python tools/c5_launch_breakdown.py [template] [batch]`.
Prints the library's own marks (DNLP_BATCH_DEBUG: slab, plan + buffers, kernel, results) and the Python-side total."""
import os, sys, time
import numpy as np
os.environ["DNLP_BATCH_DEBUG"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from dnlp_amd.batch import ParametricBatch

which = sys.argv[1] if len(sys.argv) > 1 else "localization"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
prob, params, sample, _ = bench.c5_template(which)
pb = ParametricBatch(prob, params)
sets = [np.stack([sample(k * B + i) for i in range(B)]) for k in range(4)]
pb.solve(sets[0], device=0)
for k in range(1, 4):
    sys.stderr.write("--- launch %d\n" % k)
    t0 = time.time()
    r = pb.solve(sets[k], device=0)
    t1 = time.time()
    sys.stderr.write("python total %.4f s   kernel %.4f s   slowest instance %d iterations, mean %.1f\n"
                     % (t1 - t0, r.kernel_seconds, int(r.iterations.max()), float(r.iterations.mean())))
