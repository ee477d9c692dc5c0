import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import batch_problems as bp
from dnlp_amd.batch import ParametricBatch
prob, params, sample, _ = bp.template_power_flow()
pb = ParametricBatch(prob, params)
for rep in range(4):
    thetas = np.stack([sample(rep * 1024 + i) for i in range(1024)])
    row = []
    for mode in ("1", "0"):
        os.environ["DNLP_WAVE_SPEC"] = mode
        r = pb.solve(thetas, least_square_init_duals="no") if len(sys.argv) > 1 else pb.solve(thetas)
        row.append({int(k): int(v) for k, v in zip(*np.unique(r.status, return_counts=True))})
        row.append(float(r.iterations.mean()))
    print(rep, "wg", row[0], "%.2f" % row[1], "generic", row[2], "%.2f" % row[3], flush=True)
