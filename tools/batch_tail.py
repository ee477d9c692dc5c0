"""Which instances make the tail of a batch launch: iteration histogram of a localization batch, and the
interior-point log (host build, print_level 5) of the slowest ones.  python tools/batch_tail.py [batch] [nlogs]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import batch_problems as bp  # noqa: E402
from dnlp_amd.batch import ParametricBatch, arrays_with_data  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
nlogs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
which = sys.argv[3] if len(sys.argv) > 3 else "localization"          # localization | circle_packing10 | power_flow | path_planning
OPTS = dict(a.split("=", 1) for a in sys.argv[4:])                       # solver options: ladder_dual_blowup_tol=1e12 ...
prob, params, sample, var = bp.template_circle_packing(10) if which == "circle_packing10" else getattr(bp, "template_" + which)()
pb = ParametricBatch(prob, params)
thetas = np.stack([sample(i) for i in range(B)])
mat = pb.data(thetas)
res = pb.solve(thetas, **OPTS)
it = res.iterations
wall = res.raw["phase_seconds"][:, 0]
print("kernel_sec", res.kernel_seconds, "iters mean", it.mean(), "max", it.max())
edges = [0, 20, 30, 40, 60, 80, 120, 160, 1000]
h, _ = np.histogram(it, edges)
print("iteration histogram", dict(zip(["<%d" % e for e in edges[1:]], h.tolist())))
print("instance wall ms: mean %.3f, p99 %.3f, max %.3f" % (1e3 * wall.mean(), 1e3 * np.quantile(wall, 0.99), 1e3 * wall.max()))
order = np.argsort(-it)
print("slowest", [(int(i), int(it[i]), round(1e3 * wall[i], 2), int(res.factorizations[i]), int(res.status[i])) for i in order[:12]])
print("statuses", {int(k): int(v) for k, v in zip(*np.unique(res.status, return_counts=True))})
ph = res.raw["phase_seconds"]
tot_it = float(it.sum())
print("per iteration (mean over all instances, device clock): wall %.3f ms = tape %.3f + factorisation %.3f + solves %.3f + rest %.3f" % (
    1e3 * ph[:, 0].sum() / tot_it, 1e3 * ph[:, 1].sum() / tot_it, 1e3 * ph[:, 2].sum() / tot_it, 1e3 * ph[:, 3].sum() / tot_it,
    1e3 * (ph[:, 0] - ph[:, 1] - ph[:, 2] - ph[:, 3]).sum() / tot_it))
print("factorisations per iteration %.2f" % (float(res.factorizations.sum()) / tot_it))
print("sum of instance walls %.3f s over %d instances; slowest instance %.3f s" % (wall.sum(), B, wall.max()))
from oracle_check import OracleProblem  # noqa: E402
from dnlp_amd.nlp_solver import HIPNLP  # noqa: E402
from dnlp_amd.tape import serialize  # noqa: E402
for i in order[:nlogs]:
    arrays = arrays_with_data(pb.arrays0, mat[i])
    orc = OracleProblem(serialize(arrays))
    for k, v in dict(HIPNLP.DEFAULT_OPTIONS, print_level=5).items():
        orc.set_option(k, v)
    info = orc.solve(arrays["x0"])
    print("=== instance", int(i), "oracle iterations", info["iterations"], "status", info["status"])
    log = orc.log().splitlines()
    full = os.environ.get("DNLP_TAIL_FULL_LOG")
    if full:
        open(full, "w").write("\n".join(log))
    print("\n".join(log[:12]))
    print("   ...")
    print("\n".join(log[-40:]))
