"""Does the IPOPT log printed in the reference's examples/nlp_examples/localization.ipynb belong to the data its committed
cells generate (np.random.seed(0); a ~ U(-5, 5), v ~ N(0, 1))?  tests/paper_examples.py pins only the log's dimensions and
says why; this script shows the evidence with numbers (CPU only: the front-end + the host build of the solver).

The log (localization.ipynb, output of the solve cell) says, for its data:
    iteration 0:  objective 1.0000000e+01,  inf_pr 7.81e+00          final: 18 iterations, objective 7.6602709641776840
and the next cell prints  x.value = [ 2.11285122 -1.6415691 ].
For the committed data this script prints the same quantities: the objective and the constraint violation at the
canonical problem's start point (what IPOPT reports at iteration 0, before any step), the optimum of the least-squares
problem and its minimiser.  Both runs solve  min sum (t - rho)^2, t = |x - a_i|:  if the minimiser agrees with the printed
one and the iteration-0 numbers do not, the log was produced with the same anchors and ranges but another start; if
neither agrees, with other data."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dnlp_amd as cp  # noqa: E402
from batch_problems import oracle_solver  # noqa: E402
from dnlp_amd.dnlp2smooth import Dnlp2Smooth  # noqa: E402
from dnlp_amd.nlp_solver import build_nlp_data  # noqa: E402
from paper_examples import nb_localization  # noqa: E402

LOG = {"obj0": 1.0000000e+01, "inf_pr0": 7.81, "objective": 7.6602709641776840, "iters": 18, "x": np.array([2.11285122, -1.6415691])}

prob = nb_localization(cp)
smooth, _ = Dnlp2Smooth().apply(prob)
data, inv = build_nlp_data(smooth)
x0 = np.asarray(data["x0"], float)
# iteration 0 of IPOPT = the start point pushed into the bounds (bound_push 1e-2): objective and max |c(x)|
from oracle.oracle_capi import OracleProblem  # noqa: E402
from dnlp_amd.tape import serialize  # noqa: E402
orc = OracleProblem(serialize(data["tape_arrays"]))
lb, ub = np.asarray(data["lb"], float), np.asarray(data["ub"], float)
xs = x0.copy()
lo = np.isfinite(lb)
xs[lo] = np.maximum(xs[lo], lb[lo] + 1e-2 * np.maximum(1.0, np.abs(lb[lo])))
f0 = orc.eval_f(xs)
g0 = orc.eval_g(xs)
cl = np.asarray(data["cl"], float)
print("committed data, canonical start point: objective %.7e   max |c| %.3e   (the log: %.7e, %.2e)" % (f0, np.max(np.abs(g0 - cl)), LOG["obj0"], LOG["inf_pr0"]))
obj, status, iters, x = oracle_solver(prob)
xv = [v for v in prob.variables() if v.name() == "x"][0]
off = inv.var_offsets[xv.id]
print("committed data, solved: status %d, %d iterations, objective %.16e, x = %s" % (status, iters, obj, x[off:off + 2]))
print("the log:                           %d iterations, objective %.16e, x = %s" % (LOG["iters"], LOG["objective"], LOG["x"]))
# the least-squares objective of the committed data AT the minimiser the notebook printed
np.random.seed(0)
a = np.random.uniform(-5, 5, (10, 2))
v = np.random.normal(0, 1, 10)
rho = np.linalg.norm(a - np.array([2.0, -1.5]), axis=1) + v
def ls(p):
    return float(np.sum((np.linalg.norm(a - p, axis=1) - rho) ** 2))
print("committed data: sum (|x - a_i| - rho_i)^2 at the log's minimiser %.10f, at this solve's minimiser %.10f" % (ls(LOG["x"]), ls(x[off:off + 2])))
same_x = np.max(np.abs(x[off:off + 2] - LOG["x"])) <= 1e-6
print("verdict: minimiser %s the printed one; objective %s the log's"
      % ("IS" if same_x else "is NOT", "matches" if abs(obj - LOG["objective"]) <= 1e-6 * LOG["objective"] else "does not match"))
