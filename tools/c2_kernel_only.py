"""The generated fused f + grad f kernel of the Rosenbrock chain alone (for rocprofv3 --pmc passes):
python tools/c2_kernel_only.py n reps"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dnlp_amd as cp  # noqa: E402
from dnlp_amd import _capi  # noqa: E402
from dnlp_amd.tape import serialize  # noqa: E402
from problem_zoo import rosenbrock_chain  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
p = rosenbrock_chain(cp, 100000)
chain = p._build_chain(None)
data, _ = chain.apply(p)
data["handle"].close()
a = dict(data["tape_arrays"])
a["fz_prog_nelem"] = np.array([n - 1], dtype=np.int64)
dims = a["fz_dims"].copy()
dims[3] = n
a["fz_dims"] = dims
h = _capi.DeviceProblem(serialize(a), data["tape"])
x = np.random.default_rng(0).uniform(0.5, 1.5, n)
print("seconds per evaluation:", h.time_fused(x, reps))
