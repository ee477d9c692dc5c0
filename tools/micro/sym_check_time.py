import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from dnlp_amd import lowering
n = 10000
rng = np.random.default_rng(0)
G = rng.standard_normal((n, n)); Q = G + G.T
for flag in ("1", "0", "1", "0"):
    os.environ["DNLP_LOWER_CXX"] = flag
    t = time.time(); ok = lowering._is_symmetric(Q); print("cxx" if flag == "1" else "numpy", ok, "%.3f s" % (time.time() - t))
