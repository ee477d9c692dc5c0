"""Time the blocked unpivoted LDL^T (csrc/ldlt_blocked.h) alone on KKT-shaped quasi-definite matrices:
python tools/time_ldlt.py n1 m [reps].  Checks the inertia and the solve residual, prints TFLOP/s of
(n1+m)^3/3."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dnlp_amd import _capi  # noqa: E402

n1, m = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 1000)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = n1 + m
rng = np.random.default_rng(0)
A = np.zeros((n, n), order="F")
R = rng.uniform(-1, 1, (n1, n1))
A[:n1, :n1] = R + R.T
A[np.arange(n1), np.arange(n1)] += 2.0 * n1 + 1.0              # diagonally dominant: positive definite
J = rng.standard_normal((m, n1))
A[n1:, :n1] = J
A[:n1, n1:] = J.T
A[np.arange(n1, n), np.arange(n1, n)] = -1e-2
b = rng.standard_normal(n)
api = _capi.require_device(0)
dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
best = None
for rep in range(reps):
    Af = np.array(A, order="F")
    ipiv = np.zeros(n, np.int32)
    nneg, nzero, sec = C.c_int(), C.c_int(), C.c_double()
    sol = np.zeros(n)
    rc = api.lib.dnlp_ldlt_host(0, dp(Af), n, n, ipiv.ctypes.data_as(C.POINTER(C.c_int32)), 0, C.byref(nneg), C.byref(nzero),
                                dp(b), dp(sol), C.byref(sec))
    assert rc == 0, api.error()
    res = float(np.linalg.norm(A @ sol - b) / np.linalg.norm(b))
    rec = {"n": n, "seconds": sec.value, "TFLOPs": n ** 3 / 3.0 / sec.value / 1e12, "nneg": nneg.value, "nzero": nzero.value,
           "rel_residual": res}
    assert nneg.value == m and nzero.value == 0 and res < 1e-8, rec
    if best is None or rec["seconds"] < best["seconds"]:
        best = rec
print(json.dumps(best))
