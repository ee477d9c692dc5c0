"""GPU probe: times the blocked FP64-MFMA LDL^T and the dense symmetric product at several
orders (run on the MI355X box; writes gpurun_out/probe.json)."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dnlp_amd import _capi  # noqa: E402
from dnlp_amd.device import symmetric_test_matrix  # noqa: E402

api = _capi.require_device(0)
out = {"ldlt": [], "symv": []}
sizes = [int(s) for s in sys.argv[1:]] or [4096, 8192, 16384, 32768]
for n in sizes:
    A = symmetric_test_matrix(n, seed=1, spike_eig=4 * np.sqrt(n))
    x = np.ones(n)
    A.symv(x)
    t = time.time()
    reps = 5
    for _ in range(reps):
        A.symv(x)
    dt = (time.time() - t) / reps
    out["symv"].append({"n": n, "sec": dt, "GBps": 8.0 * n * n / dt / 1e9})
    # make it positive definite-ish: factor A + shift on the diagonal is not available here, so
    # factor the indefinite matrix as is (unpivoted; inertia is reported, timing is what matters)
    nneg, nzero = C.c_int(), C.c_int()
    sec, upd = C.c_double(), C.c_double()
    rc = api.lib.dnlp_ldlt_device(0, C.c_void_p(A.ptr), n, A.ld, C.byref(nneg), C.byref(nzero),
                                  C.byref(sec), C.byref(upd))
    flops = n ** 3 / 3.0
    out["ldlt"].append({"n": n, "rc": rc, "sec": sec.value, "update_sec": upd.value,
                        "TFLOPs": flops / sec.value / 1e12, "nneg": nneg.value, "nzero": nzero.value})
    print(out["symv"][-1], out["ldlt"][-1], flush=True)
    A.free()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "probe.json"), "w"), indent=1)
