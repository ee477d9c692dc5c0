"""TEST INFRASTRUCTURE — ctypes access to oracle/_build/liboracle_dnlp.so (orc_* symbols).

Host instantiation of the solver core (see host_exec.h).  Only tests/, smoke() and bench.py's
cpu_baseline leg may import this.  `build()` compiles the library with g++.
"""
import os
import subprocess

from dnlp_amd._capi import CApi, ProblemHandle

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "liboracle_dnlp.so")
_api = None


def build(force=False):
    if force or not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return LIB


def api() -> CApi:
    global _api
    if _api is None:
        build()
        _api = CApi(LIB, "orc_")
    return _api


class OracleProblem(ProblemHandle):
    def __init__(self, blob: bytes):
        super().__init__(api(), blob, 0)


def use_lapack(threads: int = 0) -> int:
    """CPU baseline only: route the pivoted dense KKT factorisation of the host build through LAPACK
    DSYTRF / DSYTRS of the OpenBLAS inside this image's scipy wheel.  Returns the BLAS thread count
    (0: no such library here — the restated DSYTF2 stays in use)."""
    import ctypes
    import glob
    import scipy
    lib = api().lib
    lib.orc_use_lapack.restype = ctypes.c_int
    lib.orc_use_lapack.argtypes = [ctypes.c_char_p, ctypes.c_int]
    base = os.path.join(os.path.dirname(os.path.dirname(scipy.__file__)), "scipy.libs")
    for path in sorted(glob.glob(os.path.join(base, "libscipy_openblas*.so"))):
        n = lib.orc_use_lapack(path.encode(), int(threads))
        if n > 0:
            return n
    return 0


def lapack_best_threads(candidates, n=3000):
    """Thread count among `candidates` with the fastest DSYTRF of order n on this host (after
    use_lapack); leaves the BLAS set to it.  Returns (threads, {threads: seconds})."""
    import ctypes
    lib = api().lib
    lib.orc_lapack_probe.restype = ctypes.c_double
    lib.orc_lapack_probe.argtypes = [ctypes.c_int, ctypes.c_int]
    times = {}
    for t in candidates:
        lib.orc_lapack_probe(n, t)                       # warm the threads
        times[t] = lib.orc_lapack_probe(n, t)
    best = min(times, key=times.get)
    lib.orc_lapack_probe(64, best)
    return best, times


def no_lapack():
    import ctypes
    lib = api().lib
    lib.orc_use_lapack.restype = ctypes.c_int
    lib.orc_use_lapack.argtypes = [ctypes.c_char_p, ctypes.c_int]
    lib.orc_use_lapack(None, 0)


def use_blocked_ldlt(on: bool = True) -> bool:
    """CPU baseline only: unpivoted dense factorisations of the host build go through the blocked LDL^T whose
    trailing update is DGEMM of the library loaded by use_lapack (all its threads)."""
    import ctypes
    lib = api().lib
    lib.orc_use_blocked_ldlt.restype = ctypes.c_int
    lib.orc_use_blocked_ldlt.argtypes = [ctypes.c_int]
    return bool(lib.orc_use_blocked_ldlt(1 if on else 0))


def set_blas_threads(threads: int) -> int:
    import ctypes
    lib = api().lib
    lib.orc_lapack_probe.restype = ctypes.c_double
    lib.orc_lapack_probe.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.orc_lapack_probe(8, int(threads))
    return int(threads)


def dgemm_gflops(n: int, threads: int) -> float:
    import ctypes
    lib = api().lib
    lib.orc_dgemm_probe.restype = ctypes.c_double
    lib.orc_dgemm_probe.argtypes = [ctypes.c_int, ctypes.c_int]
    return float(lib.orc_dgemm_probe(int(n), int(threads)))


def blocked_ldlt_phases():
    """Seconds of the last blocked LDL^T's phases (needs DNLP_HOST_LDLT_TIMING in the environment)."""
    import ctypes
    lib = api().lib
    out = (ctypes.c_double * 4)()
    lib.orc_blocked_ldlt_phases(out)
    return {"diagonal_blocks": out[0], "dtrsm": out[1], "w_copy": out[2], "dgemm": out[3]}
