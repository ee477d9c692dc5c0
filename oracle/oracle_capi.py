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
