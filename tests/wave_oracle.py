"""TEST INFRASTRUCTURE — the wavefront batch solver's algorithm text (dnlp_amd/csrc/wave_ipm.h) on ONE host lane, next to
the generic algorithm text (ipm_core.h over the host space) set up the way the generic batch kernel sets it up: both
through oracle/oracle_lib.cpp `orc_wave_solve_batch`.  Only tests/ may import this."""
import ctypes as C

import numpy as np

from dnlp_amd.nlp_solver import HIPNLP
from dnlp_amd.tape import serialize
from oracle.oracle_capi import OracleProblem, api

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class HostBatch:
    """The template of a ParametricBatch in the test oracle."""

    def __init__(self, pb, opts=None):
        self.pb = pb
        self.handle = OracleProblem(serialize(pb.arrays0))
        o = dict(HIPNLP.DEFAULT_OPTIONS)
        o.update(opts or {})
        for k, v in o.items():
            self.handle.set_option(k, v)
        self.N, self.m = int(pb.arrays0["dims"][0]), int(pb.arrays0["dims"][1])
        lib = api().lib
        lib.orc_wave_solve_batch.restype = C.c_int
        lib.orc_wave_solve_batch.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int64, C.c_int, _dp, _dp, _ip, _ip, _ip, _dp, _dp, _dp]
        lib.orc_last_error.restype = C.c_char_p
        self.lib = lib

    def solve(self, thetas, which):
        """which = 0: wave_ipm.h on one host lane; 1: ipm_core.h (generic) with the template's plan."""
        return self.solve_rows(self.pb.data(np.atleast_2d(thetas)), which)

    def solve_rows(self, mat, which):
        """The same for instance data rows (BATCH_DATA_KEYS order)."""
        mat = np.ascontiguousarray(mat)
        B = mat.shape[0]
        out = {"x": np.zeros((B, self.N)), "obj": np.zeros(B), "status": np.zeros(B, np.int32), "iters": np.zeros(B, np.int32),
               "nfact": np.zeros(B, np.int32), "mult_g": np.zeros((B, max(self.m, 1))), "zl": np.zeros((B, self.N)), "zu": np.zeros((B, self.N))}
        rc = self.lib.orc_wave_solve_batch(self.handle.ptr, B, mat.ctypes.data_as(_dp), mat.shape[1], which, out["x"].ctypes.data_as(_dp),
                                           out["obj"].ctypes.data_as(_dp), out["status"].ctypes.data_as(_ip), out["iters"].ctypes.data_as(_ip),
                                           out["nfact"].ctypes.data_as(_ip), out["mult_g"].ctypes.data_as(_dp), out["zl"].ctypes.data_as(_dp),
                                           out["zu"].ctypes.data_as(_dp))
        if rc != 0:
            raise RuntimeError("orc_wave_solve_batch: %d %s" % (rc, self.lib.orc_last_error().decode()))
        return out
