"""TEST INFRASTRUCTURE — the per-template straight-line LDL^T phases (dnlp_amd/csrc/wave_gen.h + wave_gen_rt.h) on the HOST:
the oracle prints a translation unit for one template (oracle_lib.cpp orc_wave_gen_host_source: wave_ipm.h compiled with
-DDNLP_WAVE_SPEC -DDNLP_WAVE_GEN, the generated phases playing 64 lanes one after the other), g++ builds it with the
oracle's own flags, and `solve` runs instances through it.  Only tests/ may import this."""
import ctypes as C
import hashlib
import os
import subprocess
import tempfile

import numpy as np

from wave_oracle import HostBatch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dnlp_amd", "csrc")
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class GenHostBatch(HostBatch):
    def __init__(self, pb, opts=None, build_dir=None, lanes=64):
        super().__init__(pb, opts)
        f = self.lib.orc_wave_gen_host_source
        f.restype = C.c_longlong
        f.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_longlong]
        n = f(self.handle.ptr, lanes, None, 0)
        if n <= 0:
            raise RuntimeError("orc_wave_gen_host_source: %d %s" % (n, self.lib.orc_last_error().decode()))
        buf = C.create_string_buffer(n + 16)
        f(self.handle.ptr, lanes, buf, len(buf))
        self.source = buf.value.decode()
        d = build_dir or os.path.join(tempfile.gettempdir(), "dnlp_wave_gen_host-%d" % os.getuid())
        os.makedirs(d, exist_ok=True)
        # (the unit includes csrc headers by name: their text is part of what is built, so it is part of the tag)
        h = hashlib.sha1(self.source.encode())
        for name in sorted(os.listdir(CSRC)):
            if name.endswith(".h"):
                with open(os.path.join(CSRC, name), "rb") as fh:
                    h.update(name.encode() + b"\0" + fh.read())
        tag = h.hexdigest()[:16]
        so = os.path.join(d, "wgen_%s.so" % tag)
        if not os.path.exists(so):
            cpp = os.path.join(d, "wgen_%s.cpp" % tag)
            with open(cpp, "w") as fh:
                fh.write(self.source)
            # the oracle's own flags (oracle/Makefile): the comparison is bit for bit
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-I", CSRC, cpp, "-o", so + ".tmp"])
            os.replace(so + ".tmp", so)
        self.gen = C.CDLL(so)
        self.gen.wgen_host_solve.restype = C.c_int
        self.gen.wgen_host_solve.argtypes = [C.c_int, _dp, C.c_longlong, C.c_void_p, C.c_longlong, C.c_int, _dp, _dp, _ip, _ip, _ip, _dp, _dp, _dp]
        e = self.lib.orc_wave_expand_rows
        e.restype = C.c_longlong
        e.argtypes = [C.c_void_p, C.c_int, _dp, C.c_int64, _dp, C.c_void_p, C.c_longlong, _ip]
        self._expand = e

    def solve_gen(self, thetas):
        """The generated phases inside the interior-point loop of wave_ipm.h, on the host."""
        mat = np.ascontiguousarray(self.pb.data(np.atleast_2d(thetas)))
        B = mat.shape[0]
        opt = C.create_string_buffer(1024)
        fb = C.c_int(0)
        width = self._expand(self.handle.ptr, B, mat.ctypes.data_as(_dp), mat.shape[1], None, opt, len(opt), C.byref(fb))
        assert width > 0, self.lib.orc_last_error()
        rows = np.zeros((B, width))
        rc = self._expand(self.handle.ptr, B, mat.ctypes.data_as(_dp), mat.shape[1], rows.ctypes.data_as(_dp), opt, len(opt), C.byref(fb))
        assert rc == width, (rc, self.lib.orc_last_error())
        out = {"x": np.zeros((B, self.N)), "obj": np.zeros(B), "status": np.zeros(B, np.int32), "iters": np.zeros(B, np.int32),
               "nfact": np.zeros(B, np.int32), "mult_g": np.zeros((B, max(self.m, 1))), "zl": np.zeros((B, self.N)), "zu": np.zeros((B, self.N))}
        # (sizeof(IpmOptions) is checked on the other side: the struct is the same text in both libraries)
        import dnlp_amd  # noqa: F401
        size = self._opt_size()
        rc = self.gen.wgen_host_solve(B, rows.ctypes.data_as(_dp), width, opt, size, fb.value, out["x"].ctypes.data_as(_dp), out["obj"].ctypes.data_as(_dp),
                                      out["status"].ctypes.data_as(_ip), out["iters"].ctypes.data_as(_ip), out["nfact"].ctypes.data_as(_ip),
                                      out["mult_g"].ctypes.data_as(_dp), out["zl"].ctypes.data_as(_dp), out["zu"].ctypes.data_as(_dp))
        if rc != 0:
            raise RuntimeError("wgen_host_solve: %d" % rc)
        return out

    def _opt_size(self):
        f = self.lib.orc_sizeof_ipm_options
        f.restype = C.c_longlong
        return f()
