"""HBM-resident dense matrices for constants that never travel through the host
(BASELINE config C4: the n = 1e5 quad_form matrix is 80 GB)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi
from .expressions import DeviceMatrix


class DeviceDense:
    """Owner of a column-major FP64 n x n matrix in HBM."""

    def __init__(self, n: int, device: int = None):
        self.device = _capi.current_device() if device is None else device
        self.api = _capi.require_device(self.device)
        self.n = int(n)
        self.ld = (self.n + 7) // 8 * 8
        ptr = C.c_void_p()
        rc = self.api.lib.dnlp_dev_alloc(self.device, C.c_size_t(self.ld * self.n * 8 + 2048), C.byref(ptr))   # +2 KiB slack (tile over-reads)
        if rc != 0:
            raise MemoryError("dnlp_dev_alloc failed: %s" % self.api.error())
        self.ptr = ptr.value
        self.handle = DeviceMatrix(self.ptr, self.n, self.n, self.ld, owner=self, symmetric=True)

    def free(self):
        if self.ptr:
            self.api.lib.dnlp_dev_free(self.device, C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def to_host(self) -> np.ndarray:
        buf = np.empty((self.n, self.ld), dtype=np.float64)   # row of buf = column of A
        rc = self.api.lib.dnlp_dev_copy(self.device, buf.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr),
                                        C.c_size_t(buf.nbytes), 1)
        if rc != 0:
            raise RuntimeError(self.api.error())
        return np.ascontiguousarray(buf[:, :self.n].T)

    def symv(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty(self.n)
        dp = C.POINTER(C.c_double)
        rc = self.api.lib.dnlp_dev_symv(self.device, C.c_void_p(self.ptr), self.n, self.ld,
                                        x.ctypes.data_as(dp), y.ctypes.data_as(dp))
        if rc != 0:
            raise RuntimeError(self.api.error())
        return y


def symmetric_test_matrix(n: int, seed: int = 0, spike_eig: float = None, device: int = None) -> DeviceDense:
    """Seeded dense symmetric matrix generated on the device:
    A[i,j] = u(min(i,j), max(i,j)) + s * v_i v_j, u ~ U[-1,1) counter-based, v ~ U[-1,1)^n.
    `spike_eig` is the eigenvalue the rank-one term contributes (s = spike_eig / ||v||^2 with
    ||v||^2 ~ n/3), which separates lambda_max from the noise bulk (radius ~ 2 sqrt(n/3))."""
    A = DeviceDense(n, device)
    if spike_eig is None:
        spike_eig = 4.0 * np.sqrt(n)
    s = spike_eig / (n / 3.0)
    rc = A.api.lib.dnlp_gen_symmetric(A.device, C.c_void_p(A.ptr), n, A.ld, C.c_uint64(seed), C.c_double(s), None)
    if rc != 0:
        raise RuntimeError(A.api.error())
    return A
