"""ctypes binding of libdnlp_hip.so (C ABI: include/dnlp_hip.h).

The library is built in-tree by `__graft_entry__.build()` (hipcc --offload-arch=gfx950).  There
is no CPU fallback: `load()` raises DeviceUnavailableError when the shared library is missing
or no HIP device is visible, and every solve goes through it.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .error import DeviceUnavailableError

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DNLP_HIP_LIB", os.path.join(_HERE, "libdnlp_hip.so"))    # (override: kernel-variant experiments)

_dbl_p = C.POINTER(C.c_double)
_i32_p = C.POINTER(C.c_int32)


def _dp(a):
    return None if a is None else a.ctypes.data_as(_dbl_p)


def _ip(a):
    return None if a is None else a.ctypes.data_as(_i32_p)


# int cb(alg_mod, iter_count, obj_value, inf_pr, inf_du, mu, d_norm, regularization_size, alpha_du,
#        alpha_pr, ls_trials, user_data)  -- include/dnlp_hip.h dnlp_intermediate_cb
INTERMEDIATE_CB = C.CFUNCTYPE(C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                              C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_void_p)

N_STATS = 24


class CApi:
    """Typed access to one shared library exporting the `<prefix>*` entry points."""

    def __init__(self, lib_path: str, prefix: str = "dnlp_"):
        self.lib = C.CDLL(lib_path)
        self.prefix = prefix
        f = self._fn
        f("last_error", C.c_char_p, [])
        f("create", C.c_void_p, [C.c_void_p, C.c_size_t, C.c_int])
        f("create_arrays", C.c_void_p, [C.c_void_p, C.c_int, C.c_int])
        f("destroy", None, [C.c_void_p])
        f("bind_dense", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64])
        f("dims", C.c_int, [C.c_void_p] + [C.POINTER(C.c_int64)] * 4)
        f("bounds", C.c_int, [C.c_void_p] + [_dbl_p] * 5)
        f("eval_f", C.c_int, [C.c_void_p, _dbl_p, C.c_int, _dbl_p])
        f("eval_grad_f", C.c_int, [C.c_void_p, _dbl_p, C.c_int, _dbl_p])
        f("eval_g", C.c_int, [C.c_void_p, _dbl_p, C.c_int, _dbl_p])
        f("eval_jac_g", C.c_int, [C.c_void_p, _dbl_p, C.c_int, _i32_p, _i32_p, _dbl_p])
        f("eval_h", C.c_int, [C.c_void_p, _dbl_p, C.c_int, C.c_double, _dbl_p, C.c_int, _i32_p,
                              _i32_p, _dbl_p])
        f("set_option", C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p])
        f("reset_options", C.c_int, [C.c_void_p])
        f("solve", C.c_int, [C.c_void_p] + [_dbl_p] * 6 + [C.POINTER(C.c_int)])
        f("solve_reduced", C.c_int, [C.c_void_p, _dbl_p, _dbl_p, C.POINTER(C.c_int), C.POINTER(C.c_int), _dbl_p])
        f("ipm_begin", C.c_int, [C.c_void_p, _dbl_p])
        f("ipm_step", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int)])
        f("ipm_finish", C.c_int, [C.c_void_p] + [_dbl_p] * 6 + [C.POINTER(C.c_int)])
        f("eval_fused", C.c_int, [C.c_void_p, _dbl_p, _dbl_p, _dbl_p])
        f("set_warm_start", C.c_int, [C.c_void_p, _dbl_p, _dbl_p, _dbl_p])
        f("kkt_info", C.c_int, [C.c_void_p, C.POINTER(C.c_int64)])
        f("kkt_mode", C.c_int, [C.c_void_p])
        f("kkt_tail_nodes", C.c_int64, [C.c_void_p])
        f("get_stats", C.c_int, [C.c_void_p, _dbl_p, C.c_int])
        f("get_log", C.c_size_t, [C.c_void_p, C.c_char_p, C.c_size_t])
        f("set_intermediate_cb", C.c_int, [C.c_void_p, INTERMEDIATE_CB, C.c_void_p])
        f("reduced_info", C.c_int, [C.c_void_p, _dbl_p, C.c_int])
        if hasattr(self.lib, prefix + "time_fused"):
            f("time_fused", C.c_int, [C.c_void_p, _dbl_p, C.c_int, _dbl_p])
        if hasattr(self.lib, prefix + "solve_batch_timed"):      # product library only (no oracle batch path)
            _int_p = C.POINTER(C.c_int)
            f("batch_stride", C.c_int64, [C.c_void_p])
            f("batch_warm_start", C.c_int, [C.c_void_p, C.c_int, _dbl_p, _dbl_p, _dbl_p])
            f("solve_batch_timed", C.c_int, [C.c_void_p, C.c_int, _dbl_p, C.c_int64] + [_dbl_p] * 5 +
              [_int_p] * 3 + [_dbl_p, _dbl_p])
            f("batch_set_affine_map", C.c_int, [C.c_void_p, C.c_int, _dbl_p, _dbl_p, C.POINTER(C.c_int64), _i32_p, _dbl_p])
            f("solve_batch_theta", C.c_int, [C.c_void_p, C.c_int, _dbl_p, C.c_int] + [_dbl_p] * 5 +
              [_int_p] * 3 + [_dbl_p, _dbl_p])
            if hasattr(self.lib, prefix + "batch_launch_info"):
                f("batch_launch_info", C.c_int, [C.c_void_p, _i32_p])
            if hasattr(self.lib, prefix + "batch_result_rows"):
                f("batch_keep_result_rows", C.c_int, [C.c_void_p, C.c_int])
                f("batch_result_rows", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)])
            if hasattr(self.lib, prefix + "batch_stream_create"):
                f("batch_stream_create", C.c_void_p, [C.c_void_p, C.c_int])
                f("batch_stream_submit", C.c_int, [C.c_void_p, C.c_int, _dbl_p, C.c_int] + [_dbl_p] * 5 + [_int_p] * 3)
                f("batch_stream_wait", C.c_int, [C.c_void_p, C.c_int, _dbl_p])
                f("batch_stream_destroy", None, [C.c_void_p])

    def _fn(self, name, restype, argtypes):
        fn = getattr(self.lib, self.prefix + name)
        fn.restype = restype
        fn.argtypes = argtypes
        setattr(self, name, fn)
        return fn

    def error(self) -> str:
        msg = self.last_error()
        return msg.decode() if msg else ""


class CsrArrays:
    """A CSR matrix as its three raw arrays — int64 row pointers, int32 columns, float64 values: the layout of the
    tape (tape.py) and of the C ABI's views (dnlp_lf_view, dnlp_lowered_csr_view), so a 1e7-entry constraint block
    travels from the lowering to the device upload without a copy.  `tocsr()` builds the scipy object on demand."""
    __slots__ = ("indptr", "indices", "data", "shape")

    def __init__(self, indptr, indices, data, shape):
        self.indptr, self.indices, self.data, self.shape = indptr, indices, data, (int(shape[0]), int(shape[1]))

    @property
    def nnz(self):
        return int(self.indices.size)

    def tocsr(self):
        import scipy.sparse as sp
        M = sp.csr_matrix((self.data, self.indices, self.indptr), shape=self.shape)
        M.has_sorted_indices = True
        return M

    def __matmul__(self, other):
        return self.tocsr() @ other


class _Owner:
    """Keeps a C-ABI handle alive for as long as a numpy view of its arrays exists."""
    __slots__ = ("h", "free")

    def __init__(self, h, free):
        self.h, self.free = h, free

    def __del__(self):
        try:
            if self.h:
                self.free(self.h)
        except Exception:
            pass


def view_array(addr, n, ctype, dtype, owner):
    """numpy array over `n` elements at `addr` (memory of a C-ABI handle); `owner` is released with the last view."""
    if not n or not addr:
        return np.zeros(0, dtype)
    buf = (ctype * int(n)).from_address(addr)
    buf._owner = owner
    return np.frombuffer(buf, dtype=dtype)


def lower_maps(N, Z, m, G, c, drow, dcol, hrow, hcol, blocks):
    """C++ construction of the constant maps of a lowered problem (include/dnlp_hip.h: dnlp_lower_maps; the role of
    cvxcore's build_matrix).  `G`: scipy CSR m x (N + Z).  Returns a dict of numpy arrays / scipy matrices, or None
    when the library (or the entry point) is not there — the caller then runs its numpy construction."""
    try:
        lib = load().lib
        fn = lib.dnlp_lower_maps
    except (DeviceUnavailableError, OSError, AttributeError):
        return None
    i64p, i32p = C.POINTER(C.c_int64), C.POINTER(C.c_int32)
    fn.restype = C.c_void_p
    fn.argtypes = [C.c_int64] * 5 + [i64p, i32p, _dbl_p, _dbl_p, i64p, i64p, i64p, i64p, C.c_int, i64p, i64p]
    lib.dnlp_lowered_free.argtypes = [C.c_void_p]
    lib.dnlp_lowered_sizes.argtypes = [C.c_void_p, i64p]
    lib.dnlp_lowered_csr_view.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
    lib.dnlp_lowered_pattern.argtypes = [C.c_void_p, C.c_int, i32p, i32p, _dbl_p]
    lib.dnlp_lowered_block.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), i64p, i64p]
    c64 = lambda a: np.ascontiguousarray(a, dtype=np.int64)     # noqa: E731
    Gp, Gi, Gv = c64(G.indptr), np.ascontiguousarray(G.indices, dtype=np.int32), np.ascontiguousarray(G.data, dtype=np.float64)
    cc = np.ascontiguousarray(c, dtype=np.float64)
    drow, dcol, hrow, hcol = c64(drow), c64(dcol), c64(hrow), c64(hcol)
    bx0, bn = c64([b[0] for b in blocks]), c64([b[1] for b in blocks])
    p64 = lambda a: a.ctypes.data_as(i64p)                      # noqa: E731
    p32 = lambda a: a.ctypes.data_as(i32p)                      # noqa: E731
    h = fn(N, Z, m, drow.size, hrow.size, p64(Gp), p32(Gi), _dp(Gv), _dp(cc), p64(drow), p64(dcol), p64(hrow), p64(hcol),
           len(blocks), p64(bx0), p64(bn))
    if not h:
        raise RuntimeError("dnlp_lower_maps failed: %s" % load().error())
    owner = _Owner(h, lib.dnlp_lowered_free)          # the maps below are views of the handle's arrays
    sz = np.zeros(9, np.int64)
    lib.dnlp_lowered_sizes(h, p64(sz))
    changed, nG, nMg, nMw, nMJ, nnzJ, nMH, nnzH, jac_is_G = (int(v) for v in sz)

    def csr(which, rows, nnz, shape):
        pp, pi, pv = C.c_void_p(), C.c_void_p(), C.c_void_p()
        lib.dnlp_lowered_csr_view(h, which, C.byref(pp), C.byref(pi), C.byref(pv))
        return CsrArrays(view_array(pp.value, rows + 1, C.c_int64, np.int64, owner), view_array(pi.value, nnz, C.c_int32, np.int32, owner),
                         view_array(pv.value, nnz, C.c_double, np.float64, owner), shape)
    nd, nh = int(drow.size), int(hrow.size)
    out = {"G": csr(0, m, nG, (m, N + Z)) if changed else None,
           "Mg": csr(1, N, nMg, (N, nd)), "Mw": csr(2, Z, nMw, (Z, 1 + m)),
           "MJ": csr(3, nnzJ, nMJ, (nnzJ, nd)), "MH": csr(4, nnzH, nMH, (nnzH, nh))}
    if jac_is_G:
        # affine rows over x only (BASELINE C3: 1e7 entries): pattern and values are G's own arrays
        jr = np.repeat(np.arange(m, dtype=np.int32), np.diff(Gp))
        jc, Jc = Gi, Gv
    else:
        jr, jc, Jc = np.zeros(nnzJ, np.int32), np.zeros(nnzJ, np.int32), np.zeros(nnzJ, np.float64)
        lib.dnlp_lowered_pattern(h, 0, p32(jr), p32(jc), _dp(Jc))
    hr, hc = np.zeros(nnzH, np.int32), np.zeros(nnzH, np.int32)
    lib.dnlp_lowered_pattern(h, 1, p32(hr), p32(hc), None)
    out.update({"jac_rows": jr, "jac_cols": jc, "Jc": Jc, "hess_rows": hr, "hess_cols": hc, "blocks": []})
    for b in range(len(blocks)):
        mode, cnt = C.c_int(), C.c_int64()
        lib.dnlp_lowered_block(h, b, C.byref(mode), C.byref(cnt), None)
        pos = np.zeros(cnt.value, np.int64)
        lib.dnlp_lowered_block(h, b, C.byref(mode), C.byref(cnt), p64(pos))
        out["blocks"].append((mode.value, pos))
    return out


_api = None


def load() -> CApi:
    """Load libdnlp_hip.so; fail loudly when it (or a GPU) is not there."""
    global _api
    if _api is not None:
        return _api
    if not os.path.exists(LIB_PATH):
        raise DeviceUnavailableError(
            "%s not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  dnlp_amd has no CPU fallback." % LIB_PATH)
    api = CApi(LIB_PATH, "dnlp_")
    lib = api.lib
    lib.dnlp_device_count.restype = C.c_int
    lib.dnlp_version.restype = C.c_char_p
    lib.dnlp_dev_alloc.restype = C.c_int
    lib.dnlp_dev_alloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.dnlp_dev_free.restype = C.c_int
    lib.dnlp_dev_free.argtypes = [C.c_int, C.c_void_p]
    lib.dnlp_dev_copy.restype = C.c_int
    lib.dnlp_dev_copy.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    lib.dnlp_gen_symmetric.restype = C.c_int
    lib.dnlp_gen_symmetric.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_uint64,
                                       C.c_double, C.c_void_p]
    lib.dnlp_dev_symv.restype = C.c_int
    lib.dnlp_dev_symv.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64, _dbl_p, _dbl_p]
    lib.dnlp_ldlt_host.restype = C.c_int
    lib.dnlp_ldlt_host.argtypes = [C.c_int, _dbl_p, C.c_int64, C.c_int64, _i32_p, C.c_int,
                                   C.POINTER(C.c_int), C.POINTER(C.c_int), _dbl_p, _dbl_p, _dbl_p]
    lib.dnlp_ldlt_device.restype = C.c_int
    lib.dnlp_ldlt_device.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_int64,
                                     C.POINTER(C.c_int), C.POINTER(C.c_int), _dbl_p, _dbl_p]
    lib.dnlp_rtc_compiler.restype = C.c_int
    lib.dnlp_rtc_compiler.argtypes = [C.c_char_p, C.c_size_t]
    _api = api
    return api


def rtc_compiler():
    """(kind, identity) of the compiler the generated kernels of this process go through (include/dnlp_hip.h
    dnlp_rtc_compiler): kind "hiprtc" — in process, whichever libhiprtc the loader bound first: the one PyTorch ships when
    torch was imported before this library — or "clang" (DNLP_RTC_COMPILER=clang: the ROCm install's clang++ as a child
    process); identity = file, size and time: what the kernel cache is keyed by."""
    buf = C.create_string_buffer(2048)
    rc = load().lib.dnlp_rtc_compiler(buf, C.c_size_t(len(buf)))
    return ("clang" if rc == 1 else "hiprtc"), buf.value.decode()


def device_count() -> int:
    try:
        return int(load().lib.dnlp_device_count())
    except (DeviceUnavailableError, OSError):
        return 0


def require_device(device: int = 0) -> CApi:
    api = load()
    n = int(api.lib.dnlp_device_count())
    if n <= device:
        raise DeviceUnavailableError(
            "no HIP device %d visible (found %d); dnlp_amd has no CPU fallback." % (device, n))
    return api


def current_device() -> int:
    """One process per GPU: LOCAL_RANK selects the device (torch.distributed launch)."""
    return int(os.environ.get("DNLP_DEVICE", os.environ.get("LOCAL_RANK", "0")))


class TapeArrayDesc(C.Structure):
    """include/dnlp_hip.h: dnlp_tape_array."""
    _fields_ = [("name", C.c_char_p), ("dtype", C.c_int32), ("reserved", C.c_int32), ("count", C.c_uint64),
                ("data", C.c_void_p)]


class ProblemHandle:
    """A created `<prefix>problem` with numpy-friendly methods (shared by the product binding
    and the test oracle binding)."""

    def __init__(self, api: CApi, blob, device: int = 0):
        self.api = api
        if isinstance(blob, dict):
            # the tape as named arrays (tape.tape_arrays): handed over in place, nothing serialised
            # (`<prefix>create_arrays`; the library copies every array into its execution space during the call)
            from .tape import DTYPE_CODES
            keep, descs = [], (TapeArrayDesc * len(blob))()
            for d, (name, arr) in zip(descs, blob.items()):
                arr = np.ascontiguousarray(arr)
                if arr.dtype not in DTYPE_CODES:
                    raise TypeError("array %s has unsupported dtype %s" % (name, arr.dtype))
                keep.append(arr)
                d.name, d.dtype, d.reserved, d.count = name.encode(), DTYPE_CODES[arr.dtype], 0, arr.size
                d.data = arr.ctypes.data if arr.size else None
            self.ptr = api.create_arrays(descs, len(blob), device)
            del keep
        else:
            # the library parses the blob in place (and copies what it keeps); bytes and bytearray are both accepted
            buf = (C.c_char * len(blob)).from_buffer(blob) if isinstance(blob, bytearray) else blob
            self.ptr = api.create(buf, len(blob), device)
        if not self.ptr:
            raise RuntimeError("%screate failed: %s" % (api.prefix, api.error()))
        n, m, nj, nh = (C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64())
        api.dims(self.ptr, C.byref(n), C.byref(m), C.byref(nj), C.byref(nh))
        self.n, self.m, self.nnz_jac, self.nnz_hess = n.value, m.value, nj.value, nh.value

    def close(self):
        st = getattr(self, "_bstream", None)
        if st is not None and self.ptr:
            self.api.batch_stream_destroy(st[0])
        self._bstream = None
        if self.ptr:
            self.api.destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("%s failed (%d): %s" % (what, rc, self.api.error()))

    @staticmethod
    def _x(x):
        return np.ascontiguousarray(np.asarray(x, dtype=np.float64).reshape(-1))

    def bind_dense(self, const_id: int, device_ptr: int, ld: int):
        self._check(self.api.bind_dense(self.ptr, const_id, C.c_void_p(device_ptr), ld), "bind_dense")

    def eval_f(self, x):
        x = self._x(x)
        out = C.c_double()
        self._check(self.api.eval_f(self.ptr, _dp(x), 1, C.byref(out)), "eval_f")
        return out.value

    def eval_grad_f(self, x):
        x = self._x(x)
        out = np.empty(self.n)
        self._check(self.api.eval_grad_f(self.ptr, _dp(x), 1, _dp(out)), "eval_grad_f")
        return out

    def eval_g(self, x):
        x = self._x(x)
        out = np.empty(self.m)
        self._check(self.api.eval_g(self.ptr, _dp(x), 1, _dp(out)), "eval_g")
        return out

    def jac_structure(self):
        r = np.empty(self.nnz_jac, np.int32)
        c = np.empty(self.nnz_jac, np.int32)
        self._check(self.api.eval_jac_g(self.ptr, None, 0, _ip(r), _ip(c), None), "eval_jac_g")
        return r, c

    def eval_jac_g(self, x):
        x = self._x(x)
        out = np.empty(self.nnz_jac)
        self._check(self.api.eval_jac_g(self.ptr, _dp(x), 1, None, None, _dp(out)), "eval_jac_g")
        return out

    def hess_structure(self):
        if self.nnz_hess < 0:
            raise RuntimeError("Hessian too large for COO output")
        r = np.empty(self.nnz_hess, np.int32)
        c = np.empty(self.nnz_hess, np.int32)
        self._check(self.api.eval_h(self.ptr, None, 0, 0.0, None, 0, _ip(r), _ip(c), None), "eval_h")
        return r, c

    def eval_h(self, x, lagrange, obj_factor):
        x = self._x(x)
        lam = self._x(lagrange) if self.m else np.zeros(1)
        out = np.empty(self.nnz_hess)
        self._check(self.api.eval_h(self.ptr, _dp(x), 1, float(obj_factor), _dp(lam), 1, None, None,
                                    _dp(out)), "eval_h")
        return out

    def eval_fused(self, xfree):
        """f and grad f of the user's variables from the fused element program (one kernel)."""
        xfree = np.ascontiguousarray(xfree, dtype=np.float64)
        f = C.c_double()
        grad = np.empty_like(xfree)
        rc = self.api.eval_fused(self.ptr, _dp(xfree), C.cast(C.byref(f), _dbl_p), _dp(grad))
        if rc != 0:
            raise RuntimeError("eval_fused failed: %s" % self.api.error())
        return float(f.value), grad

    def time_fused(self, xfree, reps=10) -> float:
        xfree = np.ascontiguousarray(xfree, dtype=np.float64)
        sec = C.c_double()
        rc = self.api.time_fused(self.ptr, _dp(xfree), int(reps), C.cast(C.byref(sec), _dbl_p))
        if rc != 0:
            raise RuntimeError("time_fused failed: %s" % self.api.error())
        return float(sec.value)

    def set_warm_start(self, mult_g, mult_x_L, mult_x_U):
        """Multipliers of a previous solution for `warm_start_init_point=yes` (None clears them)."""
        if mult_g is None:
            self.api.set_warm_start(self.ptr, None, None, None)
            return
        mg = np.ascontiguousarray(mult_g, dtype=np.float64) if self.m else np.zeros(1)
        zl = np.ascontiguousarray(mult_x_L, dtype=np.float64)
        zu = np.ascontiguousarray(mult_x_U, dtype=np.float64)
        if (self.m and mg.size != self.m) or zl.size != self.n or zu.size != self.n:
            raise ValueError("warm-start multipliers have the wrong size")
        rc = self.api.set_warm_start(self.ptr, _dp(mg), _dp(zl), _dp(zu))
        if rc != 0:
            raise RuntimeError("set_warm_start failed: %s" % self.api.error())

    def stats(self):
        st = np.zeros(N_STATS)
        self.api.get_stats(self.ptr, _dp(st), N_STATS)
        return st

    def set_intermediate(self, fn):
        """Per-iteration callback `fn(alg_mod, iter_count, obj_value, inf_pr, inf_du, mu, d_norm,
        regularization_size, alpha_du, alpha_pr, ls_trials)` (the reference's Oracles.intermediate,
        nlp_solver.py:423-427).  A return value of False stops the solve with status 5
        (User_Requested_Stop); None / True continue.  `None` removes the callback."""
        if fn is None:
            self._cb_keep = None
            self.api.set_intermediate_cb(self.ptr, C.cast(None, INTERMEDIATE_CB), None)
            return

        def tramp(alg, it, obj, ipr, idu, mu, dn, reg, adu, apr, ls, _user):
            try:
                r = fn(alg, it, obj, ipr, idu, mu, dn, reg, adu, apr, ls)
            except Exception:            # an exception inside a C callback cannot propagate: stop the solve
                import traceback
                traceback.print_exc()
                return 0
            return 0 if r is False else 1
        self._cb_keep = INTERMEDIATE_CB(tramp)       # keep the thunk alive as long as the handle uses it
        self.api.set_intermediate_cb(self.ptr, self._cb_keep, None)

    def kkt_info(self):
        """Linear-solver plan of this handle: sparse static-pattern LDL^T or dense."""
        out = (C.c_int64 * 8)()
        rc = self.api.kkt_info(self.ptr, out)
        if rc != 0:
            raise RuntimeError("kkt_info failed: %s" % self.api.error())
        return {"sparse": bool(out[0]), "factor_values": int(out[1]), "pivot_blocks": int(out[2]),
                "max_struct": int(out[3]), "pairs_2x2": int(out[4]), "update_triples": int(out[5]),
                "levels": int(out[6]), "dense_pivoted": bool(out[7])}

    KKT_MODES = {-1: None, 0: "sparse", 1: "bunch-kaufman", 2: "unpivoted", 3: "paired", 4: "paired-then-bunch-kaufman"}

    def kkt_mode(self):
        """The handle's linear solver as it stands after the last solve (csrc/kkt_dense.h): 'sparse', 'bunch-kaufman',
        'unpivoted' (blocked LDL^T), 'paired' (unpivoted on rotated static pairs), 'paired-then-bunch-kaufman' (the static
        sequence lost digits on the way and the handle was demoted); None before the first solve."""
        return self.KKT_MODES[int(self.api.kkt_mode(self.ptr))]

    def kkt_tail_nodes(self) -> int:
        """Nodes of the dense tail of the sparse plan (0: none): such a plan is the host-driven loop's."""
        return int(self.api.kkt_tail_nodes(self.ptr))

    def reset_options(self):
        self.api.reset_options(self.ptr)

    def set_option(self, key, val):
        if isinstance(val, bool):
            val = "yes" if val else "no"
        rc = self.api.set_option(self.ptr, str(key).encode(), str(val).encode())
        if rc != 0:
            raise ValueError("Invalid solver option %s=%s" % (key, val))

    def _outputs(self):
        return (np.empty(self.n), C.c_double(), np.empty(max(self.m, 1)), np.empty(max(self.m, 1)),
                np.empty(self.n), np.empty(self.n), C.c_int())

    def _info(self, status, x, obj, g, mg, zl, zu, iters):
        st = np.zeros(N_STATS)
        self.api.get_stats(self.ptr, _dp(st), N_STATS)
        return {"status": int(status), "x": x, "obj_val": float(obj.value), "g": g[:self.m],
                "mult_g": mg[:self.m], "mult_x_L": zl, "mult_x_U": zu, "iterations": int(iters.value),
                "solve_time": float(st[2]), "stats": st}

    def solve(self, x0):
        x, obj, g, mg, zl, zu, iters = self._outputs()
        x[:] = self._x(x0)
        status = self.api.solve(self.ptr, _dp(x), C.byref(obj), _dp(g), _dp(mg), _dp(zl), _dp(zu),
                                C.byref(iters))
        if status == -199:
            raise RuntimeError("solve failed: %s" % self.api.error())
        return self._info(status, x, obj, g, mg, zl, zu, iters)

    def solve_reduced(self, x0):
        """Reduced-space L-BFGS (tape f / grad f + line search only; BASELINE config C2)."""
        import time
        x = np.array(self._x(x0))
        obj, gn = C.c_double(), C.c_double()
        iters, evals = C.c_int(), C.c_int()
        t0 = time.time()
        st = self.api.solve_reduced(self.ptr, _dp(x), C.byref(obj), C.byref(iters), C.byref(evals), C.byref(gn))
        if st == -199:
            raise RuntimeError("solve_reduced failed: %s" % self.api.error())
        status = {0: 0, -1: -1, 3: 3}.get(st, st)
        ri = np.zeros(6)
        self.api.reduced_info(self.ptr, _dp(ri), 6)
        return {"status": status, "x": x, "obj_val": obj.value, "iterations": iters.value,
                "device_loop": bool(ri[0]), "device_loop_seconds": float(ri[1]), "device_loop_slots": int(ri[2]),
                "fused_objective_used": bool(ri[3]), "library_seconds": float(ri[4]),
                "device_loop_persistent": bool(ri[5]),
                "evaluations": evals.value, "grad_inf_norm": gn.value, "solve_time": time.time() - t0,
                "g": self.eval_g(x) if self.m else np.zeros(0), "mult_g": np.zeros(self.m),
                "mult_x_L": np.zeros(self.n), "mult_x_U": np.zeros(self.n), "stats": np.zeros(N_STATS)}

    def ipm_begin(self, x0):
        x0 = self._x(x0)
        rc = self.api.ipm_begin(self.ptr, _dp(x0))
        if rc == -199:
            raise RuntimeError("ipm_begin failed: %s" % self.api.error())
        return rc

    def ipm_step(self, max_steps=1):
        done = C.c_int()
        rc = self.api.ipm_step(self.ptr, int(max_steps), C.byref(done))
        if rc == -199:
            raise RuntimeError("ipm_step failed: %s" % self.api.error())
        return rc, done.value

    def ipm_finish(self):
        x, obj, g, mg, zl, zu, iters = self._outputs()
        status = self.api.ipm_finish(self.ptr, _dp(x), C.byref(obj), _dp(g), _dp(mg), _dp(zl), _dp(zu),
                                     C.byref(iters))
        return self._info(status, x, obj, g, mg, zl, zu, iters)

    def set_batch_affine_map(self, d0, theta0, D):
        """Hand the affine parameter -> instance-data map to the device once (`D`: scipy CSR,
        stride x P): later `solve_batch(thetas=...)` calls move only the parameter rows."""
        need = int(self.api.batch_stride(self.ptr))
        if getattr(self, "_bstream", None) is not None:          # (its slots carry copies of the previous map)
            self.api.batch_stream_destroy(self._bstream[0])
            self._bstream = None
        d0 = np.ascontiguousarray(d0, dtype=np.float64)
        theta0 = np.ascontiguousarray(theta0, dtype=np.float64)
        if d0.size != need or D.shape != (need, theta0.size):
            raise ValueError("set_batch_affine_map: the map does not match the tape's instance stride")
        D = D.tocsr()
        D.sort_indices()
        indptr = np.ascontiguousarray(D.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(D.indices, dtype=np.int32)
        vals = np.ascontiguousarray(D.data, dtype=np.float64)
        rc = self.api.batch_set_affine_map(self.ptr, int(theta0.size), _dp(d0), _dp(theta0 if theta0.size else np.zeros(1)),
                                           indptr.ctypes.data_as(C.POINTER(C.c_int64)),
                                           _ip(indices if indices.size else np.zeros(1, np.int32)),
                                           _dp(vals if vals.size else np.zeros(1)))
        if rc != 0:
            raise RuntimeError("batch_set_affine_map failed: %s" % self.api.error())
        self._affine_P = int(theta0.size)

    def solve_batch(self, data=None, want_duals: bool = False, warm=None, thetas=None):
        """Solve `data.shape[0]` instances that share this handle's tape structure in ONE kernel
        launch (one workgroup per instance, csrc/batch.h).  `data`: (B, stride) float64, rows laid
        out as dnlp_amd.batch.BATCH_DATA_KEYS — or `thetas`: (B, P) parameter rows after
        set_batch_affine_map (the instance data is then generated on the device).  Returns arrays over
        the batch."""
        if thetas is not None:
            thetas = np.ascontiguousarray(thetas, dtype=np.float64)
            B = thetas.shape[0]
            if getattr(self, "_affine_P", None) != thetas.shape[1]:
                raise ValueError("solve_batch: parameter rows do not match the affine map of this handle")
        else:
            data = np.ascontiguousarray(data, dtype=np.float64)
            B, stride = data.shape
            need = int(self.api.batch_stride(self.ptr))
            if need < 0:
                raise RuntimeError("solve_batch: %s" % self.api.error())
            if stride != need:
                raise ValueError("solve_batch: rows have %d values, the tape needs %d" % (stride, need))
        if warm is not None:
            mg, wl, wu = (np.ascontiguousarray(a, dtype=np.float64) for a in warm)
            if mg.shape != (B, self.m) or wl.shape != (B, self.n) or wu.shape != (B, self.n):
                raise ValueError("solve_batch: warm-start multipliers must be (B, m), (B, N), (B, N)")
            if self.api.batch_warm_start(self.ptr, B, _dp(mg if self.m else np.zeros(1)), _dp(wl), _dp(wu)) != 0:
                raise RuntimeError("batch_warm_start failed: %s" % self.api.error())
        x = np.empty((B, self.n))
        obj = np.empty(B)
        mg = np.empty((B, max(self.m, 1))) if want_duals else None
        zl = np.empty((B, self.n)) if want_duals else None
        zu = np.empty((B, self.n)) if want_duals else None
        status = np.empty(B, dtype=np.int32)
        iters = np.empty(B, dtype=np.int32)
        nfact = np.empty(B, dtype=np.int32)
        sec = C.c_double()
        _int_p = C.POINTER(C.c_int)
        ip = lambda a: a.ctypes.data_as(_int_p)
        times = np.zeros((B, 4))
        outs = (_dp(x), _dp(obj), _dp(mg) if want_duals else None, _dp(zl) if want_duals else None,
                _dp(zu) if want_duals else None, ip(status), ip(iters), ip(nfact), C.cast(C.byref(sec), _dbl_p), _dp(times))
        if thetas is not None:
            rc = self.api.solve_batch_theta(self.ptr, B, _dp(thetas if thetas.size else np.zeros(1)), thetas.shape[1], *outs)
        else:
            rc = self.api.solve_batch_timed(self.ptr, B, _dp(data), stride, *outs)
        if rc != 0:
            raise RuntimeError("solve_batch failed (code %d): %s" % (rc, self.api.error()))
        out = {"x": x, "obj_val": obj, "status": status, "iterations": iters, "factorizations": nfact,
               "kernel_seconds": float(sec.value), "phase_seconds": times}
        if hasattr(self.api, "batch_launch_info"):
            info = np.zeros(8, dtype=np.int32)
            if self.api.batch_launch_info(self.ptr, info.ctypes.data_as(_i32_p)) == 0:
                # (lds_mode of a wavefront-solver launch: 2 x state in LDS + plan in LDS, + 4 when the kernel was the one
                #  compiled for this template at run time — csrc/wave_codegen.h — + 8 when that was the workgroup-per-instance
                #  kernel: `lanes` / 64 wavefronts per instance, `per_cu` workgroups per compute unit)
                out["launch"] = {"grid": int(info[0]), "lanes": int(info[1]), "lds_mode": int(info[2]) & 3, "per_cu": int(info[3]),
                                 "packed": bool(info[4]), "longest_first": bool(info[5]), "wave_form": int(info[6]),
                                 "wave_refused": int(info[7]), "wave_spec": bool(int(info[6]) and (int(info[2]) & 4)),
                                 "wave_wg": bool(int(info[6]) and (int(info[2]) & 8))}
        if want_duals:
            out.update({"mult_g": mg[:, :self.m], "mult_x_L": zl, "mult_x_U": zu})
        return out

    def keep_batch_result_rows(self, on: bool = True):
        """dnlp_batch_keep_result_rows: the launches that follow also leave their result rows packed on the device."""
        if not hasattr(self.api, "batch_keep_result_rows"):
            return False
        if self.api.batch_keep_result_rows(self.ptr, 1 if on else 0) != 0:
            raise RuntimeError("batch_keep_result_rows failed: %s" % self.api.error())
        return True

    def batch_result_rows(self):
        """dnlp_batch_result_rows: (device pointer, rows, width) of the last launch's packed rows {index, objective,
        status, iterations, x*}; the memory belongs to the handle and is valid until its next launch."""
        ptr, rows, width = C.c_void_p(), C.c_int64(), C.c_int64()
        if self.api.batch_result_rows(self.ptr, C.byref(ptr), C.byref(rows), C.byref(width)) != 0:
            raise RuntimeError("batch_result_rows failed: %s" % self.api.error())
        return (ptr.value or 0), int(rows.value), int(width.value)

    def solve_batch_stream(self, batches, slots: int = 2, want_duals: bool = False):
        """A stream of parameter-row batches with `slots` launches in flight INSIDE the library (include/dnlp_hip.h
        dnlp_batch_stream_*: own HIP stream, buffers and host thread per slot — no Python threads, one handle).  Returns the
        solve_batch dicts in order; bit for bit what solve_batch(thetas=...) returns one launch at a time."""
        batches = [np.ascontiguousarray(t, dtype=np.float64) for t in batches]
        P = getattr(self, "_affine_P", None)
        if any(t.ndim != 2 or t.shape[1] != P for t in batches):
            raise ValueError("solve_batch_stream: parameter rows do not match the affine map of this handle")
        # (the stream — its slots' device buffers and plan copies — is kept with the handle for the next call)
        cached = getattr(self, "_bstream", None)
        if cached is not None and cached[1] != int(slots):
            self.api.batch_stream_destroy(cached[0])
            cached = self._bstream = None
        if cached is None:
            st = self.api.batch_stream_create(self.ptr, int(slots))
            if not st:
                raise RuntimeError("batch_stream_create failed: %s" % self.api.error())
            self._bstream = (st, int(slots))
        st = self._bstream[0]
        _int_p = C.POINTER(C.c_int)
        ip = lambda a: a.ctypes.data_as(_int_p)      # noqa: E731
        outs, tickets, res = [], [], [None] * len(batches)

        def collect(k):
            sec = C.c_double()
            rc = self.api.batch_stream_wait(st, tickets[k], C.cast(C.byref(sec), _dbl_p))
            if rc != 0:
                raise RuntimeError("batch stream: submission %d failed (code %d): %s" % (k, rc, self.api.error()))
            o = outs[k]
            res[k] = {"x": o["x"], "obj_val": o["obj"], "status": o["status"], "iterations": o["iters"], "factorizations": o["nfact"],
                      "kernel_seconds": float(sec.value), "phase_seconds": np.zeros((o["x"].shape[0], 4))}
            if want_duals:
                res[k].update({"mult_g": o["mg"][:, :self.m], "mult_x_L": o["zl"], "mult_x_U": o["zu"]})
        try:
            for k, th in enumerate(batches):
                B = th.shape[0]
                o = {"x": np.empty((B, self.n)), "obj": np.empty(B), "status": np.empty(B, np.int32), "iters": np.empty(B, np.int32),
                     "nfact": np.empty(B, np.int32), "mg": np.empty((B, max(self.m, 1))) if want_duals else None,
                     "zl": np.empty((B, self.n)) if want_duals else None, "zu": np.empty((B, self.n)) if want_duals else None}
                outs.append(o)
                if k >= slots:
                    collect(k - slots)
                t = self.api.batch_stream_submit(st, B, _dp(th if th.size else np.zeros(1)), th.shape[1], _dp(o["x"]), _dp(o["obj"]),
                                                 _dp(o["mg"]) if want_duals else None, _dp(o["zl"]) if want_duals else None,
                                                 _dp(o["zu"]) if want_duals else None, ip(o["status"]), ip(o["iters"]), ip(o["nfact"]))
                if t < 0:
                    raise RuntimeError("batch_stream_submit failed (code %d): %s" % (t, self.api.error()))
                tickets.append(t)
            for k in range(max(0, len(batches) - slots), len(batches)):
                collect(k)
        except BaseException:
            self.api.batch_stream_destroy(st)        # (waits for what is in flight: the output arrays die with this frame)
            self._bstream = None
            raise
        return res

    def log(self) -> str:
        need = self.api.get_log(self.ptr, None, 0)
        buf = C.create_string_buffer(need + 1)
        self.api.get_log(self.ptr, buf, need + 1)
        return buf.value.decode()


class DeviceProblem(ProblemHandle):
    """Product handle: created on the MI355X selected by LOCAL_RANK / DNLP_DEVICE."""

    def __init__(self, blob: bytes, tape=None, device: int = None):
        dev = current_device() if device is None else device
        api = require_device(dev)
        super().__init__(api, blob, dev)
        self.device = dev
        if tape is not None:
            for k, dc in enumerate(tape.dense_consts):
                if dc.device is not None:
                    self.bind_dense(k, dc.device.ptr, dc.device.ld)
