"""Lower a canonicalised (smooth) problem to the flat device tape — once per solve.

This replaces, for the nlp=True path, three pieces of the reference that run on EVERY
interior-point callback: the recursive `expr.jacobian()/hess_vec()` tree walks
(atoms/atom.py:501-561 + per-atom rules), the Python COO bookkeeping of `Oracles`
(nlp_solver.py:246-276, 337-372: parse_*_dict / sum_coo / insert_missing_zeros) and the
NaN-based structure discovery (nlp_solver.py:309-335, 374-392).  It also plays the role the
cvxcore lin_ops builder has on the conic path: affine sub-DAGs are folded ONCE into constant
sparse blocks.

Normal form ("tape").  After dnlp2smooth every nonlinear atom is applied to bare variables,
so with x in R^N and z in R^Z the concatenated outputs of all nonlinear atom segments,

    f(x) = c0 + c . [x; z(x)]                   g(x) = b + G [x; z(x)]      (G: m x (N+Z) CSR)

and all derivative values are constant-sparse-matrix images of per-segment element arrays
that the HIP kernels fill:

    dvals  (nnzD)  first derivatives  dz_r/dx_c        one entry per (segment element, arg)
    hvals  (nnzD2) weighted second derivatives  w_r * d2 z_r / dx_a dx_b   (lower oriented)
    w      = Mw [sigma; lambda]          (Z)        pull-back of the multipliers
    grad f = c_x + Mg  dvals             (N)
    J vals = Jc  + MJ  dvals             (nnzJ, row-major sorted COO, pattern fixed here)
    H vals =       MH  hvals (+ dense quad_form blocks)   (nnzH, lower triangle, row-major)

Segments are contiguous runs of identical elementwise work (coalesced x reads); dense
quad_form matrices stay dense (and may live in HBM only, see `DeviceMatrix`).
"""
from __future__ import annotations

import ctypes as C
import os

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import scipy.sparse as sp

from . import atoms as at
from .expressions import Constant, Expression, Parameter, Variable

# ---- opcodes (shared with csrc/tape.h) ---------------------------------------------------
OP_EXP, OP_LOG, OP_ENTR, OP_LOGISTIC, OP_POWER = 1, 2, 3, 4, 5
OP_SIN, OP_COS, OP_TAN, OP_SINH, OP_TANH, OP_ASINH, OP_ATANH, OP_XEXP = 6, 7, 8, 9, 10, 11, 12, 13
OP_MUL, OP_REL_ENTR = 20, 21
OP_QUAD_FORM_DENSE, OP_QUAD_FORM_SPARSE, OP_QUAD_OVER_LIN, OP_MATMUL = 30, 31, 32, 33

UNARY_OPS = {
    at.exp: OP_EXP, at.log: OP_LOG, at.entr: OP_ENTR, at.logistic: OP_LOGISTIC,
    at.sin: OP_SIN, at.cos: OP_COS, at.tan: OP_TAN, at.sinh: OP_SINH, at.tanh: OP_TANH,
    at.asinh: OP_ASINH, at.atanh: OP_ATANH, at.xexp: OP_XEXP,
}

# dense quad_form blocks up to this order are also listed in the COO Hessian pattern
DENSE_COO_MAX_N = 8192
# host-resident dense P up to this order is expanded into per-entry tape work
DENSE_EXPAND_MAX_N = 48


class LinForm:
    """rows x (N+Z) sparse coefficient matrix plus constant: value = A @ [x; z] + b.

    scipy.sparse implementation (the fallback when libdnlp_hip.so is not built, and the checker of the C++ one:
    tests/test_lower_maps.py).  `CLinForm` below has the same interface over handles of the C ABI."""
    __slots__ = ("A", "b")

    def __init__(self, A, b):
        self.A = A.tocsr()
        self.b = np.asarray(b, dtype=float).reshape(-1)

    @property
    def rows(self):
        return self.A.shape[0]

    # -- constructors --
    @classmethod
    def const(cls, ncol, values):
        values = np.asarray(values, dtype=float).reshape(-1)
        return cls(sp.csr_matrix((values.size, ncol)), values)

    @classmethod
    def range(cls, ncol, n, col0):
        A = sp.csr_matrix((np.ones(n), (np.arange(n), col0 + np.arange(n))), shape=(n, ncol))
        return cls(A, np.zeros(n))

    # -- operations --
    def select(self, sel):
        sel = np.asarray(sel, dtype=np.int64).reshape(-1)
        return LinForm(self.A[sel, :], self.b[sel])

    def neg(self):
        return LinForm(-self.A, -self.b)

    def scale(self, cvals):
        cvals = np.asarray(cvals, dtype=float).reshape(-1)
        return LinForm(sp.diags(cvals) @ self.A, cvals * self.b)

    @staticmethod
    def add(forms):
        A = forms[0].A
        b = forms[0].b.copy()
        for f in forms[1:]:
            if f.A.nnz:                      # (a constant term has no coefficients: no sparse sum, no copy)
                A = A + f.A if A.nnz else f.A
            b = b + f.b
        return LinForm(A, b)

    @staticmethod
    def vstack(forms):
        if len(forms) == 1:
            return forms[0]
        return LinForm(sp.vstack([f.A for f in forms], format="csr"), np.concatenate([f.b for f in forms]))

    def gather(self, N):
        """Column of every row when the form is a plain selection of variables (columns below N), else None."""
        A = self.A
        ok = (A.nnz == self.rows and np.all(np.diff(A.indptr) == 1) and np.all(A.data == 1.0)
              and np.all(self.b == 0.0) and (A.indices.size == 0 or A.indices.max() < N))
        return A.indices.astype(np.int64) if ok else None

    def csr(self):
        """(CSR matrix, constants): the canonical form is NOT guaranteed (lower_problem canonicalises)."""
        return self.A, self.b

    def apply_dense(self, M):
        """Left-multiply by a dense constant matrix."""
        return self.apply(_dense_csr(M))

    def apply(self, S):
        """Left-multiply by a constant sparse matrix S (out_rows x rows)."""
        S = sp.csr_matrix(S)
        A = self.A
        if (A.nnz == A.shape[0] == S.shape[1] and S.nnz > 4096 and not self.b.any()
                and bool(np.all(A.data == 1.0)) and A.indptr[-1] == A.shape[0]
                and bool(np.all(np.diff(A.indptr) == 1))):
            # a plain (sub)vector of variables: S @ A is S with its columns renamed -- one gather over S's
            # indices instead of a sparse product (a dense 1e3 x 1e4 constraint block is 1e7 entries)
            cols = A.indices.astype(S.indices.dtype, copy=False)[S.indices]
            if bool(np.all(np.diff(A.indices) > 0)):
                # monotone renaming: S's value / row-pointer arrays are shared, read-only from here on
                R = sp.csr_matrix((S.data, cols, S.indptr), shape=(S.shape[0], A.shape[1]))
            else:
                # x[::-1], x[perm], hstack([y, x]): the rows must be re-sorted, and sort_indices permutes
                # the value array IN PLACE — S.data may be the caller's own constant (a dense matrix without
                # zeros is wrapped as a view), so the sort works on a private copy
                R = sp.csr_matrix((S.data.copy(), cols, S.indptr.copy()), shape=(S.shape[0], A.shape[1]))
                R.has_sorted_indices = False
                R.sort_indices()
            return LinForm(R, np.zeros(S.shape[0]))
        return LinForm(S @ A, S @ self.b)


class CLinForm:
    """The same affine form behind the C ABI (include/dnlp_hip.h: dnlp_lf_*, csrc/linform.h): the DAG walk composes
    handles instead of scipy.sparse objects."""
    __slots__ = ("h", "rows", "ncol", "__weakref__")
    _lib = None

    @classmethod
    def available(cls):
        if cls._lib is None:
            try:
                from . import _capi
                lib = _capi.load().lib
                i64p, i32p, dp = C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_double)
                for name, args in (("dnlp_lf_const", [C.c_int64, C.c_int64, dp]), ("dnlp_lf_range", [C.c_int64] * 3),
                                   ("dnlp_lf_select", [C.c_void_p, i64p, C.c_int64]), ("dnlp_lf_add", [C.c_void_p, C.c_void_p]),
                                   ("dnlp_lf_scale", [C.c_void_p, dp]),
                                   ("dnlp_lf_apply_csr", [C.c_void_p, C.c_int64, i64p, i32p, dp]),
                                   ("dnlp_lf_vstack", [C.POINTER(C.c_void_p), C.c_int])):
                    fn = getattr(lib, name)
                    fn.restype = C.c_void_p
                    fn.argtypes = args
                lib.dnlp_lf_free.argtypes = [C.c_void_p]
                lib.dnlp_lf_free.restype = None
                lib.dnlp_lf_info.argtypes = [C.c_void_p, i64p]
                lib.dnlp_lf_export.argtypes = [C.c_void_p, i64p, i32p, dp, dp]
                lib.dnlp_lf_view.argtypes = [C.c_void_p] + [C.POINTER(C.c_void_p)] * 4
                lib.dnlp_lf_apply_dense.restype = C.c_void_p
                lib.dnlp_lf_apply_dense.argtypes = [C.c_void_p, C.c_int64, dp]
                lib.dnlp_lf_gather.argtypes = [C.c_void_p, C.c_int64, i64p]
                lib.dnlp_last_error.restype = C.c_char_p
                cls._lib = lib
            except Exception:
                cls._lib = False
        return bool(cls._lib)

    def __init__(self, h, rows, ncol):
        if not h:
            raise RuntimeError("linform: %s" % (self._lib.dnlp_last_error() or b"").decode())
        self.h, self.rows, self.ncol = h, int(rows), int(ncol)

    def __del__(self):
        try:
            if self.h:
                self._lib.dnlp_lf_free(self.h)
        except Exception:
            pass

    @staticmethod
    def _p(a, ct):
        return a.ctypes.data_as(C.POINTER(ct))

    @classmethod
    def const(cls, ncol, values):
        v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        return cls(cls._lib.dnlp_lf_const(ncol, v.size, cls._p(v, C.c_double)), v.size, ncol)

    @classmethod
    def range(cls, ncol, n, col0):
        return cls(cls._lib.dnlp_lf_range(ncol, n, col0), n, ncol)

    def select(self, sel):
        sel = np.ascontiguousarray(sel, dtype=np.int64).reshape(-1)
        return CLinForm(self._lib.dnlp_lf_select(self.h, self._p(sel, C.c_int64), sel.size), sel.size, self.ncol)

    def neg(self):
        return CLinForm(self._lib.dnlp_lf_scale(self.h, None), self.rows, self.ncol)

    def scale(self, cvals):
        v = np.ascontiguousarray(cvals, dtype=np.float64).reshape(-1)
        if v.size != self.rows:
            raise ValueError("scale: one factor per row")
        return CLinForm(self._lib.dnlp_lf_scale(self.h, self._p(v, C.c_double)), self.rows, self.ncol)

    @staticmethod
    def add(forms):
        out = forms[0]
        for f in forms[1:]:
            out = CLinForm(CLinForm._lib.dnlp_lf_add(out.h, f.h), out.rows, out.ncol)
        return out

    @staticmethod
    def vstack(forms):
        if len(forms) == 1:
            return forms[0]
        arr = (C.c_void_p * len(forms))(*[f.h for f in forms])
        return CLinForm(CLinForm._lib.dnlp_lf_vstack(arr, len(forms)), sum(f.rows for f in forms), forms[0].ncol)

    def apply(self, S):
        S = sp.csr_matrix(S)
        ptr = np.ascontiguousarray(S.indptr, dtype=np.int64)
        idx = np.ascontiguousarray(S.indices, dtype=np.int32)
        val = np.ascontiguousarray(S.data, dtype=np.float64)
        if S.shape[1] != self.rows:
            raise ValueError("apply: shape mismatch")
        return CLinForm(self._lib.dnlp_lf_apply_csr(self.h, S.shape[0], self._p(ptr, C.c_int64), self._p(idx, C.c_int32),
                                                    self._p(val, C.c_double)), S.shape[0], self.ncol)

    def _info(self):
        info = np.zeros(4, np.int64)
        self._lib.dnlp_lf_info(self.h, self._p(info, C.c_int64))
        return info

    def apply_dense(self, M):
        M = np.ascontiguousarray(M, dtype=np.float64)
        if M.shape[1] != self.rows:
            raise ValueError("apply: shape mismatch")
        return CLinForm(self._lib.dnlp_lf_apply_dense(self.h, M.shape[0], self._p(M, C.c_double)), M.shape[0], self.ncol)

    def csr(self):
        """(A, b) in canonical form as views of the handle's arrays (no copy; the handle lives as long as they do)."""
        from ._capi import CsrArrays, view_array
        pp, pi, pv, pb = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._lib.dnlp_lf_view(self.h, C.byref(pp), C.byref(pi), C.byref(pv), C.byref(pb))
        nnz = int(self._info()[2])
        A = CsrArrays(view_array(pp.value, self.rows + 1, C.c_int64, np.int64, self), view_array(pi.value, nnz, C.c_int32, np.int32, self),
                      view_array(pv.value, nnz, C.c_double, np.float64, self), (self.rows, self.ncol))
        return A, view_array(pb.value, self.rows, C.c_double, np.float64, self)

    def gather(self, N):
        out = np.zeros(self.rows, np.int64)
        return out if self._lib.dnlp_lf_gather(self.h, N, self._p(out, C.c_int64)) else None

    @property
    def A(self):
        return self.csr()[0].tocsr()

    @property
    def b(self):
        return self.csr()[1]


@dataclass
class Segment:
    op: int
    n: int                      # elements of the segment (inputs for reductions)
    a0: np.ndarray              # x indices of argument 0 (length n, or n_in)
    a1: Optional[np.ndarray]    # x indices of argument 1 (or None)
    param: float = 0.0          # derivative exponent (power) / unused
    param2: float = 0.0         # forward exponent (power)
    zoff: int = 0
    zcount: int = 0
    doff: int = 0
    dcount: int = 0
    hoff: int = 0
    hcount: int = 0
    aux: int = -1               # constant-matrix id (quad_form) / inner dimension (matmul)
    dims: tuple = (0, 0, 0)     # matmul (m, k, p)


@dataclass
class DenseConst:
    n: int
    host: Optional[np.ndarray] = None       # column-major dense, when it travels in the blob
    device: Optional[object] = None         # DeviceMatrix, when resident in HBM only


@dataclass
class Tape:
    """The lowered problem (host arrays, ready for `tape.serialize`)."""
    N: int
    m: int
    Z: int
    segments: List[Segment]
    dense_consts: List[DenseConst]
    sparse_consts: List[sp.csr_matrix]
    c0: float
    c: np.ndarray
    G: sp.csr_matrix
    b: np.ndarray
    drow: np.ndarray
    dcol: np.ndarray
    hrow: np.ndarray
    hcol: np.ndarray
    hz: np.ndarray
    Mg: sp.csr_matrix
    Mw: sp.csr_matrix
    MJ: sp.csr_matrix
    Jc: np.ndarray
    jac_rows: np.ndarray
    jac_cols: np.ndarray
    MH: sp.csr_matrix
    hess_rows: np.ndarray
    hess_cols: np.ndarray
    dense_blocks: List[dict] = field(default_factory=list)
    hess_coo_complete: bool = True     # False when a dense block is too large to list as COO
    var_offsets: Dict[int, int] = field(default_factory=dict)

    @property
    def nnzJ(self):
        return int(self.jac_rows.size)

    @property
    def nnzH(self):
        return int(self.hess_rows.size)


class Lowerer:
    def __init__(self, variables: List[Variable], exprs: List[Expression]):
        self.variables = variables
        self.var_offsets = {}
        off = 0
        for v in variables:
            self.var_offsets[id(v)] = off
            off += v.size
        self.N = off
        self.Zcap = self._count_outputs(exprs)
        self.ncol = self.N + self.Zcap
        self.Z = 0
        self.segments: List[Segment] = []
        self.dense_consts: List[DenseConst] = []
        self.sparse_consts: List[sp.csr_matrix] = []
        self.dense_blocks: List[dict] = []
        self._d = ([], [])          # drow (z index), dcol (x index)
        self._h = ([], [], [])      # hrow, hcol (lower oriented x indices), hz (z index)
        self.nd = 0
        self.nh = 0
        self._memo = {}
        # affine forms behind the C ABI (csrc/linform.h) unless the library is not built or DNLP_LOWER_CXX=0
        self.LF = CLinForm if (os.environ.get("DNLP_LOWER_CXX", "1") != "0" and CLinForm.available()) else LinForm

    # -- sizing ---------------------------------------------------------------------
    def _count_outputs(self, exprs):
        seen = set()
        total = 0

        def walk(e):
            nonlocal total
            if id(e) in seen:
                return
            seen.add(id(e))
            if isinstance(e, at.Atom) and not e.is_constant() and self._is_nonlinear(e):
                total += e.size
            for a in e.args:
                walk(a)
        for e in exprs:
            walk(e)
        return total

    @staticmethod
    def _is_nonlinear(e) -> bool:
        if isinstance(e, (at.MulExpression,)):   # incl. multiply
            return not (e.args[0].is_constant() or e.args[1].is_constant())
        if isinstance(e, at.AffAtom):
            return False
        return True

    # -- leaves ---------------------------------------------------------------------
    def _const_form(self, value, size):
        v = value.toarray() if sp.issparse(value) else np.asarray(value, dtype=float)
        return self.LF.const(self.ncol, v.reshape(-1, order="F"))

    def _var_form(self, v: Variable):
        off = self.var_offsets.get(id(v))
        if off is None:
            raise ValueError("Variable %s is not part of the problem." % v.name())
        return self.LF.range(self.ncol, v.size, off)

    def _z_form(self, zoff, n):
        return self.LF.range(self.ncol, n, self.N + zoff)

    # -- main recursion ---------------------------------------------------------------
    def lower(self, e: Expression) -> LinForm:
        key = id(e)
        if key in self._memo:
            return self._memo[key]
        out = self._lower(e)
        if out.rows != e.size:
            raise AssertionError("lowering of %s produced %d rows for size %d"
                                 % (type(e).__name__, out.rows, e.size))
        self._memo[key] = out
        return out

    def _lower(self, e):
        if isinstance(e, Variable):
            return self._var_form(e)
        if isinstance(e, Parameter):
            if e.value is None:
                raise ValueError("Parameter %s has no value." % e.name())
            return self._const_form(e.value, e.size)
        if isinstance(e, Constant):
            if e.is_device:
                raise ValueError("A device-resident constant can only be the matrix of quad_form.")
            return self._const_form(e.value, e.size)
        if e.is_constant():
            return self._const_form(e.value, e.size)
        fn = getattr(self, "_lower_" + type(e).__name__, None)
        if fn is None:
            if type(e) in UNARY_OPS:
                return self._lower_unary(e, UNARY_OPS[type(e)], 0.0)
            raise NotImplementedError("No tape lowering for atom %s." % type(e).__name__)
        return fn(e)

    # -- affine atoms ---------------------------------------------------------------
    @staticmethod
    def _idx(e):
        return np.arange(e.size, dtype=np.int64).reshape(e.shape, order="F")

    def _lower_AddExpression(self, e):
        forms = []
        for a in e.args:
            f = self.lower(a)
            if a.shape != e.shape:   # numpy-style broadcast of an argument
                sel = np.broadcast_to(self._idx(a), e.shape).reshape(-1, order="F")
                f = f.select(sel)
            forms.append(f)
        return self.LF.add(forms)

    def _lower_NegExpression(self, e):
        return self.lower(e.args[0]).neg()

    def _scale_rows(self, f, cvals):
        return f.scale(np.asarray(cvals, dtype=float).reshape(-1))

    def _bcast_const(self, cexpr, shape):
        v = cexpr.value
        v = v.toarray() if sp.issparse(v) else np.asarray(v, dtype=float)
        return np.broadcast_to(v, shape).reshape(-1, order="F")

    def _lower_multiply(self, e):
        x, y = e.args
        if x.is_constant():
            f = self.lower(y)
            if y.shape != e.shape:
                f = f.select(np.broadcast_to(self._idx(y), e.shape).reshape(-1, order="F"))
            return self._scale_rows(f, self._bcast_const(x, e.shape))
        if y.is_constant():
            f = self.lower(x)
            if x.shape != e.shape:
                f = f.select(np.broadcast_to(self._idx(x), e.shape).reshape(-1, order="F"))
            return self._scale_rows(f, self._bcast_const(y, e.shape))
        return self._lower_bilinear_elementwise(e)

    def _lower_DivExpression(self, e):
        x, y = e.args
        if not y.is_constant():
            raise ValueError("Division by a non-constant must be canonicalised first.")
        f = self.lower(x)
        if x.shape != e.shape:
            f = f.select(np.broadcast_to(self._idx(x), e.shape).reshape(-1, order="F"))
        return self._scale_rows(f, 1.0 / self._bcast_const(y, e.shape))

    @staticmethod
    def _matmul_dims(X, Y):
        """(m, k, p) of X @ Y with numpy's 1-D promotion: a 1-D left operand is a row."""
        if X.ndim == 1:
            m, k = 1, X.shape[0]
        else:
            m, k = at.MulExpression.get_dimensions(X)
        k2, p = at.MulExpression.get_dimensions(Y)
        if k != k2:
            raise ValueError("Incompatible dimensions %s %s" % (X.shape, Y.shape))
        return m, k, p

    def _lower_MulExpression(self, e):
        X, Y = e.args
        m, k, p = self._matmul_dims(X, Y)
        if X.is_constant():
            C = X.value
            if p == 1 and not sp.issparse(C):
                return self.lower(Y).apply_dense(np.asarray(C, dtype=float).reshape(m, k))
            C = sp.csr_matrix(C) if sp.issparse(C) else _dense_csr(np.asarray(C, dtype=float).reshape(m, k))
            S = C if p == 1 else sp.kron(sp.identity(p, format="csr"), C, format="csr")
            return self.lower(Y).apply(S)
        if Y.is_constant():
            C = Y.value
            C = sp.csr_matrix(C) if sp.issparse(C) else _dense_csr(np.asarray(C, dtype=float).reshape(k, p))
            S = sp.csr_matrix(C.T) if m == 1 else sp.kron(C.T, sp.identity(m, format="csr"), format="csr")
            return self.lower(X).apply(S)
        return self._lower_bilinear_matmul(e, m, k, p)

    def _lower_index(self, e):
        sel = self._idx(e.args[0])[e.key]
        return self.lower(e.args[0]).select(np.asarray(sel).reshape(-1, order="F"))

    _lower_special_index = _lower_index

    def _lower_Promote(self, e):
        return self.lower(e.args[0]).select(np.zeros(e.size, dtype=np.int64))

    def _lower_broadcast_to(self, e):
        sel = np.broadcast_to(self._idx(e.args[0]), e.shape).reshape(-1, order="F")
        return self.lower(e.args[0]).select(sel)

    def _lower_reshape(self, e):
        f = self.lower(e.args[0])
        if e.order == "F":
            return f
        sel = np.reshape(self._idx(e.args[0]), e.shape, order="C").reshape(-1, order="F")
        return f.select(sel)

    def _lower_transpose(self, e):
        sel = self._idx(e.args[0]).T.reshape(-1, order="F")
        return self.lower(e.args[0]).select(sel)

    def _lower_Sum(self, e):
        a = e.args[0]
        if e.axis is None or a.ndim == 0:
            tgt = np.zeros(a.size, dtype=np.int64)
        else:
            ax = e.axis if e.axis >= 0 else e.axis + a.ndim
            red_shape = tuple(d for i, d in enumerate(a.shape) if i != ax)
            idx_out = np.arange(e.size, dtype=np.int64).reshape(red_shape, order="F")
            tgt = np.broadcast_to(np.expand_dims(idx_out, ax), a.shape).reshape(-1, order="F")
        src = np.arange(a.size, dtype=np.int64)
        S = sp.csr_matrix((np.ones(src.size), (tgt, src)), shape=(e.size, a.size))
        return self.lower(a).apply(S)

    def _stack(self, e, fn):
        pieces, off = [], 0
        for a in e.args:
            pieces.append(off + self._idx(a))
            off += a.size
        sel = fn([np.atleast_1d(p) for p in pieces]).reshape(-1, order="F")
        forms = [self.lower(a) for a in e.args]
        return self.LF.vstack(forms).select(sel)

    def _lower_Hstack(self, e):
        return self._stack(e, np.hstack)

    def _lower_Vstack(self, e):
        return self._stack(e, np.vstack)

    # -- nonlinear atoms ------------------------------------------------------------
    def _gather(self, arg: Expression) -> np.ndarray:
        """x index of every entry of a nonlinear atom's argument.  After dnlp2smooth the
        argument is a Variable; pure selections of variables (index / reshape / promote of a
        variable) are accepted too."""
        idx = self.lower(arg).gather(self.N)
        if idx is None:
            raise ValueError("Argument of a nonlinear atom is not a bare variable; run "
                             "dnlp2smooth first (got %s)." % type(arg).__name__)
        return idx

    def _new_segment(self, seg: Segment, drow, dcol, hrow, hcol, hz):
        seg.zoff = self.Z
        seg.doff = self.nd
        seg.hoff = self.nh
        seg.dcount = len(drow)
        seg.hcount = len(hrow)
        self.Z += seg.zcount
        if self.Z > self.Zcap:
            raise AssertionError("atom output count exceeded the pre-pass estimate")
        self.nd += seg.dcount
        self.nh += seg.hcount
        self._d[0].append(np.asarray(drow, dtype=np.int64))
        self._d[1].append(np.asarray(dcol, dtype=np.int64))
        hrow = np.asarray(hrow, dtype=np.int64)
        hcol = np.asarray(hcol, dtype=np.int64)
        self._h[0].append(np.maximum(hrow, hcol))
        self._h[1].append(np.minimum(hrow, hcol))
        self._h[2].append(np.asarray(hz, dtype=np.int64))
        self.segments.append(seg)
        return seg

    def _lower_unary(self, e, op, param):
        a0 = self._gather(e.args[0])
        n = a0.size
        seg = Segment(op=op, n=n, a0=a0, a1=None, param=param, zcount=n)
        z = self.Z + np.arange(n)
        self._new_segment(seg, z, a0, a0, a0, z)
        return self._z_form(seg.zoff, n)

    def _lower_power(self, e):
        p = e.p_rational
        if p == 0:
            return self._const_form(np.ones(e.shape), e.size)
        if p == 1:
            return self.lower(e.args[0])
        # value uses float(p.value); derivatives use the rational approximation
        # (reference power.py:188 vs :410-419, :433-450).  They differ by < 1e-6 relative
        # only when p is not representable with denominator <= 1024.
        seg_form = self._lower_unary(e, OP_POWER, float(p))
        self.segments[-1].param2 = float(e.p_value)
        return seg_form

    def _lower_bilinear_elementwise(self, e):
        x, y = e.args
        a0 = self._gather(x)
        a1 = self._gather(y)
        if a0.size != e.size:
            a0 = np.broadcast_to(a0.reshape(x.shape, order="F"), e.shape).reshape(-1, order="F")
        if a1.size != e.size:
            a1 = np.broadcast_to(a1.reshape(y.shape, order="F"), e.shape).reshape(-1, order="F")
        if np.any(a0 == a1):
            raise ValueError("multiply of a variable with itself must be written as a power.")
        n = e.size
        seg = Segment(op=OP_MUL, n=n, a0=a0, a1=a1, zcount=n)
        z = self.Z + np.arange(n)
        self._new_segment(seg, np.concatenate([z, z]), np.concatenate([a0, a1]), a0, a1, z)
        return self._z_form(seg.zoff, n)

    def _lower_rel_entr(self, e):
        x, y = e.args
        a0 = self._gather(x)
        a1 = self._gather(y)
        if a0.size != e.size:
            a0 = np.broadcast_to(a0.reshape(x.shape, order="F"), e.shape).reshape(-1, order="F")
        if a1.size != e.size:
            a1 = np.broadcast_to(a1.reshape(y.shape, order="F"), e.shape).reshape(-1, order="F")
        n = e.size
        seg = Segment(op=OP_REL_ENTR, n=n, a0=a0, a1=a1, zcount=n)
        z = self.Z + np.arange(n)
        self._new_segment(seg, np.concatenate([z, z]), np.concatenate([a0, a1]),
                          np.concatenate([a0, a1, a0]), np.concatenate([a0, a1, a1]),
                          np.concatenate([z, z, z]))
        return self._z_form(seg.zoff, n)

    def _lower_bilinear_matmul(self, e, m, k, p):
        X, Y = e.args
        ax = self._gather(X).reshape((m, k), order="F")
        ay = self._gather(Y).reshape((k, p), order="F")
        if np.intersect1d(ax, ay).size:
            raise ValueError("matmul of an expression with itself is not supported.")
        n = m * p
        z = (self.Z + np.arange(n)).reshape((m, p), order="F")
        # entry order: for every output (i,j) [F-order], l = 0..k-1: dU then dV
        I, J, L = np.meshgrid(np.arange(m), np.arange(p), np.arange(k), indexing="ij")
        # F-order over (i,j), l fastest inside
        order = np.lexsort((L.reshape(-1), I.reshape(-1), J.reshape(-1)))
        I, J, L = I.reshape(-1)[order], J.reshape(-1)[order], L.reshape(-1)[order]
        zz = z[I, J]
        u_idx = ax[I, L]
        v_idx = ay[L, J]
        seg = Segment(op=OP_MATMUL, n=n, a0=ax.reshape(-1, order="F"),
                      a1=ay.reshape(-1, order="F"), zcount=n, dims=(m, k, p))
        self._new_segment(seg, np.concatenate([zz, zz]), np.concatenate([u_idx, v_idx]),
                          u_idx, v_idx, zz)
        return self._z_form(seg.zoff, n)

    def _lower_QuadForm(self, e):
        x, P = e.args
        a0 = self._gather(x)
        n = a0.size
        z = np.array([self.Z])
        if isinstance(P, Constant) and P.is_device:
            cid = len(self.dense_consts)
            self.dense_consts.append(DenseConst(n=n, device=P.device_matrix))
            dense = True
        else:
            Pv = P.value
            if sp.issparse(Pv):
                dense = False
                Pm = sp.csr_matrix(Pv)
            else:
                Pm = np.asarray(Pv, dtype=float)
                dense = n > DENSE_EXPAND_MAX_N
                if dense:
                    cid = len(self.dense_consts)
                    # quad_form's P is symmetric: the transposed view of a C-ordered P is the
                    # column-major matrix itself, no 8 n^2-byte copy
                    Pf = Pm.T if (Pm.flags.c_contiguous and not Pm.flags.f_contiguous
                                  and _is_symmetric(Pm)) else np.asfortranarray(Pm)
                    self.dense_consts.append(DenseConst(n=n, host=Pf))
                else:
                    Pm = sp.csr_matrix(Pm)
                    # keep explicit zeros out but make sure the pattern is symmetric
        if dense:
            if not (n == 1 or np.all(np.diff(a0) == 1)):
                raise ValueError("dense quad_form needs a contiguous variable argument.")
            seg = Segment(op=OP_QUAD_FORM_DENSE, n=n, a0=a0, a1=None, zcount=1, aux=cid)
            self._new_segment(seg, np.repeat(z, n), a0, [], [], [])
            self.dense_blocks.append({"seg": len(self.segments) - 1, "const": cid,
                                      "x0": int(a0[0]), "n": n, "z": int(z[0])})
            return self._z_form(seg.zoff, 1)
        # sparse / small P: second derivatives are per-entry tape work on sym(P) = P + P^T
        S = sp.coo_matrix(Pm + Pm.T)
        S.sum_duplicates()
        keep = S.row >= S.col
        r, c, v = S.row[keep], S.col[keep], S.data[keep]
        cid = len(self.sparse_consts)
        self.sparse_consts.append((sp.csr_matrix(Pm), r.astype(np.int64), c.astype(np.int64),
                                   np.asarray(v, dtype=float)))
        seg = Segment(op=OP_QUAD_FORM_SPARSE, n=n, a0=a0, a1=None, zcount=1, aux=cid)
        self._new_segment(seg, np.repeat(z, n), a0, a0[r], a0[c], np.repeat(z, r.size))
        return self._z_form(seg.zoff, 1)

    def _lower_quad_over_lin(self, e):
        x, y = e.args
        a0 = self._gather(x)
        a1 = self._gather(y)
        n = a0.size
        z = np.array([self.Z])
        seg = Segment(op=OP_QUAD_OVER_LIN, n=n, a0=a0, a1=a1, zcount=1)
        # d: n entries d/dx_i then 1 entry d/dy ; h: n (xx diag), 1 (yy), n (xy)
        self._new_segment(seg, np.repeat(z, n + 1), np.concatenate([a0, a1]),
                          np.concatenate([a0, a1, a0]),
                          np.concatenate([a0, a1, np.repeat(a1, n)]),
                          np.repeat(z, 2 * n + 1))
        return self._z_form(seg.zoff, 1)


def _is_symmetric(P: np.ndarray) -> bool:
    """Exact symmetry test, tile against mirrored tile (cache-sized, no n x n temporary).  Large constants go to the
    threaded tile loop behind the C ABI (dnlp_is_symmetric): BASELINE C3's 1e4 x 1e4 block 5-18 ms on the GPU box's
    host against 62 ms for the numpy loop below (build host: 40 against 130 ms; tools/micro/sym_check_time.py) — it
    was the largest single cost of that problem's lowering."""
    n, b = P.shape[0], 256
    if n >= 1024 and P.dtype == np.float64 and P.flags.c_contiguous and os.environ.get("DNLP_LOWER_CXX", "1") != "0":
        try:
            from . import _capi
            fn = _capi.load().lib.dnlp_is_symmetric
            fn.argtypes = [C.POINTER(C.c_double), C.c_int64, C.c_int64]
            return bool(fn(P.ctypes.data_as(C.POINTER(C.c_double)), n, n))
        except Exception:
            pass
    for i in range(0, n, b):
        for j in range(0, i + 1, b):
            if not np.array_equal(P[i:i + b, j:j + b], P[j:j + b, i:i + b].T):
                return False
    return True


def _dense_csr(C: np.ndarray) -> sp.csr_matrix:
    """CSR of a dense constant; a matrix without zero entries is wrapped without a search."""
    C = np.ascontiguousarray(C, dtype=float)
    r, k = C.shape
    if C.size and np.count_nonzero(C) == C.size:
        idx_t = np.int32 if C.size < 2 ** 31 - 1 else np.int64
        return sp.csr_matrix((C.reshape(-1), np.tile(np.arange(k, dtype=idx_t), r),
                              np.arange(0, C.size + 1, k, dtype=idx_t)), shape=(r, k))
    return sp.csr_matrix(C)


def _coo_unique(keys, return_first=False):
    """Sorted unique keys and the position of every input key among them (and, on request, the index of the
    first input key of every unique one)."""
    keys = np.asarray(keys)
    if keys.size < 2 or bool(np.all(keys[1:] > keys[:-1])):
        # already strictly increasing (pure CSR / dense lower-triangle patterns): nothing to merge
        ident = np.arange(keys.size, dtype=np.int64)
        return (keys, ident, ident) if return_first else (keys, ident)
    if return_first:
        uniq, first, inv = np.unique(keys, return_index=True, return_inverse=True)
        return uniq, inv.astype(np.int64), first
    uniq, inv = np.unique(keys, return_inverse=True)
    return uniq, inv.astype(np.int64)


def _maps_numpy(N, Z, m, G, c, drow, dcol, hrow, hcol, listed, single_form=None):
    """The constant maps of a lowered problem in numpy / scipy: what csrc/lower_maps.h computes (same arrays)."""
    if not G.has_canonical_format:
        # sum_duplicates / sort_indices rewrite the arrays in place; with a single constraint block and
        # shared column ranges (head_cols, LinForm.apply) those arrays can still be a constant the user holds
        if single_form is not None and np.shares_memory(G.data, single_form.data):
            G = G.copy()
        G.sum_duplicates()
        G.sort_indices()
    nd, nh = drow.size, hrow.size

    def head_cols(A, k):
        A = sp.csr_matrix(A)
        if A.shape[1] == k:
            return A
        if A.nnz == 0 or int(A.indices.max()) < k:
            return sp.csr_matrix((A.data, A.indices, A.indptr), shape=(A.shape[0], k))
        return sp.csr_matrix(A[:, :k])

    cz = c[N:]
    if Z:
        Gx = head_cols(G, N)
        Gz = sp.csr_matrix(G[:, N:]) if Gx.nnz != G.nnz else sp.csr_matrix((m, Z))
    else:                                    # no intermediate values: G is Gx
        Gx = G
        Gz = sp.csr_matrix((m, 0))

    # gradient map: grad = c_x + Mg @ dvals
    coef = cz[drow] if nd else np.zeros(0)
    nzm = coef != 0
    Mg = sp.csr_matrix((coef[nzm], (dcol[nzm], np.nonzero(nzm)[0])), shape=(N, nd))

    # multiplier pull-back: w = Mw @ [sigma; lambda]
    Mw = sp.hstack([sp.csr_matrix(cz.reshape(-1, 1)), Gz.T], format="csr") if Z else \
        sp.csr_matrix((0, 1 + m))

    # Jacobian: J = Gx + Gz @ D ; positions are row-major sorted unique (row, col)
    E = sp.csr_matrix((np.ones(nd), (drow, np.arange(nd))), shape=(Z, nd))
    Cm = sp.coo_matrix(Gz @ E) if (m and nd) else sp.coo_matrix((m, nd))
    gx = sp.coo_matrix(Gx)
    if Cm.nnz == 0:
        # only the affine part: G was canonicalised above (sum_duplicates + sort_indices), so its entries ARE
        # the row-major sorted unique pattern -- no keys, no divisions (2 s of int64 work at BASELINE C3's 1e7)
        nnzJ = gx.nnz
        jac_rows = gx.row.astype(np.int32, copy=False)
        jac_cols = gx.col.astype(np.int32, copy=False)
        Jc = np.asarray(gx.data, dtype=np.float64)
        MJ = sp.csr_matrix((np.zeros(0), np.zeros(0, np.int32), np.zeros(nnzJ + 1, np.int32)), shape=(nnzJ, nd))
    else:
        rows_all = np.concatenate([gx.row.astype(np.int64), Cm.row.astype(np.int64)])
        cols_all = np.concatenate([gx.col.astype(np.int64), dcol[Cm.col]])
        uniq, inv, first = _coo_unique(rows_all * N + cols_all, return_first=True)
        nnzJ = uniq.size
        jac_rows = rows_all[first].astype(np.int32)
        jac_cols = cols_all[first].astype(np.int32)
        # (np.add.at is an order of magnitude slower than bincount)
        Jc = np.bincount(inv[:gx.nnz], weights=gx.data, minlength=nnzJ) if gx.nnz else np.zeros(nnzJ)
        MJ = sp.csr_matrix((Cm.data, (inv[gx.nnz:], Cm.col)), shape=(nnzJ, nd))
        MJ.sum_duplicates()

    # Hessian: lower-oriented positions, row-major sorted unique
    hkeys = hrow * N + hcol
    dense_pos_keys = []
    for x0, nb in listed:
        ii, jj = np.tril_indices(nb)
        dense_pos_keys.append((x0 + ii).astype(np.int64) * N + (x0 + jj))
    allkeys = np.concatenate([hkeys] + dense_pos_keys) if (nh or dense_pos_keys) else \
        np.zeros(0, np.int64)
    huniq, hinv = _coo_unique(allkeys)
    nnzH = huniq.size
    hess_rows = (huniq // N).astype(np.int32) if N else np.zeros(0, np.int32)
    hess_cols = (huniq % N).astype(np.int32) if N else np.zeros(0, np.int32)
    MH = sp.csr_matrix((np.ones(nh), (hinv[:nh], np.arange(nh))), shape=(nnzH, nh))
    off = nh
    blocks = []
    for x0, nb in listed:
        cnt = nb * (nb + 1) // 2
        pos = hinv[off:off + cnt]                               # tril_indices order
        if cnt and int(pos[-1]) - int(pos[0]) == cnt - 1 and bool(np.all(pos[1:] > pos[:-1])):
            # the block's entries are a contiguous run of the sorted pattern (the usual case: the
            # quad_form block is the only Hessian contribution of its rows): base + q, no table
            blocks.append((2, np.array([int(pos[0])], dtype=np.int64)))
        else:
            blocks.append((1, pos.astype(np.int64)))
        off += cnt
    return {"G": G,
            "Mg": Mg, "Mw": Mw, "MJ": MJ, "MH": MH, "Jc": Jc, "jac_rows": jac_rows, "jac_cols": jac_cols,
            "hess_rows": hess_rows, "hess_cols": hess_cols, "blocks": blocks}


def lower_problem(objective_expr: Expression, constraint_exprs: List[Expression],
                  variables: List[Variable]) -> Tape:
    """Flatten objective + constraint expressions (already smooth-canonical, constraints
    already lowered to `expr == 0` / `expr >= 0` residual form) over `variables`."""
    lw = Lowerer(variables, [objective_expr] + list(constraint_exprs))
    fobj = lw.lower(objective_expr)
    forms = [lw.lower(c) for c in constraint_exprs]
    N, Z = lw.N, lw.Z
    ncol = N + Z

    def trim(A):
        """A[:, :ncol] as CSR; when no entry lies beyond column ncol the arrays are shared (no 1e7-entry copy)."""
        from ._capi import CsrArrays
        if A.shape[1] == ncol:
            return A
        if A.nnz == 0 or int(A.indices.max()) < ncol:
            if isinstance(A, CsrArrays):
                return CsrArrays(A.indptr, A.indices, A.data, (A.shape[0], ncol))
            return sp.csr_matrix((A.data, A.indices, A.indptr), shape=(A.shape[0], ncol))
        return sp.csr_matrix(A.tocsr()[:, :ncol])

    fA, fb = fobj.csr()
    c = np.asarray(trim(fA).tocsr().todense()).reshape(-1) if fA.nnz else np.zeros(ncol)
    c0 = float(fb[0])
    single_A = None
    if forms:
        GA, b = lw.LF.vstack(forms).csr()
        if len(forms) == 1:
            single_A = GA
        G = trim(GA)
    else:
        G = sp.csr_matrix((0, ncol))
        b = np.zeros(0)
    m = G.shape[0]
    cat = lambda lst, dt: (np.concatenate(lst).astype(dt) if lst else np.zeros(0, dt))  # noqa
    drow, dcol = cat(lw._d[0], np.int64), cat(lw._d[1], np.int64)
    hrow, hcol, hz = cat(lw._h[0], np.int64), cat(lw._h[1], np.int64), cat(lw._h[2], np.int64)
    # The constant maps (canonical G, Mg, Mw, MJ / Jc, MH, the two patterns) are built in C++ behind the C ABI
    # (csrc/lower_maps.h, dnlp_lower_maps: the role of cvxcore's build_matrix); the numpy / scipy construction
    # below is the same computation, kept as the fallback where the library is not built and as the checker
    # (tests/test_lower_maps.py compares every array of the two).  DNLP_LOWER_CXX=0 forces it.
    listed = [(blk["x0"], blk["n"]) for blk in lw.dense_blocks if blk["n"] <= DENSE_COO_MAX_N]
    maps = None
    if os.environ.get("DNLP_LOWER_CXX", "1") != "0":
        from . import _capi
        maps = _capi.lower_maps(N, Z, m, G, c, drow, dcol, hrow, hcol, listed)
    if maps is None:
        maps = _maps_numpy(N, Z, m, G.tocsr(), c, drow, dcol, hrow, hcol, listed,
                           single_form=None if single_A is None else single_A.tocsr())
    if maps["G"] is not None:
        G = maps["G"]
    Mg, Mw, MJ, MH, Jc = maps["Mg"], maps["Mw"], maps["MJ"], maps["MH"], maps["Jc"]
    jac_rows, jac_cols, hess_rows, hess_cols = maps["jac_rows"], maps["jac_cols"], maps["hess_rows"], maps["hess_cols"]
    hess_coo_complete = True
    it = iter(maps["blocks"])
    for blk in lw.dense_blocks:
        if blk["n"] <= DENSE_COO_MAX_N:
            mode, pos = next(it)
            blk["coo_pos"] = np.asarray(pos, dtype=np.int64)
            if mode == 2:
                blk["coo_pos_identity"] = True
        else:
            blk["coo_pos"] = None
            hess_coo_complete = False

    return Tape(N=N, m=m, Z=Z, segments=lw.segments, dense_consts=lw.dense_consts,
                sparse_consts=lw.sparse_consts, c0=c0, c=c, G=G, b=b, drow=drow, dcol=dcol,
                hrow=hrow, hcol=hcol, hz=hz, Mg=Mg, Mw=Mw, MJ=MJ, Jc=Jc, jac_rows=jac_rows,
                jac_cols=jac_cols, MH=MH, hess_rows=hess_rows, hess_cols=hess_cols,
                dense_blocks=lw.dense_blocks, hess_coo_complete=hess_coo_complete,
                var_offsets=dict(lw.var_offsets))
