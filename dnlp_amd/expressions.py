"""Expression tree for the nlp=True path: leaves, operator overloading, DNLP rule engine.

Host-side mirror of the part of the reference's DSL that `Problem.solve(nlp=True)` touches
(reference: cvxpy/expressions/expression.py:300-400,600-830 for rules and operators,
cvxpy/expressions/{variable,leaf}.py, cvxpy/expressions/constants/constant.py).  Only what
the disciplined-NLP path needs is here: shapes, signs, curvature/ESR/HSR composition,
numeric forward values (F-order everywhere) and the tree structure that
`dnlp_amd.dnlp2smooth` rewrites and `dnlp_amd.lowering` flattens into the device tape.

Derivatives are NOT computed on these objects at solve time; they are lowered once to the
device tape (see lowering.py); the tape's numpy restatement used by the tests is
`oracle/tape_eval.py` (test infrastructure, never imported from this package).
"""
from __future__ import annotations

import itertools
from typing import List, Optional, Tuple

import numpy as np

from . import settings as s

_id_counter = itertools.count(1)


def get_id() -> int:
    return next(_id_counter)


def size_from_shape(shape) -> int:
    # (shapes are tuples of at most a few ints: a Python product is 20x cheaper than np.prod, and the
    # lowering asks for sizes thousands of times)
    n = 1
    for d in shape:
        n *= int(d)
    return n


def unique_list(items):
    seen = set()
    out = []
    for it in items:
        if id(it) not in seen:
            seen.add(id(it))
            out.append(it)
    return out


def _cast_other(fn):
    def wrapped(self, other):
        return fn(self, Expression.cast_to_const(other))
    wrapped.__name__ = fn.__name__
    return wrapped


class Expression:
    """Base of every node.  Subclasses provide `args`, `shape` and the rule predicates."""

    __array_priority__ = 100   # numpy defers to our reflected operators
    args: List["Expression"] = []

    # ---- structure ---------------------------------------------------------------
    @property
    def shape(self) -> Tuple[int, ...]:
        raise NotImplementedError

    @property
    def size(self) -> int:
        return size_from_shape(self.shape)

    @property
    def ndim(self) -> int:
        return len(self.shape)

    def is_scalar(self) -> bool:
        return all(d == 1 for d in self.shape)

    def is_vector(self) -> bool:
        return self.ndim <= 1 or (self.ndim == 2 and min(self.shape) == 1)

    def is_matrix(self) -> bool:
        return self.ndim == 2 and self.shape[0] > 1 and self.shape[1] > 1

    def variables(self) -> List["Variable"]:
        """Leaf variables in depth-first argument order, first occurrence kept
        (reference: utilities/canonical.py variables() + unique_list)."""
        return unique_list([v for a in self.args for v in a.variables()])

    def parameters(self):
        return unique_list([p for a in self.args for p in a.parameters()])

    def name(self) -> str:
        return f"{type(self).__name__}({', '.join(a.name() for a in self.args)})"

    def __repr__(self):
        return f"Expression({self.curvature}, {self.sign}, {self.shape})"

    def __str__(self):
        return self.name()

    # ---- values ------------------------------------------------------------------
    @property
    def value(self):
        return self._value_impl()

    def _value_impl(self):
        raise NotImplementedError

    # ---- sign --------------------------------------------------------------------
    def is_nonneg(self) -> bool:
        raise NotImplementedError

    def is_nonpos(self) -> bool:
        raise NotImplementedError

    def is_zero(self) -> bool:
        return self.is_nonneg() and self.is_nonpos()

    @property
    def sign(self) -> str:
        if self.is_zero():
            return s.ZERO
        if self.is_nonneg():
            return s.NONNEG
        if self.is_nonpos():
            return s.NONPOS
        return s.UNKNOWN

    # ---- curvature / DNLP rules (reference expression.py:316-383) -----------------
    def is_constant(self) -> bool:
        return 0 in self.shape or all(a.is_constant() for a in self.args)

    def is_affine(self) -> bool:
        return self.is_constant() or (self.is_convex() and self.is_concave())

    def is_smooth(self) -> bool:
        return self.is_constant() or (self.is_esr() and self.is_hsr())

    def is_convex(self) -> bool:
        raise NotImplementedError

    def is_concave(self) -> bool:
        raise NotImplementedError

    def is_esr(self) -> bool:
        raise NotImplementedError

    def is_hsr(self) -> bool:
        raise NotImplementedError

    def is_dcp(self) -> bool:
        return self.is_convex() or self.is_concave()

    def is_dnlp(self) -> bool:
        return self.is_esr() or self.is_hsr()

    @property
    def curvature(self) -> str:
        if self.is_constant():
            return "CONSTANT"
        if self.is_affine():
            return "AFFINE"
        if self.is_convex():
            return "CONVEX"
        if self.is_concave():
            return "CONCAVE"
        return "UNKNOWN"

    def is_complex(self) -> bool:
        return False

    # ---- operators (reference expression.py:600-830) ------------------------------
    @staticmethod
    def cast_to_const(expr):
        if isinstance(expr, list):
            for elem in expr:
                if isinstance(elem, Expression):
                    raise ValueError(
                        "The input must be a single Expression, not a list. "
                        "Combine Expressions using atoms such as hstack and vstack.")
        return expr if isinstance(expr, Expression) else Constant(expr)

    cast = cast_to_const

    @staticmethod
    def broadcast(lh_expr, rh_expr):
        """Operand pair of a binary elementwise operator, brought to a common shape.  Same resulting
        trees as the reference's rule (expression.py:680-715): a scalar next to a non-scalar is
        promoted; two matrices are stretched along their unit dimensions by products with vectors of
        ones (rows first, then columns); operands of different rank or rank >= 3 go through
        broadcast_to."""
        from . import atoms as at
        pair = [Expression.cast_to_const(lh_expr), Expression.cast_to_const(rh_expr)]
        scalar = [e.is_scalar() for e in pair]
        if all(scalar):
            return pair[0], pair[1]
        for k in (0, 1):
            if scalar[k]:
                pair[k] = at.promote(pair[k], pair[1 - k].shape)
        if pair[0].ndim == 2 and pair[1].ndim == 2:
            for axis in (0, 1):
                full = max(pair[0].shape[axis], pair[1].shape[axis])
                for k in (0, 1):
                    if pair[k].shape[axis] == 1 and full > 1:
                        ones = np.ones((full, 1)) if axis == 0 else np.ones((1, full))
                        pair[k] = ones @ pair[k] if axis == 0 else pair[k] @ ones
        elif pair[0].ndim != pair[1].ndim or max(pair[0].ndim, pair[1].ndim) >= 3:
            target = np.broadcast_shapes(pair[0].shape, pair[1].shape)
            pair = [e if e.shape == target else at.broadcast_to(e, target) for e in pair]
        return pair[0], pair[1]

    def __getitem__(self, key):
        from . import atoms as at
        if isinstance(key, tuple) and len(key) == 0:
            return self
        if at.is_special_slice(key):
            return at.special_index(self, key)
        return at.index(self, key)

    @property
    def T(self):
        from . import atoms as at
        if self.ndim <= 1:
            return self
        return at.transpose(self)

    def flatten(self, order="F"):
        from . import atoms as at
        return at.vec(self, order)

    def __pow__(self, power):
        from . import atoms as at
        return at.power(self, power)

    def __rpow__(self, base):
        raise NotImplementedError("Variables on the right side of ** are not supported; "
                                  "use exp(multiply(log(a), x)).")

    @_cast_other
    def __add__(self, other):
        from . import atoms as at
        if isinstance(other, Constant) and other.is_zero():
            return self
        lhs, rhs = self.broadcast(self, other)
        return at.AddExpression([lhs, rhs])

    @_cast_other
    def __radd__(self, other):
        if isinstance(other, Constant) and other.is_zero():
            return self
        return other + self

    @_cast_other
    def __sub__(self, other):
        return self + -other

    @_cast_other
    def __rsub__(self, other):
        return other - self

    @_cast_other
    def __mul__(self, other):
        from . import atoms as at
        if self.shape == () or other.shape == ():
            return at.multiply(self, other)
        if self.shape[-1] != other.shape[0] and (self.is_scalar() or other.is_scalar()):
            return at.multiply(self, other)
        return at.MulExpression(self, other)

    @_cast_other
    def __rmul__(self, other):
        return other * self

    @_cast_other
    def __matmul__(self, other):
        from . import atoms as at
        if self.shape == () or other.shape == ():
            raise ValueError("Scalar operands are not allowed, use '*' instead")
        if isinstance(self, at.MulExpression) and not isinstance(self, at.multiply):
            # x.T @ A @ x with constant A and the same x on both sides is a QuadForm
            # (reference expression.py:795-801).
            if self.args[0] is other and not other.is_constant() and self.args[1].is_constant():
                return at.QuadForm(other, self.args[1])
        return at.MulExpression(self, other)

    @_cast_other
    def __rmatmul__(self, other):
        from . import atoms as at
        if self.shape == () or other.shape == ():
            raise ValueError("Scalar operands are not allowed, use '*' instead")
        return at.MulExpression(other, self)

    @_cast_other
    def __truediv__(self, other):
        from . import atoms as at
        lhs, rhs = self.broadcast(self, other)
        if (lhs.is_scalar() or rhs.is_scalar()) or rhs.shape == lhs.shape:
            return at.DivExpression(lhs, rhs)
        raise ValueError("Incompatible shapes for division (%s / %s)" % (lhs.shape, rhs.shape))

    @_cast_other
    def __rtruediv__(self, other):
        return other / self

    def __neg__(self):
        from . import atoms as at
        return at.NegExpression(self)

    @_cast_other
    def __eq__(self, other):
        from .constraints import Equality
        return Equality(self, other)

    @_cast_other
    def __le__(self, other):
        from .constraints import Inequality
        return Inequality(self, other)

    @_cast_other
    def __ge__(self, other):
        from .constraints import Inequality
        return Inequality(other, self)

    def __lt__(self, other):
        raise NotImplementedError("Strict inequalities are not allowed.")

    __gt__ = __lt__

    __hash__ = object.__hash__

    # numpy interop: ndarray <op> Expression is routed to our reflected operators
    __array_ufunc__ = None


# =====================================================================================
# Leaves
# =====================================================================================
class Leaf(Expression):
    """A leaf node: Variable, Constant or Parameter (reference expressions/leaf.py)."""

    def __init__(self, shape, value=None, nonneg=False, nonpos=False, bounds=None):
        if isinstance(shape, (int, np.integer)):
            shape = (int(shape),)
        self._shape = tuple(int(d) for d in shape)
        for d in self._shape:
            if d < 0:
                raise ValueError("Invalid dimensions %s." % (shape,))
        self.args = []
        self.attributes = {"nonneg": bool(nonneg), "nonpos": bool(nonpos)}
        self.bounds = self._ensure_valid_bounds(bounds)
        self.attributes["bounds"] = self.bounds
        self._value = None
        if value is not None:
            self.value = value

    @property
    def shape(self):
        return self._shape

    def variables(self):
        return []

    def parameters(self):
        return []

    def copy(self, args=None):
        return self

    def is_convex(self):
        return True

    def is_concave(self):
        return True

    def is_esr(self):
        return True

    def is_hsr(self):
        return True

    def is_nonneg(self):
        # sign comes from the nonneg/pos attribute only, not from `bounds`
        # (reference leaf.py:270-277)
        return self.attributes["nonneg"]

    def is_nonpos(self):
        return self.attributes["nonpos"]

    def _ensure_valid_bounds(self, value):
        """Promote [lb, ub] (None / scalar / array each) to two arrays of the leaf's shape
        (reference leaf.py:647-685)."""
        if value is None:
            return None
        if not hasattr(value, "__len__") or len(value) != 2:
            raise ValueError("Bounds should be a list of two items.")
        value = list(value)
        none_bounds = [-np.inf, np.inf]
        for idx in range(2):
            if value[idx] is None:
                value[idx] = np.full(self._shape, none_bounds[idx])
            else:
                arr = np.asarray(value[idx], dtype=float)
                if arr.shape != self._shape:
                    if arr.ndim == 0:
                        arr = np.full(self._shape, float(arr))
                    else:
                        raise ValueError("Bounds must be scalars or arrays matching the "
                                         "variable's shape.")
                value[idx] = arr
        if np.any(value[0] > value[1]):
            raise ValueError("Invalid bounds: some upper bounds are less than "
                             "corresponding lower bounds.")
        if np.any(np.isnan(value[0])) or np.any(np.isnan(value[1])):
            raise ValueError("np.nan is not feasible as lower or upper bound.")
        return value

    @property
    def value(self):
        return self._value

    @value.setter
    def value(self, val):
        self.save_value(self._validate_value(val))

    def save_value(self, val):
        self._value = val

    def _value_impl(self):
        return self._value

    def _validate_value(self, val):
        if val is None:
            return None
        val = np.asarray(val, dtype=float)
        if val.shape != self._shape:
            if val.size == self.size and (val.ndim <= 1 or self.ndim <= 1):
                val = val.reshape(self._shape)
            else:
                raise ValueError("Invalid dimensions %s for %s value." %
                                 (val.shape, type(self).__name__))
        if self.ndim == 0:
            return np.asarray(float(val))
        return val


class Variable(Leaf):
    """Optimization variable (reference expressions/variable.py:30-80)."""

    def __init__(self, shape=(), name: Optional[str] = None, var_id: Optional[int] = None,
                 nonneg=False, nonpos=False, bounds=None, pos=False, neg=False, **kwargs):
        unsupported = [k for k, v in kwargs.items() if v]
        if unsupported:
            raise NotImplementedError(
                "Variable attributes %s are not part of the nlp=True path." % unsupported)
        self.id = get_id() if var_id is None else var_id
        self._name = name if name is not None else "var%d" % self.id
        # bounds for sampling initial points (reference variable.py:52-53)
        self.sample_bounds = None
        super().__init__(shape, nonneg=nonneg or pos, nonpos=nonpos or neg, bounds=bounds)

    def name(self):
        return self._name

    def variables(self):
        return [self]

    def is_constant(self):
        return False

    def __repr__(self):
        return "Variable(%s, %s)" % (self.shape, self._name)


class Constant(Leaf):
    """Numeric constant (reference expressions/constants/constant.py)."""

    def __init__(self, value, name: Optional[str] = None):
        import scipy.sparse as sp
        if isinstance(value, DeviceMatrix):
            self._device = value
            arr = None
            shape = value.shape
        else:
            self._device = None
            if sp.issparse(value):
                arr = value.tocsr().astype(float)
                shape = arr.shape
            else:
                arr = np.asarray(value, dtype=float)
                shape = arr.shape
        self._name = name
        self.id = get_id()
        self._shape = tuple(int(d) for d in shape)
        self.args = []
        self.attributes = {"nonneg": False, "nonpos": False, "bounds": None}
        self.bounds = None
        self._value = arr
        self._sign_cache = None

    @property
    def is_device(self) -> bool:
        return self._device is not None

    @property
    def device_matrix(self):
        return self._device

    @property
    def is_sparse(self) -> bool:
        import scipy.sparse as sp
        return sp.issparse(self._value)

    def name(self):
        if self._name is not None:
            return self._name
        if self.is_device:
            return "DeviceMatrix%s" % (self.shape,)
        if self.size == 1 and not self.is_sparse:
            return str(float(np.asarray(self._value).reshape(-1)[0]))
        return "Constant%s" % (self.shape,)

    def is_constant(self):
        return True

    @property
    def value(self):
        return self._value

    @value.setter
    def value(self, val):
        raise AttributeError("Cannot set the value of a Constant.")

    def _signs(self):
        if self._sign_cache is None:
            import scipy.sparse as sp
            if self.is_device:
                self._sign_cache = (False, False)
            elif sp.issparse(self._value):
                d = self._value.data
                self._sign_cache = (bool(np.all(d >= 0)), bool(np.all(d <= 0)))
            else:
                v = self._value
                self._sign_cache = (bool(np.all(v >= 0)), bool(np.all(v <= 0)))
        return self._sign_cache

    def is_nonneg(self):
        return self._signs()[0]

    def is_nonpos(self):
        return self._signs()[1]

    def __repr__(self):
        return "Constant(%s, %s)" % (self.sign, self.shape)


class Parameter(Leaf):
    """Named constant whose value may change between solves (reference
    expressions/constants/parameter.py).  On the nlp=True path a parameter is a constant at
    lowering time: re-solving after `p.value = ...` re-lowers the tape with the new value, which
    is what the reference does too (it re-canonicalises on every solve, problem.py:1243)."""

    def __init__(self, shape=(), name: Optional[str] = None, value=None, nonneg=False, nonpos=False):
        self.id = get_id()
        self._name = name if name is not None else "param%d" % self.id
        super().__init__(shape, value=value, nonneg=nonneg, nonpos=nonpos)

    def name(self):
        return self._name

    def parameters(self):
        return [self]

    def is_constant(self):
        return True

    @property
    def is_device(self):
        return False

    def __repr__(self):
        return "Parameter(%s, %s)" % (self.shape, self._name)


class DeviceMatrix:
    """Handle to a dense FP64 column-major matrix resident in MI355X HBM.

    Used for constants too large to travel through the host (BASELINE config C4: the
    n=1e5 quad_form matrix is 80 GB).  `ptr` is a raw device address owned by the creator
    (e.g. `dnlp_amd.device.symmetric_test_matrix`); the tape references it by handle.
    """

    def __init__(self, ptr: int, rows: int, cols: int, ld: Optional[int] = None, owner=None,
                 symmetric: bool = False):
        self.ptr = int(ptr)
        self.shape = (int(rows), int(cols))
        self.ld = int(ld if ld is not None else rows)
        self.owner = owner
        self.symmetric = symmetric
