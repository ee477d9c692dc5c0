"""Atoms of the nlp=True path: forward values, shapes, signs and DNLP rule tags.

Host-side mirror of the reference atom set that carries `_jacobian/_hess_vec` rules or a
dnlp2smooth canonicaliser (reference cvxpy/atoms/**).  Every atom
keeps the reference's class name, argument meaning, F-order vectorisation and rule tags
(`is_atom_esr/hsr`, `is_incr/is_decr`, `is_atom_convex/concave`, `sign_from_args`).
Derivative *values* are not computed here at solve time: `lowering.py` turns the
canonicalised tree into the flat device tape once (affine chains folded to CSR blocks,
nonlinear atoms to tape segments) and the HIP kernels evaluate f/∇f/g/J/∇²L.
"""
from __future__ import annotations

import numbers
from fractions import Fraction
from typing import List, Tuple

import numpy as np
import scipy.sparse as sp
import scipy.special as special

from .expressions import Constant, Expression, Variable, size_from_shape

# =====================================================================================
# Atom base
# =====================================================================================


class Atom(Expression):
    """Base class of function nodes (reference atoms/atom.py:36-300)."""

    def __init__(self, *args):
        if len(args) == 0:
            raise TypeError("No arguments given to %s." % type(self).__name__)
        self.args = [Expression.cast_to_const(a) for a in args]
        self._cache = {}
        self.validate_arguments()
        self._shape = tuple(int(d) for d in self.shape_from_args())
        if len(self._shape) > 2:
            raise ValueError("Atoms must be at most 2D.")
        self.id = None

    # -- structure -----------------------------------------------------------------
    def validate_arguments(self):
        pass

    def shape_from_args(self) -> Tuple[int, ...]:
        raise NotImplementedError

    @property
    def shape(self):
        return self._shape

    def get_data(self):
        return None

    def copy(self, args=None):
        if args is None:
            args = self.args
        data = self.get_data()
        if data is not None:
            return type(self)(*(list(args) + list(data)))
        return type(self)(*args)

    def name(self):
        data = self.get_data() or []
        return "%s(%s)" % (type(self).__name__,
                           ", ".join([a.name() for a in self.args] + [str(d) for d in data]))

    def _memo(self, key, fn):
        if key not in self._cache:
            self._cache[key] = fn()
        return self._cache[key]

    # -- sign ----------------------------------------------------------------------
    def sign_from_args(self) -> Tuple[bool, bool]:
        raise NotImplementedError

    def is_nonneg(self):
        return self._memo("nn", lambda: bool(self.sign_from_args()[0]))

    def is_nonpos(self):
        return self._memo("np", lambda: bool(self.sign_from_args()[1]))

    # -- atom-level tags (defaults as in reference expression.py:385-420) ------------
    def is_atom_convex(self):
        return True

    def is_atom_concave(self):
        return True

    def is_atom_affine(self):
        return self.is_atom_concave() and self.is_atom_convex()

    def is_atom_esr(self):
        return True

    def is_atom_hsr(self):
        return True

    def is_incr(self, idx):
        return True

    def is_decr(self, idx):
        return False

    # -- composition rules (reference atoms/atom.py:196-273) -------------------------
    def is_convex(self):
        return self._memo("cvx", self._is_convex)

    def _is_convex(self):
        if self.is_constant():
            return True
        if self.is_atom_convex():
            for idx, arg in enumerate(self.args):
                if not (arg.is_affine() or (arg.is_convex() and self.is_incr(idx)) or
                        (arg.is_concave() and self.is_decr(idx))):
                    return False
            return True
        return False

    def is_concave(self):
        return self._memo("ccv", self._is_concave)

    def _is_concave(self):
        if self.is_constant():
            return True
        if self.is_atom_concave():
            for idx, arg in enumerate(self.args):
                if not (arg.is_affine() or (arg.is_concave() and self.is_incr(idx)) or
                        (arg.is_convex() and self.is_decr(idx))):
                    return False
            return True
        return False

    def is_esr(self):
        return self._memo("esr", self._is_esr)

    def _is_esr(self):
        if self.is_constant():
            return True
        if self.is_atom_esr():
            for idx, arg in enumerate(self.args):
                if not (arg.is_smooth() or (arg.is_esr() and self.is_incr(idx)) or
                        (arg.is_hsr() and self.is_decr(idx))):
                    return False
            return True
        return False

    def is_hsr(self):
        return self._memo("hsr", self._is_hsr)

    def _is_hsr(self):
        if self.is_constant():
            return True
        if self.is_atom_hsr():
            for idx, arg in enumerate(self.args):
                if not (arg.is_smooth() or (arg.is_hsr() and self.is_incr(idx)) or
                        (arg.is_esr() and self.is_decr(idx))):
                    return False
            return True
        return False

    def is_constant(self):
        return self._memo("const", lambda: Expression.is_constant(self))

    def is_affine(self):
        return self._memo("aff", lambda: Expression.is_affine(self))

    # -- values --------------------------------------------------------------------
    def numeric(self, values):
        raise NotImplementedError

    def _value_impl(self):
        if 0 in self.shape:
            return np.array([])
        vals = []
        for arg in self.args:
            v = arg._value_impl()
            if v is None and not self.is_constant():
                return None
            vals.append(v)
        out = self.numeric(vals)
        if sp.issparse(out):
            out = out.toarray()
        out = np.asarray(out, dtype=float)
        if out.shape != self.shape and out.size == self.size:
            out = out.reshape(self.shape, order="F")
        return out

    def point_in_domain(self):
        return np.ones(self.shape)


def _dense(v):
    return v.toarray() if sp.issparse(v) else np.asarray(v, dtype=float)


# =====================================================================================
# Affine atoms
# =====================================================================================
class AffAtom(Atom):
    """Affine atom: ESR and HSR, sign by interval arithmetic
    (reference atoms/affine/affine_atom.py:30-100)."""

    def sign_from_args(self):
        # default: if all args have the same sign so does the output
        return (all(a.is_nonneg() for a in self.args), all(a.is_nonpos() for a in self.args))


class AddExpression(AffAtom):
    """Sum of any number of expressions (reference atoms/affine/add_expr.py)."""

    def __init__(self, arg_groups):
        self._arg_groups = arg_groups
        flat = []
        for g in arg_groups:
            flat += self.expand_args(g)
        super().__init__(*flat)

    @staticmethod
    def expand_args(expr):
        # nested sums are flattened (reference add_expr.py:52-58)
        if isinstance(expr, AddExpression):
            return list(expr.args)
        return [expr]

    def shape_from_args(self):
        return np.broadcast_shapes(*[a.shape for a in self.args])

    def name(self):
        return " + ".join(a.name() for a in self.args)

    def numeric(self, values):
        out = 0.0
        for v in values:
            out = out + _dense(v)
        return out

    def copy(self, args=None):
        return AddExpression(list(self.args if args is None else args))


class NegExpression(AffAtom):
    """Negation (reference atoms/affine/unary_operators.py:84-140)."""

    def shape_from_args(self):
        return self.args[0].shape

    def sign_from_args(self):
        return (self.args[0].is_nonpos(), self.args[0].is_nonneg())

    def is_incr(self, idx):
        return False

    def is_decr(self, idx):
        return True

    def name(self):
        return "-(%s)" % self.args[0].name()

    def numeric(self, values):
        return -_dense(values[0])


def _mul_sign(lh, rh):
    zero = lh.is_zero() or rh.is_zero()
    pos = zero or (lh.is_nonneg() and rh.is_nonneg()) or (lh.is_nonpos() and rh.is_nonpos())
    neg = zero or (lh.is_nonneg() and rh.is_nonpos()) or (lh.is_nonpos() and rh.is_nonneg())
    return pos, neg


def _mul_shapes(lh, rh):
    """numpy matmul shape rule incl. 1-D promotion (reference utilities/shape.py mul_shapes)."""
    l2 = lh if len(lh) == 2 else ((1, lh[0]) if len(lh) == 1 else lh)
    r2 = rh if len(rh) == 2 else ((rh[0], 1) if len(rh) == 1 else rh)
    if len(lh) == 0 or len(rh) == 0:
        raise ValueError("Scalar operands are not allowed, use '*' instead")
    if l2[1] != r2[0]:
        raise ValueError("Incompatible dimensions %s %s" % (lh, rh))
    out = (l2[0], r2[1])
    if len(lh) == 1 and len(rh) == 1:
        return ()
    if len(lh) == 1:
        return (out[1],)
    if len(rh) == 1:
        return (out[0],)
    return out


class MulExpression(AffAtom):
    """Matrix product X @ Y (reference atoms/affine/binary_operators.py:96-400).
    Affine when one side is constant; bilinear (smooth, non-convex) otherwise."""

    def shape_from_args(self):
        return _mul_shapes(self.args[0].shape, self.args[1].shape)

    def sign_from_args(self):
        return _mul_sign(self.args[0], self.args[1])

    def is_atom_convex(self):
        return self.args[0].is_constant() or self.args[1].is_constant()

    is_atom_concave = is_atom_convex

    def is_incr(self, idx):
        return self.args[1 - idx].is_nonneg()

    def is_decr(self, idx):
        return self.args[1 - idx].is_nonpos()

    def name(self):
        return "%s @ %s" % (self.args[0].name(), self.args[1].name())

    def numeric(self, values):
        a, b = values
        if sp.issparse(a) or sp.issparse(b):
            return a @ b
        return np.asarray(a) @ np.asarray(b)

    @staticmethod
    def get_dimensions(X):
        if len(X.shape) == 0:
            return (1, 1)
        if len(X.shape) == 1:
            return (X.shape[0], 1)
        return X.shape


class multiply(MulExpression):
    """Elementwise product (reference binary_operators.py:403-620)."""

    def __init__(self, lh_expr, rh_expr):
        lh_expr, rh_expr = Expression.broadcast(lh_expr, rh_expr)
        super().__init__(lh_expr, rh_expr)

    def shape_from_args(self):
        return np.broadcast_shapes(self.args[0].shape, self.args[1].shape)

    def name(self):
        return "multiply(%s, %s)" % (self.args[0].name(), self.args[1].name())

    def numeric(self, values):
        a, b = values
        if sp.issparse(a):
            return a.multiply(b)
        if sp.issparse(b):
            return b.multiply(a)
        return np.multiply(a, b)


class DivExpression(AffAtom):
    """Elementwise division (reference binary_operators.py:623-720).  Affine only with a
    constant denominator; otherwise canonicalised to a bilinear equality (div_canon)."""

    def __init__(self, lh_expr, rh_expr):
        lh_expr, rh_expr = Expression.broadcast(lh_expr, rh_expr)
        super().__init__(lh_expr, rh_expr)

    def shape_from_args(self):
        return np.broadcast_shapes(self.args[0].shape, self.args[1].shape)

    def sign_from_args(self):
        return _mul_sign(self.args[0], self.args[1])

    def is_atom_convex(self):
        return self.args[1].is_constant()

    is_atom_concave = is_atom_convex

    def is_atom_esr(self):
        return True

    def is_atom_hsr(self):
        return True

    def is_incr(self, idx):
        if idx == 0:
            return self.args[1].is_nonneg()
        return self.args[0].is_nonpos()

    def is_decr(self, idx):
        if idx == 0:
            return self.args[1].is_nonpos()
        return self.args[0].is_nonneg()

    def name(self):
        return "%s / %s" % (self.args[0].name(), self.args[1].name())

    def numeric(self, values):
        return _dense(values[0]) / _dense(values[1])

    def point_in_domain(self):
        return np.ones(self.args[1].shape)


def is_special_slice(key) -> bool:
    """Does the key contain a list / ndarray (fancy indexing)?  (reference key_utils.py:203-211)"""
    if not isinstance(key, tuple):
        key = (key,)
    for elem in key:
        if not (isinstance(elem, (numbers.Number, slice)) or np.isscalar(elem)):
            return True
    return False


class index(AffAtom):
    """Basic slicing x[key] (reference atoms/affine/index.py:34-150)."""

    def __init__(self, expr, key):
        self.key = key
        super().__init__(expr)

    def get_data(self):
        return [self.key]

    def _select(self):
        idx = np.arange(self.args[0].size).reshape(self.args[0].shape, order="F")
        return idx[self.key]

    def shape_from_args(self):
        return self._select().shape

    def sign_from_args(self):
        return (self.args[0].is_nonneg(), self.args[0].is_nonpos())

    def name(self):
        return "%s[%s]" % (self.args[0].name(), self.key)

    def numeric(self, values):
        return _dense(values[0])[self.key]


class special_index(index):
    """Fancy indexing x[list / ndarray / bool mask] (reference index.py:153-290)."""


class Promote(AffAtom):
    """Scalar -> array of a given shape (reference atoms/affine/promote.py)."""

    def __init__(self, expr, shape):
        self.promoted_shape = tuple(shape)
        super().__init__(expr)

    def get_data(self):
        return [self.promoted_shape]

    def validate_arguments(self):
        if not self.args[0].is_scalar():
            raise ValueError("Only scalars may be promoted.")

    def shape_from_args(self):
        return self.promoted_shape

    def sign_from_args(self):
        return (self.args[0].is_nonneg(), self.args[0].is_nonpos())

    def numeric(self, values):
        return np.ones(self.promoted_shape) * np.asarray(values[0]).reshape(-1)[0]


def promote(expr, shape):
    expr = Expression.cast_to_const(expr)
    if expr.shape != tuple(shape):
        if not expr.is_scalar():
            raise ValueError("Only scalars may be promoted.")
        return Promote(expr, shape)
    return expr


class broadcast_to(AffAtom):
    """numpy-style broadcast (reference atoms/affine/broadcast_to.py)."""

    def __init__(self, expr, shape):
        self.broadcast_shape = tuple(shape)
        super().__init__(expr)

    def get_data(self):
        return [self.broadcast_shape]

    def shape_from_args(self):
        np.broadcast_shapes(self.args[0].shape, self.broadcast_shape)
        return self.broadcast_shape

    def sign_from_args(self):
        return (self.args[0].is_nonneg(), self.args[0].is_nonpos())

    def numeric(self, values):
        return np.broadcast_to(_dense(values[0]), self.broadcast_shape)


class reshape(AffAtom):
    """Reshape with explicit order (reference atoms/affine/reshape.py)."""

    def __init__(self, expr, shape, order="F"):
        if isinstance(shape, (int, np.integer)):
            shape = (int(shape),)
        if order not in ("F", "C"):
            raise ValueError("order must be 'F' or 'C'")
        self._req_shape = tuple(shape)
        self.order = order
        super().__init__(expr)

    def get_data(self):
        return [self._req_shape, self.order]

    def shape_from_args(self):
        shape = list(self._req_shape)
        if -1 in shape:
            known = int(np.prod([d for d in shape if d != -1], dtype=np.int64))
            shape[shape.index(-1)] = self.args[0].size // max(known, 1)
        if size_from_shape(shape) != self.args[0].size:
            raise ValueError("Invalid reshape dimensions %s." % (self._req_shape,))
        return tuple(shape)

    def sign_from_args(self):
        return (self.args[0].is_nonneg(), self.args[0].is_nonpos())

    def numeric(self, values):
        return np.reshape(_dense(values[0]), self.shape, order=self.order)


def vec(expr, order="F"):
    expr = Expression.cast_to_const(expr)
    return reshape(expr, (expr.size,), order)


class transpose(AffAtom):
    """Matrix transpose (reference atoms/affine/transpose.py)."""

    def shape_from_args(self):
        return self.args[0].shape[::-1]

    def sign_from_args(self):
        return (self.args[0].is_nonneg(), self.args[0].is_nonpos())

    def name(self):
        return "%s.T" % self.args[0].name()

    def numeric(self, values):
        return _dense(values[0]).T


class _AxisMixin:
    def _axis_shape(self):
        shp = self.args[0].shape
        if self.axis is None:
            return tuple([1] * len(shp)) if self.keepdims else ()
        ax = self.axis if self.axis >= 0 else self.axis + len(shp)
        if self.keepdims:
            return tuple(1 if i == ax else d for i, d in enumerate(shp))
        return tuple(d for i, d in enumerate(shp) if i != ax)


class Sum(AffAtom, _AxisMixin):
    """Sum of entries, optionally along an axis (reference atoms/affine/sum.py)."""

    def __init__(self, expr, axis=None, keepdims=False):
        self.axis = axis
        self.keepdims = keepdims
        super().__init__(expr)

    def get_data(self):
        return [self.axis, self.keepdims]

    def shape_from_args(self):
        return self._axis_shape()

    def sign_from_args(self):
        return (self.args[0].is_nonneg(), self.args[0].is_nonpos())

    def numeric(self, values):
        return np.sum(_dense(values[0]), axis=self.axis, keepdims=self.keepdims)


def sum(expr, axis=None, keepdims=False):  # noqa: A001  (reference name)
    if isinstance(expr, list):
        out = expr[0]
        for e in expr[1:]:
            out = out + e
        return out
    return Sum(Expression.cast_to_const(expr), axis, keepdims)


class Hstack(AffAtom):
    """Horizontal concatenation (reference atoms/affine/hstack.py)."""

    def shape_from_args(self):
        return np.hstack([np.empty(a.shape) for a in self.args]).shape

    def numeric(self, values):
        return np.hstack([_dense(v) for v in values])


class Vstack(AffAtom):
    """Vertical concatenation (reference atoms/affine/vstack.py)."""

    def shape_from_args(self):
        return np.vstack([np.empty(a.shape) for a in self.args]).shape

    def numeric(self, values):
        return np.vstack([_dense(v) for v in values])


def hstack(arg_list):
    return Hstack(*[Expression.cast_to_const(a) for a in arg_list])


def vstack(arg_list):
    return Vstack(*[Expression.cast_to_const(a) for a in arg_list])


def matmul(lh, rh):
    return Expression.cast_to_const(lh) @ Expression.cast_to_const(rh)


# =====================================================================================
# Smooth elementwise atoms
# =====================================================================================
class Elementwise(Atom):
    def shape_from_args(self):
        return np.broadcast_shapes(*[a.shape for a in self.args])


class _Unary(Elementwise):
    """Unary smooth elementwise atom; subclasses give numeric + tags."""
    CONVEX = False
    CONCAVE = False
    INCR = False
    DECR = False

    def is_atom_convex(self):
        return self.CONVEX

    def is_atom_concave(self):
        return self.CONCAVE

    def is_incr(self, idx):
        return self.INCR

    def is_decr(self, idx):
        return self.DECR

    def sign_from_args(self):
        return (False, False)


class exp(_Unary):
    """e^x (reference atoms/elementwise/exp.py)."""
    CONVEX, INCR = True, True

    def numeric(self, values):
        return np.exp(values[0])

    def sign_from_args(self):
        return (True, False)


class log(_Unary):
    """ln x, domain x>0 (reference atoms/elementwise/log.py)."""
    CONCAVE, INCR = True, True

    def numeric(self, values):
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.log(values[0])


class entr(_Unary):
    """-x ln x (reference atoms/elementwise/entr.py)."""
    CONCAVE = True

    def numeric(self, values):
        x = np.asarray(values[0], dtype=float)
        with np.errstate(divide="ignore", invalid="ignore"):
            out = -special.xlogy(x, x)
        out = np.where(x < 0, -np.inf, out)
        return out


class logistic(_Unary):
    """log(1+e^x) (reference atoms/elementwise/logistic.py)."""
    CONVEX, INCR = True, True

    def numeric(self, values):
        return np.logaddexp(0, values[0])

    def sign_from_args(self):
        return (True, False)


class xexp(_Unary):
    """x e^x (reference atoms/elementwise/xexp.py)."""

    def is_atom_convex(self):
        return self.args[0].is_nonneg()

    def is_incr(self, idx):
        return self.args[0].is_nonneg()

    def numeric(self, values):
        return values[0] * np.exp(values[0])

    def sign_from_args(self):
        return (self.args[0].is_nonneg(), self.args[0].is_nonpos())


class sin(_Unary):
    """sin x (reference atoms/elementwise/trig.py:25-105)."""

    def numeric(self, values):
        return np.sin(values[0])


class cos(_Unary):
    """cos x (reference trig.py:108-186)."""

    def numeric(self, values):
        return np.cos(values[0])


class tan(_Unary):
    """tan x on (-pi/2, pi/2) (reference trig.py:189-270)."""

    def numeric(self, values):
        return np.tan(values[0])


class sinh(_Unary):
    """sinh x (reference atoms/elementwise/hyperbolic.py:25-100)."""

    def numeric(self, values):
        return np.sinh(values[0])


class tanh(_Unary):
    """tanh x (reference hyperbolic.py:103-175)."""

    def numeric(self, values):
        return np.tanh(values[0])


class asinh(_Unary):
    """asinh x (reference hyperbolic.py:178-235)."""

    def numeric(self, values):
        return np.arcsinh(values[0])


class atanh(_Unary):
    """atanh x on (-1,1) (reference hyperbolic.py:238-295)."""

    def numeric(self, values):
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.arctanh(values[0])


def _rational_power(p, max_denom=1024):
    """Rational approximation used by the reference for derivative exponents
    (reference power.py:152-178, utilities/power_tools.py:106-149)."""
    if p > 1:
        q = Fraction(1 / Fraction(p)).limit_denominator(max_denom)
        if 1 / q == int(1 / q):
            return int(1 / q)
        return 1 / q
    if 0 < p < 1:
        return Fraction(p).limit_denominator(max_denom)
    if p < 0:
        q = Fraction(p)
        q = Fraction(q / (q - 1)).limit_denominator(max_denom)
        return q / (q - 1)
    return p


class power(Elementwise):
    """x^p for constant p (reference atoms/elementwise/power.py).

    `numeric` uses float(p); derivative rules use `p_rational` (the rational approximation
    with max_denom=1024) exactly as the reference does (power.py:188 vs :410-419)."""

    def __init__(self, x, p, max_denom: int = 1024):
        p_expr = Expression.cast_to_const(p)
        if not isinstance(p_expr, Constant) or p_expr.size != 1:
            raise ValueError("The exponent `p` must be a scalar Constant.")
        self._p_orig = p if not isinstance(p, Expression) else float(np.asarray(p_expr.value))
        self.p = p_expr
        self.max_denom = max_denom
        pv = self._p_orig
        if isinstance(pv, np.ndarray):
            pv = pv.item()
        pr = _rational_power(pv, max_denom)
        if pr == 1:
            pr = 1
        if pr == 0:
            pr = 0
        self.p_rational = pr
        super().__init__(x)

    @property
    def p_value(self) -> float:
        return float(np.asarray(self.p.value).reshape(-1)[0])

    def get_data(self):
        return [self._p_orig, self.max_denom]

    def name(self):
        return "power(%s, %s)" % (self.args[0].name(), self.p_value)

    def numeric(self, values):
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.power(np.asarray(values[0], dtype=float), self.p_value)

    def sign_from_args(self):
        if self.p_value == 1:
            return (self.args[0].is_nonneg(), self.args[0].is_nonpos())
        return (True, False)

    def is_atom_convex(self):
        return self.p_value <= 0 or self.p_value >= 1

    def is_atom_concave(self):
        return 0 <= self.p_value <= 1

    def _is_power2(self):
        p = self.p_rational
        return isinstance(p, int) and p > 0 and (p & (p - 1)) == 0

    def is_incr(self, idx):
        # reference power.py:277-292 (rule keyed on p_rational and is_power2)
        p = self.p_rational
        if 0 <= p <= 1:
            return True
        if p > 1:
            return self.args[idx].is_nonneg() if self._is_power2() else True
        return False

    def is_decr(self, idx):
        # reference power.py:294-309
        p = self.p_rational
        if p <= 0:
            return True
        if p > 1:
            return self.args[idx].is_nonpos() if self._is_power2() else False
        return False

    def point_in_domain(self):
        return np.ones(self.shape)


def square(x):
    return power(x, 2)


def sqrt(x):
    return power(x, Fraction(1, 2))


class rel_entr(Elementwise):
    """x log(x/y) (reference atoms/elementwise/rel_entr.py)."""

    def __init__(self, x, y):
        super().__init__(x, y)

    def numeric(self, values):
        with np.errstate(divide="ignore", invalid="ignore"):
            return special.rel_entr(np.asarray(values[0], float), np.asarray(values[1], float))

    def sign_from_args(self):
        return (False, False)

    def is_atom_convex(self):
        return True

    def is_atom_concave(self):
        return False

    def is_incr(self, idx):
        return False

    def is_decr(self, idx):
        return idx == 1

    def point_in_domain(self, argument=0):
        return np.ones(self.args[argument].shape)


class kl_div(Elementwise):
    """x log(x/y) - x + y (reference atoms/elementwise/kl_div.py)."""

    def __init__(self, x, y):
        super().__init__(x, y)

    def numeric(self, values):
        with np.errstate(divide="ignore", invalid="ignore"):
            return special.kl_div(np.asarray(values[0], float), np.asarray(values[1], float))

    def sign_from_args(self):
        return (True, False)

    def is_atom_convex(self):
        return True

    def is_atom_concave(self):
        return False

    def is_incr(self, idx):
        return False


# =====================================================================================
# Nonsmooth ESR / HSR atoms (rewritten to epigraph form by dnlp2smooth)
# =====================================================================================
class abs(_Unary):  # noqa: A001
    """|x| (reference atoms/elementwise/abs.py): ESR only."""
    CONVEX = True

    def is_atom_hsr(self):
        return False

    def is_incr(self, idx):
        return self.args[0].is_nonneg()

    def is_decr(self, idx):
        return self.args[0].is_nonpos()

    def numeric(self, values):
        return np.abs(values[0])

    def sign_from_args(self):
        return (True, False)


class maximum(Elementwise):
    """Elementwise max of several args (reference atoms/elementwise/maximum.py): ESR only."""

    def __init__(self, arg1, arg2, *args):
        super().__init__(arg1, arg2, *args)

    def numeric(self, values):
        out = _dense(values[0])
        for v in values[1:]:
            out = np.maximum(out, _dense(v))
        return out

    def sign_from_args(self):
        return (any(a.is_nonneg() for a in self.args), all(a.is_nonpos() for a in self.args))

    def is_atom_convex(self):
        return True

    def is_atom_concave(self):
        return False

    def is_atom_hsr(self):
        return False


class minimum(Elementwise):
    """Elementwise min (reference atoms/elementwise/minimum.py): HSR only."""

    def __init__(self, arg1, arg2, *args):
        super().__init__(arg1, arg2, *args)

    def numeric(self, values):
        out = _dense(values[0])
        for v in values[1:]:
            out = np.minimum(out, _dense(v))
        return out

    def sign_from_args(self):
        return (all(a.is_nonneg() for a in self.args), any(a.is_nonpos() for a in self.args))

    def is_atom_convex(self):
        return False

    def is_atom_concave(self):
        return True

    def is_atom_esr(self):
        return False


class huber(Elementwise):
    """Huber function with threshold M (reference atoms/elementwise/huber.py): ESR only."""

    def __init__(self, x, M=1):
        self.M = float(np.asarray(Expression.cast_to_const(M).value))
        if self.M < 0:
            raise ValueError("M must be a non-negative scalar constant.")
        super().__init__(x)

    def get_data(self):
        return [self.M]

    def numeric(self, values):
        x = np.asarray(values[0], float)
        a = np.abs(x)
        return np.where(a <= self.M, x * x, 2 * self.M * a - self.M ** 2)

    def sign_from_args(self):
        return (True, False)

    def is_atom_convex(self):
        return True

    def is_atom_concave(self):
        return False

    def is_atom_hsr(self):
        return False

    def is_incr(self, idx):
        return self.args[0].is_nonneg()

    def is_decr(self, idx):
        return self.args[0].is_nonpos()


class _AxisAtom(Atom, _AxisMixin):
    def __init__(self, expr, axis=None, keepdims=False):
        self.axis = axis
        self.keepdims = keepdims
        super().__init__(expr)

    def get_data(self):
        return [self.axis, self.keepdims]

    def shape_from_args(self):
        return self._axis_shape()


class max(_AxisAtom):  # noqa: A001
    """Largest entry (reference atoms/max.py): ESR only, never smooth."""

    def numeric(self, values):
        return np.max(_dense(values[0]), axis=self.axis, keepdims=self.keepdims)

    def sign_from_args(self):
        return (self.args[0].is_nonneg(), self.args[0].is_nonpos())

    def is_atom_convex(self):
        return True

    def is_atom_concave(self):
        return False

    def is_atom_hsr(self):
        return False

    def is_smooth(self):
        return False


class min(_AxisAtom):  # noqa: A001
    """Smallest entry (reference atoms/min.py): HSR only."""

    def numeric(self, values):
        return np.min(_dense(values[0]), axis=self.axis, keepdims=self.keepdims)

    def sign_from_args(self):
        return (self.args[0].is_nonneg(), self.args[0].is_nonpos())

    def is_atom_convex(self):
        return False

    def is_atom_concave(self):
        return True

    def is_atom_esr(self):
        return False

    def is_smooth(self):
        return False


class norm1(_AxisAtom):
    """Sum of absolute values (reference atoms/norm1.py): ESR only."""

    def numeric(self, values):
        return np.sum(np.abs(_dense(values[0])), axis=self.axis, keepdims=self.keepdims)

    def sign_from_args(self):
        return (True, False)

    def is_atom_convex(self):
        return True

    def is_atom_concave(self):
        return False

    def is_atom_hsr(self):
        return False

    def is_incr(self, idx):
        return self.args[0].is_nonneg()

    def is_decr(self, idx):
        return self.args[0].is_nonpos()


class norm_inf(_AxisAtom):
    """Largest absolute value (reference atoms/norm_inf.py): ESR only."""

    def numeric(self, values):
        return np.max(np.abs(_dense(values[0])), axis=self.axis, keepdims=self.keepdims)

    def sign_from_args(self):
        return (True, False)

    def is_atom_convex(self):
        return True

    def is_atom_concave(self):
        return False

    def is_atom_hsr(self):
        return False

    def is_incr(self, idx):
        return self.args[0].is_nonneg()

    def is_decr(self, idx):
        return self.args[0].is_nonpos()


class Pnorm(_AxisAtom):
    """Vector p-norm, p not in {1, inf} (reference atoms/pnorm.py:50-300).
    Only p=2 has a dnlp2smooth rule (pnorm_canon.py:29-34)."""

    def __init__(self, x, p=2, axis=None, keepdims=False, max_denom=1024):
        self.p = p
        self.max_denom = max_denom
        super().__init__(x, axis, keepdims)

    def get_data(self):
        return [self.p, self.axis, self.keepdims, self.max_denom]

    def numeric(self, values):
        return np.linalg.norm(_dense(values[0]).reshape(-1, order="F") if self.axis is None
                              else _dense(values[0]), float(self.p), axis=self.axis,
                              keepdims=self.keepdims if self.axis is not None else False)

    def sign_from_args(self):
        return (True, False)

    def is_atom_convex(self):
        return self.p > 1

    def is_atom_concave(self):
        return self.p < 1

    def is_atom_esr(self):
        return self.p > 1

    def is_atom_hsr(self):
        return self.p < 1

    def is_incr(self, idx):
        return self.p < 1 or (self.p > 1 and self.args[0].is_nonneg())

    def is_decr(self, idx):
        return self.p > 1 and self.args[0].is_nonpos()


def pnorm(x, p=2, axis=None, keepdims=False, max_denom=1024):
    if p == 1:
        return norm1(x, axis=axis, keepdims=keepdims)
    if p in [np.inf, "inf", "Inf"]:
        return norm_inf(x, axis=axis, keepdims=keepdims)
    return Pnorm(x, p=p, axis=axis, keepdims=keepdims, max_denom=max_denom)


def norm(x, p=2, axis=None, keepdims=False):
    """Wrapper on the norm atoms (reference atoms/norm.py:30-82)."""
    x = Expression.cast_to_const(x)
    num_nontrivial = np.sum([d > 1 for d in x.shape]) if x.ndim else 0
    if axis is None and x.ndim == 2:
        if p == 1:
            return max(norm1(x, axis=0))
        if p == "fro" or (p == 2 and num_nontrivial == 1):
            return pnorm(vec(x, "F"), 2)
        if p in [np.inf, "inf", "Inf"]:
            return max(norm1(x, axis=1))
        raise RuntimeError("Unsupported matrix norm on the nlp=True path.")
    if p == 1 or x.is_scalar():
        return norm1(x, axis=axis, keepdims=keepdims)
    if str(p).lower() == "inf":
        return norm_inf(x, axis=axis, keepdims=keepdims)
    if str(p).lower() == "fro":
        return pnorm(vec(x, "F"), 2, axis)
    if isinstance(p, str):
        raise RuntimeError("Unsupported norm option %s for non-matrix." % p)
    return pnorm(x, p, axis=axis, keepdims=keepdims)


def norm2(x, axis=None):
    return norm(x, p=2, axis=axis)


class sum_largest(Atom):
    """Sum of the k largest entries (reference atoms/sum_largest.py): ESR only."""

    def __init__(self, x, k):
        self.k = k
        super().__init__(x)

    def get_data(self):
        return [self.k]

    def shape_from_args(self):
        return ()

    def numeric(self, values):
        v = np.sort(_dense(values[0]).reshape(-1))[::-1]
        k = int(np.floor(self.k))
        out = v[:k].sum()
        if k < self.k and k < v.size:
            out += (self.k - k) * v[k]
        return out

    def sign_from_args(self):
        return (self.args[0].is_nonneg(), self.args[0].is_nonpos())

    def is_atom_convex(self):
        return True

    def is_atom_concave(self):
        return False

    def is_atom_hsr(self):
        return False


def sum_smallest(x, k):
    """-sum_largest(-x, k) (reference atoms/sum_smallest.py)."""
    x = Expression.cast_to_const(x)
    return -sum_largest(-x, k)


class geo_mean(Atom):
    """Weighted geometric mean prod x_i^{w_i} (reference atoms/geo_mean.py)."""

    def __init__(self, x, p=None, max_denom=1024):
        x = Expression.cast_to_const(x)
        if p is None:
            p = [1] * x.size
        p = np.asarray(p, dtype=float).reshape(-1)
        if p.size != x.size or np.any(p < 0) or p.sum() <= 0:
            raise ValueError("Invalid weights for geo_mean.")
        self.p = p
        # rational weights summing to one (reference utilities/power_tools.py fracify)
        fr = [Fraction(v / p.sum()).limit_denominator(max_denom) for v in p]
        tot = np.sum(fr)
        self.w = tuple(f / tot for f in fr)
        self.max_denom = max_denom
        super().__init__(x)

    def get_data(self):
        return [self.p, self.max_denom]

    def shape_from_args(self):
        return ()

    def numeric(self, values):
        v = np.asarray(values[0], float).reshape(-1, order="F")
        w = np.array([float(x) for x in self.w])
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.prod(np.power(v, w))

    def sign_from_args(self):
        return (True, False)

    def is_atom_convex(self):
        return False

    def is_atom_concave(self):
        return True


# =====================================================================================
# Quadratic atoms
# =====================================================================================
class QuadForm(Atom):
    """x^T P x with constant symmetric P (reference atoms/quad_form.py:33-170)."""

    def __init__(self, x, P):
        super().__init__(x, P)

    def validate_arguments(self):
        x, P = self.args
        n = P.shape[0]
        if P.ndim != 2 or n != P.shape[1] or x.size != n:
            raise ValueError("Invalid dimensions for arguments.")
        if not P.is_constant():
            raise ValueError("P must be constant in QuadForm.")

    def shape_from_args(self):
        return ()

    def name(self):
        return "QuadForm(%s, %s)" % (self.args[0].name(), self.args[1].name())

    def numeric(self, values):
        x = np.asarray(values[0], float).reshape(-1, order="F")
        P = values[1]
        return float(x @ (P @ x))

    def sign_from_args(self):
        return (False, False)

    def is_atom_convex(self):
        return False

    def is_atom_concave(self):
        return False

    def is_incr(self, idx):
        return False

    def is_decr(self, idx):
        return False

    def _value_impl(self):
        P = self.args[1]
        if isinstance(P, Constant) and P.is_device:
            return None   # evaluated on the device only
        return super()._value_impl()


def quad_form(x, P, assume_PSD: bool = False):
    """Alias for x^T P x (reference atoms/quad_form.py:270-292)."""
    x = Expression.cast_to_const(x)
    P = Expression.cast_to_const(P)
    if P.ndim != 2 or P.shape[0] != P.shape[1] or (x.shape or (1,))[0] != P.shape[0]:
        raise Exception("Invalid dimensions for arguments to quad_form.")
    if x.is_constant():
        return x.T @ P @ x
    if P.is_constant():
        return QuadForm(x, P)
    raise Exception("At least one argument to quad_form must be non-variable.")


class quad_over_lin(Atom):
    """sum(x^2)/y (reference atoms/quad_over_lin.py:30-200)."""

    def __init__(self, x, y):
        super().__init__(x, y)

    def validate_arguments(self):
        if not self.args[1].is_scalar():
            raise ValueError("The second argument to quad_over_lin must be a scalar.")

    def shape_from_args(self):
        return ()

    def numeric(self, values):
        return np.square(_dense(values[0])).sum() / np.asarray(values[1]).reshape(-1)[0]

    def sign_from_args(self):
        return (True, False)

    def is_atom_convex(self):
        return True

    def is_atom_concave(self):
        return False

    def is_incr(self, idx):
        return idx == 0 and self.args[0].is_nonneg()

    def is_decr(self, idx):
        return (idx == 0 and self.args[0].is_nonpos()) or idx == 1


def sum_squares(expr):
    """quad_over_lin(expr, 1) (reference atoms/sum_squares.py:33)."""
    return quad_over_lin(expr, 1)
