"""Constraint kinds of the nlp=True path.

Mirror of reference cvxpy/constraints/{zero,nonpos}.py restricted to what
`Bounds.get_constraint_bounds` (nlp_solver.py:89-114) handles: Equality, Inequality, NonPos
(plus the lowered forms Zero and NonNeg produced by reductions/utilities.py:36-49).
DNLP rules: zero.py:61-63,140-142; nonpos.py:72-77,138-143,228-233.
"""
from __future__ import annotations

import numpy as np

from .expressions import Expression, get_id, unique_list


class Constraint:
    def __init__(self, args, constr_id=None):
        self.args = list(args)
        self.id = get_id() if constr_id is None else constr_id
        self.dual_value = None

    @property
    def shape(self):
        return self.expr.shape

    @property
    def size(self):
        return self.expr.size

    def variables(self):
        return unique_list([v for a in self.args for v in a.variables()])

    def parameters(self):
        return unique_list([p for a in self.args for p in a.parameters()])

    def copy(self, args=None):
        if args is None:
            args = self.args
        return type(self)(*args)

    def is_dnlp(self) -> bool:
        raise NotImplementedError

    def residual_value(self):
        raise NotImplementedError

    def __bool__(self):
        raise ValueError("Constraints have no truth value; chained comparisons such as "
                         "a <= x <= b are not supported.")

    def __repr__(self):
        return "%s(%s)" % (type(self).__name__, ", ".join(repr(a) for a in self.args))

    def __str__(self):
        return self.name()


class Equality(Constraint):
    """lhs == rhs; DNLP iff lhs - rhs is smooth (reference zero.py:140-142)."""

    def __init__(self, lhs, rhs, constr_id=None):
        lhs = Expression.cast_to_const(lhs)
        rhs = Expression.cast_to_const(rhs)
        self._expr = lhs - rhs
        super().__init__([lhs, rhs], constr_id)

    @property
    def expr(self):
        return self._expr

    def name(self):
        return "%s == %s" % (self.args[0].name(), self.args[1].name())

    def is_dnlp(self):
        return self.expr.is_smooth()

    def violation(self):
        v = self.expr.value
        return None if v is None else np.abs(v)


class Inequality(Constraint):
    """lhs <= rhs; DNLP iff lhs - rhs is ESR (reference nonpos.py:228-233)."""

    def __init__(self, lhs, rhs, constr_id=None):
        lhs = Expression.cast_to_const(lhs)
        rhs = Expression.cast_to_const(rhs)
        self._expr = lhs - rhs
        super().__init__([lhs, rhs], constr_id)

    @property
    def expr(self):
        return self._expr

    def name(self):
        return "%s <= %s" % (self.args[0].name(), self.args[1].name())

    def is_dnlp(self):
        return self.expr.is_esr()

    def violation(self):
        v = self.expr.value
        return None if v is None else np.maximum(v, 0)


class _Unary(Constraint):
    def __init__(self, expr, constr_id=None):
        super().__init__([Expression.cast_to_const(expr)], constr_id)

    @property
    def expr(self):
        return self.args[0]


class Zero(_Unary):
    """expr == 0 (reference zero.py:30-63)."""

    def name(self):
        return "%s == 0" % self.args[0].name()

    def is_dnlp(self):
        return self.args[0].is_smooth()


class NonPos(_Unary):
    """expr <= 0 (reference nonpos.py:30-77)."""

    def name(self):
        return "%s <= 0" % self.args[0].name()

    def is_dnlp(self):
        return self.args[0].is_esr()


class NonNeg(_Unary):
    """expr >= 0 (reference nonpos.py:100-143)."""

    def name(self):
        return "%s >= 0" % self.args[0].name()

    def is_dnlp(self):
        return self.args[0].is_hsr()


# ---- lowering helpers (reference reductions/utilities.py:36-49) ------------------------
def lower_equality(c: Equality) -> Zero:
    return Zero(c.args[0] - c.args[1], constr_id=c.id)


def lower_ineq_to_nonneg(c: Inequality) -> NonNeg:
    return NonNeg(c.args[1] - c.args[0], constr_id=c.id)


def nonpos2nonneg(c: NonPos) -> NonNeg:
    return NonNeg(-c.args[0], constr_id=c.id)
