"""NLP canonicaliser: disciplined nonlinear program -> smooth program.

Own implementation of the reduction the reference runs once per solve
(reductions/dnlp2smooth/dnlp2smooth.py:27-111 + the 23 rules of
reductions/dnlp2smooth/canonicalizers/ and eliminate_pwl/canonicalizers/, SURVEY.md
Appendix B).  The rewrite is bottom-up: every nonlinear atom ends up applied to a bare
`Variable` (`t == arg` equalities are added, with the domain bounds and initial values the
reference uses), nonsmooth ESR/HSR atoms are replaced by epigraph variables, and a few
algebraic rewrites remove atoms that have no derivative rule (div, geo_mean, kl_div, pnorm).

Parity matters here: N, m, the aux-variable order, bounds and x0 define the local optimum
the interior-point loop lands in, so each rule reproduces the reference's choice of new
variables, constraint order and initial values (file:line cited per rule).
"""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

from . import atoms as at
from .constraints import Constraint
from .expressions import Constant, Expression, Variable

MIN_INIT = 1e-4


def _maybe_value(dst: Variable, val):
    if val is not None:
        dst.value = val


# ---- smooth elementwise atoms ------------------------------------------------------
def _keep_or_alias(expr, args):
    """exp / logistic / sin / cos / sinh / tanh / asinh: keep when the argument already is a
    Variable, otherwise t == arg with t initialised at arg.value
    (exp_canon.py:19-26, logistic_canon.py:19-26, trig_canon.py:19-35,
    hyperbolic_canon.py:19-44)."""
    if isinstance(args[0], Variable):
        return expr.copy([args[0]]), []
    t = Variable(args[0].shape)
    _maybe_value(t, args[0].value)
    return expr.copy([t]), [t == args[0]]


def _log_like(expr, args):
    """log / entr: always a new t >= 0 with t == arg, initialised at max(arg.value, 1e-4)
    or the atom's point in the domain (log_canon.py:23-31, entr_canon.py:23-30)."""
    t = Variable(args[0].shape, bounds=[0, None])
    if args[0].value is not None:
        t.value = np.maximum(args[0].value, MIN_INIT)
    else:
        t.value = expr.point_in_domain()
    return expr.copy([t]), [t == args[0]]


def _tan_canon(expr, args):
    """tan: new t in (-3.14159/2, 3.14159/2) (trig_canon.py:37-41; the truncated pi is the
    reference's)."""
    t = Variable(args[0].shape, bounds=[-3.14159 / 2, 3.14159 / 2])
    _maybe_value(t, args[0].value)
    return expr.copy([t]), [t == args[0]]


def _atanh_canon(expr, args):
    """atanh: new t in [-1, 1] (hyperbolic_canon.py:46-50)."""
    t = Variable(args[0].shape, bounds=[-1, 1])
    _maybe_value(t, args[0].value)
    return expr.copy([t]), [t == args[0]]


def power_canon(expr, args):
    """power (power_canon.py:24-54): p=0 -> ones; p=1 -> x; integer p>1 keeps a Variable
    argument else t == x; other p>0 -> nonneg t == x; p<0 unsupported."""
    x = args[0]
    p = expr.p_rational
    shape = expr.shape
    if p == 0:
        return Constant(np.ones(shape)), []
    if p == 1:
        return x, []
    if isinstance(p, int) and p > 1:
        if isinstance(x, Variable):
            return expr.copy(args), []
        t = Variable(shape)
        t.value = x.value if x.value is not None else expr.point_in_domain()
        return expr.copy([t]), [t == x]
    if p > 0:
        t = Variable(shape, nonneg=True)
        t.value = x.value if x.value is not None else expr.point_in_domain()
        return expr.copy([t]), [t == x]
    raise NotImplementedError("The power %s is not yet supported." % p)


def quad_over_lin_canon(expr, args):
    """quad_over_lin (quad_over_lin_canon.py:25-54): constant denominator ->
    (1/c) * Sum(power(x, 2)); otherwise numerator variable if needed and ALWAYS a new
    nonneg denominator variable initialised at y.value (> 1e-4) or 1."""
    if args[1].is_constant():
        pw = at.power(args[0], 2)
        var, constr = power_canon(pw, pw.args)
        summation = at.Sum(var)
        return 1 / args[1].value * summation, constr
    t1, t2 = args
    constraints = []
    if not isinstance(t1, Variable):
        t1 = Variable(t1.shape)
        constraints += [t1 == args[0]]
        _maybe_value(t1, args[0].value)
    t2 = Variable(t2.shape, nonneg=True)
    constraints += [t2 == args[1]]
    if args[1].value is not None and np.all(args[1].value > MIN_INIT):
        t2.value = args[1].value
    else:
        t2.value = np.ones(t2.shape)
    return expr.copy([t1, t2]), constraints


def pnorm_canon(expr, args):
    """Pnorm, p=2 only: t >= 0 with quad_over_lin(x, t) <= t (pnorm_canon.py:22-34)."""
    x = args[0]
    if expr.p != 2:
        raise ValueError("Only p=2 is supported as Pnorm.")
    t = Variable(expr.shape, nonneg=True)
    q = at.quad_over_lin(x, t)
    new_expr, constr = quad_over_lin_canon(q, q.args)
    return t, constr + [new_expr <= t]


def div_canon(expr, args):
    """f/g -> z with z*y == f, y == g, y >= 0; z sign-bounded by f's sign
    (div_canon.py:26-55; assumes a positive denominator)."""
    dim = args[0].shape
    sgn = args[0].sign
    if sgn == "NONNEGATIVE":
        z = Variable(dim, bounds=[0, None])
    elif sgn == "NONPOSITIVE":
        z = Variable(dim, bounds=[None, 0])
    else:
        z = Variable(dim)
    y = Variable(args[1].shape, bounds=[0, None])
    if args[1].value is not None:
        y.value = np.maximum(args[1].value, MIN_INIT)
    else:
        y.value = expr.point_in_domain()
    if args[0].value is not None:
        val = args[0].value / y.value
    else:
        val = expr.point_in_domain()
    val = np.asarray(val)
    if dim == () and val.shape == (1,):
        z.value = val[0]
    else:
        z.value = val
    return z, [at.multiply(z, y) == args[0], y == args[1]]


def _bilinear_canon(expr, args):
    """multiply / matmul (multiply_canon.py:23-63): unchanged when a side is constant,
    otherwise each non-Variable side becomes t == side initialised at side.value."""
    t1, t2 = args
    constraints = []
    if t1.is_constant() or t2.is_constant():
        return expr.copy([t1, t2]), []
    if not isinstance(t1, Variable):
        t1 = Variable(t1.shape)
        constraints += [t1 == args[0]]
        _maybe_value(t1, args[0].value)
    if not isinstance(t2, Variable):
        t2 = Variable(t2.shape)
        constraints += [t2 == args[1]]
        _maybe_value(t2, args[1].value)
    return expr.copy([t1, t2]), constraints


def rel_entr_canon(expr, args):
    """rel_entr (rel_entr_canon.py:29-61)."""
    if args[0].is_constant():
        lg = at.log(args[1])
        log_expr, constr_log = _log_like(lg, lg.args)
        x = args[0].value
        return x * np.log(x) - at.multiply(x, log_expr), constr_log
    if args[1].is_constant():
        en = at.entr(args[0])
        entr_expr, constr_entr = _log_like(en, en.args)
        mu = at.multiply(args[0], np.log(args[1].value))
        mult_expr, constr_mult = _bilinear_canon(mu, mu.args)
        return -entr_expr - mult_expr, constr_entr + constr_mult
    t1 = Variable(args[0].shape, bounds=[0, None])
    t2 = Variable(args[1].shape, bounds=[0, None])
    constraints = [t1 == args[0], t2 == args[1]]
    if args[0].value is not None:
        t1.value = np.maximum(args[0].value, MIN_INIT)
    else:
        t1.value = expr.point_in_domain(argument=0)
    if args[1].value is not None:
        t2.value = np.maximum(args[1].value, MIN_INIT)
    else:
        t2.value = expr.point_in_domain(argument=1)
    return expr.copy([t1, t2]), constraints


def kl_div_canon(expr, args):
    """kl_div = rel_entr - x + y (kl_div_canon.py:21-24)."""
    re = at.rel_entr(args[0], args[1])
    re_expr, constr = rel_entr_canon(re, re.args)
    return re_expr - args[0] + args[1], constr


def geo_mean_canon(expr, args):
    """geo_mean: t >= 0 with log(t) == sum_i w_i log(x_i) (geo_mean_canon.py:27-41)."""
    t = Variable(expr.shape, nonneg=True)
    if args[0].value is not None and np.all(args[0].value > MIN_INIT):
        t.value = expr.numeric([args[0].value])
    else:
        t.value = np.ones(expr.shape)
    weights = np.array([float(w) for w in expr.w])
    lg = at.log(args[0])
    var, constr = _log_like(lg, expr.args)
    return t, [at.log(t) == at.sum(at.multiply(weights, var))] + constr


# ---- nonsmooth ESR / HSR atoms -------------------------------------------------------
def abs_canon(expr, args):
    """|x| -> t with t >= x, t >= -x (eliminate_pwl abs_canon.py:20-24)."""
    x = args[0]
    t = Variable(expr.shape)
    return t, [t >= x, t >= -x]


def maximum_canon(expr, args):
    """maximum(args) -> t with t >= arg for every arg (maximum_canon.py:21-31)."""
    t = Variable(expr.shape)
    return t, [t >= elem for elem in args]


def minimum_canon(expr, args):
    """minimum(args) = -maximum(-args) (minimum_canon.py:23-27)."""
    tmp = at.maximum(*[-arg for arg in args])
    canon, constr = maximum_canon(tmp, tmp.args)
    return -canon, constr


def _promote_axis(t, x, axis):
    if axis is None:
        return at.promote(t, x.shape)
    if axis == 0:
        return Constant(np.ones((x.shape[0], 1))) @ at.reshape(t, (1, x.shape[1]), order="F")
    return at.reshape(t, (x.shape[0], 1), order="F") @ Constant(np.ones((1, x.shape[1])))


def max_canon(expr, args):
    """max(x, axis) -> t with x <= promote(t) (max_canon.py:24-38)."""
    x = args[0]
    t = Variable(expr.shape)
    return t, [x <= _promote_axis(t, x, expr.axis)]


def min_canon(expr, args):
    """min(x) = -max(-x) (min_canon.py:21-28)."""
    tmp = at.max(-args[0], axis=expr.axis, keepdims=expr.keepdims)
    canon, constr = max_canon(tmp, tmp.args)
    return -canon, constr


def norm1_canon(expr, args):
    """norm1(x) -> sum(t), t >= x, t >= -x (norm1_canon.py:22-34)."""
    ab = at.abs(args[0])
    abs_x, constr = abs_canon(ab, ab.args)
    return at.sum(abs_x, axis=expr.axis), constr


def norm_inf_canon(expr, args):
    """norm_inf(x) -> t with x <= T, x + T >= 0 (norm_inf_canon.py:24-37)."""
    x = args[0]
    t = Variable(expr.shape)
    T = _promote_axis(t, x, expr.axis)
    return t, [x <= T, x + T >= 0]


def huber_canon(expr, args):
    """huber(x, M) -> n^2 + 2M|s| with x == s + n (huber_canon.py:27-44)."""
    M = expr.M
    x = args[0]
    shape = expr.shape
    n = Variable(shape)
    s = Variable(shape)
    pw = at.power(n, 2)
    n2, constr_sq = power_canon(pw, pw.args)
    ab = at.abs(s)
    abs_s, constr_abs = abs_canon(ab, ab.args)
    obj = n2 + 2 * M * abs_s
    constraints = constr_sq + constr_abs
    constraints.append(x == s + n)
    return obj, constraints


def sum_largest_canon(expr, args):
    """sum_largest(x, k) -> sum(t) + k q with x <= t + q, t >= 0
    (sum_largest_canon.py:21-32)."""
    x = args[0]
    k = expr.k
    t = Variable(x.shape)
    q = Variable()
    return at.sum(t) + k * q, [x <= t + q, t >= 0]


SMOOTH_CANON_METHODS = {
    at.log: _log_like,
    at.exp: _keep_or_alias,
    at.logistic: _keep_or_alias,
    at.sin: _keep_or_alias,
    at.cos: _keep_or_alias,
    at.tan: _tan_canon,
    at.sinh: _keep_or_alias,
    at.asinh: _keep_or_alias,
    at.tanh: _keep_or_alias,
    at.atanh: _atanh_canon,
    at.quad_over_lin: quad_over_lin_canon,
    at.power: power_canon,
    at.Pnorm: pnorm_canon,
    at.DivExpression: div_canon,
    at.entr: _log_like,
    at.rel_entr: rel_entr_canon,
    at.kl_div: kl_div_canon,
    at.multiply: _bilinear_canon,
    at.MulExpression: _bilinear_canon,
    at.geo_mean: geo_mean_canon,
    # ESR atoms
    at.abs: abs_canon,
    at.maximum: maximum_canon,
    at.max: max_canon,
    at.norm1: norm1_canon,
    at.norm_inf: norm_inf_canon,
    at.huber: huber_canon,
    at.sum_largest: sum_largest_canon,
    # HSR atoms
    at.minimum: minimum_canon,
    at.min: min_canon,
}


class Dnlp2Smooth:
    """Bottom-up tree rewrite (reference dnlp2smooth.py:27-111)."""

    def apply(self, problem):
        from .problem import Problem
        canon_objective, canon_constraints = self.canonicalize_tree(problem.objective, True)
        cons_id_map = {}
        for constraint in problem.constraints:
            canon_constr, aux_constr = self.canonicalize_tree(constraint, False)
            canon_constraints += aux_constr + [canon_constr]
            cons_id_map[constraint.id] = canon_constr.id
        new_problem = Problem(canon_objective, canon_constraints)
        return new_problem, {"cons_id_map": cons_id_map}

    def canonicalize_tree(self, expr, affine_above: bool) -> Tuple[object, List[Constraint]]:
        affine_atom = type(expr) not in SMOOTH_CANON_METHODS
        canon_args, constrs = [], []
        for arg in expr.args:
            canon_arg, c = self.canonicalize_tree(arg, affine_atom and affine_above)
            canon_args.append(canon_arg)
            constrs += c
        canon_expr, c = self.canonicalize_expr(expr, canon_args, affine_above)
        constrs += c
        return canon_expr, constrs

    def canonicalize_expr(self, expr, args, affine_above: bool):
        # constant trees are collapsed (reference dnlp2smooth.py:104-106)
        if isinstance(expr, Expression) and expr.is_constant() and not expr.parameters():
            return expr, []
        rule = SMOOTH_CANON_METHODS.get(type(expr))
        if rule is not None:
            return rule(expr, args)
        return expr.copy(args), []
