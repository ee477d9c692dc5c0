"""NLP canonicaliser: disciplined nonlinear program -> smooth program.

What the reference does once per solve in reductions/dnlp2smooth/dnlp2smooth.py:27-111 with the 23
rule files of reductions/dnlp2smooth/canonicalizers/ and eliminate_pwl/canonicalizers/ (SURVEY.md
Appendix B), organised here as DATA plus three generic builders instead of one function per atom:

* **alias rules** (`ALIAS`): every argument of a smooth atom must end up a bare `Variable`.  A rule is
  one `Aux` description per argument — bounds of the new variable, how it is initialised, whether an
  argument that already is a Variable is kept — and `_alias_arguments` does the rest
  (`t == arg` rows in argument order).
* **epigraph rules** (`EPIGRAPH`): a nonsmooth ESR / HSR atom becomes fresh variable(s) plus the rows
  that pin them from the right side; a rule returns (replacement expression, rows).
* **rewrite rules** (`REWRITE`): atoms without derivative rules are expressed through atoms that have
  them (minimum through maximum, norm1 through abs, kl_div through rel_entr, division through a
  bilinear row, ...); the pieces are then canonicalised by their own rules (`_canon`).

Parity is structural: N, m, the order of the rows, the bounds and the start values decide which local
optimum the interior-point loop reaches, so each table entry cites the reference rule whose OUTPUT it
reproduces (tests/test_frontend_golden.py compares all of that with vectors captured from the
reference on 24 problems).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import atoms as at
from .constraints import Constraint
from .expressions import Constant, Expression, Variable

MIN_INIT = 1e-4                 # smallest start value given to a variable with a logarithmic domain
PI_TRUNCATED = 3.14159          # the reference's bound of tan's argument (trig_canon.py:38), not math.pi

Rows = List[Constraint]


# ---- start-value policies of auxiliary variables ---------------------------------------------------
def init_from_argument(atom, arg, k):
    """the argument's current value, when it has one"""
    return arg.value


def init_clipped(atom, arg, k):
    """max(value, 1e-4); without a value: the atom's own point in its domain"""
    if arg.value is not None:
        return np.maximum(arg.value, MIN_INIT)
    return atom.point_in_domain()


def init_clipped_per_argument(atom, arg, k):
    if arg.value is not None:
        return np.maximum(arg.value, MIN_INIT)
    return atom.point_in_domain(argument=k)


def init_or_domain_point(atom, arg, k):
    return arg.value if arg.value is not None else atom.point_in_domain()


def init_if_safely_positive(atom, arg, k):
    """the value when every entry exceeds 1e-4, else ones"""
    if arg.value is not None and np.all(arg.value > MIN_INIT):
        return arg.value
    return np.ones(arg.shape)


class Aux:
    """One auxiliary variable standing in for an argument."""

    def __init__(self, lower=None, upper=None, nonneg=False, init: Callable = init_from_argument,
                 keep_variables=False):
        self.lower, self.upper, self.nonneg = lower, upper, nonneg
        self.init = init
        self.keep_variables = keep_variables        # an argument that already is a Variable stays

    def make(self, atom, arg, k) -> Variable:
        kwargs = {}
        if self.lower is not None or self.upper is not None:
            kwargs["bounds"] = [self.lower, self.upper]
        if self.nonneg:
            kwargs["nonneg"] = True
        t = Variable(arg.shape, **kwargs)
        val = self.init(atom, arg, k)
        if val is not None:
            t.value = val
        return t


def _alias_arguments(atom, args, specs: Sequence[Optional[Aux]]) -> Tuple[Expression, Rows]:
    """atom(args) -> atom(t_0, t_1, ...) with a row t_k == args[k] per replaced argument."""
    new_args, rows = [], []
    for k, (arg, spec) in enumerate(zip(args, specs)):
        if spec is None or (spec.keep_variables and isinstance(arg, Variable)):
            new_args.append(arg)
            continue
        t = spec.make(atom, arg, k)
        new_args.append(t)
        rows.append(t == arg)
    return atom.copy(new_args), rows


PLAIN = Aux(keep_variables=True)
LOG_DOMAIN = Aux(lower=0, init=init_clipped)

ALIAS = {
    # exp_canon.py:19-26, logistic_canon.py:19-26, trig_canon.py:19-35, hyperbolic_canon.py:19-44
    at.exp: (PLAIN,), at.logistic: (PLAIN,), at.sin: (PLAIN,), at.cos: (PLAIN,),
    at.sinh: (PLAIN,), at.tanh: (PLAIN,), at.asinh: (PLAIN,),
    # log_canon.py:23-31, entr_canon.py:23-30: always a new variable on [0, inf)
    at.log: (LOG_DOMAIN,), at.entr: (LOG_DOMAIN,),
    # trig_canon.py:37-41, hyperbolic_canon.py:46-50: always a new variable on the open domain
    at.tan: (Aux(lower=-PI_TRUNCATED / 2, upper=PI_TRUNCATED / 2),),
    at.atanh: (Aux(lower=-1, upper=1),),
}


def _canon(atom) -> Tuple[Expression, Rows]:
    """Canonicalise an atom built inside a rule (its arguments are already canonical)."""
    return RULES[type(atom)](atom, atom.args)


def _alias_rule(atom, args):
    return _alias_arguments(atom, args, ALIAS[type(atom)])


# ---- atoms whose alias description depends on the instance -------------------------------------------
def _power(atom, args):
    """power_canon.py:24-54 by exponent class."""
    p = atom.p_rational
    if p == 0:
        return Constant(np.ones(atom.shape)), []
    if p == 1:
        return args[0], []
    if p < 0:
        raise NotImplementedError("The power %s is not yet supported." % p)
    whole = isinstance(p, int) and p > 1
    spec = Aux(init=init_or_domain_point, keep_variables=True) if whole else Aux(nonneg=True, init=init_or_domain_point)
    return _alias_arguments(atom, args, (spec,))


def _bilinear(atom, args):
    """multiply_canon.py:23-63 (elementwise and matrix product): nothing to do next to a constant."""
    if args[0].is_constant() or args[1].is_constant():
        return atom.copy(list(args)), []
    return _alias_arguments(atom, args, (PLAIN, PLAIN))


def _quad_over_lin(atom, args):
    """quad_over_lin_canon.py:25-54."""
    num, den = args
    if den.is_constant():                      # sum of squares over a number
        squares, rows = _canon(at.power(num, 2))
        return 1 / den.value * at.Sum(squares), rows
    return _alias_arguments(atom, args, (PLAIN, Aux(nonneg=True, init=init_if_safely_positive)))


def _rel_entr(atom, args):
    """rel_entr_canon.py:29-61: x log(x / y)."""
    x, y = args
    if x.is_constant():                         # c log c - c log(y)
        log_y, rows = _canon(at.log(y))
        c = x.value
        return c * np.log(c) - at.multiply(c, log_y), rows
    if y.is_constant():                         # -entr(x) - x log(c)
        entr_x, rows = _canon(at.entr(x))
        scaled, rows2 = _canon(at.multiply(x, np.log(y.value)))
        return -entr_x - scaled, rows + rows2
    both = Aux(lower=0, init=init_clipped_per_argument)
    return _alias_arguments(atom, args, (both, both))


# ---- rewrites through other atoms ---------------------------------------------------------------------
def _kl_div(atom, args):
    """kl_div_canon.py:21-24: rel_entr(x, y) - x + y."""
    core, rows = _canon(at.rel_entr(args[0], args[1]))
    return core - args[0] + args[1], rows


def _quotient(atom, args):
    """div_canon.py:26-55: f / g -> z with z y == f, y == g, y on [0, inf) (the denominator is taken
    to be positive); z inherits a sign bound from f."""
    f, g = args
    lower, upper = {"NONNEGATIVE": (0, None), "NONPOSITIVE": (None, 0)}.get(f.sign, (None, None))
    z = Variable(f.shape, bounds=[lower, upper]) if (lower, upper) != (None, None) else Variable(f.shape)
    y = Variable(g.shape, bounds=[0, None])
    y.value = init_clipped(atom, g, 1)
    start = np.asarray(f.value / y.value if f.value is not None else atom.point_in_domain())
    z.value = start[0] if (f.shape == () and start.shape == (1,)) else start
    return z, [at.multiply(z, y) == f, y == g]


def _geo_mean(atom, args):
    """geo_mean_canon.py:27-41: t >= 0 with log t == sum_i w_i log x_i."""
    x = args[0]
    t = Variable(atom.shape, nonneg=True)
    t.value = atom.numeric([x.value]) if (x.value is not None and np.all(x.value > MIN_INIT)) else np.ones(atom.shape)
    weights = np.array([float(w) for w in atom.w])
    log_x, rows = _alias_arguments(at.log(x), atom.args, (LOG_DOMAIN,))
    return t, [at.log(t) == at.sum(at.multiply(weights, log_x))] + rows


def _negated(make_atom):
    """minimum_canon.py:23-27 / min_canon.py:21-28: the mirrored atom of the negated arguments."""
    def rule(atom, args):
        mirrored, rows = _canon(make_atom(atom, args))
        return -mirrored, rows
    return rule


def _norm1(atom, args):
    """norm1_canon.py:22-34: sum of the absolute values."""
    magnitude, rows = _canon(at.abs(args[0]))
    return at.sum(magnitude, axis=atom.axis), rows


# ---- epigraph forms --------------------------------------------------------------------------------------
def _spread(t, like, axis):
    """t repeated to the shape of `like` along the reduced axis (max_canon.py:24-38 uses promote for a
    full reduction and products with vectors of ones for the axis forms)."""
    if axis is None:
        return at.promote(t, like.shape)
    rows, cols = like.shape
    if axis == 0:
        return Constant(np.ones((rows, 1))) @ at.reshape(t, (1, cols), order="F")
    return at.reshape(t, (rows, 1), order="F") @ Constant(np.ones((1, cols)))


def _abs(atom, args):                           # eliminate_pwl abs_canon.py:20-24
    t = Variable(atom.shape)
    return t, [t >= args[0], t >= -args[0]]


def _maximum(atom, args):                       # maximum_canon.py:21-31
    t = Variable(atom.shape)
    return t, [t >= a for a in args]


def _max(atom, args):                           # max_canon.py:24-38
    t = Variable(atom.shape)
    return t, [args[0] <= _spread(t, args[0], atom.axis)]


def _norm_inf(atom, args):                      # norm_inf_canon.py:24-37
    t = Variable(atom.shape)
    T = _spread(t, args[0], atom.axis)
    return t, [args[0] <= T, args[0] + T >= 0]


def _sum_largest(atom, args):                   # sum_largest_canon.py:21-32
    t = Variable(args[0].shape)
    q = Variable()
    return at.sum(t) + atom.k * q, [args[0] <= t + q, t >= 0]


def _pnorm(atom, args):
    """pnorm_canon.py:22-34 (p = 2 only): t >= 0 with x'x / t <= t."""
    if atom.p != 2:
        raise ValueError("Only p=2 is supported as Pnorm.")
    t = Variable(atom.shape, nonneg=True)
    ratio, rows = _canon(at.quad_over_lin(args[0], t))
    return t, rows + [ratio <= t]


def _huber(atom, args):
    """huber_canon.py:27-44: x == s + n, value n^2 + 2 M |s|."""
    n = Variable(atom.shape)
    s = Variable(atom.shape)
    square, rows_sq = _canon(at.power(n, 2))
    magnitude, rows_abs = _canon(at.abs(s))
    return square + 2 * atom.M * magnitude, rows_sq + rows_abs + [args[0] == s + n]


RULES = {kind: _alias_rule for kind in ALIAS}
RULES.update({
    at.power: _power,
    at.multiply: _bilinear,
    at.MulExpression: _bilinear,
    at.quad_over_lin: _quad_over_lin,
    at.rel_entr: _rel_entr,
    # rewrites
    at.kl_div: _kl_div,
    at.DivExpression: _quotient,
    at.geo_mean: _geo_mean,
    at.minimum: _negated(lambda atom, args: at.maximum(*[-a for a in args])),
    at.min: _negated(lambda atom, args: at.max(-args[0], axis=atom.axis, keepdims=atom.keepdims)),
    at.norm1: _norm1,
    # epigraphs
    at.abs: _abs,
    at.maximum: _maximum,
    at.max: _max,
    at.norm_inf: _norm_inf,
    at.sum_largest: _sum_largest,
    at.Pnorm: _pnorm,
    at.huber: _huber,
})
SMOOTH_CANON_METHODS = RULES          # the reference's name for the table (dnlp2smooth.py:24)


class Dnlp2Smooth:
    """Post-order rewrite of the objective and of every constraint (dnlp2smooth.py:27-111): auxiliary
    rows come before the row that needed them."""

    def apply(self, problem):
        from .problem import Problem
        objective, rows = self.canonicalize_tree(problem.objective, True)
        cons_id_map = {}
        for constraint in problem.constraints:
            canon, aux_rows = self.canonicalize_tree(constraint, False)
            rows = rows + aux_rows + [canon]
            cons_id_map[constraint.id] = canon.id
        return Problem(objective, rows), {"cons_id_map": cons_id_map}

    def canonicalize_tree(self, node, affine_above: bool) -> Tuple[object, Rows]:
        below_affine = affine_above and type(node) not in RULES
        children, rows = [], []
        for child in node.args:
            canon_child, child_rows = self.canonicalize_tree(child, below_affine)
            children.append(canon_child)
            rows.extend(child_rows)
        canon, own_rows = self.canonicalize_expr(node, children, affine_above)
        return canon, rows + own_rows

    def canonicalize_expr(self, node, args, affine_above: bool):
        # parameter-free constant subtrees stay as they are (dnlp2smooth.py:104-106)
        if isinstance(node, Expression) and node.is_constant() and not node.parameters():
            return node, []
        rule = RULES.get(type(node))
        return rule(node, args) if rule is not None else (node.copy(args), [])
