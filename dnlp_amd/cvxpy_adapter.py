"""Adapter for the real (reference) `cvxpy` package: solve a cvxpy Problem on the MI355X path.

Where the reference fork of cvxpy is installed, this module is the literal drop-in behind
`cp.Problem.solve(...)`: it uses CVXPY's own plugin hook `Problem.register_solve`
(problems/problem.py:622-648) so nothing in the reference tree has to change:

    import cvxpy as cp
    from dnlp_amd import cvxpy_adapter
    cvxpy_adapter.register(cp)                    # once
    prob.solve(method="dnlp_hip", tol=1e-8)       # instead of prob.solve(nlp=True)

The reference's own reductions run unchanged (FlipObjective -> CvxAttr2Constr -> Dnlp2Smooth ->
NLPsolver.apply, problems/problem.py:1220-1243); the canonicalised cvxpy expression trees of
`data["problem"]` are then translated node by node into dnlp_amd nodes (same class names and
data by construction), lowered to the device tape and solved by `dnlp_solve`.  Results are
unpacked through the reference's own `IPOPT.invert` (ipopt_nlpif.py:75-102) and
`Problem.unpack_results`.

cvxpy is not needed (and not present) on the GPU box; this module imports it lazily.
"""
from __future__ import annotations

import numpy as np

from . import atoms as at
from .expressions import Constant, Variable
from .lowering import lower_problem
from .tape import serialize, tape_arrays

_SIMPLE = {
    "NegExpression": at.NegExpression, "MulExpression": at.MulExpression, "multiply": at.multiply,
    "DivExpression": at.DivExpression, "transpose": at.transpose, "exp": at.exp, "log": at.log,
    "entr": at.entr, "logistic": at.logistic, "sin": at.sin, "cos": at.cos, "tan": at.tan,
    "sinh": at.sinh, "tanh": at.tanh, "asinh": at.asinh, "atanh": at.atanh, "xexp": at.xexp,
    "rel_entr": at.rel_entr, "QuadForm": at.QuadForm, "quad_over_lin": at.quad_over_lin,
    "Hstack": at.Hstack, "Vstack": at.Vstack,
}


class Translator:
    """cvxpy expression tree -> dnlp_amd expression tree (memoised by node identity)."""

    def __init__(self):
        self.memo = {}
        self.var_map = {}      # cvxpy variable id -> dnlp_amd Variable

    def variable(self, v):
        if v.id not in self.var_map:
            nv = Variable(v.shape, name=v.name())
            if v.value is not None:
                nv.value = np.asarray(v.value, dtype=float)
            self.var_map[v.id] = nv
        return self.var_map[v.id]

    def convert(self, e):
        key = id(e)
        if key in self.memo:
            return self.memo[key]
        out = self._convert(e)
        if tuple(out.shape) != tuple(e.shape):
            raise AssertionError("shape mismatch translating %s: %s vs %s"
                                 % (type(e).__name__, out.shape, e.shape))
        self.memo[key] = out
        return out

    def _convert(self, e):
        name = type(e).__name__
        if name == "Variable":
            return self.variable(e)
        if name in ("Constant", "Parameter") or (e.is_constant() and not e.variables()):
            val = e.value
            if val is None:
                raise ValueError("constant / parameter without a value")
            return Constant(val)
        args = [self.convert(a) for a in e.args]
        if name in _SIMPLE:
            return _SIMPLE[name](*args)
        if name == "AddExpression":
            return at.AddExpression(args)
        if name == "index":
            # the reference keeps the user's key in _orig_key (index.py:60-66, numeric :88-90)
            return at.index(args[0], e._orig_key)
        if name == "special_index":
            return at.special_index(args[0], e.key)
        if name == "Promote":
            return at.Promote(args[0], e.shape)
        if name == "broadcast_to":
            return at.broadcast_to(args[0], e.shape)
        if name == "reshape":
            return at.reshape(args[0], e.shape, getattr(e, "order", "F"))
        if name == "Sum":
            return at.Sum(args[0], e.axis, e.keepdims)
        if name == "power":
            p = e._p_orig if not hasattr(e._p_orig, "value") else e.p.value
            return at.power(args[0], p, e.max_denom)
        if name in ("psd_wrap", "nonneg_wrap", "nonpos_wrap", "symmetric_wrap"):
            return args[0]
        raise NotImplementedError("no dnlp_amd counterpart for cvxpy atom %s" % name)


def tape_from_cvxpy(data):
    """`data` = the dict built by the reference's NLPsolver.apply (nlp_solver.py:62-79).
    Returns (tape, tape_arrays) for the same canonical problem and variable order."""
    problem = data["problem"]
    tr = Translator()
    variables = [tr.variable(v) for v in problem.variables()]
    obj = tr.convert(problem.objective.expr)
    cons = [tr.convert(c.args[0]) for c in problem.constraints]
    tape = lower_problem(obj, cons, variables)
    arrays = tape_arrays(tape, np.asarray(data["x0"], float), np.asarray(data["lb"], float),
                         np.asarray(data["ub"], float), np.asarray(data["cl"], float),
                         np.asarray(data["cu"], float))
    return tape, arrays


def tape_blob_from_cvxpy(data) -> bytes:
    return serialize(tape_from_cvxpy(data)[1])


def _device_solve(blob, tape, x0, options):
    from . import _capi
    h = _capi.DeviceProblem(blob, tape)
    for k, v in options.items():
        h.set_option(k, v)
    info = h.solve(x0)
    h.close()
    return info


def make_solve_method(cp, solve_fn=None):
    """Build the function handed to `Problem.register_solve`.  `solve_fn(blob, tape, x0, opts)`
    defaults to the MI355X library; tests inject the CPU oracle."""
    from cvxpy.reductions.cvx_attr2constr import CvxAttr2Constr
    from cvxpy.reductions.dnlp2smooth.dnlp2smooth import Dnlp2Smooth
    from cvxpy.reductions.flip_objective import FlipObjective
    from cvxpy.reductions.solvers.nlp_solvers.ipopt_nlpif import IPOPT
    from cvxpy.reductions.solvers.solving_chain import SolvingChain
    from cvxpy import error

    solve_fn = solve_fn or _device_solve
    defaults = {"mu_strategy": "adaptive", "tol": 1e-7, "bound_relax_factor": 0.0,
                "hessian_approximation": "exact", "derivative_test": "none",
                "least_square_init_duals": "yes"}      # ipopt_nlpif.py:153-160

    def solve_dnlp_hip(self, verbose=False, **kwargs):
        if not self.is_dnlp():
            raise error.DNLPError("The problem you specified is not DNLP.")
        kwargs.pop("nlp", None)
        red = ([FlipObjective()] if type(self.objective) == cp.Maximize else []) + \
            [CvxAttr2Constr(reduce_bounds=False), Dnlp2Smooth(), IPOPT()]
        chain = SolvingChain(reductions=red)
        data, inverse_data = chain.apply(problem=self)
        tape, arrays = tape_from_cvxpy(data)
        opts = dict(defaults)
        opts.update(kwargs)
        if verbose and "print_level" not in opts:
            opts["print_level"] = 5
        info = solve_fn(serialize(arrays), tape, np.asarray(data["x0"], float), opts)
        solution = {"status": info["status"], "obj_val": info["obj_val"], "x": info["x"],
                    "iterations": info["iterations"]}
        self.unpack_results(solution, chain, inverse_data)
        return self.value

    return solve_dnlp_hip


def register(cp, name="dnlp_hip", solve_fn=None):
    cp.Problem.register_solve(name, make_solve_method(cp, solve_fn))
    return name
