"""Import the reference cvxpy fork in place (this container only; never on the GPU box).

Recipe from SURVEY.md Appendix C: py3.10 `typing.Self` shim + generated version module.
Nothing is copied from or written into /root/reference.
"""
import sys
import types
import typing

import typing_extensions

REF = "/root/reference"


def import_reference():
    if "cvxpy" in sys.modules and getattr(sys.modules["cvxpy"], "__file__", "").startswith(REF):
        return sys.modules["cvxpy"]
    typing.Self = typing_extensions.Self
    v = types.ModuleType("cvxpy.version")
    v.version = v.full_version = "1.8.0.dev0"
    v.short_version = "1.8.0"
    v.git_revision = "Unknown"
    v.commit_count = "0"
    v.release = False
    sys.modules["cvxpy.version"] = v
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import cvxpy as cp
    return cp


def ref_reductions(cp, prob):
    """The reduction list of problem.py:1220-1228 (the last entry is the NLPsolver)."""
    from cvxpy.reductions.cvx_attr2constr import CvxAttr2Constr
    from cvxpy.reductions.dnlp2smooth.dnlp2smooth import Dnlp2Smooth
    from cvxpy.reductions.flip_objective import FlipObjective
    from cvxpy.reductions.solvers.nlp_solvers.ipopt_nlpif import IPOPT
    return ([FlipObjective()] if type(prob.objective) == cp.Maximize else []) + \
        [CvxAttr2Constr(reduce_bounds=False), Dnlp2Smooth(), IPOPT()]


def ref_chain_apply(cp, prob):
    """Mirror of problem.py:1220-1243 without the cyipopt call."""
    from cvxpy.reductions.cvx_attr2constr import CvxAttr2Constr
    from cvxpy.reductions.dnlp2smooth.dnlp2smooth import Dnlp2Smooth
    from cvxpy.reductions.flip_objective import FlipObjective
    from cvxpy.reductions.solvers.nlp_solvers.ipopt_nlpif import IPOPT
    from cvxpy.reductions.solvers.solving_chain import SolvingChain
    red = ([FlipObjective()] if type(prob.objective) == cp.Maximize else []) + \
        [CvxAttr2Constr(reduce_bounds=False), Dnlp2Smooth(), IPOPT()]
    chain = SolvingChain(reductions=red)
    data, inv = chain.apply(problem=prob)
    return data, inv, chain
