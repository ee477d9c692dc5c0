"""dnlp_amd — MI355X-native disciplined-NLP solve path.

Drop-in for the one path of cvxgrp/DNLP behind `Problem.solve(nlp=True)`: the atom DAG is
canonicalised (dnlp2smooth), lowered once to a flat device tape, and f/∇f/g/Jac_g/∇²L plus
the interior-point Newton loop run as hand-written HIP kernels for gfx950 behind the C-ABI
declared in include/dnlp_hip.h.  There is no CPU fallback: solving without the built
extension and an MI355X raises `DeviceUnavailableError`.

The public names below mirror the reference's `cvxpy` namespace for this path, so a test
written against the reference reads the same here (`import dnlp_amd as cp`).
"""
from .settings import (  # noqa: F401
    COPT, HIP, INFEASIBLE, IPOPT, KNITRO, OPTIMAL, OPTIMAL_INACCURATE, SOLVER_ERROR,
    UNBOUNDED, USER_LIMIT,
)
from .error import DeviceUnavailableError, DNLPError, SolverError  # noqa: F401
from .expressions import Constant, DeviceMatrix, Expression, Parameter, Variable  # noqa: F401
from .atoms import (  # noqa: F401
    AddExpression, DivExpression, MulExpression, NegExpression, Pnorm, Promote, QuadForm, Sum,
    abs, asinh, atanh, broadcast_to, cos, entr, exp, geo_mean, hstack, huber, index, kl_div,
    log, logistic, matmul, max, maximum, min, minimum, multiply, norm, norm1, norm2, norm_inf,
    pnorm, power, promote, quad_form, quad_over_lin, rel_entr, reshape, sin, sinh,
    special_index, sqrt, square, sum, sum_largest, sum_smallest, sum_squares, tan, tanh,
    transpose, vec, vstack, xexp,
)
from .constraints import Equality, Inequality, NonNeg, NonPos, Zero  # noqa: F401
from .problem import Maximize, Minimize, Problem  # noqa: F401

__version__ = "0.1.0"
