"""Status strings and solver names for the nlp=True path.

Mirrors the subset of the reference's cvxpy/settings.py:49-68,101-108,153-156 that the
NLP path touches (status strings returned by Problem.solve, solver name constants).
"""
OPTIMAL = "optimal"
OPTIMAL_INACCURATE = "optimal_inaccurate"
INFEASIBLE = "infeasible"
INFEASIBLE_INACCURATE = "infeasible_inaccurate"
UNBOUNDED = "unbounded"
UNBOUNDED_INACCURATE = "unbounded_inaccurate"
INFEASIBLE_OR_UNBOUNDED = "infeasible_or_unbounded"
USER_LIMIT = "user_limit"
SOLVER_ERROR = "solver_error"

SOLUTION_PRESENT = [OPTIMAL, OPTIMAL_INACCURATE, USER_LIMIT]
INF_OR_UNB = [INFEASIBLE, INFEASIBLE_INACCURATE, UNBOUNDED, UNBOUNDED_INACCURATE,
              INFEASIBLE_OR_UNBOUNDED]
INACCURATE = [OPTIMAL_INACCURATE, INFEASIBLE_INACCURATE, UNBOUNDED_INACCURATE, USER_LIMIT]
ERROR = [SOLVER_ERROR]

# solver names (reference: cvxpy/settings.py + reductions/solvers/defines.py:87-91,114)
IPOPT = "IPOPT"      # reference default NLP solver name; here it selects the HIP interior point path
HIP = "HIP"          # explicit name of the MI355X-native solver
KNITRO = "KNITRO"
COPT = "COPT"

NUM_ITERS = "num_iters"
SOLVE_TIME = "solve_time"
EXTRA_STATS = "solver_specific_stats"

NONNEG = "NONNEGATIVE"
NONPOS = "NONPOSITIVE"
ZERO = "ZERO"
UNKNOWN = "UNKNOWN"
