"""Objectives and Problem with the `solve(nlp=True)` entry point.

Mirror of the NLP branch of the reference's Problem (problems/problem.py:298-303 `is_dnlp`,
:1219-1277 `_solve` nlp branch, :1603-1633 `unpack_results`, :1643-1697
`set_random_NLP_initial_point`) and of problems/objective.py:159-163,236-240.  The
reduction chain is the reference's — [FlipObjective] -> (CvxAttr2Constr: nothing to lower
for the attributes this path supports, bounds stay on variables) -> Dnlp2Smooth ->
NLPsolver.apply — but the last stage lowers to the MI355X device tape and the solver is the
on-device interior-point loop instead of cyipopt callbacks.
"""
from __future__ import annotations

import warnings

import numpy as np

from . import settings as s
from .constraints import Constraint
from .error import DNLPError, SolverError
from .expressions import Expression, unique_list


class Objective:
    NAME = "objective"

    def __init__(self, expr):
        self.args = [Expression.cast_to_const(expr)]
        if not self.args[0].is_scalar():
            raise ValueError("The '%s' objective must resolve to a scalar." % self.NAME)

    @property
    def expr(self):
        return self.args[0]

    @property
    def value(self):
        v = self.args[0].value
        if v is None:
            return None
        return float(np.asarray(v).reshape(-1)[0])

    def variables(self):
        return self.args[0].variables()

    def parameters(self):
        return self.args[0].parameters()

    def copy(self, args=None):
        return type(self)(*(self.args if args is None else args))

    def is_constant(self):
        return False

    def __str__(self):
        return "%s %s" % (self.NAME, self.args[0].name())


class Minimize(Objective):
    NAME = "minimize"

    def is_dnlp(self):
        """Minimize needs an ESR expression (reference objective.py:159-163)."""
        return self.args[0].is_esr()


class Maximize(Objective):
    NAME = "maximize"

    def is_dnlp(self):
        """Maximize needs an HSR expression (reference objective.py:236-240)."""
        return self.args[0].is_hsr()


class SolverStats:
    def __init__(self, solver_name, num_iters=None, solve_time=None, extra_stats=None):
        self.solver_name = solver_name
        self.num_iters = num_iters
        self.solve_time = solve_time
        self.extra_stats = extra_stats


class Problem:
    """An optimization problem solved through the disciplined-NLP path."""

    def __init__(self, objective, constraints=None):
        if constraints is None:
            constraints = []
        if not isinstance(objective, (Minimize, Maximize)):
            raise Exception("Problem objective must be Minimize or Maximize.")
        for c in constraints:
            if not isinstance(c, Constraint):
                raise ValueError("Problem has an invalid constraint of type %s" % type(c))
        self._objective = objective
        self._constraints = list(constraints)
        self._value = None
        self._status = None
        self._solver_stats = None
        self._solution = None

    @property
    def objective(self):
        return self._objective

    @property
    def constraints(self):
        return self._constraints[:]

    @property
    def value(self):
        return self._value

    @property
    def status(self):
        return self._status

    @property
    def solver_stats(self):
        return self._solver_stats

    def variables(self):
        """Objective variables first, then each constraint's, first occurrence kept
        (reference problem.py `variables`)."""
        vars_ = list(self.objective.variables())
        for c in self._constraints:
            vars_ += c.variables()
        return unique_list(vars_)

    def parameters(self):
        ps = list(self.objective.parameters())
        for c in self._constraints:
            ps += c.parameters()
        return unique_list(ps)

    def copy(self, args=None):
        if args is None:
            args = [self.objective, self._constraints]
        return Problem(args[0], args[1])

    def is_dnlp(self) -> bool:
        """reference problem.py:298-303."""
        return all(e.is_dnlp() for e in self._constraints + [self.objective])

    def __str__(self):
        out = str(self.objective)
        if self._constraints:
            out += "\nsubject to " + "\n           ".join(c.name() for c in self._constraints)
        return out

    # ------------------------------------------------------------------------------
    def solve(self, solver=None, warm_start=True, verbose=False, nlp=False, **kwargs):
        """Solve the problem.  Only `nlp=True` is implemented (this package is the
        disciplined-NLP hot path, not a general CVXPY)."""
        if not nlp:
            raise NotImplementedError(
                "dnlp_amd implements only the solve(nlp=True) path of the reference.")
        return self._solve_nlp(solver, warm_start, verbose, **kwargs)

    def _build_chain(self, solver):
        from .nlp_solver import HIPNLP
        if solver in (None, s.IPOPT, s.HIP):
            nlp_solver = HIPNLP()
        elif solver == s.KNITRO:
            raise SolverError("KNITRO is a commercial solver and is not part of this build.")
        elif solver == s.COPT:
            raise NotImplementedError("COPT NLP interface is a stub in the reference too.")
        else:
            raise SolverError("Solver %s is not supported for NLP problems." % solver)
        return NLPChain(type(self.objective) == Maximize, nlp_solver)

    def _solve_nlp(self, solver, warm_start, verbose, **kwargs):
        """reference problem.py:1219-1277."""
        if not self.is_dnlp():
            raise DNLPError("The problem you specified is not DNLP.")
        chain = self._build_chain(solver)
        best_of = kwargs.pop("best_of", 1)
        if not isinstance(best_of, int) or best_of < 1:
            raise ValueError("best_of must be a positive integer.")
        if best_of == 1:
            direct = str(kwargs.get("algorithm", "")) in ("lbfgs", "reduced-lbfgs") and \
                str(kwargs.get("fused_objective", "yes")) not in ("no", "False", "0")
            canon_problem, inverse_data = chain.apply(self, direct_fused=direct)
            # Dual warm start (IPOPT warm_start_init_point=yes; the reference accepts `warm_start` and
            # drops it, ipopt_nlpif.py:126-127): the previous solve's canonical point and multipliers
            # are reused when the canonical dimensions are unchanged.
            prev = getattr(self, "_nlp_last", None)
            if str(kwargs.get("warm_start_init_point", "no")) in ("yes", "True", "1") and prev is not None \
                    and prev["x"].size == len(canon_problem["x0"]) and prev["mult_g"].size == len(canon_problem["cl"]):
                canon_problem["x0"] = prev["x"]
                canon_problem["warm_duals"] = (prev["mult_g"], prev["mult_x_L"], prev["mult_x_U"])
            solution = chain.solver.solve_via_data(canon_problem, warm_start, verbose,
                                                   solver_opts=kwargs)
            if "mult_g" in solution:
                self._nlp_last = {k: np.array(solution[k], dtype=float) for k in ("x", "mult_g", "mult_x_L", "mult_x_U")}
            self.unpack_results(solution, chain, inverse_data)
            return self.value
        best_obj, best_solution, all_objs = float("inf"), None, np.zeros(best_of)
        best_inv = None
        # Multistart as a batch dimension: the runs differ only in the start
        # point, so on the device path all of them are solved by ONE kernel launch
        # (dnlp_solve_batch); ranking and bookkeeping stay exactly the reference's.
        batched = self._best_of_batched(chain, best_of, kwargs) if kwargs.pop("batch", True) else None
        for run in range(best_of):
            if batched is None:
                print("Starting NLP solve %d of %d" % (run + 1, best_of))
                self.set_random_NLP_initial_point(run)
                canon_problem, inverse_data = chain.apply(self)
                solution = chain.solver.solve_via_data(canon_problem, warm_start, verbose,
                                                       solver_opts=kwargs)
            else:
                solution, inverse_data = batched[run]
            # the reference ranks runs by the ORIGINAL objective at the unpacked point.  Only a run
            # that returned a point can be ranked: a failed run writes no variable values, and the
            # objective read through the shared Variable state would be some other run's (or the
            # random start's) — such a run gets +inf and can never win
            obj_value = None
            if chain.solver.STATUS_MAP.get(solution["status"]) in s.SOLUTION_PRESENT:
                self.unpack_results(solution, chain, inverse_data, raise_on_error=False)
                obj_value = self.objective.value
            if type(self.objective) == Maximize and obj_value is not None:
                obj_value = -obj_value
            all_objs[run] = np.inf if obj_value is None else obj_value
            if obj_value is not None and obj_value < best_obj:
                best_obj, best_solution, best_inv = obj_value, solution, inverse_data
        if best_solution is None:
            best_solution, best_inv = solution, inverse_data
        if type(self.objective) == Maximize:
            all_objs = -all_objs
        best_solution["all_objs_from_best_of"] = all_objs
        self.unpack_results(best_solution, chain, best_inv)
        return self.value

    def _best_of_batched(self, chain, best_of, opts):
        """All `best_of` runs in one device launch; None when the chain's solver has no batch
        entry point or the lowered runs do not share one tape (then the serial loop runs)."""
        solve_rows = getattr(chain.solver, "solve_batch_via_data", None)
        if solve_rows is None:
            return None
        if str(opts.get("hessian_approximation", "exact")) == "limited-memory":
            return None          # the quasi-Newton mode belongs to the host-driven loop: the starts run one by one
        from .batch import instance_data, same_structure
        lowered = []
        for run in range(best_of):
            self.set_random_NLP_initial_point(run)
            data, inv = chain.apply(self, make_handle=False)
            if lowered and not same_structure(lowered[0][0]["tape_arrays"], data["tape_arrays"]):
                return None
            lowered.append((data, inv))
        if lowered[0][0]["tape"].dense_blocks:
            return None
        rows = np.stack([instance_data(d["tape_arrays"]) for d, _ in lowered])
        print("Solving %d NLP starts in one batched launch" % best_of)
        infos = solve_rows(lowered[0][0], rows, opts)
        return [(infos[k], lowered[k][1]) for k in range(best_of)]

    def set_random_NLP_initial_point(self, run):
        """Uniform sample inside sample_bounds / finite bounds for variables the user did
        not initialise (reference problem.py:1643-1697)."""
        if run == 0:
            self._user_initials = {}
            for var in self.variables():
                if var.value is not None:
                    self._user_initials[id(var)] = var.value
            for var in self.variables():
                if var.sample_bounds is not None:
                    continue
                if var.bounds is None:
                    if id(var) in self._user_initials:
                        continue
                    raise ValueError("Variable %s has no sample_bounds, bounds or initial "
                                     "value for best_of sampling." % var.name())
                lb, ub = var.bounds
                if np.all(np.isfinite(lb)) and np.all(np.isfinite(ub)):
                    var.sample_bounds = [lb, ub]
                elif id(var) not in self._user_initials:
                    raise ValueError("Variable %s needs finite sample_bounds." % var.name())
        for var in self.variables():
            if id(var) in self._user_initials:
                var.value = self._user_initials[id(var)]
            elif var.sample_bounds is not None:
                low, high = var.sample_bounds
                var.value = np.random.uniform(low=low, high=high, size=var.shape)

    def unpack_results(self, solution, chain, inverse_data, raise_on_error=True):
        """reference problem.py:1603-1633 + chain.invert."""
        sol = chain.invert(solution, inverse_data)
        if sol["status"] in s.INACCURATE:
            warnings.warn("Solution may be inaccurate. Try another solver, adjusting the "
                          "solver settings, or solve with verbose=True for more information.")
        if sol["status"] in s.ERROR and raise_on_error:
            raise SolverError("Solver '%s' failed. Try another solver, or solve with "
                              "verbose=True for more information." % chain.solver.name())
        self._status = sol["status"]
        if sol["status"] in s.SOLUTION_PRESENT:
            for var in self.variables():
                if var.id in sol["primal_vars"]:
                    var.save_value(sol["primal_vars"][var.id])
            self._value = sol["opt_val"]
        elif sol["status"] in s.INF_OR_UNB:
            for var in self.variables():
                var.save_value(None)
            self._value = sol["opt_val"]
        self._solution = sol
        attr = sol["attr"]
        self._solver_stats = SolverStats(chain.solver.name(), attr.get(s.NUM_ITERS),
                                         attr.get(s.SOLVE_TIME), attr.get(s.EXTRA_STATS))


class NLPChain:
    """[FlipObjective] -> Dnlp2Smooth -> NLPsolver.apply (reference problem.py:1220-1238,
    reductions/chain.py:54-85, flip_objective.py:29-66)."""

    def __init__(self, flip: bool, solver):
        self.flip = flip
        self.solver = solver

    @staticmethod
    def _signature(problem):
        """What the lowered tape depends on besides variable VALUES: the expression objects and the
        parameter values.  (Constants are taken as immutable, as the reference's own caches do.)"""
        params = tuple((id(p), None if p.value is None else np.asarray(p.value, dtype=float).tobytes())
                       for p in problem.parameters())
        return (id(problem.objective.expr), tuple(id(c) for c in problem._constraints), params)

    def apply(self, problem, make_handle=True, direct_fused=False):
        from .dnlp2smooth import Dnlp2Smooth
        original = problem
        if self.flip:
            problem = Problem(Minimize(-problem.objective.expr), problem.constraints)
        if direct_fused and make_handle:
            out = self._apply_direct(original, problem)
            if out is not None:
                return out
        smooth, _ = Dnlp2Smooth().apply(problem)
        if not make_handle:
            return self.solver.apply(smooth, user_variables=problem.variables(), make_handle=False)
        # Lowering, tape upload, sparse plan and generated kernels depend on the problem's structure and
        # constants only: a second solve of the same Problem object (another start, other options, a
        # warm start) reuses them and rebuilds just the start point.  The reference re-runs its whole
        # chain on every solve (problem.py:1243-1246, :1256-1269).
        sig = self._signature(original)
        cached = getattr(original, "_nlp_cache", None)
        reapply = getattr(self.solver, "reapply", None)
        if cached is not None and cached["sig"] == sig and cached["solver"] is type(self.solver) and reapply is not None:
            hit = reapply(smooth, cached)
            if hit is not None:
                return hit
        from .fused import build_fused_spec
        try:
            out = self.solver.apply(smooth, user_variables=problem.variables(), fused_spec=build_fused_spec(problem))
        except TypeError:      # a solver interface without the fused-objective hook
            out = self.solver.apply(smooth, user_variables=problem.variables())
        if cached is not None and cached["data"].get("handle") is not None:
            cached["data"]["handle"].close()
        original._nlp_cache = {"sig": sig, "solver": type(self.solver), "data": out[0]}
        return out

    def _apply_direct(self, original, problem):
        """algorithm='lbfgs' on an unconstrained elementwise-sum objective (BASELINE config C2): the solve
        evaluates f and grad f through the fused native-form program only (dnlp_amd/fused.py), so the
        smooth canonical form -- 7 n variables and 3 n defining equalities for the Rosenbrock chain, 0.22 s
        of lowering and an 80 MB tape at n = 1e5 -- is never needed.  The tape that is uploaded holds the
        user's variables, a zero objective and the fused program.  None when the objective has no fused form
        (the canonical path then serves the reduced solve as before)."""
        from .atoms import sum as sum_atom
        from .fused import build_fused_spec
        if getattr(self.solver, "reapply", None) is None:
            return None
        sig = ("direct",) + self._signature(original)
        cached = getattr(original, "_nlp_cache", None)
        if cached is not None and cached["sig"] == sig and cached["solver"] is type(self.solver):
            # (a cached direct handle: the fused program need not be emitted again to know that it exists)
            light = cached["light"]
            hit = self.solver.reapply(light, cached)
            if hit is not None:
                return hit
        spec = build_fused_spec(problem)
        if spec is None:
            return None
        zero = None
        for v in problem.variables():
            term = sum_atom(0.0 * v)
            zero = term if zero is None else zero + term
        light = Problem(Minimize(zero), [])
        out = self.solver.apply(light, user_variables=problem.variables(), fused_spec=spec)
        if not out[0].get("fused"):
            if out[0].get("handle") is not None:
                out[0]["handle"].close()
            return None
        if cached is not None and cached["data"].get("handle") is not None:
            cached["data"]["handle"].close()
        original._nlp_cache = {"sig": sig, "solver": type(self.solver), "data": out[0], "light": light}
        return out

    def invert(self, solution, inverse_data):
        sol = self.solver.invert(solution, inverse_data)
        if self.flip and sol["opt_val"] is not None:
            sol["opt_val"] = -sol["opt_val"]
        return sol
