"""NLP standard form, device oracles and the HIP solver interface.

Mirror of the reference's `NLPsolver` / `Bounds` / `Oracles`
(reductions/solvers/nlp_solvers/nlp_solver.py:34-427) and of the `IPOPT` solver interface
(nlp_solvers/ipopt_nlpif.py:25-184) for the MI355X path:

* `Bounds` reproduces cl/cu per constraint kind, lb/ub from `bounds` ∩ nonneg/nonpos and the
  initial point rule (value, else midpoint / bound±1 / 0) — nlp_solver.py:81-178.
* `DeviceOracles` has the reference's seven callback names (`objective`, `gradient`,
  `constraints`, `jacobian`, `jacobianstructure`, `hessian`, `hessianstructure`,
  `intermediate`), but every evaluation is one call into libdnlp_hip.so (HIP kernels over
  the tape lowered once by `lowering.lower_problem`), not a Python tree walk.
* `HIPNLP.solve_via_data / invert` keep the reference's signatures, option names, returned
  record (`status`, `obj_val`, `x`, `iterations`, plus cyipopt's `g`, `mult_g`,
  `mult_x_L`, `mult_x_U`) and IPOPT status integers; the interior-point loop itself runs
  on the device (`dnlp_solve`).
"""
from __future__ import annotations

import numpy as np

import warnings

from . import settings as s
from .constraints import (Equality, Inequality, NonPos, lower_equality,
                          lower_ineq_to_nonneg, nonpos2nonneg)
from .lowering import lower_problem
from .tape import serialize, tape_arrays


class Bounds:
    """reference nlp_solver.py:81-178, with ONE variable order.  The reference builds lb / ub / x0 in
    the order of `problem.variables()` (:84,116,163) while its oracles — and `InverseData` — walk the
    problem whose inequalities were rewritten to `rhs - lhs >= 0` (:200), which can list the variables
    in another order; bounds then land on the wrong variables.  Here everything follows the lowered
    problem's order, the one the tape and the returned `x` use."""

    def __init__(self, problem):
        self.problem = problem
        self.get_constraint_bounds()
        self.main_var = self.new_problem.variables()
        self.get_variable_bounds()
        self.construct_initial_point()

    def get_constraint_bounds(self):
        lower, upper, new_constr = [], [], []
        for constraint in self.problem.constraints:
            if isinstance(constraint, Equality):
                upper.append(np.zeros(constraint.size))
                new_constr.append(lower_equality(constraint))
            elif isinstance(constraint, Inequality):
                upper.append(np.full(constraint.size, np.inf))
                new_constr.append(lower_ineq_to_nonneg(constraint))
            elif isinstance(constraint, NonPos):
                upper.append(np.full(constraint.size, np.inf))
                new_constr.append(nonpos2nonneg(constraint))
            else:
                raise ValueError("Constraint type %s is not supported on the NLP path."
                                 % type(constraint).__name__)
            lower.append(np.zeros(constraint.size))
        self.new_problem = self.problem.copy([self.problem.objective, new_constr])
        self.cl = np.concatenate(lower) if lower else np.zeros(0)
        self.cu = np.concatenate(upper) if upper else np.zeros(0)

    def get_variable_bounds(self):
        lows, ups = [], []
        for var in self.main_var:
            size = var.size
            if var.bounds:
                lb = np.asarray(var.bounds[0], dtype=float).flatten(order="F")
                ub = np.asarray(var.bounds[1], dtype=float).flatten(order="F")
                if var.attributes["nonneg"]:
                    lb = np.maximum(lb, 0)
                if var.attributes["nonpos"]:
                    ub = np.minimum(ub, 0)
            else:
                lb = np.zeros(size) if var.is_nonneg() else np.full(size, -np.inf)
                ub = np.zeros(size) if var.is_nonpos() else np.full(size, np.inf)
            lows.append(lb)
            ups.append(ub)
        self.lb = np.concatenate(lows) if lows else np.zeros(0)
        self.ub = np.concatenate(ups) if ups else np.zeros(0)

    def construct_initial_point(self):
        initial_values = []
        offset = 0
        for var in self.main_var:
            if var.value is not None:
                initial_values.append(np.atleast_1d(var.value).flatten(order="F"))
            else:
                lb = self.lb[offset:offset + var.size]
                ub = self.ub[offset:offset + var.size]
                lb_finite = np.isfinite(lb)
                ub_finite = np.isfinite(ub)
                lb0 = np.where(lb_finite, lb, 0.0)
                ub0 = np.where(ub_finite, ub, 0.0)
                init = (lb_finite * ub_finite * 0.5 * (lb0 + ub0) +
                        lb_finite * (~ub_finite) * (lb0 + 1.0) +
                        (~lb_finite) * ub_finite * (ub0 - 1.0))
                initial_values.append(init)
            offset += var.size
        self.x0 = np.concatenate(initial_values, axis=0) if initial_values else np.zeros(0)


class InverseData:
    """Variable offsets/shapes of the lowered problem (reference inverse_data.py)."""

    def __init__(self, problem):
        self.var_offsets, self.var_shapes = {}, {}
        off = 0
        for v in problem.variables():
            self.var_offsets[v.id] = off
            self.var_shapes[v.id] = v.shape
            off += v.size
        self.x_length = off
        self.offset = 0.0


def build_nlp_data(problem, user_variables=None, fused_spec=None):
    """Bounds + tape lowering for a smooth-canonical problem.  Returns the data dict without
    touching the device (used by CPU tests and by `HIPNLP.apply`).  `user_variables` (the
    variables of the problem as the user wrote it) enables the reduced-space arrays."""
    bounds = Bounds(problem)
    new_problem = bounds.new_problem
    variables = bounds.main_var
    tape = lower_problem(new_problem.objective.expr,
                         [c.args[0] for c in new_problem.constraints], variables)
    inverse_data = InverseData(new_problem)
    data = {
        "problem": new_problem,
        "cl": bounds.cl, "cu": bounds.cu, "lb": bounds.lb, "ub": bounds.ub, "x0": bounds.x0,
        "tape": tape,
    }
    data["tape_arrays"] = tape_arrays(tape, bounds.x0, bounds.lb, bounds.ub, bounds.cl, bounds.cu)
    if user_variables is not None:
        from .reduced import reduction_arrays
        red = reduction_arrays(new_problem, tape, {id(v) for v in user_variables})
        if red is not None:
            data["tape_arrays"].update(red)
            if fused_spec is not None:
                # fused native-form objective (dnlp_amd/fused.py): variable references become
                # positions in the free-variable vector of the reduced-space solve
                from .fused import fused_arrays
                free = np.asarray(red["free_idx"], dtype=np.int64)
                off, base = 0, {}
                for v in new_problem.variables():
                    if any(v is u for u in user_variables):
                        base[v.id] = int(np.searchsorted(free, off))
                    off += v.size
                data["tape_arrays"].update(fused_arrays(fused_spec, base, free.size))
        data["reducible"] = red is not None
        data["fused"] = red is not None and fused_spec is not None
    return data, inverse_data


class DeviceOracles:
    """The reference's `Oracles` protocol (nlp_solver.py:181-427) over the device tape."""

    def __init__(self, handle, n, m):
        self._h = handle
        self.n, self.m = n, m
        self.iterations = 0
        self._jac_struct = None
        self._hess_struct = None

    def objective(self, x):
        return self._h.eval_f(x)

    def gradient(self, x):
        return self._h.eval_grad_f(x)

    def constraints(self, x):
        return self._h.eval_g(x)

    def jacobianstructure(self):
        if self._jac_struct is None:
            self._jac_struct = self._h.jac_structure()
        return self._jac_struct

    def jacobian(self, x):
        return self._h.eval_jac_g(x)

    def hessianstructure(self):
        if self._hess_struct is None:
            self._hess_struct = self._h.hess_structure()
        return self._hess_struct

    def hessian(self, x, duals, obj_factor):
        return self._h.eval_h(x, duals, obj_factor)

    def intermediate(self, alg_mod, iter_count, obj_value, inf_pr, inf_du, mu, d_norm,
                     regularization_size, alpha_du, alpha_pr, ls_trials):
        self.iterations = iter_count


class HIPNLP:
    """NLP solver interface for the on-device interior-point method.  Same role, method
    names and record layout as the reference's `IPOPT(NLPsolver)` (ipopt_nlpif.py:25-184)."""

    # IPOPT ApplicationReturnStatus -> status string (reference ipopt_nlpif.py:31-61)
    STATUS_MAP = {
        0: s.OPTIMAL, 1: s.OPTIMAL_INACCURATE, 6: s.OPTIMAL,
        2: s.INFEASIBLE, 4: s.UNBOUNDED,
        3: s.SOLVER_ERROR, -2: s.SOLVER_ERROR, -3: s.SOLVER_ERROR, -13: s.SOLVER_ERROR,
        -100: s.SOLVER_ERROR, -101: s.SOLVER_ERROR, -199: s.SOLVER_ERROR,
        5: s.USER_LIMIT, -1: s.USER_LIMIT, -4: s.USER_LIMIT, -5: s.USER_LIMIT,
        -102: s.USER_LIMIT,
        -10: s.SOLVER_ERROR, -11: s.SOLVER_ERROR, -12: s.SOLVER_ERROR,
    }

    # defaults of the reference (ipopt_nlpif.py:153-160)
    DEFAULT_OPTIONS = {
        "mu_strategy": "adaptive",
        "tol": 1e-7,
        "bound_relax_factor": 0.0,
        "hessian_approximation": "exact",
        "derivative_test": "none",
        "least_square_init_duals": "yes",
    }

    def name(self):
        return s.IPOPT

    def import_solver(self):
        from . import _capi
        _capi.load()

    def accepts(self, problem):
        return problem.is_dnlp()

    def apply(self, problem, user_variables=None, make_handle=True, fused_spec=None):
        """reference nlp_solver.py:47-79: builds the data dict incl. the oracles."""
        from . import _capi
        data, inverse_data = build_nlp_data(problem, user_variables, fused_spec)
        if not make_handle:          # front-end only: the batched multistart needs just the tape arrays
            return data, inverse_data
        arrays = data["tape_arrays"]
        if sum(a.nbytes for a in arrays.values()) >= self.LARGE_TAPE_BYTES:
            # dense constant blocks of hundreds of MB: the arrays go to the library where they are
            # (dnlp_create_arrays) instead of through one more gigabyte-sized copy
            handle = _capi.DeviceProblem(arrays, data["tape"])
        else:
            handle = _capi.DeviceProblem(serialize(arrays), data["tape"])
        oracles = DeviceOracles(handle, len(data["x0"]), len(data["cl"]))
        data["handle"] = handle
        data["oracles"] = oracles
        for k in ("objective", "gradient", "constraints", "jacobian", "jacobianstructure",
                  "hessian", "hessianstructure"):
            data[k] = getattr(oracles, k)
        return data, inverse_data

    def reapply(self, problem, cached):
        """Second and later solves of an unchanged problem: the lowered tape AND the device handle
        (uploaded tape, sparse plan, generated kernels) of the first solve are reused; only the start
        point is rebuilt from the variables' current values.  Returns None when anything that shapes the
        tape differs (then the caller lowers again)."""
        bounds = Bounds(problem)
        old = cached["data"]
        handle = old.get("handle")
        if handle is None or handle.ptr is None:
            return None
        same = (bounds.x0.size == old["x0"].size and np.array_equal(bounds.lb, old["lb"]) and
                np.array_equal(bounds.ub, old["ub"]) and np.array_equal(bounds.cl, old["cl"]) and
                np.array_equal(bounds.cu, old["cu"]) and
                [v.size for v in bounds.main_var] == [v.size for v in old["problem"].variables()])
        if not same:
            return None
        data = dict(old)
        data["problem"] = bounds.new_problem
        data["x0"] = bounds.x0
        data.pop("warm_duals", None)
        handle.reset_options()
        return data, InverseData(bounds.new_problem)

    def solve_via_data(self, data, warm_start: bool, verbose: bool, solver_opts,
                       solver_cache=None):
        """reference ipopt_nlpif.py:104-174.  `warm_start` is accepted and unused there too."""
        options = dict(self.DEFAULT_OPTIONS)
        if solver_opts:
            options.update(solver_opts)
        if "print_level" not in options:
            # the reference forces print_level 3 when not verbose (ipopt_nlpif.py:164-165); here
            # 0 is silent and 5 prints the IPOPT-style iteration table
            options["print_level"] = 5 if verbose else 0
        handle = data["handle"]
        # hessian_approximation='limited-memory' (ipopt_nlpif.py:153-168 passes it to IPOPT): the quasi-Newton
        # interior-point mode of csrc/ipm_core.h (BFGS pairs in compact form, no second derivatives); it lives in
        # the host-driven loop, so the in-kernel loop is not taken for such a solve (_use_device_loop)
        intermediate = options.pop("intermediate_callback", None)
        algorithm = options.pop("algorithm", "interior-point")
        device_loop = options.pop("device_loop", "auto")
        for k, v in options.items():
            handle.set_option(k, v)
        # Oracles.intermediate (nlp_solver.py:423-427): cyipopt calls it once per iteration; so does
        # the host-driven loop here.  `intermediate_callback=fn` adds a user hook that may return
        # False to stop the solve (status 5, User_Requested_Stop -> "user_limit").
        oracles = data["oracles"]
        data["_intermediate"] = intermediate
        if intermediate is not None:
            def _cb(*a):
                oracles.intermediate(*a)
                return intermediate(*a)
            handle.set_intermediate(_cb)
        else:
            handle.set_intermediate(None)
        if algorithm in ("lbfgs", "reduced-lbfgs"):
            if not data.get("reducible"):
                raise ValueError("algorithm='lbfgs' needs an unconstrained smooth problem whose "
                                 "canonical constraints only define auxiliary variables")
            info = handle.solve_reduced(data["x0"])
        elif data.get("warm_duals") is not None and str(options.get("warm_start_init_point", "no")) in ("yes", "True", "1") \
                and not self._use_device_loop(data, options, device_loop):
            handle.set_warm_start(*data["warm_duals"])
            info = handle.solve(data["x0"])
            handle.set_warm_start(None, None, None)
        elif self._use_device_loop(data, options, device_loop):
            # Small problem, nothing to print: the whole interior-point loop runs inside one kernel
            # (the batch path with a batch of one, csrc/batch.h) instead of being driven from the
            # host with a stream synchronisation per scalar.  Same algorithm text, same result.
            from .batch import instance_data
            row = instance_data(data["tape_arrays"])
            o = row.size - sum(data["tape_arrays"][k].size for k in ("x0", "lb", "ub", "cl", "cu"))
            row[o:o + len(data["x0"])] = data["x0"]                     # a warm start replaces x0
            wd = data.get("warm_duals")
            warm = tuple(np.asarray(a, dtype=float)[None, :] for a in wd) \
                if wd is not None and str(options.get("warm_start_init_point", "no")) in ("yes", "True", "1") else None
            raw = handle.solve_batch(row[None, :], want_duals=True, warm=warm)
            info = {"status": int(raw["status"][0]), "x": raw["x"][0], "obj_val": float(raw["obj_val"][0]),
                    "g": handle.eval_g(raw["x"][0]) if handle.m else np.zeros(0), "mult_g": raw["mult_g"][0], "mult_x_L": raw["mult_x_L"][0],
                    "mult_x_U": raw["mult_x_U"][0], "iterations": int(raw["iterations"][0]),
                    "solve_time": raw["kernel_seconds"], "stats": np.zeros(24), "device_loop": True}
        else:
            info = handle.solve(data["x0"])
        data["oracles"].iterations = info["iterations"]
        return info

    LARGE_TAPE_BYTES = 1 << 20           # tapes above this are created from their arrays in place (dnlp_create_arrays), not from a blob:
                                         # one copy less (the canonical Rosenbrock chain at n = 1e5 is a 60 MB tape: 0.05 s of serialising)
    DEVICE_LOOP_MAX_ORDER = 256          # dense KKT: order up to which one wavefront runs the whole solve
    DEVICE_LOOP_MAX_ORDER_SPARSE = 20000  # sparse static-pattern KKT (csrc/sparse_plan.h)
    DEVICE_LOOP_MAX_TRIPLES = 150000      # ... whose update program one workgroup can walk in ~0.5 ms

    def _use_device_loop(self, data, options, mode) -> bool:
        if mode in (False, "no", "host") or data.get("_intermediate") is not None:
            return False                      # a per-iteration callback needs the host-driven loop
        if str(options.get("hessian_approximation", "exact")) == "limited-memory":
            return False                      # the quasi-Newton mode is the host-driven loop's
        tape = data["tape"]
        order = len(data["x0"]) + len(data["cl"])
        fits = not tape.dense_blocks and not tape.dense_consts
        if fits and order > self.DEVICE_LOOP_MAX_ORDER:
            info = data["handle"].kkt_info()
            # one workgroup factors in-kernel: a long update program (dense-ish fronts) belongs to the
            # host-driven loop, whose level kernels use the whole chip (small NMF: 1.9 s in-kernel, 0.5 s host-driven)
            # (a plan with a dense tail — NMF, phase retrieval — lists no triples for the tail: the count says nothing
            #  about what ONE workgroup would have to walk, and the dense tail is the host-driven loop's anyway)
            fits = order <= self.DEVICE_LOOP_MAX_ORDER_SPARSE and info["sparse"] and \
                info.get("update_triples", 0) <= self.DEVICE_LOOP_MAX_TRIPLES and data["handle"].kkt_tail_nodes() == 0
        if mode in (True, "yes", "device"):
            if tape.dense_blocks or tape.dense_consts:
                raise ValueError("device_loop='yes' needs a tape without dense quad_form blocks")
            return True
        return fits and int(options.get("print_level", 0)) == 0

    def solve_batch_via_data(self, data0, rows, solver_opts):
        """`rows[k]` = instance data (dnlp_amd.batch.BATCH_DATA_KEYS) of run k on data0's tape:
        one dnlp_solve_batch launch, one info dict per run (same keys as solve_via_data)."""
        from .batch import _device_handle
        opts = dict(solver_opts or {})
        opts.setdefault("print_level", 0)
        h = _device_handle(data0["tape_arrays"], data0["tape"], None, opts)
        try:
            raw = h.solve_batch(rows, want_duals=True)
        finally:
            h.close()
        return [{"status": int(raw["status"][k]), "x": raw["x"][k], "obj_val": float(raw["obj_val"][k]),
                 "mult_g": raw["mult_g"][k], "mult_x_L": raw["mult_x_L"][k], "mult_x_U": raw["mult_x_U"][k],
                 "iterations": int(raw["iterations"][k]), "solve_time": raw["kernel_seconds"] / len(rows),
                 "stats": np.zeros(24)} for k in range(len(rows))]

    def invert(self, solution, inverse_data):
        """reference ipopt_nlpif.py:75-102 (duals are not surfaced there either)."""
        attr = {s.NUM_ITERS: solution["iterations"]}
        if "solve_time" in solution:
            attr[s.SOLVE_TIME] = solution["solve_time"]
        if "all_objs_from_best_of" in solution:
            attr[s.EXTRA_STATS] = {"all_objs_from_best_of": solution["all_objs_from_best_of"]}
        status = self.STATUS_MAP[solution["status"]]
        if status in s.SOLUTION_PRESENT:
            opt_val = solution["obj_val"] + inverse_data.offset
            primal_vars = {}
            x_opt = solution["x"]
            for vid, offset in inverse_data.var_offsets.items():
                shape = inverse_data.var_shapes[vid]
                size = int(np.prod(shape, dtype=int))
                primal_vars[vid] = np.reshape(x_opt[offset:offset + size], shape, order="F")
            return {"status": status, "opt_val": opt_val, "primal_vars": primal_vars,
                    "dual_vars": {}, "attr": attr}
        opt_val = {s.INFEASIBLE: np.inf, s.UNBOUNDED: -np.inf}.get(status)
        return {"status": status, "opt_val": opt_val, "primal_vars": {}, "dual_vars": {},
                "attr": attr}

    def cite(self, data):
        return ("Interior-point filter line-search algorithm of Waechter & Biegler, Math. "
                "Prog. 106(1), 2006 (IPOPT), re-implemented on-device for MI355X.")
