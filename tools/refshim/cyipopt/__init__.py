"""Stand-in for the third-party `cyipopt` module, used ONLY by tools/run_reference_tests.py in
the build container: it lets the reference's own test-suite (cvxpy/tests/NLP_tests, run in
place from /root/reference) drive THIS repository's solver.

`cyipopt.Problem(n, m, problem_obj, lb, ub, cl, cu)` receives the reference's `Oracles` object;
instead of calling its Python callbacks, `solve(x0)` translates the canonical cvxpy problem it
holds into the device tape (dnlp_amd.cvxpy_adapter) and runs the interior-point loop — on the
MI355X when one is visible, otherwise through the CPU oracle build of the same algorithm text.
This is test tooling, not part of the product, and it is not cyipopt: only the four calls the
reference makes (ipopt_nlpif.py:143-170) exist.
"""
import os

import numpy as np

__version__ = "0.0-dnlp-shim"


def _backend():
    if os.environ.get("DNLP_SHIM_BACKEND", "auto") != "oracle":
        try:
            from dnlp_amd import _capi
            if _capi.device_count() > 0:
                return "device"
        except Exception:
            pass
    return "oracle"


class Problem:
    def __init__(self, n, m, problem_obj=None, lb=None, ub=None, cl=None, cu=None):
        self.n, self.m, self.obj = n, m, problem_obj
        self.lb, self.ub, self.cl, self.cu = lb, ub, cl, cu
        self.opts = {}

    def add_option(self, name, value):
        self.opts[name] = value

    addOption = add_option

    def solve(self, x0, lagrange=None, zl=None, zu=None):
        from dnlp_amd.cvxpy_adapter import tape_from_cvxpy
        from dnlp_amd.tape import serialize
        data = {"problem": self.obj.problem, "x0": np.asarray(x0, float), "lb": self.lb, "ub": self.ub,
                "cl": self.cl, "cu": self.cu}
        tape, arrays = tape_from_cvxpy(data)
        blob = serialize(arrays)
        if _backend() == "device":
            from dnlp_amd._capi import DeviceProblem
            h = DeviceProblem(blob, tape)
        else:
            import sys
            _t = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))), "tests")
            if _t not in sys.path:
                sys.path.insert(0, _t)
            from oracle_check import OracleProblem          # tests/oracle_check.py
            h = OracleProblem(blob)
        for k, v in self.opts.items():
            if k in ("print_level",):
                continue
            h.set_option(k, v)
        info = h.solve(data["x0"])
        self.obj.iterations = info["iterations"]
        # IPOPT's last callbacks leave the cvxpy variables at the final iterate; the reference's
        # best_of loop reads self.objective.value right after the solve (problem.py:1262)
        self.obj.set_variable_value(np.asarray(info["x"], float))
        out = {"x": info["x"], "g": info["g"], "obj_val": info["obj_val"], "mult_g": info["mult_g"],
               "mult_x_L": info["mult_x_L"], "mult_x_U": info["mult_x_U"], "status": info["status"],
               "status_msg": b"dnlp_amd interior point"}
        return info["x"], out
