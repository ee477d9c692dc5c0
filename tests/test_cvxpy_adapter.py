"""Adapter tests — run only where the reference tree is importable (the build container);
skipped on the GPU box.  The reference's OWN cvxpy objects are canonicalised by the reference's
own reductions, translated by dnlp_amd.cvxpy_adapter, and (a) the tape oracles are compared
in-process with the reference's `Oracles`, (b) `prob.solve(method=...)` is driven end to end
through CVXPY's register_solve hook with the CPU oracle injected as the solver."""
import os
import sys
import warnings

import numpy as np
import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "cvxpy")),
                                reason="reference tree not present")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def cpref():
    from ref_import import import_reference
    return import_reference()


def _oracle_solve(blob, tape, x0, options):
    from oracle.oracle_capi import OracleProblem
    h = OracleProblem(blob)
    for k, v in options.items():
        h.set_option(k, v)
    return h.solve(x0)


@pytest.mark.parametrize("name", ["hs071", "socp", "localization", "elementwise_zoo", "bilinear_matmul",
                                  "nonsmooth_zoo", "sphere60", "broadcast_div"])
def test_translated_tape_matches_reference_oracles(cpref, name):
    from ref_import import ref_chain_apply
    from dnlp_amd.cvxpy_adapter import tape_from_cvxpy
    from oracle.tape_eval import TapeEvaluator
    from problem_zoo import ZOO
    from golden_util import coo_dense
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        prob = ZOO[name](cpref)
        data, inv, chain = ref_chain_apply(cpref, prob)
        tape, arrays = tape_from_cvxpy(data)
    ev = TapeEvaluator(arrays)
    o = data["oracles"]
    N, m = len(data["x0"]), len(data["cl"])
    jr, jc = o.jacobianstructure()
    hr, hc = o.hessianstructure()
    rng = np.random.default_rng(0)
    x = np.asarray(data["x0"], float)
    lam = rng.standard_normal(m)
    np.testing.assert_allclose(ev.objective(x), o.objective(x), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(ev.gradient(x), np.array(o.gradient(x)), rtol=1e-12, atol=1e-12)
    if m:
        np.testing.assert_allclose(ev.constraints(x), o.constraints(x), rtol=1e-12, atol=1e-10)
        J = coo_dense(*ev.jacobianstructure(), ev.jacobian(x), (m, N))
        Jr = coo_dense(jr, jc, np.asarray(o.jacobian(x)).ravel(), (m, N))
        np.testing.assert_allclose(J, Jr, rtol=1e-12, atol=1e-12)
    H = coo_dense(*ev.hessianstructure(), ev.hessian(x, lam, 0.7), (N, N))
    Hr = coo_dense(hr, hc, np.asarray(o.hessian(x, lam, 0.7)).ravel(), (N, N))
    np.testing.assert_allclose(H, Hr, rtol=1e-12, atol=1e-12)


def test_register_solve_hook_end_to_end(cpref):
    """cp.Problem.solve(method=...) on the reference's own objects (problem.py:622-648)."""
    from dnlp_amd import cvxpy_adapter
    from problem_zoo import hs071, readme_toy
    name = cvxpy_adapter.register(cpref, "dnlp_hip_test", solve_fn=_oracle_solve)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        p = hs071(cpref)
        p.solve(method=name)
        assert p.status == cpref.OPTIMAL
        assert np.allclose(p.variables()[0].value, [0.75450865, 4.63936861, 3.78856881, 1.88513184])
        p = readme_toy(cpref)
        val = p.solve(method=name, tol=1e-9)
        assert abs(val - 11.95081085398) <= 1e-6 * 11.95
        assert p.solver_stats.num_iters > 0
