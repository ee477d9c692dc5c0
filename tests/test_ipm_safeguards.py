"""Interior-point safeguards beyond the published algorithm (csrc/ipm_core.h), on the CPU oracle build
of the same source: the degenerate-Jacobian heuristic of the inertia loop, the bracketed
quality-function search and the retry ladder."""
import numpy as np
import pytest

import dnlp_amd as cp
from dnlp_amd.batch import ParametricBatch, arrays_with_data
from dnlp_amd.dnlp2smooth import Dnlp2Smooth
from dnlp_amd.nlp_solver import HIPNLP, build_nlp_data
from dnlp_amd.tape import serialize

import batch_problems as bp


def _oracle(arrays, **opts):
    from oracle.oracle_capi import OracleProblem
    orc = OracleProblem(serialize(arrays))
    for k, v in dict(HIPNLP.DEFAULT_OPTIONS, **opts).items():
        orc.set_option(k, v)
    return orc


@pytest.mark.parametrize("inst", [0, 2])
def test_power_flow_template_needs_few_iterations_and_factorisations(inst):
    """AC power flow with perturbed loads: rank-deficient Jacobian at the iterates (static pivots report
    a wrong inertia that only delta_c repairs) and a non-unimodal quality function.  Without the two
    safeguards the same instances took 71 / 22 lucky or ~300 regular iterations and up to 28
    factorisations per iteration."""
    prob, params, sample, _ = bp.template_power_flow()
    pb = ParametricBatch(prob, params)
    mat = pb.data(np.stack([sample(inst)]))
    arr = arrays_with_data(pb.arrays0, mat[0])
    for ls in ("sparse", "dense"):
        if ls == "dense" and inst != 2:
            continue                       # the dense host factorisation of order 1723 takes seconds per iteration
        orc = _oracle(arr, linear_solver=ls, print_level=6)
        info = orc.solve(arr["x0"])
        assert info["status"] == 0
        assert info["iterations"] <= 30, info["iterations"]
        attempts = orc.log().count("factor attempt")
        assert attempts <= 4 * info["iterations"], (attempts, info["iterations"])


def test_retry_ladder_rescues_a_run_that_diverges():
    """Risk parity written with a bilinear division (reference test_risk_parity.py:25-42): off the
    feasible set the canonical objective is unbounded below; the adaptive and the monotone run from
    the default start both end badly, the last rung (monotone, mu_0 = 1) converges."""
    Sigma = 1e-5 * np.array([
        [41.16, 22.03, 18.64, -4.74, 6.27, 10.1, 14.52, 3.18], [22.03, 58.57, 32.92, -5.04, 4.02, 3.7, 26.76, 2.17],
        [18.64, 32.92, 81.02, 0.53, 6.05, 2.02, 25.52, 1.56], [-4.74, -5.04, 0.53, 20.6, 2.52, 0.57, 0.2, 3.6],
        [6.27, 4.02, 6.05, 2.52, 10.13, 2.59, 4.32, 3.13], [10.1, 3.7, 2.02, 0.57, 2.59, 22.89, 3.97, 3.26],
        [14.52, 26.76, 25.52, 0.2, 4.32, 3.97, 29.91, 3.25], [3.18, 2.17, 1.56, 3.6, 3.13, 3.26, 3.25, 13.63]])
    n = 8
    b = np.ones(n) / n
    w = cp.Variable(n, nonneg=True)
    t = cp.Variable(n)
    term1 = cp.sum(cp.multiply(cp.square(w), cp.square(t))) / cp.quad_form(w, Sigma)
    term2 = float(b @ b) * cp.quad_form(w, Sigma)
    term3 = -2 * cp.sum(cp.multiply(b, cp.multiply(w, t)))
    prob = cp.Problem(cp.Minimize(term1 + term2 + term3), [cp.sum(w) == 1, t == Sigma @ w])
    smooth, _ = Dnlp2Smooth().apply(prob)
    data, _ = build_nlp_data(smooth)
    orc = _oracle(data["tape_arrays"], print_level=5)
    info = orc.solve(data["x0"])
    log = orc.log()
    if "restarting in monotone mode" not in log:
        pytest.skip("this build converges without the ladder")
    assert info["status"] == 0, info["status"]
    off = _oracle(data["tape_arrays"], adaptive_fallback="no")
    assert off.solve(data["x0"])["status"] != 0


def test_lazy_dense_fallback_keeps_the_sparse_factorisation_in_the_first_run():
    """Localization instance 57288 of the C5 template repeatedly meets singular static pivots.  With the
    eager rule it switches to Bunch-Kaufman in its first run (in the batch kernel: iterations of 1.1 ms
    instead of 0.18 ms, the tail of a 65536-instance launch); with lazy_dense_fallback (the batch
    default) it stays on the sparse factorisation and still reaches the optimum."""
    prob, params, sample, _ = bp.template_localization()
    pb = ParametricBatch(prob, params)
    arr = arrays_with_data(pb.arrays0, pb.data(np.stack([sample(57288)]))[0])
    eager = _oracle(arr, print_level=5)
    info_e = eager.solve(arr["x0"])
    lazy = _oracle(arr, print_level=5, lazy_dense_fallback="yes")
    info_l = lazy.solve(arr["x0"])
    assert info_e["status"] == 0 and info_l["status"] == 0
    assert abs(info_l["obj_val"]) <= 1e-8 and abs(info_e["obj_val"]) <= 1e-8
    assert "Bunch-Kaufman from here on" in eager.log()
    first_run = lazy.log().split("restarting in monotone mode")[0]
    assert "Bunch-Kaufman from here on" not in first_run


def test_stall_guard_hands_a_crawling_run_to_the_monotone_rung():
    """One of 8192 circle-packing instances (start placed on the right variables) crawls in
    free-mu mode: alpha_pr = 1e-4 with ||d|| ~ 500 until max_iter (3000 iterations: the tail of the whole
    batch launch).  Forty consecutive accepted steps below 1e-3 of the Newton step now end the run with the
    tiny-step status, which the retry ladder answers with the monotone rung (23 iterations)."""
    import batch_problems as bp
    from dnlp_amd.batch import ParametricBatch, arrays_with_data
    from dnlp_amd.nlp_solver import HIPNLP
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    prob, params, sample, _ = bp.template_circle_packing()
    pb = ParametricBatch(prob, params)
    a = arrays_with_data(pb.arrays0, pb.data(np.stack([sample(6272)]))[0])
    opts = dict(HIPNLP.DEFAULT_OPTIONS)
    opts["lazy_dense_fallback"] = "yes"            # the batch path's setting
    o = OracleProblem(serialize(a))
    for k, v in opts.items():
        o.set_option(k, v)
    r = o.solve(a["x0"])
    assert r["status"] == 0 and r["iterations"] < 150
    assert "stall guard" in o.log()
    # without the ladder but with the guard asked for, its verdict is visible: IPOPT's tiny-step status
    o2 = OracleProblem(serialize(a))
    opts["adaptive_fallback"] = "no"
    opts["stall_guard"] = "yes"
    for k, v in opts.items():
        o2.set_option(k, v)
    r2 = o2.solve(a["x0"])
    assert r2["status"] == 3 and r2["iterations"] < 150
    # default for a run that no rung can take over: no guard — IPOPT has no such rule, the run
    # goes on to the iteration limit as IPOPT's would
    o3 = OracleProblem(serialize(a))
    opts["stall_guard"] = "auto"
    opts["max_iter"] = 300
    for k, v in opts.items():
        o3.set_option(k, v)
    r3 = o3.solve(a["x0"])
    assert r3["status"] == -1 and r3["iterations"] == 300 and "stall guard" not in o3.log()
