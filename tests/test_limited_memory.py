"""hessian_approximation='limited-memory' (the reference passes the option through to IPOPT's quasi-Newton
interior-point mode, cvxpy/reductions/solvers/nlp_solvers/ipopt_nlpif.py:153-168; its own suite uses it in
cvxpy/tests/NLP_tests/test_entropy_related.py:40).  Here: csrc/ipm_core.h lm_* — BFGS pairs in compact form, the KKT
system of the diagonal Hessian factorised as usual, the low-rank part by Sherman-Morrison-Woodbury; no second
derivative is evaluated.  A quasi-Newton run follows its own iterate path, so the checks are: the optimum of the
exact-Hessian run is reached (convex and benign non-convex members of the zoo), the run says what it did, and on the
device the same text gives the same answer as its host build."""
import numpy as np
import pytest

from problem_zoo import GOLDEN_ZOO

QN_ZOO = ["dense_eq_qp", "hs071", "elementwise_zoo", "nonsmooth_zoo", "nb_localization", "mle", "nb_power_flow",
          "rosenbrock_chain50", "sphere60", "socp", "bilinear_matmul", "portfolio_qp"]


def _lowered(name):
    import dnlp_amd as cp
    from dnlp_amd.dnlp2smooth import Dnlp2Smooth
    from dnlp_amd.nlp_solver import build_nlp_data
    from dnlp_amd.tape import serialize
    prob = GOLDEN_ZOO[name](cp) if isinstance(name, str) else name(cp)
    if isinstance(prob.objective, cp.Maximize):
        prob = cp.Problem(cp.Minimize(-prob.objective.expr), prob.constraints)
    smooth, _ = Dnlp2Smooth().apply(prob)
    data, _ = build_nlp_data(smooth)
    return data, serialize(data["tape_arrays"])


def _solve(handle, x0, mode, **opts):
    from dnlp_amd.nlp_solver import HIPNLP
    for k, v in HIPNLP.DEFAULT_OPTIONS.items():
        handle.set_option(k, v)
    handle.set_option("hessian_approximation", mode)
    for k, v in opts.items():
        handle.set_option(k, v)
    return handle.solve(x0)


@pytest.mark.parametrize("name", QN_ZOO)
def test_quasi_newton_run_reaches_the_exact_run_optimum(name):
    from oracle.oracle_capi import OracleProblem
    data, blob = _lowered(name)
    exact = _solve(OracleProblem(blob), data["x0"], "exact")
    h = OracleProblem(blob)
    qn = _solve(h, data["x0"], "limited-memory")
    assert exact["status"] == 0 and qn["status"] == 0
    assert abs(qn["obj_val"] - exact["obj_val"]) <= 1e-6 * max(1.0, abs(exact["obj_val"]))
    log = h.log()
    assert "limited-memory quasi-Newton:" in log and "no second derivatives evaluated" in log
    pairs = int(log.split("limited-memory quasi-Newton:")[1].split("pairs accepted")[0])
    assert 1 <= pairs <= qn["iterations"]
    assert qn["iterations"] <= 12 * max(exact["iterations"], 10)          # superlinear-ish, not a gradient crawl


def _min_entropy(cp):
    """cvxpy/tests/NLP_tests/test_entropy_related.py:30-42 (non-convex: the minimisers are the simplex vertices)."""
    np.random.seed(0)
    n = 10
    q = cp.Variable((n,), nonneg=True)
    q.value = np.random.rand(n)
    q.value = q.value / np.sum(q.value)
    return cp.Problem(cp.Minimize(cp.sum(cp.entr(q))), [cp.sum(q) == 1])


def test_reference_limited_memory_case_on_the_host_build():
    from oracle.oracle_capi import OracleProblem
    data, blob = _lowered(_min_entropy)
    info = _solve(OracleProblem(blob), data["x0"], "limited-memory")
    assert info["status"] == 0
    assert int(np.sum(info["x"][:10] > 1e-8)) == 1                        # the reference's own assertion (:42)


def test_history_length_and_option_validation():
    from oracle.oracle_capi import OracleProblem
    data, blob = _lowered("rosenbrock_chain50")
    runs = {}
    for hist in (2, 6, 12):
        h = OracleProblem(blob)
        runs[hist] = _solve(h, data["x0"], "limited-memory", limited_memory_max_history=hist)
        assert runs[hist]["status"] == 0 and abs(runs[hist]["obj_val"]) <= 1e-8
    assert len({r["iterations"] for r in runs.values()}) > 1              # the history length is really used
    h = OracleProblem(blob)
    with pytest.raises(Exception):
        h.set_option("hessian_approximation", "sr1-please")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["dense_eq_qp", "hs071", "nb_localization", "mle", "nb_power_flow", "sphere60"])
def test_device_quasi_newton_run_against_the_host_build(gpu_required, name):
    """Same algorithm text on the MI355X (host-driven loop: dense Bunch-Kaufman or static-pattern sparse KKT of the
    diagonal Hessian, 2k extra solves per factorisation) and on the host: same optimum, and the iteration counts
    stay together (the inner products are summed in a different order, so a run may differ by a few iterations)."""
    from dnlp_amd import _capi
    from oracle.oracle_capi import OracleProblem
    data, blob = _lowered(name)
    host = _solve(OracleProblem(blob), data["x0"], "limited-memory")
    dev_h = _capi.DeviceProblem(blob, data["tape"], device=0)
    dev = _solve(dev_h, data["x0"], "limited-memory")
    assert host["status"] == 0 and dev["status"] == 0
    assert abs(dev["obj_val"] - host["obj_val"]) <= 1e-6 * max(1.0, abs(host["obj_val"]))
    assert abs(dev["iterations"] - host["iterations"]) <= max(5, host["iterations"] // 3)
    assert "no second derivatives evaluated" in dev_h.log()


@pytest.mark.gpu
def test_device_front_end_takes_the_quasi_newton_mode(gpu_required):
    """Problem.solve(nlp=True, hessian_approximation='limited-memory'): the reference's test case; the in-kernel loop
    is not taken for such a solve (it has no quasi-Newton mode), and best_of runs its starts one by one."""
    import dnlp_amd as cp
    prob = _min_entropy(cp)
    prob.solve(nlp=True, hessian_approximation="limited-memory")
    q = prob.variables()[0].value
    assert prob.status == cp.OPTIMAL and int(np.sum(q > 1e-8)) == 1
    # the same call one level down: which loop ran, and what the handle logged
    prob2 = _min_entropy(cp)
    chain = prob2._build_chain(None)
    data, _ = chain.apply(prob2)
    info = chain.solver.solve_via_data(data, True, False, {"hessian_approximation": "limited-memory"})
    assert info["status"] == 0 and info.get("device_loop") is not True
    assert "no second derivatives evaluated" in data["handle"].log()
    exact = chain.solver.solve_via_data(data, True, False, {})
    assert exact.get("device_loop") is True            # (the default solve of this small problem is the in-kernel loop)


@pytest.mark.gpu
def test_batch_launch_refuses_the_quasi_newton_mode(gpu_required):
    """The reference hands hessian_approximation to IPOPT for every solve (ipopt_nlpif.py:153-168).  The in-kernel
    solver of a batch launch has no quasi-Newton mode: it answers with IPOPT's Invalid_Option (-12) instead of
    silently running the exact Hessian; the same batch with the exact Hessian still solves."""
    import batch_problems as bp
    from dnlp_amd.batch import ParametricBatch
    tprob, params, sample, _ = bp.template_localization()
    pb = ParametricBatch(tprob, params)
    thetas = np.stack([sample(i) for i in range(8)])
    with pytest.raises(RuntimeError, match=r"code -12.*not available inside a batch launch"):
        pb.solve(thetas, hessian_approximation="limited-memory")
    res = pb.solve(thetas)
    assert np.all(res.status == 0)
    pb.close()
