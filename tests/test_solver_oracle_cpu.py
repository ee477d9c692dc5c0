"""CPU tests: the restated interior-point algorithm (host instantiation in oracle/) pinned
against the known optima the reference's own tests hold, and the
C-ABI library's exported surface.  No GPU compute is attempted here."""
import ctypes as C
import os
import re
import warnings

import numpy as np
import pytest

from golden_util import build_canonical
from problem_zoo import ZOO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _solve(name, **opts):
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        data, inv = build_canonical(name)
    h = OracleProblem(serialize(data["tape_arrays"]))
    for k, v in opts.items():
        h.set_option(k, v)
    info = h.solve(data["x0"])
    vals = {}
    for v in data["problem"].variables():
        off = inv.var_offsets[v.id]
        vals[v.name()] = info["x"][off:off + v.size].reshape(v.shape, order="F")
    return info, vals, data


KNOWN_OBJ = {
    # name: (objective of the canonical MINIMISATION problem, rtol)
    "readme_toy": (-11.95081085398, 1e-6),                 # -lambda_max(A), README.md:50-52
    "socp": (-13.548638814247532, 1e-6),                   # test_nlp_solvers.py:151
    "qcp": (-0.32699284, 1e-6),                            # test_nlp_solvers.py:111
    "geo_mean": (-1.0 / 3.0, 1e-6),                        # test_nlp_solvers.py:269 (x = 1/3)
    "rosenbrock2": (0.0, 1e-9),
    "rosenbrock_chain50": (0.0, 1e-9),
    "localization": (0.0, 1e-9),
}


@pytest.mark.parametrize("strategy", ["adaptive", "monotone"])
@pytest.mark.parametrize("name", sorted(ZOO))
def test_oracle_ipm_converges(name, strategy):
    if name == "mle" and strategy == "monotone":
        pytest.skip("n=2011 dense host factorisation: one strategy is enough on CPU")
    info, vals, data = _solve(name, mu_strategy=strategy)
    assert info["status"] == 0
    if name in KNOWN_OBJ:
        ref, tol = KNOWN_OBJ[name]
        assert abs(info["obj_val"] - ref) <= tol * max(1.0, abs(ref))


def test_oracle_ipm_known_points():
    info, vals, _ = _solve("hs071")
    x = [v for k, v in vals.items() if v.shape == (4,)][0]
    assert np.allclose(x, [0.75450865, 4.63936861, 3.78856881, 1.88513184])      # test_nlp_solvers.py:37
    info, vals, _ = _solve("portfolio_qp")
    x = [v for k, v in vals.items() if v.shape == (3,)][0]
    assert np.allclose(x, [497.045504, 0.0, 502.954496], atol=1e-4)               # :86
    info, vals, _ = _solve("mle")
    assert np.allclose(vals["sigma"], 0.77079388)                                 # :59-60
    assert np.allclose(vals["mu"], 0.59412321)
    info, vals, _ = _solve("localization")
    assert np.allclose(vals["x"], [2.0, -1.5])                                    # :189
    info, vals, _ = _solve("circle_packing")
    c = vals["c"]
    ref = np.array([[1.73655994, -1.98685738, 2.57208783], [1.99273311, -1.67415425, -2.57208783]])
    assert np.allclose(c, ref, atol=1e-5)                                         # :211-213


def test_dense_eq_qp_closed_form():
    """BASELINE C3 shape: one Newton step solves an equality-constrained convex QP exactly."""
    info, vals, data = _solve("dense_eq_qp")
    rng = np.random.default_rng(0)
    n, m = 40, 6
    Gm = rng.standard_normal((n, n))
    Q = Gm.T @ Gm / n + np.eye(n)
    c = rng.standard_normal(n)
    A = rng.standard_normal((m, n))
    b = A @ rng.standard_normal(n)
    K = np.block([[Q, A.T], [A, np.zeros((m, m))]])
    sol = np.linalg.solve(K, np.concatenate([-c, b]))
    assert info["iterations"] <= 2
    np.testing.assert_allclose(info["x"], sol[:n], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(info["mult_g"], sol[n:], rtol=1e-7, atol=1e-9)


def test_invalid_option_and_status_map():
    from dnlp_amd.nlp_solver import HIPNLP
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    data, _ = build_canonical("hs071")
    h = OracleProblem(serialize(data["tape_arrays"]))
    with pytest.raises(ValueError):
        h.set_option("no_such_option", 1)
    h.set_option("max_iter", 2)
    info = h.solve(data["x0"])
    assert info["status"] == -1                       # Maximum_Iterations_Exceeded
    assert HIPNLP.STATUS_MAP[-1] == "user_limit"      # ipopt_nlpif.py:55
    assert HIPNLP.STATUS_MAP[0] == "optimal" and HIPNLP.STATUS_MAP[2] == "infeasible"


def test_capi_library_exports_every_declared_symbol():
    """libdnlp_hip.so must load without a GPU and export every symbol include/dnlp_hip.h
    declares."""
    lib_path = os.path.join(ROOT, "dnlp_amd", "libdnlp_hip.so")
    if not os.path.exists(lib_path):
        import __graft_entry__ as g
        g.build()
    header = open(os.path.join(ROOT, "include", "dnlp_hip.h")).read()
    names = sorted(set(re.findall(r"\b(dnlp_[a-z_0-9]+)\s*\(", header)))
    assert len(names) >= 25
    lib = C.CDLL(lib_path)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_product_path_fails_loudly_without_device():
    """No CPU fallback: without an MI355X the front-end raises instead of solving."""
    import dnlp_amd as cp
    from dnlp_amd import _capi
    if _capi.device_count() > 0:
        pytest.skip("a GPU is present")
    from problem_zoo import readme_toy
    with pytest.raises(cp.DeviceUnavailableError):
        readme_toy(cp).solve(nlp=True)


def test_dnlp_rules():
    """DNLP rule engine (reference tests/NLP_tests/test_dnlp.py)."""
    import dnlp_amd as cp
    x = cp.Variable(3)
    y = cp.Variable(3)
    assert cp.log(x).is_smooth() and cp.exp(x).is_esr() and cp.exp(x).is_hsr()
    assert cp.abs(x).is_esr() and not cp.abs(x).is_hsr()
    assert cp.minimum(x, y).is_hsr() and not cp.minimum(x, y).is_esr()
    assert cp.Problem(cp.Minimize(cp.sum(cp.abs(x)) + cp.sum(cp.exp(cp.sin(y))))).is_dnlp()
    assert not cp.Problem(cp.Minimize(cp.sum(cp.minimum(x, y)))).is_dnlp()
    assert cp.Problem(cp.Maximize(cp.sum(cp.minimum(x, y)))).is_dnlp()
    assert not cp.Problem(cp.Maximize(cp.max(cp.maximum(x, y)))).is_dnlp()
    assert not cp.Problem(cp.Minimize(cp.sum(x)), [cp.abs(x) == 1]).is_dnlp()      # equality needs smooth
    with pytest.raises(cp.DNLPError):
        cp.Problem(cp.Maximize(cp.max(x))).solve(nlp=True)


def _oracle_solve_problem(prob, **opts):
    import dnlp_amd as cp
    from dnlp_amd.dnlp2smooth import Dnlp2Smooth
    from dnlp_amd.nlp_solver import build_nlp_data
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    flip = isinstance(prob.objective, cp.Maximize)
    p = cp.Problem(cp.Minimize(-prob.objective.expr), prob.constraints) if flip else prob
    smooth, _ = Dnlp2Smooth().apply(p)
    data, inv = build_nlp_data(smooth)
    h = OracleProblem(serialize(data["tape_arrays"]))
    for k, v in opts.items():
        h.set_option(k, v)
    info = h.solve(data["x0"])
    vals = {}
    for v in data["problem"].variables():
        off = inv.var_offsets[v.id]
        vals[v.name()] = info["x"][off:off + v.size].reshape(v.shape, order="F")
    return info, vals


def test_parameter_and_fixed_variable():
    """Parameters are constants at lowering time; a variable with lb == ub is pinned."""
    import dnlp_amd as cp
    a = cp.Parameter(3, value=np.array([1.0, 2.0, 3.0]))
    x = cp.Variable(3, name="x")
    y = cp.Variable(name="y", bounds=[2.0, 2.0])          # fixed
    prob = cp.Problem(cp.Minimize(cp.sum(cp.square(x - a)) + cp.square(y - 5) + cp.sum(cp.exp(x)) * 0.0),
                      [cp.sum(x) == y])
    info, vals = _oracle_solve_problem(prob)
    assert info["status"] == 0
    assert abs(vals["y"] - 2.0) < 1e-12
    np.testing.assert_allclose(vals["x"], np.array([1.0, 2.0, 3.0]) - 4.0 / 3.0, atol=1e-6)
    a.value = np.array([0.0, 0.0, 6.0])
    info, vals = _oracle_solve_problem(prob)
    np.testing.assert_allclose(vals["x"], np.array([0.0, 0.0, 6.0]) - 4.0 / 3.0, atol=1e-6)


def test_infeasible_problem_reports_infeasible_or_restoration_failure():
    import dnlp_amd as cp
    x = cp.Variable(2, bounds=[0, 1])
    prob = cp.Problem(cp.Minimize(cp.sum(cp.square(x))), [cp.sum(x) == 5])
    info, _ = _oracle_solve_problem(prob)
    assert info["status"] in (2, -2)       # Infeasible_Problem_Detected / Restoration_Failed


@pytest.mark.parametrize("n", [2, 50, 3000])
def test_reduced_lbfgs_rosenbrock_chain(n):
    """BASELINE config C2 (scaled): unconstrained Rosenbrock chain through the reduced-space
    path — tape f / grad f evaluations and a line search only; optimum x* = 1, f* = 0."""
    import dnlp_amd as cp
    from dnlp_amd.dnlp2smooth import Dnlp2Smooth
    from dnlp_amd.nlp_solver import build_nlp_data
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    from problem_zoo import rosenbrock_chain
    p = rosenbrock_chain(cp, n)
    smooth, _ = Dnlp2Smooth().apply(p)
    data, inv = build_nlp_data(smooth, p.variables())
    assert data["reducible"]
    h = OracleProblem(serialize(data["tape_arrays"]))
    info = h.solve_reduced(data["x0"])
    assert info["status"] == 0
    xv = [v for v in data["problem"].variables() if v.size == n][0]
    off = inv.var_offsets[xv.id]
    assert np.max(np.abs(info["x"][off:off + n] - 1.0)) <= 1e-6
    assert abs(info["obj_val"]) <= 1e-10
    if n <= 50:
        # the reduced-space optimum must agree with the interior-point solution of the canonical form
        ip = h.solve(data["x0"])
        assert ip["status"] == 0
        np.testing.assert_allclose(ip["x"][off:off + n], info["x"][off:off + n], atol=1e-6)


def test_reduction_structure_rejects_constrained_problems():
    import dnlp_amd as cp
    from dnlp_amd.dnlp2smooth import Dnlp2Smooth
    from dnlp_amd.nlp_solver import build_nlp_data
    from problem_zoo import hs071, socp
    for f in (hs071, socp):
        p = f(cp)
        smooth, _ = Dnlp2Smooth().apply(p)
        data, _ = build_nlp_data(smooth, p.variables())
        assert data["reducible"] is False and "def_var" not in data["tape_arrays"]


def test_intermediate_callback_and_user_requested_stop():
    """Oracles.intermediate (nlp_solver.py:423-427): called at iteration 0 and after every
    iteration; a False return ends the solve with status 5 (ipopt_nlpif.py:31-61)."""
    from dnlp_amd.nlp_solver import HIPNLP
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    data, _ = build_canonical("hs071")
    h = OracleProblem(serialize(data["tape_arrays"]))
    seen = []
    h.set_intermediate(lambda *a: seen.append(a) or True)
    info = h.solve(data["x0"])
    assert info["status"] == 0
    assert [a[1] for a in seen] == list(range(info["iterations"] + 1))       # iter_count 0..K
    assert len(seen[0]) == 11                                                 # cyipopt's eleven values
    assert abs(seen[-1][2] - info["obj_val"]) <= 1e-12 * abs(info["obj_val"])
    # stop after the third iteration
    h.set_intermediate(lambda alg, it, *rest: it < 3)
    info = h.solve(data["x0"])
    assert info["status"] == 5 and info["iterations"] == 3
    assert HIPNLP.STATUS_MAP[5] == "user_limit"
    h.set_intermediate(None)
    assert h.solve(data["x0"])["status"] == 0


def test_ipm_step_counts_only_iterations_that_were_carried_out():
    """bench.py's step count: the step() call that merely detects convergence adds nothing, and the
    cumulative counters survive ipm_begin (stats[16..18])."""
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    data, _ = build_canonical("hs071")
    h = OracleProblem(serialize(data["tape_arrays"]))
    total = h.solve(data["x0"])["iterations"]
    h.ipm_begin(data["x0"])
    rc, k = h.ipm_step(3)
    assert (rc, k) == (99, 3)
    rc, k2 = h.ipm_step(1000)
    assert rc == 0 and k + k2 == total
    rc, k3 = h.ipm_step(5)                 # already converged: nothing is carried out, nothing counted
    assert rc == 0 and k3 == 0
    st = h.stats()
    assert st[16] == total and st[18] == 1
    h.ipm_begin(data["x0"])
    h.ipm_step(2)
    st2 = h.stats()
    assert st2[16] == total + 2 and st2[18] == 2 and st2[17] > st[17]
    assert st2[0] == 2                     # per-solve statistics were reset by begin


def test_convergence_is_tested_before_the_iteration_limit():
    """A point that converges exactly at max_iter is a success (IPOPT's order of tests)."""
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    data, _ = build_canonical("hs071")
    h = OracleProblem(serialize(data["tape_arrays"]))
    n = h.solve(data["x0"])["iterations"]
    h.set_option("max_iter", n)
    assert h.solve(data["x0"])["status"] == 0
    h.set_option("max_iter", n - 1)
    assert h.solve(data["x0"])["status"] == -1


def test_best_of_never_ranks_a_failed_run(monkeypatch):
    """a SOLVER_ERROR run writes no variable values; it must enter all_objs as +inf and
    never win, whatever objective the shared Variable state happens to show."""
    import dnlp_amd as cp
    from dnlp_amd.nlp_solver import HIPNLP
    from dnlp_amd.problem import NLPChain
    x = cp.Variable(2, bounds=[-1, 1])
    prob = cp.Problem(cp.Minimize(cp.sum_squares(x - 0.25)))
    chain = NLPChain(False, HIPNLP())
    objs = [0.5, None, 0.125]              # run 1 fails

    def fake_apply(self, problem, make_handle=True):
        from dnlp_amd.dnlp2smooth import Dnlp2Smooth
        from dnlp_amd.nlp_solver import build_nlp_data
        smooth, _ = Dnlp2Smooth().apply(problem)
        data, inv = build_nlp_data(smooth)
        data["_x_offset"] = inv.var_offsets[x.id]
        return data, inv
    calls = {"k": 0}

    def fake_solve(self, data, warm_start, verbose, solver_opts=None, solver_cache=None):
        k = calls["k"]
        calls["k"] += 1
        N = len(data["x0"])
        if objs[k] is None:
            return {"status": -3, "x": np.full(N, 9.0), "obj_val": -1e9, "iterations": 1}
        xv = np.zeros(N)
        xv[data["_x_offset"]:data["_x_offset"] + 2] = 0.25 + np.sqrt(objs[k] / 2.0)
        return {"status": 0, "x": xv, "obj_val": objs[k], "iterations": 1}
    monkeypatch.setattr(NLPChain, "apply", fake_apply)
    monkeypatch.setattr(HIPNLP, "solve_via_data", fake_solve)
    monkeypatch.setattr(cp.Problem, "_build_chain", lambda self, solver: chain)
    prob.solve(nlp=True, best_of=3, batch=False)
    allobjs = prob.solver_stats.extra_stats["all_objs_from_best_of"]
    assert np.isinf(allobjs[1]) and np.allclose(allobjs[[0, 2]], [0.5, 0.125])
    assert abs(prob.value - 0.125) < 1e-12


def test_second_solve_reuses_the_lowered_tape_and_handle():
    """A Problem solved again (other start, other options) keeps its tape and solver handle: only the
    start point is rebuilt; options of the previous solve do not leak; a changed Parameter lowers again."""
    import dnlp_amd as cp
    from oracle_frontend import oracle_engine
    from problem_zoo import hs071
    with oracle_engine():
        p = hs071(cp)
        x = p.variables()[0]
        p.solve(nlp=True, max_iter=2)
        assert p.status == cp.USER_LIMIT
        h1 = p._nlp_cache["data"]["handle"]
        x.value = np.array([1.0, 5.0, 5.0, 1.0])
        p.solve(nlp=True)                                   # max_iter = 2 must not survive
        assert p.status == cp.OPTIMAL
        assert p._nlp_cache["data"]["handle"] is h1
        assert np.allclose(x.value, [0.75450865, 4.63936861, 3.78856881, 1.88513184])
        # a parameter value is part of the signature
        a = cp.Parameter(value=1.0)
        y = cp.Variable(2)
        q = cp.Problem(cp.Minimize(cp.sum_squares(y - a)), [cp.sum(y) == 1])
        q.solve(nlp=True)
        h2 = q._nlp_cache["data"]["handle"]
        assert np.allclose(y.value, [0.5, 0.5], atol=1e-6)
        a.value = 3.0
        q.solve(nlp=True)
        assert q._nlp_cache["data"]["handle"] is not h2
        assert np.allclose(y.value, [0.5, 0.5], atol=1e-6) and abs(q.value - 2 * 2.5 ** 2) < 1e-6


def test_pivoted_order_limit_is_a_property_of_the_execution_space():
    """`kkt_pivot_max_n` above 4096 is clamped on the device space only (its one-workgroup Bunch-Kaufman solve
    keeps the vector in LDS).  The host space has no such limit: bench.py's cpu_baseline relies on the pivoted
    (LAPACK DSYTRF) factorisation at n = 1e4 -- clamped there, it silently fell back to the scalar unpivoted
    LDL^T and the default bench ran into its time limit."""
    import dnlp_amd as cp
    from dnlp_amd.dnlp2smooth import Dnlp2Smooth
    from dnlp_amd.nlp_solver import build_nlp_data
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    n = 4200
    rng = np.random.default_rng(0)
    d = rng.uniform(1.0, 2.0, n)
    x = cp.Variable(n)
    x.value = np.ones(n) / np.sqrt(n)
    Q = np.diag(d)
    Q[0, 1] = Q[1, 0] = 0.25                           # a dense constant block, not a diagonal the lowering could split
    prob = cp.Problem(cp.Minimize(-cp.quad_form(x, Q)), [cp.sum_squares(x) == 1])
    smooth, _ = Dnlp2Smooth().apply(prob)
    data, _ = build_nlp_data(smooth)
    orc = OracleProblem(serialize(data["tape_arrays"]))
    try:
        assert not orc.kkt_info()["dense_pivoted"]     # default kkt_pivot_max_n = 2048 < order 4201
        orc.set_option("kkt_pivot_max_n", 10 ** 9)
        info = orc.kkt_info()
        assert not info["sparse"] and info["dense_pivoted"]
    finally:
        orc.close()


def _arrays_problem():
    import dnlp_amd as cp
    rng = np.random.default_rng(3)
    n, m = 40, 7
    G = rng.standard_normal((n, n))
    Q = G.T @ G / n + np.eye(n)
    A = rng.standard_normal((m, n))
    x = cp.Variable(n)
    prob = cp.Problem(cp.Minimize(0.5 * cp.quad_form(x, Q) + rng.standard_normal(n) @ x + cp.sum(cp.exp(x))),
                      [A @ x == A @ rng.standard_normal(n)])
    data, _ = prob._build_chain(None).apply(prob, make_handle=False)
    return data, rng.standard_normal(n), rng.standard_normal(m)


def _same_oracles(h1, h2, z, lam):
    assert (h1.n, h1.m, h1.nnz_jac, h1.nnz_hess) == (h2.n, h2.m, h2.nnz_jac, h2.nnz_hess)
    assert h1.eval_f(z) == h2.eval_f(z)
    for a, b in ((h1.eval_grad_f(z), h2.eval_grad_f(z)), (h1.eval_g(z), h2.eval_g(z)),
                 (h1.eval_jac_g(z), h2.eval_jac_g(z)), (h1.eval_h(z, lam, 1.0), h2.eval_h(z, lam, 1.0))):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    for a, b in zip(h1.jac_structure() + h1.hess_structure(), h2.jac_structure() + h2.hess_structure()):
        assert np.array_equal(a, b)


def test_create_from_arrays_is_create_from_the_blob():
    """`<prefix>create_arrays` (the tape's arrays handed over in place, for gigabyte-sized tapes) builds the same
    problem as `<prefix>create` on the serialised blob; bad descriptors are refused with a message."""
    from dnlp_amd import _capi
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem, api
    data, z, lam = _arrays_problem()
    arrays = data["tape_arrays"]
    _same_oracles(OracleProblem(serialize(arrays)), OracleProblem(arrays), z, lam)
    a = api()
    bad = (_capi.TapeArrayDesc * 1)()
    bad[0].name, bad[0].dtype, bad[0].count, bad[0].data = b"dims", 7, 1, 8
    assert not a.create_arrays(bad, 1, 0) and "dtype" in a.error()
    buf = np.zeros(4)
    bad[0].dtype, bad[0].data = 0, buf.ctypes.data + 4
    assert not a.create_arrays(bad, 1, 0) and "aligned" in a.error()
    assert not a.create_arrays(None, 0, 0) and "empty" in a.error()
