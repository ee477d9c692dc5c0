"""Fused native-form objective programs (dnlp_amd/fused.py, csrc/fused_obj.h): BASELINE config C2.
CPU: program builder + numpy interpreter + host instantiation against closed forms, finite
differences and the canonical-tape reduced gradient.  GPU: the LDS-register-file kernel."""
import numpy as np
import pytest

import dnlp_amd as cp
from dnlp_amd.dnlp2smooth import Dnlp2Smooth
from dnlp_amd.fused import build_fused_spec
from dnlp_amd.nlp_solver import build_nlp_data
from dnlp_amd.tape import serialize
from problem_zoo import rosenbrock_chain


def _data(prob):
    smooth, _ = Dnlp2Smooth().apply(prob)
    data, _ = build_nlp_data(smooth, user_variables=prob.variables(), fused_spec=build_fused_spec(prob))
    return data


def _rosenbrock(x):
    f = np.sum((1 - x[:-1]) ** 2) + 100 * np.sum((x[1:] - x[:-1] ** 2) ** 2)
    g = np.zeros_like(x)
    g[:-1] += -2 * (1 - x[:-1]) - 400 * (x[1:] - x[:-1] ** 2) * x[:-1]
    g[1:] += 200 * (x[1:] - x[:-1] ** 2)
    return f, g


def _mixed(n=40):
    rng = np.random.default_rng(5)
    c = rng.standard_normal(n)
    w = rng.uniform(0.5, 1.5, n)
    x = cp.Variable(n)
    y = cp.Variable(n)
    x.value = rng.uniform(0.5, 1.5, n)
    y.value = rng.uniform(0.5, 1.5, n)
    f = (cp.sum(cp.exp(cp.multiply(w, x))) + 0.5 * cp.sum_squares(x - c) + 0.3 * cp.sum(cp.multiply(x[:-1], y[1:]))
         + 2 * cp.sum(cp.logistic(y)) + (1.0 / 3) * cp.sum(cp.sin(x[::2])) + 0.25 * cp.sum_squares(y) + 7.0)
    return cp.Problem(cp.Minimize(f), []), (x, y), (c, w)


def test_builder_accepts_elementwise_sums_and_rejects_the_rest():
    assert build_fused_spec(rosenbrock_chain(cp, 30)) is not None
    prob, _, _ = _mixed()
    fb = build_fused_spec(prob)
    assert fb is not None and abs(fb.c0 - 7.0) < 1e-15
    x = cp.Variable(5)
    assert build_fused_spec(cp.Problem(cp.Minimize(cp.sum(cp.exp(x))), [cp.sum(x) == 1])) is None      # constraints
    A = np.random.default_rng(0).standard_normal((5, 5))
    assert build_fused_spec(cp.Problem(cp.Minimize(cp.quad_form(x, A @ A.T)), [])) is None              # not elementwise
    assert build_fused_spec(cp.Problem(cp.Minimize(cp.sum(cp.exp(A @ x))), [])) is None                 # matrix product
    xb = cp.Variable(5, nonneg=True)
    assert build_fused_spec(cp.Problem(cp.Minimize(cp.sum(cp.exp(xb))), [])) is None                    # bounds -> IPM


def test_program_interpreters_match_closed_form_and_tape():
    from oracle.fused_eval import numpy_eval
    from oracle.oracle_capi import OracleProblem
    n = 300
    data = _data(rosenbrock_chain(cp, n))
    assert data["fused"]
    ta = data["tape_arrays"]
    xv = np.random.default_rng(1).standard_normal(n)
    f0, g0 = _rosenbrock(xv)
    f1, g1 = numpy_eval(ta, xv)
    orc = OracleProblem(serialize(ta))
    f2, g2 = orc.eval_fused(xv)
    np.testing.assert_allclose([f1, f2], f0, rtol=1e-13)
    np.testing.assert_allclose(g1, g0, rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(g2, g1, rtol=1e-14, atol=1e-12)


def test_mixed_objective_against_finite_differences_and_reduced_tape_solve():
    from oracle.oracle_capi import OracleProblem
    prob, (x, y), _ = _mixed()
    data = _data(prob)
    assert data["fused"]
    ta = data["tape_arrays"]
    orc = OracleProblem(serialize(ta))
    free = np.asarray(ta["free_idx"])
    z = np.asarray(data["x0"])[free]
    f, g = orc.eval_fused(z)
    for k in (0, 7, 41, 79):
        e = np.zeros_like(z)
        e[k] = 1e-6
        fp, _ = orc.eval_fused(z + e)
        fm, _ = orc.eval_fused(z - e)
        assert abs((fp - fm) / 2e-6 - g[k]) <= 1e-6 * max(1.0, abs(g[k]))
    res = {}
    for mode in ("yes", "no"):
        o = OracleProblem(serialize(ta))
        o.set_option("fused_objective", mode)
        o.set_option("tol", 1e-6)
        res[mode] = o.solve_reduced(data["x0"])
        assert res[mode]["status"] == 0
    assert abs(res["yes"]["obj_val"] - res["no"]["obj_val"]) <= 1e-9 * abs(res["no"]["obj_val"])
    np.testing.assert_allclose(res["yes"]["x"], res["no"]["x"], rtol=1e-4, atol=1e-5)


def _random_objective(seed, n=24, vector_constants=True):
    """Random sums of elementwise trees over slices of two variables: exercises every rule of the
    host-side reverse sweep (products, quotients, sub / scale / shift chains, constant adjoints)."""
    rng = np.random.default_rng(seed)
    x = cp.Variable(n)
    y = cp.Variable(n)
    x.value = rng.uniform(0.6, 1.4, n)
    y.value = rng.uniform(0.6, 1.4, n)
    m = n - 2

    def leaf():
        v = x if rng.random() < 0.5 else y
        o = int(rng.integers(0, 3))
        return v[o:o + m]

    def tree(depth):
        r = rng.random()
        if depth == 0 or r < 0.15:
            return leaf()
        if r < 0.30:
            return cp.multiply(tree(depth - 1), tree(depth - 1))
        if r < 0.40:
            return tree(depth - 1) / (1.5 + cp.square(tree(depth - 1)))
        if r < 0.50:
            return float(rng.uniform(-2, 2)) * tree(depth - 1) + float(rng.uniform(-1, 1))
        if r < 0.60:
            return float(rng.uniform(0.5, 2)) - tree(depth - 1)
        if r < 0.70:
            return tree(depth - 1) - tree(depth - 1)
        if r < 0.78 and vector_constants:
            return cp.multiply(rng.uniform(0.5, 1.5, m), tree(depth - 1))
        if r < 0.86:
            return cp.square(tree(depth - 1))
        if r < 0.93:
            return cp.exp(0.3 * tree(depth - 1))
        return cp.sin(tree(depth - 1))

    f = 0
    for _ in range(int(rng.integers(1, 4))):
        f = f + float(rng.uniform(-2, 2)) * cp.sum(tree(3))
    return cp.Problem(cp.Minimize(f + 0.1 * cp.sum_squares(x) + 0.1 * cp.sum_squares(y)), [])


@pytest.mark.parametrize("seed", range(16))
def test_slot_program_matches_tree_interpreter_on_random_trees(seed):
    """csrc/fused_obj.h differentiates the tree on the host and reallocates its registers; the numpy
    interpreter (oracle/fused_eval.py) walks the tree itself: same f and gradient."""
    from oracle.fused_eval import numpy_eval
    from oracle.oracle_capi import OracleProblem
    prob = _random_objective(seed)
    if build_fused_spec(prob) is None:
        pytest.skip("tree outside the fused grammar")
    data = _data(prob)
    ta = data["tape_arrays"]
    if not data.get("fused") or "free_idx" not in ta:
        pytest.skip("program beyond the fused capacities")
    free = np.asarray(ta["free_idx"])
    z = np.asarray(data["x0"])[free]
    f1, g1 = numpy_eval(ta, z)
    f2, g2 = OracleProblem(serialize(ta)).eval_fused(z)
    assert abs(f1 - f2) <= 1e-12 * max(1.0, abs(f1))
    np.testing.assert_allclose(g2, g1, rtol=1e-11, atol=1e-11)


@pytest.mark.gpu
def test_device_fused_kernel_matches_interpreters(gpu_required):
    from dnlp_amd import _capi
    from oracle.fused_eval import numpy_eval
    for prob in (rosenbrock_chain(cp, 5000), _mixed(1000)[0]):
        data = _data(prob)
        ta = data["tape_arrays"]
        dev = _capi.DeviceProblem(serialize(ta), data["tape"], device=0)
        z = np.asarray(data["x0"])[np.asarray(ta["free_idx"])] + 0.1
        f, g = dev.eval_fused(z)
        f1, g1 = numpy_eval(ta, z)
        assert abs(f - f1) <= 1e-12 * max(1.0, abs(f1))
        np.testing.assert_allclose(g, g1, rtol=1e-12, atol=1e-10)
        dev.close()


@pytest.mark.gpu
def test_c2_full_size_fused_lbfgs(gpu_required):
    """BASELINE config C2 at n = 1e5 through the front-end with the fused evaluator: x* = 1."""
    n = 100000
    prob = rosenbrock_chain(cp, n)
    prob.solve(nlp=True, algorithm="lbfgs", tol=1e-9)
    assert prob.status == "optimal"
    assert np.max(np.abs(prob.variables()[0].value - 1.0)) <= 1e-6
    assert prob.value <= 1e-10
    # same answer without the fused evaluator (canonical-tape reduced gradient)
    prob2 = rosenbrock_chain(cp, n)
    prob2.solve(nlp=True, algorithm="lbfgs", tol=1e-9, fused_objective="no")
    assert np.max(np.abs(prob2.variables()[0].value - prob.variables()[0].value)) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_device_slot_kernel_on_random_trees_at_scale(gpu_required, seed):
    """The device kernel on random slot programs at a size that takes the four-elements-per-lane
    body, the LDS gradient window and a partial last tile: the programs are index-affine, so the
    ones lowered at n = 24 are re-targeted to 700 003 elements by patching the element counts (the
    variables then alias one flat vector — the same function for both interpreters)."""
    from dnlp_amd import _capi
    from oracle.fused_eval import numpy_eval
    prob = _random_objective(seed, vector_constants=False)
    if build_fused_spec(prob) is None:
        pytest.skip("tree outside the fused grammar")
    data = _data(prob)
    ta = dict(data["tape_arrays"])
    if not data.get("fused") or "free_idx" not in ta:
        pytest.skip("program beyond the fused capacities")
    data["handle"].close() if data.get("handle") is not None else None
    E = 700003
    ta["fz_prog_nelem"] = np.full_like(ta["fz_prog_nelem"], E)
    dims = ta["fz_dims"].copy()
    dims[3] = E + 128
    ta["fz_dims"] = dims
    z = np.random.default_rng(seed).uniform(0.6, 1.4, E + 128)
    dev = _capi.DeviceProblem(serialize(ta), data["tape"], device=0)
    f, g = dev.eval_fused(z)
    dev.close()
    f1, g1 = numpy_eval(ta, z)
    assert abs(f - f1) <= 1e-10 * max(1.0, abs(f1))
    np.testing.assert_allclose(g, g1, rtol=1e-10, atol=1e-10)


def _direct_path_checks(engine_ctx):
    """algorithm='lbfgs' on an unconstrained elementwise-sum objective never lowers the canonical form: the tape
    holds the user's variables only (m = 0, no segments) and the objective value comes from the fused program."""
    with engine_ctx:
        p = rosenbrock_chain(cp, 60)
        p.solve(nlp=True, algorithm="lbfgs", tol=1e-9)
        d = p._nlp_cache["data"]
        assert p._nlp_cache["sig"][0] == "direct" and d["tape"].N == 60 and d["tape"].m == 0
        assert p.status == "optimal" and np.max(np.abs(p.variables()[0].value - 1.0)) <= 1e-6 and 0.0 <= p.value <= 1e-10
        p.variables()[0].value = np.full(60, 0.5)
        p.solve(nlp=True, algorithm="lbfgs", tol=1e-9)          # cached handle, new start
        assert p.status == "optimal" and np.max(np.abs(p.variables()[0].value - 1.0)) <= 1e-6
        # a constant term and a Maximize flip travel with the fused program
        x = cp.Variable(7)
        x.value = np.zeros(7)
        q = cp.Problem(cp.Maximize(-cp.sum(cp.square(x - 2.0)) - 3.0), [])
        q.solve(nlp=True, algorithm="lbfgs")
        assert q.status == "optimal" and abs(q.value + 3.0) <= 1e-9 and np.max(np.abs(x.value - 2.0)) <= 1e-6
        # fused_objective="no" asks for the canonical reduced gradient: the canonical form is lowered
        p2 = rosenbrock_chain(cp, 60)
        p2.solve(nlp=True, algorithm="lbfgs", tol=1e-9, fused_objective="no")
        assert p2._nlp_cache["data"]["tape"].m > 0 and np.max(np.abs(p2.variables()[0].value - 1.0)) <= 1e-6
        # a constrained problem has no fused form: algorithm='lbfgs' is refused as before
        y = cp.Variable(3)
        y.value = np.ones(3)
        with pytest.raises(ValueError):
            cp.Problem(cp.Minimize(cp.sum(cp.square(y))), [cp.sum(y) == 1]).solve(nlp=True, algorithm="lbfgs")


def test_lbfgs_direct_path_on_the_host_engine():
    from oracle_frontend import oracle_engine
    _direct_path_checks(oracle_engine())


@pytest.mark.gpu
def test_lbfgs_direct_path_on_the_device(gpu_required):
    import contextlib
    _direct_path_checks(contextlib.nullcontext())


@pytest.mark.gpu
def test_c2_persistent_single_launch_kernel_against_the_slot_kernels(gpu_required, monkeypatch):
    """BASELINE C2 at its stated size: the whole L-BFGS solve in ONE launch (a slice of x and of the history per
    compute unit in LDS, one grid barrier per trial point) — the analytic optimum x* = 1, and the same iteration /
    evaluation counts (up to summation order) as the four-kernel slot sequence it replaces at this size."""
    from problem_zoo import rosenbrock_chain
    out = {}
    for persist in ("1", "0"):
        monkeypatch.setenv("DNLP_LBFGS_PERSIST", persist)
        p = rosenbrock_chain(cp, 100000)
        chain = p._build_chain(None)
        data, inv = chain.apply(p)
        info = chain.solver.solve_via_data(data, True, False, {"algorithm": "lbfgs"})
        assert info["status"] == 0 and info["device_loop"]
        assert info["device_loop_persistent"] == (persist == "1")
        p.unpack_results(info, chain, inv)
        assert np.max(np.abs(p.variables()[0].value - 1.0)) <= 1e-5
        assert abs(info["obj_val"]) <= 1e-10
        out[persist] = info
        data["handle"].close()
    assert abs(out["1"]["iterations"] - out["0"]["iterations"]) <= 6
    assert abs(out["1"]["evaluations"] - out["0"]["evaluations"]) <= 8
    # the same solve with the slice's history (mode 1) or the whole slice (mode 2) in the workgroup's strip of global
    # memory instead of LDS — what sizes beyond n ~ 2e5 take: same optimum, iteration counts within summation order
    for mode in ("1", "2"):
        monkeypatch.setenv("DNLP_LBFGS_PERSIST", "1")
        monkeypatch.setenv("DNLP_LBFGS_PERSIST_MODE", mode)
        p = rosenbrock_chain(cp, 100000)
        chain = p._build_chain(None)
        data, inv = chain.apply(p)
        info = chain.solver.solve_via_data(data, True, False, {"algorithm": "lbfgs"})
        assert info["status"] == 0 and info["device_loop"] and info["device_loop_persistent"]
        p.unpack_results(info, chain, inv)
        assert np.max(np.abs(p.variables()[0].value - 1.0)) <= 1e-5
        assert abs(info["iterations"] - out["1"]["iterations"]) <= 6
        data["handle"].close()
    monkeypatch.delenv("DNLP_LBFGS_PERSIST_MODE")
    # n = 3e5: the history no longer fits LDS and lives in the strip (one launch); n = 2e6: the slot kernels
    monkeypatch.setenv("DNLP_LBFGS_PERSIST", "1")
    for n, persistent in ((300000, True), (2000000, False)):
        p = rosenbrock_chain(cp, n)
        chain = p._build_chain(None)
        data, inv = chain.apply(p)
        info = chain.solver.solve_via_data(data, True, False, {"algorithm": "lbfgs"})
        assert info["status"] == 0 and info["device_loop"] and info["device_loop_persistent"] == persistent
        p.unpack_results(info, chain, inv)
        assert np.max(np.abs(p.variables()[0].value - 1.0)) <= 1e-5
        data["handle"].close()
