"""SURVEY.md Appendix D as a fixture table: every known answer the reference's own NLP tests hold,
written once for both modelling namespaces (`cp` = dnlp_amd, or the reference's cvxpy in the build
container).  Each row = (builder, check, source file:line under cvxpy/tests/NLP_tests/).

    builder(cp)            -> (problem, handles)      handles: dict of the variables the check reads
    check(problem, handles) asserts the reference-held answer on the solved problem

tests/test_appendix_d.py solves every row through the CPU oracle (CPU suite) and through the product
front-end `Problem.solve(nlp=True)` on the MI355X (`-m gpu`).
"""
import numpy as np
import numpy.linalg as LA

import problem_zoo as zoo


def _row(builder, check, src, **solve_kwargs):
    return {"build": builder, "check": check, "src": src, "kwargs": solve_kwargs}


# ---- problems already in the golden zoo ------------------------------------------------------------
def _hs071(cp):
    p = zoo.hs071(cp)
    return p, {"x": p.variables()[0]}


def _chk_hs071(p, h):
    assert np.allclose(h["x"].value, [0.75450865, 4.63936861, 3.78856881, 1.88513184])


def _mle(cp):
    p = zoo.mle(cp)
    return p, {v.name(): v for v in p.variables()}


def _chk_mle(p, h):
    assert np.allclose(h["sigma"].value, 0.77079388)
    assert np.allclose(h["mu"].value, 0.59412321)


def _portfolio_qp(cp):
    p = zoo.portfolio_qp(cp)
    return p, {"x": p.variables()[0]}


def _chk_portfolio_qp(p, h):
    assert np.allclose(h["x"].value, [497.045504, 0.0, 502.954496], atol=1e-4)


def _rosenbrock(cp):
    p = zoo.rosenbrock2(cp)
    return p, {"x": p.variables()[0]}


def _chk_rosenbrock(p, h):
    assert np.allclose(h["x"].value, [1.0, 1.0])


def _qcp(cp):
    x = cp.Variable(1)
    y = cp.Variable(1, bounds=[0, np.inf])
    z = cp.Variable(1, bounds=[0, np.inf])
    p = cp.Problem(cp.Maximize(x), [x + y + z == 1, x ** 2 + y ** 2 - z ** 2 <= 0, x ** 2 - cp.multiply(y, z) <= 0])
    return p, {"x": x, "y": y, "z": z}


def _chk_qcp(p, h):
    assert np.allclose(h["x"].value, [0.32699284])
    assert np.allclose(h["y"].value, [0.25706586])
    assert np.allclose(h["z"].value, [0.4159413])


def _socp(cp):
    x = cp.Variable(3)
    y = cp.Variable()
    p = cp.Problem(cp.Minimize(3 * x[0] + 2 * x[1] + x[2]),
                   [cp.norm(x, 2) <= y, x[0] + x[1] + 3 * x[2] >= 1.0, y <= 5])
    return p, {"x": x, "y": y}


def _chk_socp(p, h):
    assert np.allclose(p.value, -13.548638814247532)
    assert np.allclose(h["x"].value, [-3.87462191, -2.12978826, 2.33480343])
    assert np.allclose(h["y"].value, 5)


def _portfolio_socp(cp):
    np.random.seed(858)
    n = 100
    x = cp.Variable(n, name="x")
    mu = np.random.randn(n)
    Sigma = np.random.randn(n, n)
    Sigma = Sigma.T @ Sigma
    gamma = 0.1
    t = cp.Variable(name="t", bounds=[0, None])
    L = np.linalg.cholesky(Sigma)
    p = cp.Problem(cp.Minimize(-mu.T @ x + gamma * t), [cp.norm(L.T @ x, 2) <= t, cp.sum(x) == 1, x >= 0])
    return p, {}


def _chk_portfolio_socp(p, h):
    assert np.allclose(p.value, -1.93414338e+00)


def _localization(cp):
    p = zoo.localization(cp)
    return p, {v.name(): v for v in p.variables()}


def _chk_localization(p, h):
    assert np.allclose(h["x"].value, [2.0, -1.5])


def _circle(formulation):
    def build(cp):
        rng = np.random.default_rng(5)
        n = 3
        radius = rng.uniform(1.0, 3.0, n)
        centers = cp.Variable((2, n), name="c")
        cons = []
        for i in range(n - 1):
            for j in range(i + 1, n):
                cons += [cp.sum(cp.square(centers[:, i] - centers[:, j])) >= (radius[i] + radius[j]) ** 2]
        centers.value = rng.uniform(-5.0, 5.0, (2, n))
        if formulation == 1:
            t = cp.Variable()
            cons += [cp.max(cp.norm_inf(centers, axis=0) + radius) <= t]
            obj = cp.Minimize(t)
        elif formulation == 2:
            obj = cp.Minimize(cp.max(cp.norm_inf(centers, axis=0) + radius))
        else:
            obj = cp.Minimize(cp.max(cp.max(cp.abs(centers), axis=0) + radius))
        return cp.Problem(obj, cons), {"c": centers}
    return build


def _chk_circle(p, h):
    true_sol = np.array([[1.73655994, -1.98685738, 2.57208783], [1.99273311, -1.67415425, -2.57208783]])
    assert np.allclose(h["c"].value, true_sol)


def _geo_mean(cp):
    x = cp.Variable(3, pos=True)
    return cp.Problem(cp.Maximize(cp.geo_mean(x)), [cp.sum(x) == 1]), {"x": x}


def _chk_geo_mean(p, h):
    assert np.allclose(h["x"].value, np.ones(3) / 3)


def _geo_mean2(cp):
    w = np.array([.07, .12, .23, .19, .39])
    x = cp.Variable(5, nonneg=True)
    return cp.Problem(cp.Maximize(cp.geo_mean(x, w)), [cp.sum(x) <= 1]), {"x": x, "w": w}


def _chk_geo_mean2(p, h):
    assert np.allclose(h["x"].value, h["w"] / h["w"].sum())


def _clnlbeam(cp):
    N = 1000
    hh = 1 / N
    alpha = 350
    t = cp.Variable(N + 1, bounds=[-1, 1])
    x = cp.Variable(N + 1, bounds=[-0.05, 0.05])
    u = cp.Variable(N + 1)
    u.value = np.zeros(N + 1)
    control = cp.multiply(0.5 * hh, cp.power(u[1:], 2) + cp.power(u[:-1], 2))
    trig = cp.multiply(0.5 * alpha * hh, cp.cos(t[1:]) + cp.cos(t[:-1]))
    cons = [x[1:] - x[:-1] - cp.multiply(0.5 * hh, cp.sin(t[1:]) + cp.sin(t[:-1])) == 0,
            t[1:] - t[:-1] - 0.5 * hh * (u[1:] + u[:-1]) == 0]
    return cp.Problem(cp.Minimize(cp.sum(control + trig)), cons), {}


def _chk_clnlbeam(p, h):
    assert np.allclose(p.value, 3.500e+02)


SIGMA8 = 1e-5 * np.array([
    [41.16, 22.03, 18.64, -4.74, 6.27, 10.1, 14.52, 3.18],
    [22.03, 58.57, 32.92, -5.04, 4.02, 3.7, 26.76, 2.17],
    [18.64, 32.92, 81.02, 0.53, 6.05, 2.02, 25.52, 1.56],
    [-4.74, -5.04, 0.53, 20.6, 2.52, 0.57, 0.2, 3.6],
    [6.27, 4.02, 6.05, 2.52, 10.13, 2.59, 4.32, 3.13],
    [10.1, 3.7, 2.02, 0.57, 2.59, 22.89, 3.97, 3.26],
    [14.52, 26.76, 25.52, 0.2, 4.32, 3.97, 29.91, 3.25],
    [3.18, 2.17, 1.56, 3.6, 3.13, 3.26, 3.25, 13.63]])
GROUPS = [[0, 1, 5], [3, 4, 2, 6, 7]]


def _risk_parity_vanilla(cp):
    n = 8
    target = np.ones(n) / n
    w = cp.Variable((n,), nonneg=True, name="w")
    t = cp.Variable((n,), name="t")
    cons = [cp.sum(w) == 1, t == SIGMA8 @ w]
    term1 = cp.sum(cp.multiply(cp.square(w), cp.square(t))) / cp.quad_form(w, SIGMA8)
    term2 = (LA.norm(target) ** 2) * cp.quad_form(w, SIGMA8)
    term3 = -2 * cp.sum(cp.multiply(target, cp.multiply(w, t)))
    return cp.Problem(cp.Minimize(term1 + term2 + term3), cons), {"w": w}


def _chk_risk_parity_vanilla(p, h):
    w = h["w"].value
    rc = w * (SIGMA8 @ w)
    rc /= rc.sum()
    assert np.linalg.norm(rc - np.ones(8) / 8) < 1e-5


def _risk_parity_group(formulation):
    def build(cp):
        n = 8
        b = np.array([0.4, 0.6])
        w = cp.Variable((n,), nonneg=True, name="w")
        t = cp.Variable((n,), name="t")
        cons = [cp.sum(w) == 1, t == SIGMA8 @ w]
        w.value = np.ones(n) / n
        if formulation == 1:
            t1 = t2 = t3 = 0
            for k, g in enumerate(GROUPS):
                t1 += cp.square(cp.sum(cp.multiply(w[g], t[g]))) / cp.quad_form(w, SIGMA8)
                t2 += (LA.norm(b[k]) ** 2) * cp.quad_form(w, SIGMA8)
                t3 += -2 * b[k] * cp.sum(cp.multiply(w[g], t[g]))
            obj = t1 + t2 + t3
        else:
            obj = 0
            for k, g in enumerate(GROUPS):
                obj += cp.square(cp.sum(cp.multiply(w[g], t[g])) / cp.quad_form(w, SIGMA8) - b[k])
        return cp.Problem(cp.Minimize(obj), cons), {"w": w}
    return build


def _chk_risk_parity_group(p, h):
    w = h["w"].value
    rc = w * (SIGMA8 @ w)
    rc /= rc.sum()
    rc = np.array([rc[g].sum() for g in GROUPS])
    assert np.linalg.norm(rc - np.array([0.4, 0.6])) < 1e-5


def _broadcast(kind):
    def build(cp):
        np.random.seed(0)
        if kind == "scalar":
            x = cp.Variable(name="x")
            A = np.random.randn(200, 6)
        elif kind == "row":
            x = cp.Variable(6, name="x")
            A = np.random.randn(5, 6)
        else:
            x = cp.Variable((5, 1), name="x")
            A = np.random.randn(5, 6)
        return cp.Problem(cp.Minimize(cp.sum(cp.square(x - A)))), {"x": x, "A": A}
    return build


def _chk_broadcast(kind):
    def chk(p, h):
        A = h["A"]
        ref = {"scalar": np.mean(A), "row": np.mean(A, axis=0), "col": np.mean(A, axis=1)}[kind]
        assert np.allclose(np.asarray(h["x"].value).flatten(), np.asarray(ref).flatten())
    return chk


def _best_of(cp):
    rng = np.random.default_rng(5)
    n = 5
    radius = rng.uniform(1.0, 3.0, n)
    centers = cp.Variable((n, 2), name="c")
    cons = []
    for i in range(n - 1):
        cons += [cp.sum((centers[i, :] - centers[i + 1:, :]) ** 2, axis=1) >= (radius[i] + radius[i + 1:]) ** 2]
    obj = cp.Minimize(cp.max(cp.norm_inf(centers, axis=1) + radius))
    centers.sample_bounds = [-5.0, 5.0]
    return cp.Problem(obj, cons), {"c": centers, "radius": radius}


def _chk_best_of(p, h):
    all_objs = p.solver_stats.extra_stats["all_objs_from_best_of"]
    assert len(all_objs) == 10
    manual = np.max(np.linalg.norm(h["c"].value, ord=np.inf, axis=1) + h["radius"])
    assert manual == p.objective.value
    assert manual == np.min(all_objs)


TABLE = {
    "hs071": _row(_hs071, _chk_hs071, "test_nlp_solvers.py:25-37"),
    "mle": _row(_mle, _chk_mle, "test_nlp_solvers.py:39-60"),
    "portfolio_qp": _row(_portfolio_qp, _chk_portfolio_qp, "test_nlp_solvers.py:62-86"),
    "rosenbrock": _row(_rosenbrock, _chk_rosenbrock, "test_nlp_solvers.py:88-94"),
    "qcp": _row(_qcp, _chk_qcp, "test_nlp_solvers.py:96-113"),
    "socp": _row(_socp, _chk_socp, "test_nlp_solvers.py:132-153"),
    "portfolio_socp": _row(_portfolio_socp, _chk_portfolio_socp, "test_nlp_solvers.py:155-173"),
    "localization": _row(_localization, _chk_localization, "test_nlp_solvers.py:175-189"),
    "circle_packing_f1": _row(_circle(1), _chk_circle, "test_nlp_solvers.py:191-213"),
    "circle_packing_f2": _row(_circle(2), _chk_circle, "test_nlp_solvers.py:215-237"),
    "circle_packing_f3": _row(_circle(3), _chk_circle, "test_nlp_solvers.py:239-259"),
    "geo_mean": _row(_geo_mean, _chk_geo_mean, "test_nlp_solvers.py:261-269"),
    "geo_mean_weighted": _row(_geo_mean2, _chk_geo_mean2, "test_nlp_solvers.py:271-283"),
    "clnlbeam": _row(_clnlbeam, _chk_clnlbeam, "test_nlp_solvers.py:285-309"),
    "risk_parity_vanilla": _row(_risk_parity_vanilla, _chk_risk_parity_vanilla, "test_risk_parity.py:26-42"),
    "risk_parity_group_f1": _row(_risk_parity_group(1), _chk_risk_parity_group, "test_risk_parity.py:49-74"),
    "risk_parity_group_f2": _row(_risk_parity_group(2), _chk_risk_parity_group, "test_risk_parity.py:77-97"),
    "broadcast_scalar": _row(_broadcast("scalar"), _chk_broadcast("scalar"), "test_broadcast.py:11-21"),
    "broadcast_row": _row(_broadcast("row"), _chk_broadcast("row"), "test_broadcast.py:23-33"),
    "broadcast_col": _row(_broadcast("col"), _chk_broadcast("col"), "test_broadcast.py:35-45"),
    "best_of": _row(_best_of, _chk_best_of, "test_best_of.py:11-32", best_of=10),
}
