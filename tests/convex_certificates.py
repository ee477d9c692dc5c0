"""The thirteen reference tests that cross-check the NLP path against a conic solver this image does not have
(CLARABEL: test_abs.py:35,58,81,104, test_entropy_related.py:25,59,75,91,107, test_huber_sum_largest.py:39,59,79,
test_Sharpe_ratio.py:40), restated with a CERTIFICATE in the conic solver's place.  All thirteen are convex
(or, for the Sharpe ratio, equivalent to a convex problem), so a point is globally optimal to within a
duality gap that can be bounded from the returned point alone:

  * lasso  ||Ax - b||^2 + lam ||x||_1 :  the dual  max -1/4 ||nu||^2 - nu'b  s.t. ||A'nu||_inf <= lam,
    with nu = 2 (Ax - b) scaled into the dual feasible set;
  * smooth convex f over the simplex {sum q = 1, q >= 0} (entropy, relative entropy, KL in either argument
    order): the Frank-Wolfe gap  grad f(q)'q - min_i grad_i f(q)  >=  f(q) - f*  (first-order lower bound
    minimised over the vertices), with grad f written out here in numpy; the reference compares the
    optimiser q itself with the conic solver's, which is done here against scipy's SLSQP (an independent
    solver, its own gap checked too);
  * Huber regression: the conjugate dual  max sum(-nu_i y_i - nu_i^2 / 4)  s.t.  X nu = 0, |nu_i| <= 2,
    with nu = huber'(residual) projected onto null(X) and scaled into the box;
  * sum_largest / sum_smallest over the simplex: small LPs, solved exactly by scipy.optimize.linprog (HiGHS);
  * Sharpe ratio: max (mu'x)^2 / x'Sx over the simplex == min y'Sy s.t. mu'y = 1, y >= 0 (homogeneity);
    the Frank-Wolfe gap of that convex QP over its vertices e_i / mu_i bounds the optimal ratio.

Each row: build(cp) -> (problem, handles); check(problem, handles) asserts the reference's tolerance on the
certified gap.  tests/test_convex_certificates.py runs every row on both engines."""
import numpy as np
import numpy.linalg as LA


def _row(build, check, src, **kw):
    return {"build": build, "check": check, "src": src, "kwargs": kw}


# ---- lasso (test_abs.py) ----------------------------------------------------------------------------------------
def _lasso_rows(m, n, src):
    rows = {}
    np_state = np.random.RandomState(0)            # the reference seeds once and draws b, A per factor
    factors = np.linspace(0.1, 1, 20)
    data = []
    for factor in factors:
        b = np_state.randn(m)
        A = np_state.randn(m, n)
        data.append((factor * 2 * LA.norm(A.T @ b, np.inf), A, b))

    def make(k):
        lmbda, A, b = data[k]

        def build(cp):
            x = cp.Variable((n,), name="x")
            obj = cp.sum(cp.square(A @ x - b)) + lmbda * cp.sum(cp.abs(x))
            return cp.Problem(cp.Minimize(obj)), {"x": x, "obj": obj}

        def check(prob, h):
            x = h["x"].value
            r = A @ x - b
            primal = float(r @ r + lmbda * np.sum(np.abs(x)))
            assert abs(primal - h["obj"].value) <= 1e-9 * max(1.0, abs(primal))
            nu = 2.0 * r
            s = LA.norm(A.T @ nu, np.inf)
            if s > lmbda:
                nu *= lmbda / s
            dual = float(-0.25 * nu @ nu - nu @ b)
            assert dual <= primal + 1e-9 * abs(primal)
            # the reference's tolerance: |obj_nlp - obj_dcp| / obj_nlp <= 1e-4, with dual <= obj_dcp <= obj_nlp
            assert (primal - dual) / primal <= 1e-4, (k, primal, dual)
        return build, check
    for k in range(len(factors)):
        bld, chk = make(k)
        rows["%s_%02d" % (src.split("::")[-1], k)] = _row(bld, chk, src, hessian_approximation="exact",
                                                          derivative_test="none")
    return rows


# ---- smooth convex objectives over the simplex (test_entropy_related.py) -----------------------------------------
def _simplex_row(n, make_obj, value, grad, src, maximize=False):
    def data():
        rs = np.random.RandomState(0)
        p = None
        if n == 40:
            p = rs.rand(n)
            p = p / np.sum(p)
        A = rs.rand(n, n)
        return A, p

    def build(cp):
        A, p = data()
        q = cp.Variable(n, nonneg=True)
        obj = make_obj(cp, A, q, p)
        prob = cp.Problem(cp.Maximize(obj) if maximize else cp.Minimize(obj), [cp.sum(q) == 1])
        return prob, {"q": q}

    def check(prob, h):
        A, p = data()
        q = np.maximum(h["q"].value, 0.0)
        assert abs(np.sum(q) - 1.0) <= 1e-8 and np.min(h["q"].value) >= -1e-9
        q = q / np.sum(q)
        f = value(A, q, p)            # the CONVEX function being minimised (sign flipped for the concave maximisation)
        g = grad(A, q, p)
        gap = float(g @ q - np.min(g))
        assert gap >= -1e-10
        assert gap <= 1e-6 * max(1.0, abs(f)), (gap, f)
        # distance to the optimiser, as the reference asserts it (||q_nlp - q_conic|| <= 1e-4).  The gap does not
        # bound it usefully (the Hessians' smallest eigenvalue over the simplex is ~1e-4 .. 1e-7, so a rigorous
        # 1e-4 in q would need a gap below FP64 resolution); the conic solver's place is taken by an independent
        # third-party solver, scipy's SLSQP on the same smooth problem, itself certified by its own gap
        from scipy.optimize import minimize
        tiny = 1e-300
        ref = minimize(lambda z: value(A, np.maximum(z, tiny), p), np.ones(n) / n,
                       jac=lambda z: grad(A, np.maximum(z, tiny), p), method="SLSQP", bounds=[(0, None)] * n,
                       constraints=[{"type": "eq", "fun": lambda z: np.sum(z) - 1, "jac": lambda z: np.ones(n)}],
                       options={"ftol": 1e-15, "maxiter": 2000})
        qs = np.maximum(ref.x, 0.0) / np.sum(np.maximum(ref.x, 0.0))
        gs = grad(A, qs, p)
        assert float(gs @ qs - np.min(gs)) <= 1e-5 * max(1.0, abs(f))
        assert LA.norm(h["q"].value - ref.x) <= 1e-4
    return _row(build, check, src, derivative_test="none")


def _ent_val(A, q, p):
    y = A @ q
    return float(np.sum(y * np.log(y)))


def _ent_grad(A, q, p):
    return A.T @ (np.log(A @ q) + 1.0)


def _relent_val(A, q, p):
    y = A @ q
    return float(np.sum(y * np.log(y / p)))


def _relent_grad(A, q, p):
    return A.T @ (np.log(A @ q / p) + 1.0)


def _relent_sw_val(A, q, p):
    return float(np.sum(p * np.log(p / (A @ q))))


def _relent_sw_grad(A, q, p):
    return -A.T @ (p / (A @ q))


def _kl_val(A, q, p):
    y = A @ q
    return float(np.sum(y * np.log(y / p) - y + p))


def _kl_grad(A, q, p):
    return A.T @ np.log(A @ q / p)


def _kl2_val(A, q, p):
    y = A @ q
    return float(np.sum(p * np.log(p / y) - p + y))


def _kl2_grad(A, q, p):
    return A.T @ (1.0 - p / (A @ q))


# ---- Huber regression (test_huber_sum_largest.py:12-39) -----------------------------------------------------------
def _huber_rows():
    rs = np.random.RandomState(1)
    n = 100
    samples = int(1.5 * n)
    beta_true = 5 * rs.normal(size=(n, 1))
    X = rs.randn(n, samples)
    v = rs.normal(size=(samples, 1))
    rows = {}
    Ys = []
    for p in np.linspace(0, 0.15, num=5):
        factor = 2 * rs.binomial(1, 1 - p, size=(samples, 1)) - 1
        Ys.append(factor * X.T.dot(beta_true) + v)

    def make(k):
        Y = Ys[k]

        def build(cp):
            beta = cp.Variable((n, 1))
            cost = cp.sum(cp.huber(X.T @ beta - Y, 1))
            return cp.Problem(cp.Minimize(cost)), {"beta": beta}

        def check(prob, h):
            r = (X.T @ h["beta"].value - Y).ravel()
            primal = float(np.sum(np.where(np.abs(r) <= 1.0, r * r, 2.0 * np.abs(r) - 1.0)))
            assert abs(primal - prob.value) <= 1e-5 * max(1.0, abs(primal))      # (epigraph slack of the canonical form)
            nu = np.clip(2.0 * r, -2.0, 2.0)                      # huber'(r)
            # dual feasibility X nu = 0: the saturated entries (|r| > 1, nu = +-2) stay where the optimum has
            # them; the stationarity residual is absorbed by the entries strictly inside the box (min-norm
            # correction over those columns), then a scaling back into the box if that overshoots
            free = np.abs(r) < 1.0
            Xf = X[:, free]
            nu[free] -= Xf.T @ LA.lstsq(Xf @ Xf.T, X @ nu, rcond=None)[0]
            s = np.max(np.abs(nu))
            if s > 2.0:
                nu *= 2.0 / s
            assert LA.norm(X @ nu, np.inf) <= 1e-9 * max(1.0, LA.norm(nu, np.inf)) * LA.norm(X, np.inf)
            dual = float(np.sum(-nu * Y.ravel() - 0.25 * nu * nu))
            assert dual <= primal + 1e-8 * abs(primal)
            assert primal - dual <= 1e-4, (k, primal, dual)        # the reference's |nlp - conic| <= 1e-4
        return build, check
    for k in range(5):
        bld, chk = make(k)
        rows["huber_%d" % k] = _row(bld, chk, "test_huber_sum_largest.py::TestNonsmoothNontrivial::test_huber")
    return rows


# ---- sum_largest / sum_smallest (test_huber_sum_largest.py:42-79): LPs, solved exactly by HiGHS ----------------------
_W = np.array([0.1, 0.2, 0.3, 0.4, 0.5])


def _lp_sum_largest(w, k):
    """min sum_largest(w o x, k) over the simplex = min k s + sum t, t >= w o x - s, t >= 0."""
    from scipy.optimize import linprog
    n = w.size
    c = np.concatenate([np.zeros(n), [k], np.ones(n)])                     # x, s, t
    A_ub = np.hstack([np.diag(w), -np.ones((n, 1)), -np.eye(n)])
    A_eq = np.concatenate([np.ones(n), [0.0], np.zeros(n)])[None, :]
    bounds = [(0, None)] * n + [(None, None)] + [(0, None)] * n
    res = linprog(c, A_ub=A_ub, b_ub=np.zeros(n), A_eq=A_eq, b_eq=[1.0], bounds=bounds, method="highs")
    assert res.status == 0
    return res.fun


def _sum_largest_build(cp):
    x = cp.Variable(5)
    prob = cp.Problem(cp.Minimize(cp.sum_largest(cp.multiply(x, _W), 2)), [cp.sum(x) == 1, x >= 0])
    return prob, {"x": x}


def _sum_largest_check(prob, h):
    x = h["x"].value
    assert abs(np.sum(x) - 1) <= 1e-7 and np.min(x) >= -1e-8
    value = np.sum(np.sort(x * _W)[-2:])                   # the function itself at the returned point
    assert abs(value - prob.value) <= 1e-5
    assert abs(value - _lp_sum_largest(_W, 2)) <= 1e-4


def _sum_smallest_build(cp):
    x = cp.Variable(5)
    prob = cp.Problem(cp.Maximize(cp.sum_smallest(cp.multiply(x, _W), 2)), [cp.sum(x) == 1, x >= 0])
    return prob, {"x": x}


def _sum_smallest_check(prob, h):
    x = h["x"].value
    assert abs(np.sum(x) - 1) <= 1e-7 and np.min(x) >= -1e-8
    value = np.sum(np.sort(x * _W)[:2])
    assert abs(value - prob.value) <= 1e-5
    # max sum_smallest(w o x, 2) = - min sum_largest(-(w o x), 2)
    assert abs(value + _lp_sum_largest(-_W, 2)) <= 1e-4


# ---- Sharpe ratio (test_Sharpe_ratio.py:14-40) -------------------------------------------------------------------
def _sharpe_data():
    rs = np.random.RandomState(0)
    n = 100
    S = rs.rand(n, n)
    return S @ S.T, rs.rand(n), n


def _sharpe_build(cp):
    Sigma, mu, n = _sharpe_data()
    x = cp.Variable((n,), nonneg=True)
    x.value = np.ones(n) / n
    obj = cp.square(mu @ x) / cp.quad_form(x, Sigma)
    return cp.Problem(cp.Maximize(obj), [cp.sum(x) == 1]), {"x": x}


def _sharpe_check(prob, h):
    Sigma, mu, n = _sharpe_data()
    x = np.maximum(h["x"].value, 0.0)
    assert abs(np.sum(h["x"].value) - 1) <= 1e-7
    ratio = mu @ x / np.sqrt(x @ Sigma @ x)
    # the equivalent convex QP  min y'Sy  s.t. mu'y = 1, y >= 0  at y = x / (mu'x); its feasible set is the
    # simplex with vertices e_i / mu_i, so the Frank-Wolfe gap bounds f(y) - f* and 1 / sqrt(f*) is the best ratio
    y = x / (mu @ x)
    f = y @ Sigma @ y
    g = 2.0 * Sigma @ y
    gap = float(g @ y - np.min(g / mu))
    assert -1e-9 * f <= gap < f
    best = 1.0 / np.sqrt(f - max(gap, 0.0))
    assert ratio <= best * (1 + 1e-12)
    assert best - ratio < 1e-6, (ratio, best)                 # the reference's |sharpe_nlp - sharpe_cvx| < 1e-6


def build_table():
    t = {}
    t.update(_lasso_rows(10, 10, "test_abs.py::TestAbs::test_lasso_square_small"))
    t.update(_lasso_rows(50, 50, "test_abs.py::TestAbs::test_lasso_square"))
    t.update(_lasso_rows(100, 200, "test_abs.py::TestAbs::test_lasso_underdetermined"))
    t.update(_lasso_rows(200, 100, "test_abs.py::TestAbs::test_lasso_overdetermined"))
    E = "test_entropy_related.py::TestEntropy::"
    t["entropy_one"] = _simplex_row(100, lambda cp, A, q, p: cp.sum(cp.entr(A @ q)), _ent_val, _ent_grad,
                                    E + "test_entropy_one", maximize=True)
    t["rel_entropy_one"] = _simplex_row(40, lambda cp, A, q, p: cp.sum(cp.rel_entr(A @ q, p)), _relent_val,
                                        _relent_grad, E + "test_rel_entropy_one")
    t["rel_entropy_one_switched_arguments"] = _simplex_row(
        40, lambda cp, A, q, p: cp.sum(cp.rel_entr(p, A @ q)), _relent_sw_val, _relent_sw_grad,
        E + "test_rel_entropy_one_switched_arguments")
    t["KL_one"] = _simplex_row(40, lambda cp, A, q, p: cp.sum(cp.kl_div(A @ q, p)), _kl_val, _kl_grad,
                               E + "test_KL_one")
    t["KL_two"] = _simplex_row(40, lambda cp, A, q, p: cp.sum(cp.kl_div(p, A @ q)), _kl2_val, _kl2_grad,
                               E + "test_KL_two")
    t.update(_huber_rows())
    H = "test_huber_sum_largest.py::TestNonsmoothNontrivial::"
    t["sum_largest"] = _row(_sum_largest_build, _sum_largest_check, H + "test_sum_largest")
    t["sum_smallest"] = _row(_sum_smallest_build, _sum_smallest_check, H + "test_sum_smallest")
    t["sharpe_ratio"] = _row(_sharpe_build, _sharpe_check, "test_Sharpe_ratio.py::TestSharpeRatio::test_formulation_one",
                             hessian_approximation="exact")
    return t


TABLE = build_table()
