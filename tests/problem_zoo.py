"""Problem builders shared by the golden-vector generator and the tests.

Every builder takes the modelling namespace `cp` — the reference's `cvxpy` (only in the
build container, through tools/ref_import.py) or `dnlp_amd` — and returns a Problem, so the
same text builds the same problem on both sides.  Problems follow the reference's own tests
(cvxpy/tests/NLP_tests/test_nlp_solvers.py, test_scalar_and_matrix_problems.py, hess/jac
unit tests) and scaled-down BASELINE configs.  Data is seeded numpy.
"""
import numpy as np


def readme_toy(cp):
    """README.md:29-52 (BASELINE C1)."""
    np.random.seed(0)
    A = np.random.randn(3, 3)
    A = A.T @ A
    x = cp.Variable(3)
    x.value = np.ones(3)
    return cp.Problem(cp.Maximize(cp.quad_form(x, A)), [cp.sum_squares(x) == 1])


def hs071(cp):
    """test_nlp_solvers.py:25-37."""
    x = cp.Variable(4, bounds=[0, 6])
    x.value = np.array([1.0, 5.0, 5.0, 1.0])
    objective = cp.Minimize(x[0] * x[3] * (x[0] + x[1] + x[2]) + x[2])
    constraints = [x[0] * x[1] * x[2] * x[3] >= 25, cp.sum(cp.square(x)) == 40]
    return cp.Problem(objective, constraints)


def mle(cp):
    """test_nlp_solvers.py:39-60."""
    n = 1000
    np.random.seed(1234)
    data = np.random.randn(n)
    mu = cp.Variable((1,), name="mu")
    mu.value = np.array([0.0])
    sigma = cp.Variable((1,), name="sigma")
    sigma.value = np.array([1.0])
    constraints = [mu == sigma ** 2]
    log_likelihood = ((n / 2) * cp.log(1 / (2 * np.pi * (sigma) ** 2))
                      - cp.sum(cp.square(data - mu)) / (2 * (sigma) ** 2))
    return cp.Problem(cp.Maximize(log_likelihood), constraints)


def portfolio_qp(cp):
    """test_nlp_solvers.py:62-86."""
    r = np.array([0.026002150277777, 0.008101316405671, 0.073715909491990])
    Q = np.array([[0.018641039983891, 0.003598532927677, 0.001309759253660],
                  [0.003598532927677, 0.006436938322676, 0.004887265158407],
                  [0.001309759253660, 0.004887265158407, 0.068682765454814]])
    x = cp.Variable(3)
    x.value = np.array([10.0, 10.0, 10.0])
    variance = cp.quad_form(x, Q)
    expected_return = r @ x
    return cp.Problem(cp.Minimize(variance),
                      [cp.sum(x) <= 1000, expected_return >= 50, x >= 0])


def rosenbrock2(cp):
    """test_nlp_solvers.py:88-94."""
    x = cp.Variable(2, name="x")
    objective = cp.Minimize((1 - x[0]) ** 2 + 100 * (x[1] - x[0] ** 2) ** 2)
    return cp.Problem(objective, [])


def qcp(cp):
    """test_nlp_solvers.py:96-113."""
    x = cp.Variable(1)
    y = cp.Variable(1, bounds=[0, np.inf])
    z = cp.Variable(1, bounds=[0, np.inf])
    objective = cp.Maximize(x)
    constraints = [x + y + z == 1, x ** 2 + y ** 2 - z ** 2 <= 0, x ** 2 - cp.multiply(y, z) <= 0]
    return cp.Problem(objective, constraints)


def socp(cp):
    """test_nlp_solvers.py:132-153."""
    x = cp.Variable(3)
    y = cp.Variable()
    objective = cp.Minimize(3 * x[0] + 2 * x[1] + x[2])
    constraints = [cp.norm(x, 2) <= y, x[0] + x[1] + 3 * x[2] >= 1.0, y <= 5]
    return cp.Problem(objective, constraints)


def localization(cp):
    """test_nlp_solvers.py:175-189."""
    np.random.seed(42)
    m = 10
    dim = 2
    x_true = np.array([2.0, -1.5])
    a = np.random.uniform(-5, 5, (m, dim))
    rho = np.linalg.norm(a - x_true, axis=1)
    x = cp.Variable(2, name="x")
    t = cp.Variable(m, name="t")
    constraints = [t == cp.sqrt(cp.sum(cp.square(x - a), axis=1))]
    objective = cp.Minimize(cp.sum_squares(t - rho))
    return cp.Problem(objective, constraints)


def circle_packing(cp):
    """test_nlp_solvers.py:191-213 (formulation 1)."""
    rng = np.random.default_rng(5)
    n = 3
    radius = rng.uniform(1.0, 3.0, n)
    centers = cp.Variable((2, n), name="c")
    constraints = []
    for i in range(n - 1):
        for j in range(i + 1, n):
            constraints += [cp.sum(cp.square(centers[:, i] - centers[:, j])) >=
                            (radius[i] + radius[j]) ** 2]
    centers.value = rng.uniform(-5.0, 5.0, (2, n))
    obj = cp.Minimize(cp.max(cp.norm_inf(centers, axis=0) + radius))
    return cp.Problem(obj, constraints)


def geo_mean_problem(cp):
    """test_nlp_solvers.py:261-269."""
    x = cp.Variable(3, pos=True)
    geo = cp.geo_mean(x)
    return cp.Problem(cp.Maximize(geo), [cp.sum(x) == 1])


def rosenbrock_chain(cp, n=50):
    """BASELINE C2 scaled down (SURVEY.md Appendix C formulation)."""
    x = cp.Variable(n)
    f = cp.sum(cp.square(1 - x[:-1])) + 100 * cp.sum(cp.square(x[1:] - cp.square(x[:-1])))
    return cp.Problem(cp.Minimize(f), [])


def dense_eq_qp(cp, n=40, m=6):
    """BASELINE C3 scaled down: min 1/2 x'Qx + c'x s.t. Ax = b."""
    rng = np.random.default_rng(0)
    Gm = rng.standard_normal((n, n))
    Q = Gm.T @ Gm / n + np.eye(n)
    c = rng.standard_normal(n)
    A = rng.standard_normal((m, n))
    xh = rng.standard_normal(n)
    b = A @ xh
    x = cp.Variable(n)
    return cp.Problem(cp.Minimize(0.5 * cp.quad_form(x, Q) + c @ x), [A @ x == b])


def sphere(cp, n=60):
    """BASELINE C4 scaled down: max x'Ax on the unit sphere, dense (n > 48) quad_form."""
    rng = np.random.default_rng(0)
    A0 = rng.standard_normal((n, n))
    A = A0.T @ A0 / n
    x = cp.Variable(n)
    x.value = np.ones(n)
    return cp.Problem(cp.Maximize(cp.quad_form(x, A)), [cp.sum_squares(x) == 1])


def elementwise_zoo(cp):
    """One constraint per smooth elementwise atom on a matrix variable (F-order check),
    modelled on test_scalar_and_matrix_problems.py / the jac & hess unit tests."""
    rng = np.random.default_rng(1)
    X = cp.Variable((3, 2), bounds=[0.2, 0.9])
    X.value = rng.uniform(0.3, 0.8, (3, 2))
    y = cp.Variable(3, bounds=[0.1, 2.0])
    y.value = rng.uniform(0.5, 1.5, 3)
    W = rng.uniform(0.5, 1.5, (3, 2))
    obj = cp.Minimize(cp.sum(cp.multiply(W, cp.exp(X))) + cp.sum(cp.entr(y)) - cp.sum(cp.log(X))
                      + cp.sum(cp.logistic(y)) + cp.sum(cp.xexp(X)))
    cons = [cp.sum(cp.sin(X), axis=0) <= 2.5,
            cp.sum(cp.cos(X), axis=1) >= 0.1,
            cp.tan(y[0:2]) <= 50,
            cp.sinh(y) + cp.tanh(y) + cp.asinh(y) <= 20,
            cp.atanh(X[0, :]) <= 3,
            cp.sum(cp.sqrt(y)) >= 0.5,
            cp.sum(cp.power(y, 3)) <= 30,
            cp.sum(cp.rel_entr(y, X[:, 0])) <= 10,
            cp.sum(cp.kl_div(X[:, 1], y)) <= 10,
            cp.quad_over_lin(X[:, 0], y[1]) <= 40,
            X.T @ y >= 0.01]
    return cp.Problem(obj, cons)


def bilinear_matmul(cp):
    """Var @ Var bilinear products (test_hess_matmul.py / test_jac_matmul.py; NMF-like)."""
    rng = np.random.default_rng(2)
    U = cp.Variable((3, 2), bounds=[0, None])
    V = cp.Variable((2, 4), bounds=[0, None])
    U.value = rng.uniform(0.5, 1.5, (3, 2))
    V.value = rng.uniform(0.5, 1.5, (2, 4))
    M = rng.uniform(0.5, 2.0, (3, 4))
    return cp.Problem(cp.Minimize(cp.sum(cp.square(U @ V - M))), [cp.sum(U) <= 20])


def nonsmooth_zoo(cp):
    """ESR/HSR atoms in epigraph form (abs, maximum, minimum, max, min, norm1, norm_inf,
    huber, sum_largest, sum_smallest) — modelled on test_abs.py / test_huber_sum_largest.py."""
    rng = np.random.default_rng(3)
    A = rng.standard_normal((8, 4))
    b = rng.standard_normal(8)
    x = cp.Variable(4)
    x.value = rng.standard_normal(4)
    r = A @ x - b
    obj = cp.Minimize(cp.sum(cp.huber(r, 0.5)) + 0.3 * cp.norm1(x) + 0.1 * cp.norm_inf(x)
                      + cp.sum(cp.maximum(x, -0.1)) + 0.2 * cp.sum_largest(r, 3)
                      + cp.max(x) + cp.sum(cp.abs(r)))
    cons = [cp.min(x) >= -3, cp.sum(cp.minimum(x, 0.5)) >= -5, cp.sum_smallest(x, 2) >= -6]
    return cp.Problem(obj, cons)


def broadcast_div(cp):
    """Division by an expression, promote and broadcast_to (test_broadcast.py, div_canon)."""
    rng = np.random.default_rng(4)
    A = rng.uniform(1.0, 2.0, (3, 2))
    x = cp.Variable((3, 1), bounds=[0.5, 3])
    x.value = np.ones((3, 1))
    s = cp.Variable(bounds=[0.5, 4])
    s.value = 1.5
    obj = cp.Minimize(cp.sum(cp.square(cp.broadcast_to(x, (3, 2)) - A)) + cp.sum(x) / s
                      + cp.sum(cp.square(s - A)))
    return cp.Problem(obj, [cp.sum(x) >= 2])


ZOO = {
    "readme_toy": readme_toy,
    "hs071": hs071,
    "mle": mle,
    "portfolio_qp": portfolio_qp,
    "rosenbrock2": rosenbrock2,
    "qcp": qcp,
    "socp": socp,
    "localization": localization,
    "circle_packing": circle_packing,
    "geo_mean": geo_mean_problem,
    "rosenbrock_chain50": rosenbrock_chain,
    "dense_eq_qp": dense_eq_qp,
    "sphere60": sphere,
    "elementwise_zoo": elementwise_zoo,
    "bilinear_matmul": bilinear_matmul,
    "nonsmooth_zoo": nonsmooth_zoo,
    "broadcast_div": broadcast_div,
}

# golden-vector coverage = the zoo above + the paper's examples at notebook size
from paper_examples import PAPER  # noqa: E402

GOLDEN_ZOO = dict(ZOO)
GOLDEN_ZOO.update(PAPER)
