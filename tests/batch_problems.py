"""Parametrised paper NLPs for the batch path (BASELINE config C5): instance i draws its data
from default_rng(i).  `oracle_solver` solves with the CPU oracle (tests only)."""
import numpy as np


def build_localization(i, m=10):
    """Sensor localization (examples/nlp_examples/localization.ipynb; test_nlp_solvers.py:175-189)
    with anchors and true position drawn per instance."""
    import dnlp_amd as cp
    rng = np.random.default_rng(i)
    x_true = rng.uniform(-3, 3, 2)
    a = rng.uniform(-5, 5, (m, 2))
    rho = np.linalg.norm(a - x_true, axis=1)
    x = cp.Variable(2, name="x")
    t = cp.Variable(m, name="t")
    prob = cp.Problem(cp.Minimize(cp.sum_squares(t - rho)),
                      [t == cp.sqrt(cp.sum(cp.square(x - a), axis=1))])
    prob._x_true = x_true
    return prob


def build_circle_packing(i, n=4):
    """Circle packing (examples/nlp_examples/circle_packing.ipynb) with radii per instance."""
    import dnlp_amd as cp
    rng = np.random.default_rng(i)
    radius = rng.uniform(1.0, 3.0, n)
    centers = cp.Variable((2, n), name="c")
    cons = []
    for a in range(n - 1):
        for b in range(a + 1, n):
            cons += [cp.sum(cp.square(centers[:, a] - centers[:, b])) >= (radius[a] + radius[b]) ** 2]
    centers.value = rng.uniform(-5.0, 5.0, (2, n))
    return cp.Problem(cp.Minimize(cp.max(cp.norm_inf(centers, axis=0) + radius)), cons)


def oracle_solver(problem, **opts):
    from dnlp_amd.dnlp2smooth import Dnlp2Smooth
    from dnlp_amd.nlp_solver import build_nlp_data
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    import dnlp_amd as cp
    flip = isinstance(problem.objective, cp.Maximize)
    if flip:
        problem = cp.Problem(cp.Minimize(-problem.objective.expr), problem.constraints)
    smooth, _ = Dnlp2Smooth().apply(problem)
    data, _ = build_nlp_data(smooth)
    h = OracleProblem(serialize(data["tape_arrays"]))
    for k, v in opts.items():
        h.set_option(k, v)
    info = h.solve(data["x0"])
    obj = -info["obj_val"] if flip else info["obj_val"]
    return obj, info["status"], info["iterations"], info["x"]


# ---- parametrised templates for the single-launch batch path (dnlp_amd.batch.ParametricBatch) ----
def template_localization(m=10):
    """Localization with anchors `a` (m x 2) and ranges `rho` (m) as Parameters.
    Returns (problem, [a, rho], sample) with sample(i) -> flattened parameter row of instance i
    (same draws as build_localization(i))."""
    import dnlp_amd as cp
    a = cp.Parameter((m, 2), name="a", value=np.zeros((m, 2)))
    rho = cp.Parameter(m, name="rho", value=np.ones(m))
    x = cp.Variable(2, name="x")
    t = cp.Variable(m, name="t")
    prob = cp.Problem(cp.Minimize(cp.sum_squares(t - rho)),
                      [t == cp.sqrt(cp.sum(cp.square(x - a), axis=1))])

    def sample(i):
        rng = np.random.default_rng(i)
        x_true = rng.uniform(-3, 3, 2)
        av = rng.uniform(-5, 5, (m, 2))
        rv = np.linalg.norm(av - x_true, axis=1)
        return np.concatenate([av.reshape(-1, order="F"), rv])

    return prob, [a, rho], sample, x


def template_circle_packing(n=4):
    """Circle packing with the squared centre distances' lower bounds R2[a,b] = (r_a + r_b)^2 and
    the radii as Parameters (the tape data is affine in both)."""
    import dnlp_amd as cp
    pairs = [(a, b) for a in range(n - 1) for b in range(a + 1, n)]
    R2 = cp.Parameter(len(pairs), name="R2", value=np.full(len(pairs), 4.0))
    rad = cp.Parameter(n, name="rad", value=np.ones(n))
    rng0 = np.random.default_rng(0)
    if n >= 10:
        # the notebook's member (circle_packing.ipynb, n = 10), in the notebook's own formulation (centres
        # n x 2, one vector constraint per circle against the circles after it): its radii are instance 0's
        # (first draw of default_rng(0)) and the start is its second draw, so instance 0 IS the published
        # problem, canonical form included
        rng0.uniform(1.0, 3.0, n)
        centers = cp.Variable((n, 2), name="c")
        cons, k = [], 0
        for i in range(n - 1):
            cons += [cp.sum((centers[i, :] - centers[i + 1:, :]) ** 2, axis=1) >= R2[k:k + n - 1 - i]]
            k += n - 1 - i
        centers.value = rng0.uniform(-5.0, 5.0, (2, n)).T
        prob = cp.Problem(cp.Minimize(cp.max(cp.norm_inf(centers, axis=1) + rad)), cons)
    else:
        centers = cp.Variable((2, n), name="c")
        cons = [cp.sum(cp.square(centers[:, a] - centers[:, b])) >= R2[k] for k, (a, b) in enumerate(pairs)]
        centers.value = rng0.uniform(-5.0, 5.0, (2, n))
        prob = cp.Problem(cp.Minimize(cp.max(cp.norm_inf(centers, axis=0) + rad)), cons)

    def sample(i):
        rng = np.random.default_rng(i)
        radius = rng.uniform(1.0, 3.0, n)
        r2 = np.array([(radius[a] + radius[b]) ** 2 for a, b in pairs])
        return np.concatenate([r2, radius])

    return prob, [R2, rad], sample, centers


def template_path_planning(n=50):
    """Path planning (examples/nlp_examples/path_planning.ipynb) with the obstacle centres `p`
    (5 x 2) and squared radii `r2` (5) as Parameters: KKT order 1636, the 256-lane batch path."""
    import dnlp_amd as cp
    l, d = 10, 2
    a = np.array([[1.25, 1.25]])
    b = np.array([[l, l]])
    p0 = np.array([[2, 4.5, 6, 7, 8.5], [2.2, 5, 8, 6, 9]]).T
    r0 = np.array([1, 0.8, 0.4, 1.4, 0.5])
    p = cp.Parameter((5, 2), name="p", value=p0)
    r2 = cp.Parameter(5, name="r2", value=r0 ** 2)
    x = cp.Variable((d, n + 1), name="x")
    L = cp.Variable(name="L")
    cons = [x[:, 0] == a, x[:, n] == b]
    cons += [cp.sum(cp.square(x[:, 1:] - x[:, :-1]), axis=0) <= (L / n) ** 2]
    for i in range(n + 1):
        cons += [cp.sum(cp.square(x[:, i] - p), axis=1) >= r2]
    x.value = (b.T - a.T) / n * np.arange(n + 1) + a.T
    prob = cp.Problem(cp.Minimize(L), cons)

    def sample(i):
        rng = np.random.default_rng(i)
        pv = p0 + (rng.uniform(-0.15, 0.15, p0.shape) if i else 0.0)
        rv = r0 * (rng.uniform(0.9, 1.1, 5) if i else 1.0)
        return np.concatenate([pv.reshape(-1, order="F"), rv ** 2])

    return prob, [p, r2], sample, x


def template_power_flow():
    """AC optimal power flow on the IEEE 9-bus case (examples/nlp_examples/power_flow.ipynb) with the
    active / reactive loads of buses 4, 6, 8 as Parameters (the notebook fixes them through equal
    lower and upper bounds; here they are equality rows so that they can vary per instance)."""
    import dnlp_amd as cp
    from paper_examples import ieee9_admittance
    N = 9
    gen, load = [0, 1, 2], [4, 6, 8]
    p_min, p_max, q_min, q_max = np.zeros(N), np.zeros(N), np.zeros(N), np.zeros(N)
    p_min[gen] = [10, 10, 10]
    p_max[gen] = [250, 300, 270]
    q_min[gen] = [-5, -5, -5]
    p_min[load] = q_min[load] = -1e3          # loads are set by the equality rows below
    G, B = ieee9_admittance()
    Pl = cp.Parameter(3, name="Pload", value=np.array([54.0, 60.0, 75.0]))
    Ql = cp.Parameter(3, name="Qload", value=np.array([18.0, 21.0, 30.0]))
    theta, P, Q = cp.Variable((N, 1)), cp.Variable((N, N)), cp.Variable((N, N))
    v = cp.Variable((N, 1), bounds=[0.9, 1.1])
    p = cp.Variable(N, bounds=[p_min, p_max])
    q = cp.Variable(N, bounds=[q_min, q_max])
    C, S = cp.cos(theta - theta.T), cp.sin(theta - theta.T)
    cons = [theta[0] == 0, p == cp.sum(P, axis=1), q == cp.sum(Q, axis=1),
            p[load] == -Pl, q[load] == -Ql,
            P == cp.multiply(v @ v.T, cp.multiply(G, C) + cp.multiply(B, S)),
            Q == cp.multiply(v @ v.T, cp.multiply(G, S) - cp.multiply(B, C))]
    cost = (0.11 * p[0] ** 2 + 5 * p[0] + 150 + 0.085 * p[1] ** 2 + 1.2 * p[1] + 600
            + 0.1225 * p[2] ** 2 + p[2] + 335)
    v.value = np.ones((N, 1))
    theta.value = np.zeros((N, 1))
    prob = cp.Problem(cp.Minimize(cost), cons)

    def sample(i):
        rng = np.random.default_rng(i)
        scale = rng.uniform(0.8, 1.2, 3) if i else np.ones(3)
        return np.concatenate([np.array([54.0, 60.0, 75.0]) * scale, np.array([18.0, 21.0, 30.0]) * scale])

    return prob, [Pl, Ql], sample, p
