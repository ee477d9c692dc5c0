"""BASELINE.json configs at their STATED sizes on the MI355X, each against an answer that does not
come from this repository's solver core: C3 (n = 1e4, m = 1e3) against LAPACK's solution of the same
KKT system, C5 (8 192 instances) against the reference-held property of the localization test (the
noise-free position is recovered, test_nlp_solvers.py:175-189) on EVERY instance and against the CPU
oracle on 256 of them.  C4 at n = 1e5 and C2 at n = 1e5 are in test_gpu_parity.py."""
import numpy as np
import pytest

import batch_problems as bp
from dnlp_amd.batch import ParametricBatch, arrays_with_data

pytestmark = pytest.mark.gpu

BATCH = 8192
CHECKED = 256


def test_c3_full_size_against_lapack(gpu_required):
    """C3: min 1/2 x'Qx + c'x s.t. Ax = b, n = 1e4, m = 1e3, Q = G'G/n + I.  One Newton
    step is exact; primal AND dual solution against numpy.linalg.solve (LAPACK dgesv) of the closed-form
    KKT system, plus the KKT residual itself (a size-independent certificate)."""
    import dnlp_amd as cp
    n, m = 10000, 1000
    rng = np.random.default_rng(0)
    Gm = rng.standard_normal((n, n))
    Q = Gm.T @ Gm / n + np.eye(n)
    del Gm
    c = rng.standard_normal(n)
    A = rng.standard_normal((m, n))
    b = A @ rng.standard_normal(n)
    x = cp.Variable(n)
    prob = cp.Problem(cp.Minimize(0.5 * cp.quad_form(x, Q) + c @ x), [A @ x == b])
    chain = prob._build_chain(None)
    data, inv = chain.apply(prob)
    assert len(data["x0"]) == n and len(data["cl"]) == m           # KKT order 11 000, dense
    info = chain.solver.solve_via_data(data, True, False, {})
    assert info["status"] == 0 and info["iterations"] <= 2
    xs, lam = info["x"], info["mult_g"]
    # certificate: stationarity and feasibility of the returned pair
    assert np.max(np.abs(Q @ xs + c + A.T @ lam)) <= 1e-8 * max(1.0, np.max(np.abs(c)))
    assert np.max(np.abs(A @ xs - b)) <= 1e-8 * max(1.0, np.max(np.abs(b)))
    # closed form through LAPACK
    K = np.block([[Q, A.T], [A, np.zeros((m, m))]])
    sol = np.linalg.solve(K, np.concatenate([-c, b]))
    np.testing.assert_allclose(xs, sol[:n], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(lam, sol[n:], rtol=1e-6, atol=1e-8)          # duals: closed form
    assert abs(info["obj_val"] - (0.5 * sol[:n] @ Q @ sol[:n] + c @ sol[:n])) <= 1e-9 * abs(info["obj_val"])


def _oracle(arrays, opts=None, batch_defaults=False):
    from dnlp_amd.nlp_solver import HIPNLP
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    orc = OracleProblem(serialize(arrays))
    options = dict(HIPNLP.DEFAULT_OPTIONS)
    if batch_defaults:
        options["lazy_dense_fallback"] = "yes"        # dnlp_amd.batch._device_handle's setting
    options.update(opts or {})
    for k, v in options.items():
        orc.set_option(k, v)
    return orc.solve(arrays["x0"])


def _kkt_residuals(arrays, x, mult_g, zl, zu):
    """Stationarity / feasibility / complementarity of a returned primal-dual point, evaluated with the CPU
    oracle's derivative oracles on the instance's own tape (an instance-level certificate that does not
    depend on the solver's path): returns (|grad f + J'y - zL + zU|_inf, constraint violation, bound
    violation, complementarity)."""
    import scipy.sparse as sp
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    o = OracleProblem(serialize(arrays))
    n, m = o.n, o.m
    g = o.eval_grad_f(x)
    r, c = o.jac_structure()
    J = sp.csr_matrix((o.eval_jac_g(x), (r, c)), shape=(m, n))
    cons = o.eval_g(x)
    o.close()
    lb, ub, cl, cu = (np.asarray(arrays[k], float).ravel() for k in ("lb", "ub", "cl", "cu"))
    free = lb < ub               # fixed variables carry no multiplier (IPOPT's make_parameter treatment, ours too)
    stat = np.max(np.abs((g + J.T @ mult_g - zl + zu)[free]), initial=0.0)
    viol = max(np.max(np.maximum(cl - cons, 0.0), initial=0.0), np.max(np.maximum(cons - cu, 0.0), initial=0.0))
    bviol = max(np.max(np.maximum(lb - x, 0.0), initial=0.0), np.max(np.maximum(x - ub, 0.0), initial=0.0))
    fin_l, fin_u = lb > -1e19, ub < 1e19
    comp = max(np.max(np.abs(zl[fin_l] * (x[fin_l] - lb[fin_l])), initial=0.0),
               np.max(np.abs(zu[fin_u] * (ub[fin_u] - x[fin_u])), initial=0.0))
    ineq = cl < cu
    act_l = np.where(cl > -1e19, cons - cl, 0.0)
    act_u = np.where(cu < 1e19, cu - cons, 0.0)
    lam_minus, lam_plus = np.maximum(-mult_g, 0.0), np.maximum(mult_g, 0.0)
    comp = max(comp, np.max(np.abs(lam_minus[ineq] * act_l[ineq]), initial=0.0),
               np.max(np.abs(lam_plus[ineq] * act_u[ineq]), initial=0.0))
    return stat, viol, bviol, comp


def test_c5_8192_localization_instances(gpu_required):
    """C5 at its stated size: 8 192 localization instances in ONE launch.  Reference-held answer
    (test_nlp_solvers.py:175-189: noise-free ranges -> the true position is recovered, objective 0)
    checked on every instance; 256 instances compared field by field with the CPU oracle."""
    prob, params, sample, xvar = bp.template_localization()
    pb = ParametricBatch(prob, params)
    thetas = np.stack([sample(i) for i in range(BATCH)])
    res = pb.solve(thetas, want_duals=True)
    assert res.status.shape == (BATCH,)
    ok = res.status == 0
    assert ok.sum() >= BATCH - 2                       # (r01: 65 535 of 65 536 optimal)
    x_true = np.stack([np.random.default_rng(i).uniform(-3, 3, 2) for i in range(BATCH)])
    xs = res.value_of(xvar)
    err = np.linalg.norm(xs - x_true, axis=1)
    # the range-fitting problem is non-convex: about 1 % of the random instances have a second local
    # minimum that the start (0, 0) falls into (the CPU oracle and the reference's IPOPT path do too);
    # everywhere else the residual is zero and the position is the true one
    solved = ok & (res.obj_val <= 1e-9)
    assert solved.sum() >= int(0.97 * BATCH)
    assert np.sum(err[solved] <= 1e-5) >= solved.sum() - 8      # (nearly collinear anchors: mirrored twin)
    mat = pb.data(thetas)
    same_iters = 0
    for i in range(0, BATCH, BATCH // CHECKED):
        oi = _oracle(arrays_with_data(pb.arrays0, mat[i]))
        assert res.status[i] == oi["status"]
        if oi["status"] != 0:
            continue
        assert abs(res.raw["obj_val"][i] - oi["obj_val"]) <= 1e-6 * max(1.0, abs(oi["obj_val"]))
        np.testing.assert_allclose(res.x[i], oi["x"], rtol=1e-5, atol=1e-6)
        same_iters += int(res.iterations[i] == oi["iterations"])
    assert same_iters >= int(0.9 * CHECKED)


def test_c5_8192_circle_packing_instances(gpu_required):
    """8 192 circle-packing instances (non-convex): every instance must end at a feasible KKT point —
    no overlap, evaluated here from the returned centres and the instance's own radii — and 256 are
    compared with the CPU oracle (a few may legitimately sit in another local optimum)."""
    n = 4
    prob, params, sample, cvar = bp.template_circle_packing(n)
    pb = ParametricBatch(prob, params)
    thetas = np.stack([sample(i) for i in range(BATCH)])
    res = pb.solve(thetas, want_duals=True)
    ok = res.status == 0
    assert ok.sum() >= int(0.995 * BATCH)
    centers = res.value_of(cvar)                        # (B, 2, n)
    npairs = n * (n - 1) // 2
    radius = thetas[:, npairs:]
    k = 0
    for a in range(n - 1):
        for b in range(a + 1, n):
            d2 = np.sum((centers[:, :, a] - centers[:, :, b]) ** 2, axis=1)
            assert np.all(d2[ok] >= thetas[ok, k] * (1 - 1e-7))           # (r_a + r_b)^2: no overlap
            k += 1
    half_side = np.max(np.max(np.abs(centers), axis=1) + radius, axis=1)
    np.testing.assert_allclose(res.obj_val[ok], half_side[ok], rtol=1e-6)  # objective = enclosing square
    mat = pb.data(thetas)
    other = loose = 0
    for i in range(0, BATCH, BATCH // CHECKED):
        oi = _oracle(arrays_with_data(pb.arrays0, mat[i]))
        if oi["status"] != 0 or res.status[i] != 0:
            assert res.status[i] == oi["status"] or res.status[i] == 0 or oi["status"] == 0
            other += int(res.status[i] != oi["status"])
            continue
        if abs(res.raw["obj_val"][i] - oi["obj_val"]) > 1e-6 * max(1.0, abs(oi["obj_val"])):
            other += 1
            continue
        # same optimum; a circle that touches nothing can slide (the optimum is a face), and with the barrier
        # going down to IPOPT's 1e-11 the two builds' rounding picks different points of it
        assert np.max(np.abs(res.x[i] - oi["x"])) <= 0.1
        loose += int(not np.allclose(res.x[i], oi["x"], rtol=1e-5, atol=1e-5))
    assert other <= CHECKED // 16 and loose <= CHECKED // 4


MEMBER_BATCH = 1024           # 8 x 1024; each remaining member at one GPU's share
MEMBER_CHECKED = 64


def test_c5_1024_circle_packing_n10_instances(gpu_required):
    """C5's circle-packing member at its STATED size (n = 10: N = 121, m = 185).  Instance 0 is
    the notebook's problem in the notebook's formulation from the start the notebook WRITES
    (circle_packing.ipynb:68-79).  The published 7.22863 is one local optimum of a non-convex problem and was
    reached from the permuted start the reference actually hands IPOPT (nlp_solver.py:84,116,163 vs :200;
    tests/test_paper_examples.py reproduces that start and ends at 7.21758); from the start as written both
    the host build and the device end at the neighbouring KKT point 7.40240 (2.4 % above), which is what is
    asserted here together with its certificate.  Every instance: no overlap and objective = half side of the
    enclosing square, from the returned centres and the instance's radii; 64 instances against the host
    build of the algorithm (the oracle library), a KKT certificate for each from the oracle's evaluators."""
    n = 10
    prob, params, sample, cvar = bp.template_circle_packing(n)
    pb = ParametricBatch(prob, params)
    assert int(pb.arrays0["dims"][0]) == 121 and int(pb.arrays0["dims"][1]) == 185
    thetas = np.stack([sample(i) for i in range(MEMBER_BATCH)])
    res = pb.solve(thetas, want_duals=True)
    ok = res.status == 0
    assert ok.sum() >= int(0.98 * MEMBER_BATCH), np.unique(res.status, return_counts=True)
    assert res.status[0] == 0 and abs(res.obj_val[0] - 7.2286302188441365) <= 0.03 * 7.2286302188441365
    centers = np.transpose(res.value_of(cvar), (0, 2, 1))          # (B, 2, n) from the notebook's (n, 2)
    npairs = n * (n - 1) // 2
    radius = thetas[:, npairs:]
    k = 0
    for a in range(n - 1):
        for b in range(a + 1, n):
            d2 = np.sum((centers[:, :, a] - centers[:, :, b]) ** 2, axis=1)
            assert np.all(d2[ok] >= thetas[ok, k] * (1 - 1e-7))
            k += 1
    half_side = np.max(np.max(np.abs(centers), axis=1) + radius, axis=1)
    np.testing.assert_allclose(res.obj_val[ok], half_side[ok], rtol=1e-6)
    mat = pb.data(thetas)
    other = 0
    for i in range(0, MEMBER_BATCH, MEMBER_BATCH // MEMBER_CHECKED):
        a = arrays_with_data(pb.arrays0, mat[i])
        if res.status[i] == 0:
            stat, viol, bviol, comp = _kkt_residuals(a, res.x[i], res.raw["mult_g"][i], res.raw["mult_x_L"][i],
                                                     res.raw["mult_x_U"][i])
            assert stat <= 1e-5 and viol <= 1e-6 and bviol <= 1e-9 and comp <= 1e-5, (i, stat, viol, bviol, comp)
        oi = _oracle(a, batch_defaults=True)
        if oi["status"] != res.status[i] or abs(res.raw["obj_val"][i] - oi["obj_val"]) > 1e-6 * abs(oi["obj_val"]):
            other += 1                      # another local optimum / rung: must still be a certified KKT point
            continue
        assert np.max(np.abs(res.x[i] - oi["x"])) <= 0.5      # same optimum; loose circles may sit elsewhere on the face
    assert other <= MEMBER_CHECKED // 8
    pb.close()


def test_c5_1024_power_flow_instances(gpu_required):
    """C5's AC power-flow member (IEEE 9-bus, N = 867 with the load rows, KKT order 1723; loads scaled per
    instance by default_rng(i) in [0.8, 1.2]).  Instance 0 carries the notebook's loads: published objective
    3087.84222847 (power_flow.ipynb:101).  Every optimal instance: the AC power-balance equations, the
    load rows, the voltage / generation bounds and objective = generation cost, all recomputed here from the
    returned v, theta, p, q; 64 instances against the host build + KKT certificates."""
    from paper_examples import ieee9_admittance
    prob, params, sample, pvar = bp.template_power_flow()
    pb = ParametricBatch(prob, params)
    thetas = np.stack([sample(i) for i in range(MEMBER_BATCH)])
    opts = {"least_square_init_duals": "no"}
    res = pb.solve(thetas, want_duals=True, **opts)
    ok = res.status == 0
    # 6.6 % of the perturbed-load instances end "locally infeasible" from the flat start.  The launch is bitwise
    # repeatable (tests/test_determinism.py), so the count is a fixed number of this build's launch plan: 956 optimal +
    # 68 infeasible at 1024 instances (round 5, four wavefronts per instance); a change of it is a change of the
    # algorithm or of a summation order and has to be looked at, not absorbed by a percentage
    # Round 6: a launch of 1024 takes the workgroup-per-instance kernel compiled for the template (csrc/wave_wg_kernel.h), whose
    # long sums go through the wavefront's reduction tree: 953 optimal + 69 infeasible + 2 "error in step computation" — the
    # instances that move sit at the edge of the inertia correction / of the infeasibility verdict (over four batches of 1024
    # the two kernels' counts differ by a handful either way: tools/pf_status_hist.py, profiles/r06_power_flow_status_hist.txt;
    # with the long sums in the host lane's order the same kernel gave 955 + 68 + 1).  The generic kernel's count stays pinned
    # for the launches that take it.
    pin = (953, 69) if res.raw["launch"].get("wave_spec") else (956, 68)
    assert MEMBER_BATCH != 1024 or (int(ok.sum()), int((res.status == 2).sum())) == pin, np.unique(res.status, return_counts=True)
    assert ok.sum() >= int(0.90 * MEMBER_BATCH), np.unique(res.status, return_counts=True)
    assert res.status[0] == 0 and abs(res.obj_val[0] - 3.0878422284732592e+03) <= 1e-6 * 3.0878e3
    vars_by_shape = {}
    for var in prob.variables():
        vars_by_shape.setdefault(tuple(var.shape), []).append(var)
    (theta, v) = vars_by_shape[(9, 1)]           # creation order: theta, then v
    (p, q) = vars_by_shape[(9,)]
    G, B = ieee9_admittance()
    th, vv = res.value_of(theta)[:, :, 0], res.value_of(v)[:, :, 0]
    pp, qq = res.value_of(p), res.value_of(q)
    dth = th[:, :, None] - th[:, None, :]
    vv2 = vv[:, :, None] * vv[:, None, :]
    p_calc = np.sum(vv2 * (G * np.cos(dth) + B * np.sin(dth)), axis=2)
    q_calc = np.sum(vv2 * (G * np.sin(dth) - B * np.cos(dth)), axis=2)
    scale = 300.0
    assert np.max(np.abs(p_calc[ok] - pp[ok])) <= 1e-5 * scale and np.max(np.abs(q_calc[ok] - qq[ok])) <= 1e-5 * scale
    load = [4, 6, 8]
    assert np.max(np.abs(pp[ok][:, load] + thetas[ok, :3])) <= 1e-6 * scale
    assert np.max(np.abs(qq[ok][:, load] + thetas[ok, 3:])) <= 1e-6 * scale
    assert np.all(vv[ok] >= 0.9 - 1e-8) and np.all(vv[ok] <= 1.1 + 1e-8) and np.max(np.abs(th[ok][:, 0])) <= 1e-8
    cost = (0.11 * pp[:, 0] ** 2 + 5 * pp[:, 0] + 150 + 0.085 * pp[:, 1] ** 2 + 1.2 * pp[:, 1] + 600
            + 0.1225 * pp[:, 2] ** 2 + pp[:, 2] + 335)
    np.testing.assert_allclose(res.obj_val[ok], cost[ok], rtol=1e-8)
    mat = pb.data(thetas)
    differ = 0
    for i in range(0, MEMBER_BATCH, MEMBER_BATCH // MEMBER_CHECKED):
        a = arrays_with_data(pb.arrays0, mat[i])
        if res.status[i] == 0:
            stat, viol, bviol, comp = _kkt_residuals(a, res.x[i], res.raw["mult_g"][i], res.raw["mult_x_L"][i],
                                                     res.raw["mult_x_U"][i])
            assert stat <= 1e-4 and viol <= 1e-5 and bviol <= 1e-9 and comp <= 1e-4, (i, stat, viol, bviol, comp)
        oi = _oracle(a, opts, batch_defaults=True)
        if oi["status"] != res.status[i]:
            differ += 1
            continue
        if oi["status"] == 0:
            assert abs(res.raw["obj_val"][i] - oi["obj_val"]) <= 1e-6 * abs(oi["obj_val"])
    assert differ <= MEMBER_CHECKED // 16
    pb.close()


def test_c5_1024_path_planning_instances(gpu_required):
    """C5's path-planning member (n = 50 segments: N = 715 free variables, KKT order 1636; obstacle centres
    and radii perturbed per instance).  Instance 0 is the notebook's problem: published 13.1368823193
    (path_planning.ipynb:103).  Every optimal instance: end points, every way-point outside every obstacle,
    every segment no longer than L / n, objective = L; 64 against the host build + KKT certificates."""
    prob, params, sample, xvar = bp.template_path_planning()
    pb = ParametricBatch(prob, params)
    thetas = np.stack([sample(i) for i in range(MEMBER_BATCH)])
    res = pb.solve(thetas, want_duals=True)
    ok = res.status == 0
    # (the exact count of this build's launch plan, as for power flow: 1006 optimal + 18 "locally infeasible" at 1024)
    assert MEMBER_BATCH != 1024 or (int(ok.sum()), int((res.status == 2).sum())) == (1006, 18), np.unique(res.status, return_counts=True)
    assert ok.sum() >= int(0.97 * MEMBER_BATCH), np.unique(res.status, return_counts=True)
    assert res.status[0] == 0 and abs(res.obj_val[0] - 1.3136882319337619e+01) <= 1e-6 * 13.14
    X = res.value_of(xvar)                                  # (B, 2, 51)
    nseg = X.shape[2] - 1
    assert np.max(np.abs(X[ok][:, :, 0] - 1.25)) <= 1e-7 and np.max(np.abs(X[ok][:, :, nseg] - 10.0)) <= 1e-7
    centres = thetas[:, :10].reshape(MEMBER_BATCH, 2, 5)   # F-order flatten of the (5, 2) parameter
    r2 = thetas[:, 10:]
    d2 = np.sum((X[:, :, :, None] - centres[:, :, None, :]) ** 2, axis=1)     # (B, 51, 5)
    assert np.all(d2[ok] >= r2[ok][:, None, :] * (1 - 1e-6))
    seg2 = np.sum((X[:, :, 1:] - X[:, :, :-1]) ** 2, axis=1)
    L = res.obj_val
    assert np.all(seg2[ok] <= (L[ok, None] / nseg) ** 2 * (1 + 1e-6) + 1e-9)
    assert np.all(L[ok] >= np.sqrt(2.0) * 8.75 - 1e-6)       # never shorter than the straight line
    mat = pb.data(thetas)
    differ = 0
    for i in range(0, MEMBER_BATCH, MEMBER_BATCH // MEMBER_CHECKED):
        a = arrays_with_data(pb.arrays0, mat[i])
        if res.status[i] == 0:
            stat, viol, bviol, comp = _kkt_residuals(a, res.x[i], res.raw["mult_g"][i], res.raw["mult_x_L"][i],
                                                     res.raw["mult_x_U"][i])
            assert stat <= 1e-5 and viol <= 1e-6 and bviol <= 1e-9 and comp <= 1e-5, (i, stat, viol, bviol, comp)
        oi = _oracle(a, batch_defaults=True)
        if oi["status"] != res.status[i] or abs(res.raw["obj_val"][i] - oi["obj_val"]) > 1e-6 * abs(oi["obj_val"]):
            differ += 1
            continue
        np.testing.assert_allclose(res.x[i], oi["x"], rtol=1e-4, atol=1e-4)
    assert differ <= MEMBER_CHECKED // 8
    pb.close()
