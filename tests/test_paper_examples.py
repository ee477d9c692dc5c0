"""The paper's example set at notebook size: dimensions, iteration counts and optimal
objectives printed by the IPOPT runs in examples/nlp_examples/*.ipynb (tests/paper_examples.py
holds the numbers and their log locations).  north_star: optima within 1e-6 relative."""
import numpy as np
import pytest

from golden_util import build_canonical, load_golden
from paper_examples import PAPER, PUBLISHED

REL_TOL = 1e-6


@pytest.mark.parametrize("name", sorted(PUBLISHED))
def test_dimensions_match_published_ipopt_log(name):
    """IPOPT's header counts free variables, equality rows, inequality rows and non-zeros."""
    pub = PUBLISHED[name]
    data, _ = build_canonical(name)
    lb, ub, cl, cu = (np.asarray(data[k]) for k in ("lb", "ub", "cl", "cu"))
    assert int(np.sum(lb != ub)) == pub["n_free"]
    assert int(np.sum(cl == cu)) == pub["n_eq"]
    assert int(np.sum(cl != cu)) == pub["n_ineq"]
    t = data["tape"]
    if "nnz_hess" in pub:
        assert len(t.hess_rows) == pub["nnz_hess"]
    if "nnz_jac" in pub:
        assert len(t.jac_rows) == pub["nnz_jac"]


def _published_start(name, data):
    """The start point of the published run.  The reference hands IPOPT `x0` in the variable order
    of the pre-lowering problem while the oracles use the lowered order (nlp_solver.py:84,163 vs
    :200); for circle packing the two differ, so the notebook's log belongs to the start the golden
    record holds verbatim (flat, as cyipopt received it) — not to the user's start, which this
    build places on the right variables.  Bounds are identical under that permutation."""
    if name == "nb_circle_packing":
        g = load_golden(name)
        assert np.array_equal(g["lb"], data["lb"]) and np.array_equal(g["ub"], data["ub"])
        return g["x0"]
    return data["x0"]


def _check(name, info):
    pub = PUBLISHED[name]
    assert info["status"] == 0
    ref = pub["objective"]
    if name == "nb_phase_retrieval":
        # exact recovery: the optimum is 0 and the log's 3.9e-9 is the final barrier residue — 384 active
        # inequality rows x the adaptive strategy's barrier floor 1e-11, which this build shares since round 3
        # (rounds 1-2 floored mu at tol / 11 and stopped at 3.5e-6 = 384 x 9e-9)
        assert abs(info["obj_val"]) <= 1e-7
    elif name == "nb_circle_packing":
        # non-convex: the log's value is one local optimum; ours must be a KKT point no worse
        assert info["obj_val"] <= ref * (1 + REL_TOL)
    else:
        assert abs(info["obj_val"] - ref) <= REL_TOL * abs(ref)
    # interior-point work of the same order as IPOPT's on the same problem
    assert info["iterations"] <= 2 * pub["iters"] + 10


@pytest.mark.parametrize("name", ["nb_path_planning", "nb_power_flow", "nb_circle_packing"])
def test_cpu_oracle_reaches_published_optimum(name):
    from dnlp_amd.nlp_solver import HIPNLP
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    data, _ = build_canonical(name)
    h = OracleProblem(serialize(data["tape_arrays"]))
    opts = dict(HIPNLP.DEFAULT_OPTIONS)
    opts.update(PUBLISHED[name].get("options", {}))
    for k, v in opts.items():
        h.set_option(k, v)
    _check(name, h.solve(_published_start(name, data)))
    if name == "nb_circle_packing":
        # from the user's own start (correct variable order) the non-convex problem ends at another
        # KKT point; it must be one (status 0) and feasible
        h2 = OracleProblem(serialize(data["tape_arrays"]))
        for k, v in opts.items():
            h2.set_option(k, v)
        own = h2.solve(data["x0"])
        assert own["status"] == 0
        g = own["g"]
        assert np.all(g >= data["cl"] - 1e-6) and np.all(g <= data["cu"] + 1e-6)


def test_cpu_power_flow_iteration_count_equals_ipopt():
    """The 9-bus OPF has a unique optimum and a benign path: the restated algorithm takes the
    same 15 iterations as IPOPT 3.14.17 (power_flow.ipynb:101) and lands within 1e-10."""
    from dnlp_amd.nlp_solver import HIPNLP
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    data, _ = build_canonical("nb_power_flow")
    h = OracleProblem(serialize(data["tape_arrays"]))
    opts = dict(HIPNLP.DEFAULT_OPTIONS)
    opts["least_square_init_duals"] = "no"
    opts["linear_solver"] = "dense"            # Bunch-Kaufman pivoting, the analogue of IPOPT's MA27 / MUMPS
    opts["kkt_optimistic_min_n"] = 1 << 30     # ... from the first factorisation on (no unpivoted attempt)
    for k, v in opts.items():
        h.set_option(k, v)
    info = h.solve(data["x0"])
    assert info["iterations"] == 15
    assert abs(info["obj_val"] - 3.0878422284732592e+03) <= 1e-9 * 3.0878422284732592e+03
    # the static-pivot sparse factorisation (the default for this pattern) regularises a few
    # singular 2x2 blocks at the start (zero Jacobian coefficients at x0): same optimum, 19 iterations
    h2 = OracleProblem(serialize(data["tape_arrays"]))
    opts["linear_solver"] = "sparse"
    for k, v in opts.items():
        h2.set_option(k, v)
    assert h2.kkt_info()["sparse"]
    info2 = h2.solve(data["x0"])
    assert info2["status"] == 0 and info2["iterations"] <= 25
    assert abs(info2["obj_val"] - info["obj_val"]) <= 1e-8 * abs(info["obj_val"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(set(PUBLISHED) - {"nb_localization", "nb_portfolio_construction"}))
def test_device_reaches_published_optimum(name, gpu_required):
    from dnlp_amd import _capi
    from dnlp_amd.nlp_solver import HIPNLP
    from dnlp_amd.tape import serialize
    data, _ = build_canonical(name)
    dev = _capi.DeviceProblem(serialize(data["tape_arrays"]), data["tape"], device=0)
    opts = dict(HIPNLP.DEFAULT_OPTIONS)
    opts.update(PUBLISHED[name].get("options", {}))
    for k, v in opts.items():
        dev.set_option(k, v)
    info = dev.solve(_published_start(name, data))
    assert info["status"] == 0, dev.log()
    _check(name, info)
    dev.close()


# ---- portfolio_construction.ipynb: published dimensions, synthetic returns ----------------------------------
def _solve_portfolio(make_handle):
    """The eighth example of the paper at its published dimensions (1 282 / 959 / 15, nnz 103 995 + 1 306 /
    51 359: test_dimensions_match_published_ipopt_log) on seeded synthetic returns.  The published objective
    belongs to the real price file, so the known answers are properties: a KKT point (certificate from the
    returned multipliers), a fully invested long-only portfolio, t1 = Sigma w and t2 = w'Sigma w at the
    solution, and every sector's risk contribution within 10 % of its budget (the constraint the example is
    about; the notebook prints [0.33, 0.225, 0.18, 0.165, 0.1] for b = [0.3, 0.25, 0.2, 0.15, 0.1])."""
    import dnlp_amd as cp
    from dnlp_amd.dnlp2smooth import Dnlp2Smooth
    from dnlp_amd.nlp_solver import HIPNLP, build_nlp_data
    from dnlp_amd.tape import serialize
    from paper_examples import nb_portfolio_construction, portfolio_data
    from test_full_size_configs import _kkt_residuals
    p = nb_portfolio_construction(cp)
    pmin = cp.Problem(cp.Minimize(-p.objective.expr), p.constraints)
    smooth, _ = Dnlp2Smooth().apply(pmin)
    data, inv = build_nlp_data(smooth, user_variables=p.variables())
    h = make_handle(serialize(data["tape_arrays"]), data["tape"])
    for k, v in HIPNLP.DEFAULT_OPTIONS.items():
        h.set_option(k, v)
    info = h.solve(data["x0"])
    assert info["status"] == 0, info["status"]
    assert info["iterations"] <= 2 * PUBLISHED["nb_portfolio_construction"]["iters"] + 10
    stat, viol, bviol, comp = _kkt_residuals(data["tape_arrays"], info["x"], info["mult_g"], info["mult_x_L"],
                                             info["mult_x_U"])
    assert stat <= 1e-6 and viol <= 1e-8 and bviol <= 1e-12 and comp <= 1e-7, (stat, viol, bviol, comp)
    Sigma, mu, groups = portfolio_data()
    n = Sigma.shape[0]
    by_size = {}
    for v in p.variables():
        by_size.setdefault(v.size, []).append(v)
    w_var, t1_var = by_size[n]                       # creation order: w, then t1
    t2_var = by_size[1][0]
    val = lambda v: info["x"][inv.var_offsets[v.id]:inv.var_offsets[v.id] + v.size]
    w, t1, t2 = val(w_var), val(t1_var), float(val(t2_var)[0])
    assert abs(np.sum(w) - 1.0) <= 1e-9 and np.min(w) >= -1e-12
    np.testing.assert_allclose(t1, Sigma @ w, rtol=1e-7, atol=1e-12)
    # (the defining row of t2 is a constraint like any other: satisfied to the solver's tolerance — 1e-12 … 6e-12 on the
    #  device from run to run — not to a relative 1e-9 of a value of 1.4e-4)
    assert abs(t2 - w @ Sigma @ w) <= 1e-9 * t2 + 1e-10
    assert abs(-info["obj_val"] - (mu @ w - t2)) <= 1e-10
    rc = w * (Sigma @ w)
    rc = np.array([np.sum(rc[g]) for g in groups]) / np.sum(rc)
    b = np.array([0.3, 0.25, 0.20, 0.15, 0.10])
    assert np.all(np.abs(rc - b) <= 0.1 * b * (1 + 1e-6)), rc
    return info


def test_cpu_portfolio_construction_at_published_dimensions():
    from oracle.oracle_capi import OracleProblem
    _solve_portfolio(lambda blob, tape: OracleProblem(blob))


@pytest.mark.gpu
def test_device_portfolio_construction_at_published_dimensions(gpu_required):
    from dnlp_amd import _capi
    _solve_portfolio(lambda blob, tape: _capi.DeviceProblem(blob, tape, device=0))


# ---- examples without a published IPOPT log: NMF.ipynb, sparse_recovery.ipynb ------------------------
def _check_sparse_recovery(x):
    """RECOVERY_TOL of the notebook: ||x - x0|| <= 1e-2 ||x0|| (here the recovery is exact)."""
    from paper_examples import sparse_recovery_data
    A, y, x0 = sparse_recovery_data()
    assert np.linalg.norm(x - x0) <= 1e-2 * np.linalg.norm(x0)
    assert np.linalg.norm(A @ x - y) <= 1e-6 * np.linalg.norm(y)


def _check_nmf(prob, X, Y, n_samples):
    """No published objective exists.  Known properties: a KKT point (status optimal) with X, Y >= 0
    whose rank-3 reconstruction denoises — it is closer to the noise-free images than the data are,
    and no rank-3 nonnegative factorisation can beat the unconstrained rank-3 SVD."""
    from paper_examples import nmf_data
    A_true, A = nmf_data(n_samples)
    R = X @ Y
    assert X.min() >= -1e-8 and Y.min() >= -1e-8
    assert abs(prob.value - np.sum((A - R) ** 2)) <= 1e-6 * prob.value
    sv = np.linalg.svd(A, compute_uv=False)
    assert prob.value >= np.sum(sv[3:] ** 2) * (1 - 1e-9)              # Eckart-Young lower bound
    assert prob.value <= 1.05 * np.sum((A - A_true) ** 2)             # fits down to the noise level
    # (12 images determine the three shapes less sharply than the notebook's 100)
    assert np.linalg.norm(R - A_true) < (0.5 if n_samples >= 100 else 0.8) * np.linalg.norm(A - A_true)


def _solve_sparse_recovery(make_handle):
    """The notebook solves this example with Knitro, not IPOPT: at the sparse solution sqrt|x| is not
    differentiable, so an interior-point run ends AT the recovered point with either the optimal status or
    IPOPT's tiny-step status (3) — which of the two depends on rounding (1e-10 perturbations of the start
    flip it, on the CPU build as well).  The known answer is the recovery property, checked on the point
    the solver returns through the C ABI."""
    import dnlp_amd as cp
    from dnlp_amd.dnlp2smooth import Dnlp2Smooth
    from dnlp_amd.nlp_solver import HIPNLP, build_nlp_data
    from dnlp_amd.tape import serialize
    from paper_examples import nb_sparse_recovery
    p = nb_sparse_recovery(cp)
    smooth, _ = Dnlp2Smooth().apply(p)
    data, inv = build_nlp_data(smooth)
    h = make_handle(serialize(data["tape_arrays"]), data["tape"])
    for k, v in HIPNLP.DEFAULT_OPTIONS.items():
        h.set_option(k, v)
    info = h.solve(data["x0"])
    assert info["status"] in (0, 1, 3), info["status"]
    off = inv.var_offsets[p.variables()[0].id]
    _check_sparse_recovery(info["x"][off:off + 100])


def test_cpu_sparse_recovery_and_small_nmf():
    import dnlp_amd as cp
    from oracle.oracle_capi import OracleProblem
    from oracle_frontend import oracle_engine
    from paper_examples import nb_nmf
    _solve_sparse_recovery(lambda blob, tape: OracleProblem(blob))
    with oracle_engine():
        p = nb_nmf(cp, 12)
        p.solve(nlp=True)
        assert p.status == cp.OPTIMAL
        X, Y = (v.value for v in p.variables())
        _check_nmf(p, X, Y, 12)


@pytest.mark.gpu
def test_device_sparse_recovery(gpu_required):
    from dnlp_amd import _capi
    _solve_sparse_recovery(lambda blob, tape: _capi.DeviceProblem(blob, tape, device=0))


@pytest.mark.gpu
def test_device_nmf_at_notebook_size(gpu_required):
    """NMF.ipynb at its own size: 100 images of 20 x 20, k = 3 -> N = 41 500, m = 40 000, the bilinear
    Var @ Var product; sparse static-pattern KKT (6e7 update triples)."""
    import dnlp_amd as cp
    from paper_examples import nb_nmf
    p = nb_nmf(cp, 100)
    p.solve(nlp=True)
    assert p.status == cp.OPTIMAL
    X, Y = (v.value for v in p.variables())
    assert X.shape == (100, 3) and Y.shape == (3, 400)
    _check_nmf(p, X, Y, 100)


# ---- run-to-run reproducibility of the host-driven loop ---------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["nb_portfolio_construction", "nb_sparse_recovery"])
def test_device_host_driven_solve_repeats_bit_for_bit(name, gpu_required):
    """IPOPT on one thread repeats a run bit for bit; so does the host-driven loop since its sparse products
    (J v, J^T v, H v) accumulate in a fixed order (csrc/exec_hip.h coo_rows_kernel).  With atomic products these two
    non-convex examples ended after 22 to 72 resp. 66 to 177 iterations from run to run
    (profiles/r03_determinism.txt).  Two fresh handles, two solves each: same iteration count, same bits."""
    import dnlp_amd as cp
    from paper_examples import PAPER, PAPER_LARGE
    make = dict(PAPER)
    make.update(PAPER_LARGE)
    runs = []
    for fresh in range(2):
        prob = make[name](cp)
        chain = prob._build_chain(None)
        data, _ = chain.apply(prob)
        for rep in range(2):
            info = chain.solver.solve_via_data(dict(data), True, False, {})
            runs.append((int(info["iterations"]), int(info["status"]), float(info["obj_val"]), np.array(info["x"])))
    assert all(r[1] == 0 for r in runs)
    assert len({r[0] for r in runs}) == 1, [r[0] for r in runs]
    assert all(r[2] == runs[0][2] for r in runs), [r[2] for r in runs]
    assert all(np.array_equal(r[3], runs[0][3]) for r in runs)


def test_published_localization_log_belongs_to_other_data():
    """tests/paper_examples.py pins only the dimensions of the localization log; tools/localization_log_check.py is the
    evidence: with the data the notebook's committed cells generate, the minimiser the notebook prints is not an optimum
    (17.37 against this solver's 14.00) and the iteration-0 infeasibility is 11.6, not the log's 7.81."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "localization_log_check.py")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "max |c| 1.161e+01" in out.stdout
    assert "at the log's minimiser 17.37207" in out.stdout and "at this solve's minimiser 14.00431" in out.stdout
    assert "minimiser is NOT the printed one; objective does not match the log's" in out.stdout
