"""Static-pattern sparse LDL^T of the KKT system (csrc/sparse_plan.h, sparse_ldl.h) against the
dense Bunch-Kaufman path: same optima on every golden problem, the plan's invariants, and the
factorisation itself against numpy on assembled matrices."""
import numpy as np
import pytest

from golden_util import build_canonical
from problem_zoo import GOLDEN_ZOO


def _solve(name, linear_solver, **extra):
    from dnlp_amd.nlp_solver import HIPNLP
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    from paper_examples import PUBLISHED
    data, _ = build_canonical(name)
    h = OracleProblem(serialize(data["tape_arrays"]))
    opts = dict(HIPNLP.DEFAULT_OPTIONS)
    opts.update(PUBLISHED.get(name, {}).get("options", {}))
    opts.update(extra)
    opts["linear_solver"] = linear_solver
    for k, v in opts.items():
        h.set_option(k, v)
    return h, h.solve(data["x0"])


def data_bounds(name):
    data, _ = build_canonical(name)
    return np.asarray(data["cl"], float), np.asarray(data["cu"], float)


@pytest.mark.parametrize("name", sorted(GOLDEN_ZOO))
def test_sparse_and_dense_kkt_reach_the_same_optimum(name):
    if name in ("mle", "nb_phase_retrieval", "nb_path_planning", "nb_power_flow", "nb_nmf_small"):
        pytest.skip("dense host factorisation of this order takes seconds to minutes; covered by test_paper_examples")
    if name == "nb_sparse_recovery":
        pytest.skip("dense measurement matrix (80 x 100 Jacobian block): the automatic choice is the dense "
                    "Bunch-Kaufman path, static pivots are not meant for this pattern")
    if name in ("sphere60", "nb_portfolio_construction"):
        pytest.skip("dense quad_form block: no sparse plan by construction")
    hs, s = _solve(name, "sparse")
    hd, d = _solve(name, "dense")
    assert hs.kkt_info()["sparse"] and not hd.kkt_info()["sparse"]
    assert s["status"] == d["status"] == 0
    assert abs(s["obj_val"] - d["obj_val"]) <= 1e-7 * max(1.0, abs(d["obj_val"]))
    if name == "nonsmooth_zoo":
        # epigraph variables of inactive max / abs pieces are not determined by the optimum (an optimal FACE):
        # the two factorisations' rounding lands on different points of it once mu goes down to IPOPT's 1e-11;
        # both are optimal (same objective above) and feasible
        for r in (s, d):
            assert np.all(r["g"] >= data_bounds(name)[0] - 1e-7) and np.all(r["g"] <= data_bounds(name)[1] + 1e-7)
    else:
        np.testing.assert_allclose(s["x"], d["x"], rtol=1e-5, atol=1e-6)
    assert abs(s["iterations"] - d["iterations"]) <= 3


def test_plan_is_sparse_for_the_paper_examples():
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    for name, max_ratio in (("nb_path_planning", 0.01), ("nb_power_flow", 0.01), ("nb_localization", 0.08),
                            ("nb_phase_retrieval", 0.12), ("mle", 0.01)):
        data, _ = build_canonical(name)
        h = OracleProblem(serialize(data["tape_arrays"]))
        if name == "nb_phase_retrieval":
            # dense measurement matrices: as a chain of triples the update program is 3.6e6 long (past the point where
            # one workgroup beats the chip-wide dense factorisation: sparse_dense_tail = no keeps the dense path); with
            # the last 128 nodes as a dense tail and the levels before it as panels it is 4.1e5, and sparse
            h0 = OracleProblem(serialize(data["tape_arrays"]))
            h0.set_option("sparse_dense_tail", "no")
            i0 = h0.kkt_info()
            assert not i0["sparse"] and i0["update_triples"] > 3000000
            assert h.kkt_info()["update_triples"] < 500000
        info = h.kkt_info()
        n = len(data["x0"]) + len(data["cl"])
        assert info["sparse"], name
        assert info["factor_values"] <= max_ratio * n * (n + 1) / 2, (name, info)
        # every equality row sits in a static 2x2 pivot block (maximum matching)
        n_eq = int(np.sum(np.asarray(data["cl"]) == np.asarray(data["cu"])))
        assert info["pairs_2x2"] == n_eq, (name, info, n_eq)


def test_dense_patterns_keep_the_dense_path():
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    data, _ = build_canonical("sphere60")         # dense quad_form block
    assert not OracleProblem(serialize(data["tape_arrays"])).kkt_info()["sparse"]
    data, _ = build_canonical("dense_eq_qp")      # dense Hessian as COO: pattern above 10 % of the triangle
    assert not OracleProblem(serialize(data["tape_arrays"])).kkt_info()["sparse"]


@pytest.mark.parametrize("seed", range(24))
def test_random_small_nlps_sparse_equals_dense(seed):
    """Random sparse NLPs (smooth objective, sparse equalities, bounds, a quadratic inequality, an
    abs-epigraph): the static-pivot sparse factorisation — forced, i.e. without the Bunch-Kaufman
    fallback — and the dense path reach the same optimum."""
    import dnlp_amd as cp
    from dnlp_amd.dnlp2smooth import Dnlp2Smooth
    from dnlp_amd.nlp_solver import HIPNLP, build_nlp_data
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    rng = np.random.default_rng(seed)
    n = int(rng.integers(3, 30))
    k = int(rng.integers(0, max(1, n // 2)))
    kind = seed % 4
    x = cp.Variable(n)
    c = rng.standard_normal(n)
    obj = (cp.sum_squares(x - c) + 0.1 * cp.sum(cp.exp(0.3 * x))) if kind != 2 \
        else (cp.sum(cp.logistic(x)) + 0.05 * cp.sum_squares(x))
    cons = []
    if k > 0:
        A = rng.standard_normal((k, n)) * (rng.random((k, n)) < 0.4)
        A[np.arange(k), rng.integers(0, n, k)] += 1.0
        cons.append(A @ x == A @ rng.standard_normal(n))
    if kind in (1, 3):
        cons += [x >= -1.0, cp.sum(cp.square(x)) <= 2.0 * n]
    if kind == 3:
        cons.append(cp.sum(cp.abs(x)) <= 0.8 * n)
    smooth, _ = Dnlp2Smooth().apply(cp.Problem(cp.Minimize(obj), cons))
    data, _ = build_nlp_data(smooth)
    out = {}
    for ls in ("sparse", "dense"):
        o = OracleProblem(serialize(data["tape_arrays"]))
        for kk, v in HIPNLP.DEFAULT_OPTIONS.items():
            o.set_option(kk, v)
        o.set_option("linear_solver", ls)
        out[ls] = o.solve(data["x0"])
    s_, d_ = out["sparse"], out["dense"]
    assert s_["status"] == d_["status"] == 0
    assert abs(s_["obj_val"] - d_["obj_val"]) <= 1e-6 * max(1.0, abs(d_["obj_val"]))
    np.testing.assert_allclose(s_["x"], d_["x"], rtol=1e-4, atol=1e-5)


def _device_solve(name, linear_solver, device_loop):
    import dnlp_amd as cp
    from paper_examples import PAPER, PUBLISHED
    from problem_zoo import GOLDEN_ZOO as Z
    prob = Z[name](cp)
    opts = dict(PUBLISHED.get(name, {}).get("options", {}))
    prob.solve(nlp=True, linear_solver=linear_solver, device_loop=device_loop, **opts)
    return prob


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["hs071", "localization", "nb_circle_packing", "nb_path_planning", "nb_power_flow"])
def test_device_sparse_paths_agree_with_dense(name, gpu_required):
    """The three ways a sparse-pattern problem can run on the device — in-kernel loop with the
    sparse factorisation (default), host-driven loop with the sparse kernels, host-driven loop with
    the chip-wide Bunch-Kaufman — land on the same optimum."""
    a = _device_solve(name, "mumps", "auto")      # IPOPT's option value: "your choice" -> sparse, in-kernel
    b = _device_solve(name, "sparse", "host")
    c = _device_solve(name, "dense", "host")
    assert a.status == b.status == c.status == "optimal"
    for p in (a, b):
        assert abs(p.value - c.value) <= 1e-6 * max(1.0, abs(c.value))


@pytest.mark.gpu
def test_c2_canonical_form_through_the_interior_point_loop(gpu_required):
    """BASELINE C2 in the form the reference hands to IPOPT: n = 1e5 -> N = 399 997, m = 299 997,
    KKT order 699 994.  Sparse factor: 2.4e6 values, 22 elimination-tree levels."""
    import dnlp_amd as cp
    from problem_zoo import rosenbrock_chain
    n = 100000
    prob = rosenbrock_chain(cp, n)
    chain = prob._build_chain(None)
    data, inv = chain.apply(prob)
    info_k = data["handle"].kkt_info()
    assert info_k["sparse"] and info_k["levels"] <= 64 and info_k["pairs_2x2"] == 3 * n - 3
    info = chain.solver.solve_via_data(data, True, False, {})
    prob.unpack_results(info, chain, inv)
    assert prob.status == "optimal"
    assert np.max(np.abs(prob.variables()[0].value - 1.0)) <= 1e-6
    assert info["iterations"] <= 60


@pytest.mark.gpu
def test_level_graph_replay_is_the_same_computation(gpu_required, monkeypatch):
    """The host-driven sparse factorisation / solves replay their level loops as HIP graphs
    (HipExec::replay_levels).  A replay launches the very same kernels with the same arguments, so the solve
    of a problem large enough for the chip-wide level kernels (small NMF: order 10 836, 9.5e5 update
    triples) must land on the same point as the direct launches (the update kernels add with FP64 atomics, so
    the two runs agree to rounding, not bit for bit)."""
    import dnlp_amd as cp
    from paper_examples import PAPER
    runs = []
    for flag in ("1", "0"):
        monkeypatch.setenv("DNLP_LEVEL_GRAPHS", flag)
        prob = PAPER["nb_nmf_small"](cp)
        chain = prob._build_chain(None)
        data, inv = chain.apply(prob)
        info_k = data["handle"].kkt_info()
        # the level-kernel path, host-driven loop: a long update program, or (since the 36-node chain at the end of this
        # plan became a dense tail) a plan with a tail, which always takes it
        assert info_k["sparse"] and (info_k["update_triples"] > 200000 or data["handle"].kkt_tail_nodes() > 0)
        info = chain.solver.solve_via_data(data, True, False, {})
        assert info.get("device_loop") is not True
        runs.append(info)
    a, b = runs
    assert a["status"] == b["status"] == 0
    assert abs(a["iterations"] - b["iterations"]) <= 2
    # (the level kernels sum their updates with atomics: run-to-run order differs, and at the barrier's 1e-11 end
    #  the ill-conditioned last steps carry that into the sixth digit of the large multipliers)
    np.testing.assert_allclose(a["x"], b["x"], rtol=1e-5, atol=1e-7)
    assert abs(a["obj_val"] - b["obj_val"]) <= 1e-9 * max(1.0, abs(b["obj_val"]))


def _nmf_blob(images):
    import dnlp_amd as cp
    from dnlp_amd.dnlp2smooth import Dnlp2Smooth
    from dnlp_amd.nlp_solver import build_nlp_data
    from dnlp_amd.tape import serialize
    from paper_examples import nb_nmf
    smooth, _ = Dnlp2Smooth().apply(nb_nmf(cp, images))
    data, _ = build_nlp_data(smooth)
    return data, serialize(data["tape_arrays"])


def _solve_with(make, blob, data, **opts):
    from dnlp_amd.nlp_solver import HIPNLP
    h = make(blob, data)
    for k, v in HIPNLP.DEFAULT_OPTIONS.items():
        h.set_option(k, v)
    for k, v in opts.items():
        h.set_option(k, v)
    return h.solve(data["x0"]), h.kkt_info()


def test_dense_tail_and_panels_are_the_same_factorisation_host_build():
    """NMF with 20 images (KKT order ~1.7e4): the elimination order ends in a chain of 60 one-block levels over a dense
    Schur complement, fed by 1 200 blocks whose structs end in those 60 nodes.  With sparse_dense_tail (default) the chain
    is ONE dense factorisation, the 1 200 blocks update it with one product per level instead of 2.2e6 triples, and the
    solves run 4 levels + one dense solve instead of 64 levels — the same LDL^T: same iterates, same optimum."""
    from oracle.oracle_capi import OracleProblem
    data, blob = _nmf_blob(20)

    def make(b, d):
        return OracleProblem(b)
    chain, ki_chain = _solve_with(make, blob, data, sparse_dense_tail="no")
    tail, ki_tail = _solve_with(make, blob, data)
    assert ki_chain["sparse"] and ki_tail["sparse"]
    assert ki_chain["update_triples"] > 2000000 and ki_tail["update_triples"] < 400000
    assert chain["status"] == 0 and tail["status"] == 0
    assert chain["iterations"] == tail["iterations"]
    assert abs(chain["obj_val"] - tail["obj_val"]) <= 1e-9 * abs(chain["obj_val"])


@pytest.mark.gpu
def test_device_dense_tail_against_the_level_chain(gpu_required):
    """The same on the MI355X (level kernels + FP64 MFMA product for the panels + dense tail): optimum of the level-chain
    run, with a fraction of the update program (the notebook-size plan: 5.5e7 triples as a chain, 1.2e6 with tail and panels)."""
    from dnlp_amd import _capi
    data, blob = _nmf_blob(20)

    def make(b, d):
        return _capi.DeviceProblem(b, d["tape"], device=0)
    chain, ki_chain = _solve_with(make, blob, data, sparse_dense_tail="no")
    tail, ki_tail = _solve_with(make, blob, data)
    assert chain["status"] == 0 and tail["status"] == 0
    assert abs(chain["iterations"] - tail["iterations"]) <= 3
    assert abs(chain["obj_val"] - tail["obj_val"]) <= 1e-7 * abs(chain["obj_val"])
    assert ki_tail["update_triples"] < 0.2 * ki_chain["update_triples"]


@pytest.mark.gpu
def test_device_tail_plans_take_the_host_driven_loop(gpu_required):
    """A plan with a dense tail lists no triples for the tail, so its triple count must not send the problem to the
    in-kernel loop (one workgroup walking the FULL program: phase retrieval took 0.62 s that way against 0.10 s)."""
    import time
    import dnlp_amd as cp
    from paper_examples import PAPER
    prob = PAPER["nb_phase_retrieval"](cp)
    chain = prob._build_chain(None)
    data, _ = chain.apply(prob)
    assert data["handle"].kkt_tail_nodes() >= 48
    chain.solver.solve_via_data(data, True, False, {})                 # (code-object load, allocations)
    t0 = time.time()
    info = chain.solver.solve_via_data(data, True, False, {})
    wall = time.time() - t0
    assert info["status"] == 0 and info.get("device_loop") is not True
    assert abs(info["obj_val"]) <= 1e-7
    assert wall < 0.4, wall


@pytest.mark.gpu
def test_device_batch_of_a_tailed_plan_gets_a_full_plan(gpu_required):
    """best_of on a problem whose handle carries a plan with a dense tail: the batch kernel cannot use that plan (its
    update program has no entries for the tail), so the C ABI builds a second, full plan for the in-kernel solver."""
    import dnlp_amd as cp
    from paper_examples import PAPER
    prob = PAPER["nb_phase_retrieval"](cp)
    for v in prob.variables():
        v.sample_bounds = (-1.0, 1.0)
    np.random.seed(3)
    prob.solve(nlp=True, best_of=2)
    # (random starts of a non-convex problem: a KKT point, not necessarily the global optimum 0 of the default start)
    assert prob.status == cp.OPTIMAL
    assert np.isfinite(prob.value) and prob.value >= -1e-9
    assert data_handle_modes(prob) == "sparse"


def data_handle_modes(prob):
    """kkt mode of the handle behind a solved Problem (the lowered data is cached on the problem)."""
    chain = prob._build_chain(None)
    data, _ = chain.apply(prob)
    return "sparse" if data["handle"].kkt_info()["sparse"] else "dense"
