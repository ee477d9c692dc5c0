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


@pytest.mark.parametrize("name", sorted(GOLDEN_ZOO))
def test_sparse_and_dense_kkt_reach_the_same_optimum(name):
    if name in ("mle", "nb_phase_retrieval", "nb_path_planning", "nb_power_flow"):
        pytest.skip("dense host factorisation of this order takes seconds; covered by test_paper_examples")
    if name == "sphere60":
        pytest.skip("dense quad_form block: no sparse plan by construction")
    hs, s = _solve(name, "sparse")
    hd, d = _solve(name, "dense")
    assert hs.kkt_info()["sparse"] and not hd.kkt_info()["sparse"]
    assert s["status"] == d["status"] == 0
    assert abs(s["obj_val"] - d["obj_val"]) <= 1e-7 * max(1.0, abs(d["obj_val"]))
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
            # dense measurement matrices: the plan exists but its update program (3.6e6 triples) is
            # past the point where one workgroup beats the chip-wide dense factorisation
            assert not h.kkt_info()["sparse"]
            h.set_option("linear_solver", "sparse")
            h2 = OracleProblem(serialize(data["tape_arrays"]))
            h2.set_option("linear_solver", "sparse")
            h = h2
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
