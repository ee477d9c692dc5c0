"""CPU tests: front-end (DSL + dnlp2smooth + Bounds + tape lowering) and the numpy tape
evaluator against golden vectors captured from the reference (tools/make_golden.py)."""
import numpy as np
import pytest

from golden_util import build_canonical, check_oracles_against_golden, load_golden
from problem_zoo import GOLDEN_ZOO


@pytest.mark.parametrize("name", sorted(GOLDEN_ZOO))
def test_standard_form_matches_reference(name):
    g = load_golden(name)
    data, inv = build_canonical(name)
    assert len(data["x0"]) == int(g["N"])
    assert len(data["cl"]) == int(g["m"])
    sizes = [v.size for v in data["problem"].variables()]
    assert sizes == g["var_sizes"].tolist()
    # bounds and x0 are compared PER VARIABLE: the reference lays them out in the pre-lowering
    # problem's variable order (bounds_var_sizes) while its oracles use the lowered problem's
    # (var_sizes); bounds_perm maps one to the other.  This build uses the oracle order for both.
    off = np.concatenate([[0], np.cumsum(g["bounds_var_sizes"])])

    def in_oracle_order(flat):
        return np.concatenate([flat[off[j]:off[j + 1]] for j in g["bounds_perm"]]) if len(flat) else flat

    np.testing.assert_allclose(data["x0"], in_oracle_order(g["x0"]), rtol=0, atol=0)
    np.testing.assert_array_equal(data["lb"], in_oracle_order(g["lb"]))
    np.testing.assert_array_equal(data["ub"], in_oracle_order(g["ub"]))
    np.testing.assert_array_equal(data["cl"], g["cl"])
    np.testing.assert_array_equal(data["cu"], g["cu"])


@pytest.mark.parametrize("name", sorted(GOLDEN_ZOO))
def test_tape_oracles_match_reference(name):
    from oracle.tape_eval import TapeEvaluator
    g = load_golden(name)
    data, _ = build_canonical(name)
    ev = TapeEvaluator(data["tape_arrays"])
    check_oracles_against_golden(g, ev)


def test_tape_blob_roundtrip():
    from dnlp_amd.tape import deserialize, serialize
    data, _ = build_canonical("hs071")
    blob = serialize(data["tape_arrays"])
    back = deserialize(blob)
    assert set(back) == set(data["tape_arrays"])
    for k, v in data["tape_arrays"].items():
        np.testing.assert_array_equal(back[k], v)


def test_bounds_follow_the_lowered_variable_order():
    """`a <= b` is lowered to `b - a >= 0`, which lists b before a; bounds of `a` must
    stay on `a` (the reference puts them on the first two entries of whatever comes first)."""
    import dnlp_amd as cp
    from dnlp_amd.dnlp2smooth import Dnlp2Smooth
    from dnlp_amd.nlp_solver import build_nlp_data
    a = cp.Variable(2, bounds=[0, 1])
    b = cp.Variable(2)
    c = cp.Variable()
    prob = cp.Problem(cp.Minimize(c), [a <= b, c >= cp.sum(b)])
    smooth, _ = Dnlp2Smooth().apply(prob)
    data, inv = build_nlp_data(smooth)
    order = data["problem"].variables()
    off = 0
    for v in order:
        lb, ub = data["lb"][off:off + v.size], data["ub"][off:off + v.size]
        if v is a:
            assert np.all(lb == 0.0) and np.all(ub == 1.0)
        else:
            assert np.all(np.isneginf(lb)) and np.all(np.isposinf(ub))
        assert inv.var_offsets[v.id] == off
        off += v.size
    assert any(v is a for v in order)


def test_lowering_leaves_the_users_constants_untouched():
    """A dense constant under a non-monotone vector of variables (x[::-1], a permutation, hstack of two
    variables in the other order): lowering re-sorts rows of the renamed matrix and must do so on a
    private copy — the constant the user holds is bit-identical afterwards, and the lowered rows are
    the permuted matrix."""
    import dnlp_amd as cp
    from dnlp_amd.dnlp2smooth import Dnlp2Smooth
    from dnlp_amd.nlp_solver import build_nlp_data
    rng = np.random.default_rng(0)
    n = 70
    perm = rng.permutation(n)
    for build in (lambda x, y, A: [A @ x[::-1] == 1],
                  lambda x, y, A: [A @ x[perm] == 1],
                  lambda x, y, A: [A @ cp.hstack([y, x])[n // 2:n // 2 + n] == 1]):
        A = rng.standard_normal((n, n)) + 3.0          # no zero entry: the CSR wrap is a view of A
        A0 = A.copy()
        x, y = cp.Variable(n), cp.Variable(n)
        prob = cp.Problem(cp.Minimize(cp.sum_squares(x) + cp.sum_squares(y)), build(x, y, A))
        smooth, _ = Dnlp2Smooth().apply(prob)
        data, _ = build_nlp_data(smooth)
        assert np.array_equal(A, A0), "lowering permuted the user's constant in place"
        arr = data["tape_arrays"]
        assert int(arr["dims"][1]) == n
