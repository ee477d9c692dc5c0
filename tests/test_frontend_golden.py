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
    np.testing.assert_allclose(data["x0"], g["x0"], rtol=0, atol=0)
    np.testing.assert_array_equal(data["lb"], g["lb"])
    np.testing.assert_array_equal(data["ub"], g["ub"])
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
