"""Runs the REFERENCE's own NLP test-suite (in place from /root/reference; nothing is copied)
against this repository's solver: `cyipopt` is replaced by tools/refshim/cyipopt, which hands
the canonical cvxpy problem to dnlp_amd.cvxpy_adapter and solves it with the host build of the
interior-point core (the same source the MI355X library compiles).  Build-container only:
skipped where the reference tree is absent (e.g. on the GPU box)."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TESTS = "/root/reference/cvxpy/tests/NLP_tests"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_TESTS), reason="reference tree not present")

# tests whose cross-check needs a conic solver this image does not have (CLARABEL): each of them is restated in
# tests/convex_certificates.py with a duality-gap certificate (or an exact LP / independent SLSQP solve) in the
# conic solver's place and runs in tests/test_convex_certificates.py on both engines
CERTIFIED_HERE = {
    "test_Sharpe_ratio.py::TestSharpeRatio::test_formulation_one",
    "test_abs.py::TestAbs::test_lasso_square_small", "test_abs.py::TestAbs::test_lasso_square",
    "test_abs.py::TestAbs::test_lasso_underdetermined", "test_abs.py::TestAbs::test_lasso_overdetermined",
    "test_entropy_related.py::TestEntropy::test_entropy_one",
    "test_entropy_related.py::TestEntropy::test_rel_entropy_one",
    "test_entropy_related.py::TestEntropy::test_rel_entropy_one_switched_arguments",
    "test_entropy_related.py::TestEntropy::test_KL_one", "test_entropy_related.py::TestEntropy::test_KL_two",
    "test_huber_sum_largest.py::TestNonsmoothNontrivial::test_huber",
    "test_huber_sum_largest.py::TestNonsmoothNontrivial::test_sum_largest",
    "test_huber_sum_largest.py::TestNonsmoothNontrivial::test_sum_smallest",
}
# ... and the known gap of this round (diverges from the all-default start)
KNOWN_GAPS = set()          # every reference test that only needs the NLP path passes


def test_reference_nlp_suite_passes_on_this_solver():
    cmd = [sys.executable, os.path.join(ROOT, "tools", "run_reference_tests.py"), REF_TESTS,
           "-k", "not Knitro and not KNITRO and not knitro", "-rf"]
    env = dict(os.environ, DNLP_SHIM_BACKEND="oracle")
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=1500).stdout
    failed = set(re.findall(r"^FAILED (\S+)", out, flags=re.M))
    m = re.search(r"(\d+) passed", out)
    assert m, out[-2000:]
    passed = int(m.group(1))
    unexpected = failed - CERTIFIED_HERE - KNOWN_GAPS
    assert not unexpected, sorted(unexpected)
    assert passed >= 205, passed
