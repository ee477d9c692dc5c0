"""The reference's thirteen CLARABEL-dependent NLP tests (tests/convex_certificates.py has the table and the
source lines) with a duality-gap certificate — or an exact LP / an independent SLSQP solve — in the conic
solver's place: through the front-end with the CPU oracle as the engine (CPU suite) and through the product
path on the MI355X (`-m gpu`).  With these, every test of cvxpy/tests/NLP_tests has a counterpart that runs
here (tests/test_reference_suite.py lists which certificate stands for which reference test)."""
import warnings

import numpy as np
import pytest

from convex_certificates import TABLE

# the 20-lambda sweeps of the four lasso shapes: every lambda on the device; on the CPU oracle the two small shapes
# in full and every fourth lambda of the 100 x 200 / 200 x 100 shapes (N = 500: ~1 s each on the host build)
CPU_ROWS = sorted(n for n in TABLE
                  if not (("underdetermined" in n or "overdetermined" in n) and int(n[-2:]) % 4 != 0))


def _solve_and_check(name):
    import dnlp_amd as cp
    row = TABLE[name]
    prob, handles = row["build"](cp)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        prob.solve(nlp=True, **row["kwargs"])
    assert prob.status == cp.OPTIMAL, (name, prob.status)
    row["check"](prob, handles)


@pytest.mark.parametrize("name", CPU_ROWS)
def test_convex_certificate_cpu_oracle(name):
    from oracle_frontend import oracle_engine
    with oracle_engine():
        _solve_and_check(name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(TABLE))
def test_convex_certificate_on_device(name, gpu_required):
    _solve_and_check(name)


def test_every_conic_cross_check_of_the_reference_has_a_certificate():
    from test_reference_suite import CERTIFIED_HERE
    srcs = {row["src"] for row in TABLE.values()}
    assert srcs == set(CERTIFIED_HERE), srcs ^ set(CERTIFIED_HERE)
    assert len(srcs) == 13
