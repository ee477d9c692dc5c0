"""Every known answer the reference's own NLP tests hold (tests/appendix_d.py
is the fixture table with the source lines): through the front-end with the CPU oracle as the engine
(CPU suite) and through the product path on the MI355X (`-m gpu`)."""
import warnings

import numpy as np
import pytest

from appendix_d import TABLE

SLOW_ON_CPU = {"clnlbeam"}      # N = 5 003 canonical variables: seconds through the host oracle, still run


def _solve_and_check(name):
    import dnlp_amd as cp
    row = TABLE[name]
    np.random.seed(0)                      # best_of draws its starts from numpy's global stream
    prob, handles = row["build"](cp)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        prob.solve(nlp=True, **row["kwargs"])
    assert prob.status in (cp.OPTIMAL, cp.OPTIMAL_INACCURATE), (name, prob.status)
    row["check"](prob, handles)


@pytest.mark.parametrize("name", sorted(TABLE))
def test_reference_known_answer_cpu_oracle(name):
    from oracle_frontend import oracle_engine
    with oracle_engine():
        _solve_and_check(name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(TABLE))
def test_reference_known_answer_on_device(name, gpu_required):
    _solve_and_check(name)
