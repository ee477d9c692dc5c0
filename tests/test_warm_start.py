"""Dual warm start (IPOPT warm_start_init_point): re-solving a perturbed instance
from the previous primal-dual point takes a fraction of the cold iterations and lands on the cold
optimum."""
import numpy as np
import pytest

import batch_problems as bp
from dnlp_amd.batch import ParametricBatch, arrays_with_data


def _perturbed_pair(tmpl, rel=0.01):
    prob, params, sample, var = tmpl()
    pb = ParametricBatch(prob, params)
    th0 = sample(0)
    th1 = th0 * (1 + rel * np.random.default_rng(1).standard_normal(th0.size))
    return pb, np.stack([th0, th1]), var


@pytest.mark.parametrize("tmpl,max_warm", [(bp.template_localization, 8), (bp.template_path_planning, 10)])
def test_oracle_warm_start_cuts_iterations(tmpl, max_warm):
    from dnlp_amd.nlp_solver import HIPNLP
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    pb, thetas, _ = _perturbed_pair(tmpl)
    mat = pb.data(thetas)
    a0, a1 = arrays_with_data(pb.arrays0, mat[0]), arrays_with_data(pb.arrays0, mat[1])

    def solve(arr, x0, ws=None, **o):
        h = OracleProblem(serialize(arr))
        for k, v in HIPNLP.DEFAULT_OPTIONS.items():
            h.set_option(k, v)
        for k, v in o.items():
            h.set_option(k, v)
        if ws is not None:
            h.set_warm_start(*ws)
        return h.solve(x0)

    base = solve(a0, a0["x0"])
    cold = solve(a1, a1["x0"])
    warm = solve(a1, base["x"], (base["mult_g"], base["mult_x_L"], base["mult_x_U"]),
                 warm_start_init_point="yes", mu_init=1e-6)
    assert base["status"] == cold["status"] == warm["status"] == 0
    assert warm["iterations"] <= max_warm < cold["iterations"]
    assert abs(warm["obj_val"] - cold["obj_val"]) <= 1e-6 * max(1.0, abs(cold["obj_val"]))
    # the option alone (no multipliers handed over) is an ordinary cold start
    plain = solve(a1, a1["x0"], warm_start_init_point="yes")
    assert plain["iterations"] == cold["iterations"]


@pytest.mark.gpu
def test_batch_warm_start_from_previous_result(gpu_required):
    prob, params, sample, x = bp.template_localization()
    pb = ParametricBatch(prob, params)
    B = 64
    th0 = np.stack([sample(i) for i in range(B)])
    th1 = th0 * (1 + 0.01 * np.random.default_rng(7).standard_normal(th0.shape))
    first = pb.solve(th0, want_duals=True)
    cold = pb.solve(th1)
    warm = pb.solve(th1, warm_from=first, mu_init=1e-6)
    ok = (first.status == 0) & (cold.status == 0) & (warm.status == 0)
    assert ok.sum() >= B - 2
    assert warm.iterations[ok].mean() <= 0.4 * cold.iterations[ok].mean()
    # same local optimum wherever the cold start stayed in the basin of the previous solution
    same = ok & (np.abs(warm.obj_val - cold.obj_val) <= 1e-6 * np.maximum(1.0, np.abs(cold.obj_val)))
    assert same.sum() >= 0.8 * ok.sum()


@pytest.mark.gpu
def test_problem_solve_warm_start_init_point(gpu_required):
    """Front-end: the second solve of the same Problem with warm_start_init_point='yes' reuses the
    stored primal-dual point (both the in-kernel and the host-driven loop)."""
    import dnlp_amd as cp
    from paper_examples import nb_path_planning
    for loop in ("auto", "host"):
        prob = nb_path_planning(cp)
        prob.solve(nlp=True, device_loop=loop)
        cold_iters, val = prob.solver_stats.num_iters, prob.value
        prob.solve(nlp=True, device_loop=loop, warm_start_init_point="yes", mu_init=1e-7)
        assert prob.status == "optimal"
        assert prob.solver_stats.num_iters <= max(3, cold_iters // 3)
        assert abs(prob.value - val) <= 1e-6 * abs(val)
