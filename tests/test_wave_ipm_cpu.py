"""The wavefront batch solver's restatement of the interior-point loop (dnlp_amd/csrc/wave_ipm.h: what
cvxpy/problems/problem.py:1256-1269 -> ipopt_nlpif.py:140-170 runs one problem at a time) pinned on the CPU.

One host lane runs the SAME text the MI355X kernel compiles (lane-strided loops degenerate to serial ones, the
reductions to plain sums), over the same plan block and instance rows.  Its serial sums are those of the host build
of the generic algorithm text (ipm_core.h), so the two must agree BIT FOR BIT — iterates, multipliers, iteration and
factorisation counts, statuses, the retry ladder included — on every instance the wavefront solver accepts; an
instance it refuses (kWaveNeedsGeneric = -197: the generic kernel's Bunch-Kaufman switch) is what the device hands to
the generic kernel."""
import numpy as np
import pytest

import batch_problems as bp
from dnlp_amd.batch import ParametricBatch
from wave_oracle import HostBatch

TEMPLATES = {"localization": (bp.template_localization, 96), "circle_packing": (bp.template_circle_packing, 64),
             "circle_packing10": (lambda: bp.template_circle_packing(10), 24),
             "path_planning": (bp.template_path_planning, 12), "power_flow": (bp.template_power_flow, 12)}
NEEDS_GENERIC = -197


@pytest.mark.parametrize("name", sorted(TEMPLATES))
def test_wave_restatement_equals_the_generic_algorithm_text_bit_for_bit(name):
    tmpl, B = TEMPLATES[name]
    prob, params, sample, _ = tmpl()
    hb = HostBatch(ParametricBatch(prob, params))
    thetas = np.stack([sample(i) for i in range(B)])
    w, g = hb.solve(thetas, 0), hb.solve(thetas, 1)
    took = w["status"] != NEEDS_GENERIC
    assert took.sum() >= 0.85 * B                       # (circle packing n = 10 hands ~6 % to the Bunch-Kaufman path)
    assert np.array_equal(w["status"][took], g["status"][took])
    assert np.array_equal(w["iters"][took], g["iters"][took])
    assert np.array_equal(w["nfact"][took], g["nfact"][took])
    for k in ("x", "obj", "mult_g", "zl", "zu"):
        assert np.array_equal(w[k][took], g[k][took]), k
    assert (g["status"][took] == 0).mean() >= 0.9


def test_wave_restatement_follows_the_options_too():
    """monotone barrier strategy, no second-order correction, a tighter tolerance: the same bits under other options."""
    prob, params, sample, _ = bp.template_localization()
    pb = ParametricBatch(prob, params)
    thetas = np.stack([sample(i) for i in range(32)])
    for opts in ({"mu_strategy": "monotone"}, {"max_soc": 0, "tol": 1e-9}, {"adaptive_fallback": "no", "max_iter": 25},
                 {"nlp_scaling_method": "none", "mu_init": 1.0}):
        hb = HostBatch(pb, opts)
        w, g = hb.solve(thetas, 0), hb.solve(thetas, 1)
        assert np.array_equal(w["status"], g["status"]), opts
        assert np.array_equal(w["iters"], g["iters"]), opts
        assert np.array_equal(w["x"], g["x"]), opts


def test_templates_outside_the_wave_solver_are_refused_with_a_reason():
    """A quad_over_lin segment is reduction-class: the plan builder says so and the generic kernel keeps the template."""
    import dnlp_amd as cp
    p = cp.Parameter(3, name="p", value=np.ones(3))
    x = cp.Variable(3)
    y = cp.Variable()
    prob = cp.Problem(cp.Minimize(cp.quad_over_lin(x - p, y) + y), [y >= 0.5, cp.sum(x) == 1])
    hb = HostBatch(ParametricBatch(prob, [p]))
    with pytest.raises(RuntimeError, match="reduction-class|no sparse plan"):
        hb.solve(np.ones((2, 3)), 0)


@pytest.mark.parametrize("n_circles,n_inst", [(4, 16), (10, 16), (11, 8)])
def test_dense_tail_in_registers_changes_no_bit(monkeypatch, n_circles, n_inst):
    """circle packing n = 10 ends in a chain of 21 one-block levels over a dense trailing matrix: the wavefront solver
    factors and solves it one row per lane in registers (wave_ipm.h tail_factor / tail_forward / tail_backward) instead of
    walking the level code per block.  With and without the tail (DNLP_WAVE_NO_TAIL) the host lane gives the generic text's
    bits — the tail performs the level code's operations on every entry, in its order.  n = 4, 10, 11 circles: tails of
    9, 21 and 23 rows — the widths 12 and 24 of the unrolled loops (from 24 rows on the symbolic analysis gives such a chain
    its own dense tail matrix, sparse_plan.h choose_tail, and the wavefront solver leaves the template to the generic kernel)."""
    prob, params, sample, _ = bp.template_circle_packing(n_circles)
    pb = ParametricBatch(prob, params)
    thetas = np.stack([sample(i) for i in range(n_inst)])
    g = HostBatch(pb).solve(thetas, 1)
    for no_tail in ("", "1"):
        if no_tail:
            monkeypatch.setenv("DNLP_WAVE_NO_TAIL", "1")
        w = HostBatch(pb).solve(thetas, 0)
        took = w["status"] != NEEDS_GENERIC
        assert took.sum() >= 0.7 * n_inst
        for k in ("x", "obj", "mult_g", "iters", "nfact", "status"):
            assert np.array_equal(w[k][took], g[k][took]), (no_tail, k)
