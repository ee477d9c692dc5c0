"""The per-template straight-line LDL^T phases (dnlp_amd/csrc/wave_gen.h writes them, wave_gen_rt.h holds their building
blocks) pinned on the CPU: what MUMPS does behind ipopt_nlpif.py:170, once per interior-point iteration and linear solve.

The generator lays every level phase of a template's factorisation and substitutions out as one task per lane and emits
calls with literal table bases / counts; the per-template MI355X kernel compiles that text (tests/test_wave_spec.py).  Here
the SAME text is built for the host (64 lanes played one after the other) inside the same interior-point loop (wave_ipm.h)
and must reproduce, BIT FOR BIT, the interpreted walk over the plan's level tables on one host lane — which
tests/test_wave_ipm_cpu.py pins against the generic algorithm text."""
import numpy as np
import pytest

import batch_problems as bp
from dnlp_amd.batch import ParametricBatch
from wave_gen_host import GenHostBatch

TEMPLATES = {"localization": (bp.template_localization, 96), "circle_packing": (bp.template_circle_packing, 48),
             "circle_packing10": (lambda: bp.template_circle_packing(10), 16),
             "path_planning": (bp.template_path_planning, 6), "power_flow": (bp.template_power_flow, 6)}
NEEDS_GENERIC = -197
KEYS = ("status", "iters", "nfact", "x", "obj", "mult_g", "zl", "zu")


@pytest.mark.parametrize("name", sorted(TEMPLATES))
def test_generated_phases_repeat_the_interpreted_host_lane_bit_for_bit(name):
    """localization: five levels of 2x2 blocks; circle packing: a dense tail in registers behind the generated levels;
    path planning / power flow (KKT order ~1 700, several 64-lane slots per phase, ragged and mixed-kind levels): not
    templates of the per-template KERNEL (their state exceeds LDS) but the generator's hardest inputs."""
    tmpl, B = TEMPLATES[name]
    prob, params, sample, _ = tmpl()
    hb = GenHostBatch(ParametricBatch(prob, params))
    thetas = np.stack([sample(i) for i in range(B)])
    g, w = hb.solve_gen(thetas), hb.solve(thetas, 0)
    for k in KEYS:
        assert np.array_equal(g[k], w[k]), k
    assert (g["status"] == 0).mean() >= 0.8


@pytest.mark.parametrize("name", ["path_planning", "power_flow"])
def test_generated_phases_for_a_workgroup_of_four_wavefronts(name):
    """The same generator with 256 lanes sharing a phase — the form of the workgroup-per-instance kernel for templates whose
    state exceeds a compute unit's LDS (wave_wg_kernel.h): wider slots, the same sums in the same order, the same bits."""
    tmpl, B = TEMPLATES[name]
    prob, params, sample, _ = tmpl()
    hb = GenHostBatch(ParametricBatch(prob, params), lanes=256)
    thetas = np.stack([sample(i) for i in range(B)])
    g, w = hb.solve_gen(thetas), hb.solve(thetas, 0)
    for k in KEYS:
        assert np.array_equal(g[k], w[k]), k
    assert "#define WG_LANES 256" in hb.source


def test_generated_phases_without_the_register_tail(monkeypatch):
    """The workgroup kernel's plan has no dense tail in registers (one-wavefront code): the chain's last levels run as narrow
    generated phases.  With the tail switched off in both texts the bits still agree (and tests/test_wave_ipm_cpu.py shows
    that the tail changes no bit of the interpreted text)."""
    monkeypatch.setenv("DNLP_WAVE_NO_TAIL", "1")
    tmpl, B = TEMPLATES["path_planning"]
    prob, params, sample, _ = tmpl()
    hb = GenHostBatch(ParametricBatch(prob, params), lanes=512)
    thetas = np.stack([sample(i) for i in range(B)])
    g, w = hb.solve_gen(thetas), hb.solve(thetas, 0)
    for k in KEYS:
        assert np.array_equal(g[k], w[k]), k
    assert "constexpr int k_tail_T = 0;" in hb.source and "#define WG_LANES 512" in hb.source


def test_generated_phases_follow_the_options_too():
    prob, params, sample, _ = bp.template_localization()
    pb = ParametricBatch(prob, params)
    thetas = np.stack([sample(i) for i in range(24)])
    for opts in ({"mu_strategy": "monotone"}, {"max_soc": 0, "tol": 1e-9}, {"nlp_scaling_method": "none", "mu_init": 1.0}):
        hb = GenHostBatch(pb, opts)
        g, w = hb.solve_gen(thetas), hb.solve(thetas, 0)
        for k in KEYS:
            assert np.array_equal(g[k], w[k]), (opts, k)


def test_generated_text_is_calls_with_literal_arguments():
    prob, params, sample, _ = bp.template_localization()
    hb = GenHostBatch(ParametricBatch(prob, params))
    src = hb.source
    assert "template <class P, class WS> DNLP_HD bool ldl_factor(WS* S)" in src
    assert "template <class P, bool TWO, class WS, class XP, class YP> DNLP_HD void ldl_solve(WS* S, XP x, YP y)" in src
    # five levels: pivots + scaling (one phase) + products + sums for the first four, pivots only for the last; forward levels 1..4,
    # D^-1 (all 52 blocks in one phase), backward levels 3..0; level 0 has 80 struct rows: two 64-lane slots of scaling
    # (pivots and row scaling share a phase: the rows' lanes recompute the inverse pivot)
    assert src.count("wgrt::piv2<") == 5 and src.count("wgrt::scl2<") == 5 and src.count("wgrt::fwd<TWO") == 2
    # (forward levels 3 and 4: ONE target each — a position variable — with 20 / 21 rows: rows across the lanes, added in order)
    assert src.count("wgrt::fwdw<P, TWO") == 2 and src.count("wgrt::fwdw_fin<P, TWO") == 2
    assert src.count("wgrt::dsol<TWO") == 1 and src.count("wgrt::bwd<TWO") == 4
