"""Wavefront batch solver on the MI355X (dnlp_amd/csrc/wave_plan.h, wave_ipm.h, wave_batch.h behind dnlp_solve_batch*):
the template-specialised kernel of BASELINE config C5 against (a) the generic batch kernel on the same batch, (b) the
same algorithm text on one host lane (tests/wave_oracle.py), (c) itself: identical bits from launch to launch.
Reference role: the serial loop of cvxpy/problems/problem.py:1256-1269."""
import os

import numpy as np
import pytest

import batch_problems as bp
from dnlp_amd.batch import ParametricBatch

pytestmark = pytest.mark.gpu
NEEDS_GENERIC = -197


def _in_lds(form, per_cu=None):
    """A launch of the wavefront solver with state and plan in LDS: form = 100 x wavefronts per compute unit + 11 (as many
    as fit, at most eight, and no more than the launch has instances per compute unit)."""
    return form % 100 == 11 and 1 <= form // 100 <= 8 and (per_cu is None or form // 100 == per_cu)


def _solve(pb, thetas, wave, **kw):
    old = os.environ.get("DNLP_BATCH_WAVE")
    os.environ["DNLP_BATCH_WAVE"] = "2" if wave else "0"        # (2: also the templates whose state lives in global memory)
    try:
        return pb.solve(thetas, **kw)
    finally:
        if old is None:
            os.environ.pop("DNLP_BATCH_WAVE", None)
        else:
            os.environ["DNLP_BATCH_WAVE"] = old


@pytest.mark.parametrize("name,tmpl,B,form", [("localization", bp.template_localization, 1024, 411),
                                              ("circle_packing", bp.template_circle_packing, 512, 211)])
def test_wave_kernel_agrees_with_the_generic_kernel(gpu_required, name, tmpl, B, form):
    prob, params, sample, _ = tmpl()
    pb = ParametricBatch(prob, params)
    thetas = np.stack([sample(i) for i in range(B)])
    w = _solve(pb, thetas, True, want_duals=True)
    g = _solve(pb, thetas, False, want_duals=True)
    assert w.raw["launch"]["wave_form"] == form and g.raw["launch"]["wave_form"] == 0
    assert np.array_equal(w.status, g.status)
    assert (w.status == 0).all()
    # the two kernels add in different orders (lane-strided partial sums over [variables], [rows] against one pass over
    # [variables | rows]): same path, last-bit differences — a few instances take one iteration more or less
    assert np.mean(w.iterations == g.iterations) >= 0.97
    same = w.iterations == g.iterations
    np.testing.assert_allclose(w.obj_val[same], g.obj_val[same], rtol=1e-8, atol=1e-9)
    if name == "localization":               # (circle packing has several optima: a different last bit may pick another one)
        np.testing.assert_allclose(w.x[same], g.x[same], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(w.raw["mult_g"][same], g.raw["mult_g"][same], rtol=1e-5, atol=1e-6)
    pb.close()


@pytest.mark.parametrize("template", ["localization", "circle_packing"])
@pytest.mark.parametrize("opts", [{}, {"mu_strategy": "monotone", "tol": 1e-6}, {"least_square_init_duals": "yes", "max_iter": 25}],
                         ids=["defaults", "monotone", "ls-duals-capped"])
def test_wave_kernel_agrees_with_its_own_text_on_one_host_lane(gpu_required, opts, template):
    """The MI355X kernel against the same text on one host lane (tests/wave_oracle.py; the host lane itself is pinned bit
    for bit on the generic text by tests/test_wave_ipm_cpu.py) — under the default options, the monotone barrier strategy
    with a loose tolerance, and least-squares multiplier starts with an iteration cap that stops most instances early
    (status -1 on both sides)."""
    from wave_oracle import HostBatch
    prob, params, sample, _ = bp.template_localization() if template == "localization" else bp.template_circle_packing()
    pb = ParametricBatch(prob, params)
    thetas = np.stack([sample(i) for i in range(96)])
    w = _solve(pb, thetas, True, **opts)
    assert _in_lds(w.raw["launch"]["wave_form"], 1)            # (96 instances: one wavefront per compute unit)
    h = HostBatch(pb, opts).solve(thetas, 0)
    took = h["status"] != NEEDS_GENERIC        # (circle packing: exact zero pivots on the host lane go to the generic kernel there)
    if template == "circle_packing":
        # non-convex, several optima: fused multiply-adds on the device may tip an instance onto another path — status and
        # iteration count of most instances agree, and where they do, so does the optimum found
        assert np.mean(w.status[took] == h["status"][took]) >= 0.9
        same = took & (w.status == h["status"]) & (w.iterations == h["iters"])
        assert same.mean() >= 0.6
        done = same & (w.status == 0)
        np.testing.assert_allclose(w.raw["obj_val"][done], h["obj"][done], rtol=1e-7, atol=1e-9)
        pb.close()
        return
    assert np.array_equal(w.status, h["status"])
    assert np.mean(w.iterations == h["iters"]) >= 0.95
    same = w.iterations == h["iters"]
    done = same & (w.status == 0)                  # converged: the optimum itself
    np.testing.assert_allclose(w.raw["obj_val"][done], h["obj"][done], rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(w.x[done], h["x"][done], rtol=1e-6, atol=1e-7)
    cut = same & (w.status != 0)                   # stopped on the way (max_iter): the same iterate up to the path's sensitivity
    if cut.any():
        np.testing.assert_allclose(w.raw["obj_val"][cut], h["obj"][cut], rtol=1e-5, atol=1e-8)
        np.testing.assert_allclose(w.x[cut], h["x"][cut], rtol=1e-4, atol=1e-5)
    assert done.sum() + cut.sum() >= 0.95 * len(thetas)
    if opts.get("max_iter"):
        assert cut.sum() >= 0.3 * len(thetas)      # (the cap did stop a good part of the batch early)
    pb.close()


def test_wave_kernel_repeats_bit_for_bit(gpu_required):
    prob, params, sample, _ = bp.template_localization()
    thetas = np.stack([sample(i) for i in range(2048)])
    runs = []
    for fresh in range(2):
        pb = ParametricBatch(prob, params)
        for rep in range(2):
            r = _solve(pb, thetas, True, want_duals=True)
            assert _in_lds(r.raw["launch"]["wave_form"])
            runs.append(r)
        pb.close()
    for r in runs[1:]:
        assert np.array_equal(r.status, runs[0].status) and np.array_equal(r.iterations, runs[0].iterations)
        for k in ("x", "obj_val", "mult_g", "mult_x_L", "mult_x_U"):
            assert np.array_equal(r.raw[k], runs[0].raw[k]), k
    # ... and the parameter-row path gives the bits of the instance-row path (the rows are generated by the same formula)
    pb = ParametricBatch(prob, params)
    a = _solve(pb, thetas[:256], True)
    raw = pb._handle.solve_batch(pb.data(thetas[:256]))
    assert np.array_equal(raw["x"], a.raw["x"]) and np.array_equal(raw["iterations"], a.iterations)
    pb.close()


def test_refused_instances_are_solved_by_the_generic_kernel_and_merged(gpu_required):
    """An instance whose static pivot sequence is structurally singular (the generic kernel's Bunch-Kaufman switch) ends
    with kWaveNeedsGeneric inside the launch; the host solves exactly those through the generic kernel and the caller sees
    one result set.  On the host lane circle packing n = 10 refuses ~6 % (exact zero pivots; tests/test_wave_ipm_cpu.py);
    with fused multiply-adds the device meets them rarely, so the merge is also driven through the test knob."""
    prob, params, sample, _ = bp.template_circle_packing(10)
    pb = ParametricBatch(prob, params)
    thetas = np.stack([sample(i) for i in range(512)])
    w = _solve(pb, thetas, True)
    g = _solve(pb, thetas, False)
    assert w.raw["launch"]["wave_form"] in (111, 110, 210, 400)
    assert not (w.status == NEEDS_GENERIC).any()
    assert np.array_equal(w.status, g.status)
    assert (w.status == 0).mean() >= 0.97
    assert (w.iterations == g.iterations).mean() >= 0.85
    pb.close()
    prob, params, sample, _ = bp.template_localization()
    pb = ParametricBatch(prob, params)
    thetas = np.stack([sample(i) for i in range(300)])
    g = _solve(pb, thetas, False, want_duals=True)
    os.environ["DNLP_WAVE_REFUSE_EVERY"] = "7"
    try:
        w = _solve(pb, thetas, True, want_duals=True)
    finally:
        os.environ.pop("DNLP_WAVE_REFUSE_EVERY")
    assert _in_lds(w.raw["launch"]["wave_form"]) and w.raw["launch"]["wave_refused"] == len(range(0, 300, 7))
    sel = np.arange(0, 300, 7)
    for k in ("x", "obj_val", "mult_g", "mult_x_L", "mult_x_U"):       # those came from the generic kernel: its bits
        assert np.array_equal(w.raw[k][sel], g.raw[k][sel]), k
    assert np.array_equal(w.iterations[sel], g.iterations[sel]) and np.array_equal(w.status, g.status)
    pb.close()


def test_warm_started_batch_through_the_wave_kernel(gpu_required):
    prob, params, sample, _ = bp.template_localization()
    pb = ParametricBatch(prob, params)
    thetas = np.stack([sample(i) for i in range(256)])
    cold = _solve(pb, thetas, True, want_duals=True)
    warm = _solve(pb, thetas, True, warm_from=cold, mu_init=1e-6)
    assert _in_lds(warm.raw["launch"]["wave_form"])
    ok = cold.status == 0
    assert (warm.status[ok] == 0).mean() >= 0.98
    assert warm.iterations[ok].mean() < 0.5 * cold.iterations[ok].mean()
    np.testing.assert_allclose(warm.obj_val[ok & (warm.status == 0)], cold.obj_val[ok & (warm.status == 0)], rtol=1e-6, atol=1e-7)
    pb.close()


def test_large_template_takes_the_global_memory_form(gpu_required):
    """power flow: 655 KB of state per instance — vectors and plan in global memory, still one wavefront per instance."""
    prob, params, sample, _ = bp.template_power_flow()
    pb = ParametricBatch(prob, params)
    thetas = np.stack([sample(i) for i in range(96)])
    w = _solve(pb, thetas, True)
    g = _solve(pb, thetas, False)
    assert w.raw["launch"]["wave_form"] == 400
    # (the generic launch of this size gives such an instance four wavefronts: other reduction trees, and the hardest
    #  instances — the ones that end "infeasible" — are sensitive to the last bit)
    assert np.mean(w.status == g.status) >= 0.93
    assert (w.status == 0).mean() >= 0.85
    both = (w.status == 0) & (g.status == 0)
    assert both.mean() >= 0.85
    np.testing.assert_allclose(w.obj_val[both], g.obj_val[both], rtol=1e-6, atol=1e-7)      # (north star: optima within 1e-6 relative)
    pb.close()


@pytest.mark.parametrize("wave", [True, False])
def test_ragged_and_empty_launches_repeat_the_instances_of_a_full_one(gpu_required, wave):
    """An instance's result does not depend on the launch it travels in: launches of 1, 3, 5 and 257 instances (a last
    workgroup with one to three idle wavefronts, a single resident instance) and of ANY sub-range give the bits of the
    512-instance launch; an empty launch returns empty arrays.  Both kernels."""
    prob, params, sample, _ = bp.template_localization()
    pb = ParametricBatch(prob, params)
    thetas = np.stack([sample(i) for i in range(512)])
    full = _solve(pb, thetas, wave, want_duals=True)
    assert _in_lds(full.raw["launch"]["wave_form"], 2) == wave           # (512 instances: two wavefronts per compute unit)
    for lo, n in [(0, 1), (7, 3), (100, 5), (255, 257), (511, 1)]:
        part = _solve(pb, thetas[lo:lo + n], wave, want_duals=True)
        assert np.array_equal(part.status, full.status[lo:lo + n])
        assert np.array_equal(part.iterations, full.iterations[lo:lo + n])
        assert np.array_equal(part.x, full.x[lo:lo + n])
        assert np.array_equal(part.obj_val, full.obj_val[lo:lo + n])
        assert np.array_equal(part.raw["mult_g"], full.raw["mult_g"][lo:lo + n])
    none = _solve(pb, thetas[:0], wave)
    assert none.x.shape == (0, thetas.shape[1] * 0 + pb.arrays0["dims"][0]) and none.status.shape == (0,)
    pb.close()
