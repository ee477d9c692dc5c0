"""Dense KKT systems whose static pairing exists (csrc/kkt_dense.h paired mode): the matrix is assembled in the
elimination order of the static analysis, every matched (variable, equality row) pair is rotated into two non-zero
diagonal entries and the rotated matrix is factorised WITHOUT pivoting by the blocked LDL^T — in place of Bunch-Kaufman
for orders from 1024 (phase retrieval: order 1472, 8.6 -> 1.2 ms per factorisation on the MI355X).  A static sequence
that loses digits (element growth above 1e8, or a zero pivot after the first two factorisations) hands the handle to
Bunch-Kaufman for good.  Checked here: same optimum as the Bunch-Kaufman run, the mode that ran, the demotion."""
import numpy as np
import pytest

from problem_zoo import GOLDEN_ZOO


def _lowered(name):
    import dnlp_amd as cp
    from dnlp_amd.dnlp2smooth import Dnlp2Smooth
    from dnlp_amd.nlp_solver import build_nlp_data
    from dnlp_amd.tape import serialize
    prob = GOLDEN_ZOO[name](cp)
    if isinstance(prob.objective, cp.Maximize):
        prob = cp.Problem(cp.Minimize(-prob.objective.expr), prob.constraints)
    smooth, _ = Dnlp2Smooth().apply(prob)
    data, _ = build_nlp_data(smooth)
    return data, serialize(data["tape_arrays"])


def _solve(handle, x0, **opts):
    from dnlp_amd.nlp_solver import HIPNLP
    for k, v in HIPNLP.DEFAULT_OPTIONS.items():
        handle.set_option(k, v)
    for k, v in opts.items():
        handle.set_option(k, v)
    return handle.solve(x0)


def _pair(make, name, **opts):
    data, blob = _lowered(name)
    hb = make(blob, data)
    bk = _solve(hb, data["x0"], kkt_paired="no", **opts)
    hp = make(blob, data)
    pr = _solve(hp, data["x0"], **opts)
    return bk, hb.kkt_mode(), pr, hp.kkt_mode()


def _host(blob, data):
    from oracle.oracle_capi import OracleProblem
    return OracleProblem(blob)


def test_phase_retrieval_runs_on_rotated_pairs_host_build():
    # (linear_solver = dense: since the dense-tail form of the sparse plan exists this problem would take the sparse path)
    bk, mode_bk, pr, mode_pr = _pair(_host, "nb_phase_retrieval", linear_solver="dense")
    assert mode_bk == "bunch-kaufman" and mode_pr in ("paired", "paired-then-bunch-kaufman")
    assert bk["status"] == 0 and pr["status"] == 0
    assert abs(pr["obj_val"]) <= 1e-7 and abs(bk["obj_val"]) <= 1e-7            # published: 3.86e-9
    assert abs(pr["iterations"] - bk["iterations"]) <= 3
    np.testing.assert_allclose(pr["x"], bk["x"], rtol=0, atol=1e-4 * np.max(np.abs(bk["x"])))


def test_unreliable_static_sequence_is_handed_to_bunch_kaufman_host_build():
    """Sparse recovery (non-convex, delta_w up to 1e7 on its path) at a lowered size threshold: the static sequence's
    multipliers pass 1e8 on the way, the handle ends on Bunch-Kaufman and at the Bunch-Kaufman run's optimum."""
    bk, mode_bk, pr, mode_pr = _pair(_host, "nb_sparse_recovery", kkt_paired_min_n=384, linear_solver="dense")
    assert mode_bk == "bunch-kaufman" and mode_pr == "paired-then-bunch-kaufman"
    assert pr["status"] in (0, 1) and bk["status"] in (0, 1)
    assert abs(pr["obj_val"] - bk["obj_val"]) <= 1e-4 * abs(bk["obj_val"])


def test_small_and_sparse_systems_keep_their_solver():
    from oracle.oracle_capi import OracleProblem
    for name, want in (("hs071", "bunch-kaufman"), ("nb_power_flow", "sparse")):
        data, blob = _lowered(name)
        h = OracleProblem(blob)
        assert h.kkt_mode() is None
        _solve(h, data["x0"])
        assert h.kkt_mode() == want, name


@pytest.mark.gpu
def test_device_phase_retrieval_runs_on_rotated_pairs(gpu_required):
    from dnlp_amd import _capi

    def dev(blob, data):
        return _capi.DeviceProblem(blob, data["tape"], device=0)
    bk, mode_bk, pr, mode_pr = _pair(dev, "nb_phase_retrieval", time_kernels="yes", linear_solver="dense")
    assert mode_bk == "bunch-kaufman" and mode_pr in ("paired", "paired-then-bunch-kaufman")
    assert bk["status"] == 0 and pr["status"] == 0
    assert abs(pr["obj_val"]) <= 1e-7 and abs(bk["obj_val"]) <= 1e-7
    assert abs(pr["iterations"] - bk["iterations"]) <= 3
    # stats[4] = seconds in factorisations, stats[1] = their number: the blocked MFMA path is several times faster
    per_bk, per_pr = bk["stats"][4] / bk["stats"][1], pr["stats"][4] / pr["stats"][1]
    assert per_pr < 0.5 * per_bk, (per_bk, per_pr)
