"""Run-to-run reproducibility on the device.  IPOPT + MUMPS as the reference calls it
(cvxpy/reductions/solvers/nlp_solvers/ipopt_nlpif.py:140-170) returns the same iterate sequence on every run;
"results identical to the reference's" cannot be claimed by a path that differs from itself.  Every accumulation
of the solve path is order-fixed (tape.h CooIdx: J v / J^T v / H v by output; sparse_plan.h gdst / fnode: the
sparse LDL^T's Schur updates and forward substitution by destination; fixed reduction trees), so two fresh
handles x two solves each must agree in iteration count, status and every bit of the objective and of x."""
import numpy as np
import pytest

import batch_problems as bp
from dnlp_amd.batch import ParametricBatch
from paper_examples import PAPER, PAPER_LARGE

pytestmark = pytest.mark.gpu

ALL = dict(PAPER)
ALL.update(PAPER_LARGE)


def _runs(build, fresh=2, reps=2, opts=None):
    import dnlp_amd as cp
    out = []
    for _ in range(fresh):
        prob = build(cp)
        chain = prob._build_chain(None)
        data, _ = chain.apply(prob)
        for _ in range(reps):
            info = chain.solver.solve_via_data(dict(data), True, False, dict(opts or {}))
            out.append((int(info["iterations"]), int(info["status"]), float(info["obj_val"]), np.array(info["x"]),
                        np.array(info["mult_g"])))
    return out


def _assert_same(name, runs):
    its = [r[0] for r in runs]
    sts = [r[1] for r in runs]
    assert len(set(its)) == 1, (name, "iteration counts differ from run to run", its)
    assert len(set(sts)) == 1, (name, "statuses differ", sts)
    for r in runs[1:]:
        assert r[2] == runs[0][2], (name, "objective bits differ", r[2], runs[0][2])
        assert np.array_equal(r[3], runs[0][3]), (name, "x bits differ", float(np.max(np.abs(r[3] - runs[0][3]))))
        assert np.array_equal(r[4], runs[0][4]), (name, "multiplier bits differ")


@pytest.mark.parametrize("name", sorted(ALL))
def test_paper_example_is_bitwise_repeatable(gpu_required, name):
    _assert_same(name, _runs(ALL[name]))


def _c3(cp, n=3000, m=300):
    rng = np.random.default_rng(0)
    Gm = rng.standard_normal((n, n))
    Q = Gm.T @ Gm / n + np.eye(n)
    c = rng.standard_normal(n)
    A = rng.standard_normal((m, n))
    b = A @ rng.standard_normal(n)
    x = cp.Variable(n)
    return cp.Problem(cp.Minimize(0.5 * cp.quad_form(x, Q) + c @ x), [A @ x == b])


@pytest.mark.parametrize("rect", [False, True])
def test_c3_dense_equality_qp_is_bitwise_repeatable(gpu_required, monkeypatch, rect):
    """BASELINE C3 at a size the suite affords twice over (n = 3000, m = 300: the blocked MFMA LDL^T, the dense
    Jacobian's products and the triangular solves on inverted blocks are the kernels of the full size).  At the
    full size the Jacobian (1e7 entries) is above the tape's index limit and takes the rectangular row-major
    products (exec_hip.h rect_mult / rect_tmult): `rect` forces that path here through the same limit."""
    if rect:
        monkeypatch.setenv("DNLP_COO_DET_MAX", "1000")
    runs = _runs(_c3)
    _assert_same("c3", runs)
    if rect:
        monkeypatch.delenv("DNLP_COO_DET_MAX")
        ref = _runs(_c3, fresh=1, reps=1)[0]
        assert runs[0][0] == ref[0] and runs[0][1] == ref[1]
        np.testing.assert_allclose(runs[0][3], ref[3], rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(runs[0][4], ref[4], rtol=1e-8, atol=1e-10)


@pytest.mark.parametrize("which", ["localization", "circle_packing10", "power_flow", "path_planning"])
def test_c5_member_launch_is_bitwise_repeatable(gpu_required, which):
    """A 1024-instance launch of each BASELINE C5 member, twice through one handle and once through a fresh one:
    the same status, iteration count and bits of x on every instance (instances are claimed from a queue in
    whatever order the hardware schedules them: the result of an instance must not depend on it)."""
    B = 1024
    res = []
    for fresh in range(2):
        prob, params, sample, _ = (bp.template_circle_packing(10) if which == "circle_packing10"
                                   else getattr(bp, "template_" + which)())
        pb = ParametricBatch(prob, params)
        thetas = np.stack([sample(i) for i in range(B)])
        for _ in range(2 - fresh):
            r = pb.solve(thetas)
            res.append((np.array(r.status), np.array(r.iterations), np.array(r.x), np.array(r.raw["obj_val"])))
        pb.close()
    for r in res[1:]:
        assert np.array_equal(r[0], res[0][0]), (which, "statuses differ on", int(np.sum(r[0] != res[0][0])), "instances")
        assert np.array_equal(r[1], res[0][1]), (which, "iteration counts differ on", int(np.sum(r[1] != res[0][1])), "instances")
        assert np.array_equal(r[3], res[0][3]), (which, "objective bits differ")
        assert np.array_equal(r[2], res[0][2]), (which, "x bits differ")


def test_c2_persistent_lbfgs_is_bitwise_repeatable(gpu_required):
    """BASELINE C2 at its stated size through the one-launch L-BFGS kernel (csrc/lbfgs_codegen.h): the sums across its
    workgroups — f, the check sum and the 3 (2M + 1) inner products of every trial point — are added in member /
    group order inside the grid barrier, not by atomics in arrival order: two fresh handles x three solves end on
    the same bits after the same number of iterations and evaluations."""
    import dnlp_amd as cp
    from problem_zoo import rosenbrock_chain
    runs = []
    for _ in range(2):
        prob = rosenbrock_chain(cp, 100000)
        chain = prob._build_chain(None)
        data, _ = chain.apply(prob)
        for _ in range(3):
            info = chain.solver.solve_via_data(dict(data), True, False, {"algorithm": "lbfgs"})
            assert info["status"] == 0 and info["device_loop"] and info["device_loop_persistent"]
            runs.append((int(info["iterations"]), int(info["evaluations"]), float(info["obj_val"]), np.array(info["x"])))
    for r in runs[1:]:
        assert (r[0], r[1]) == (runs[0][0], runs[0][1]), ("iterations / evaluations differ", [(q[0], q[1]) for q in runs])
        assert r[2] == runs[0][2], ("objective bits differ", r[2], runs[0][2])
        assert np.array_equal(r[3], runs[0][3]), ("x bits differ", float(np.max(np.abs(r[3] - runs[0][3]))))
