"""Helpers shared by the CPU and GPU parity tests."""
import os

import numpy as np
import scipy.sparse as sp

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
K_POINTS = 3


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz")))


def coo_dense(rows, cols, vals, shape):
    return np.asarray(sp.coo_matrix((np.asarray(vals, float), (rows, cols)), shape=shape).todense())


def build_canonical(name):
    """Front-end side: problem -> (flip) -> dnlp2smooth -> Bounds + tape."""
    import dnlp_amd as cp
    from dnlp_amd.dnlp2smooth import Dnlp2Smooth
    from dnlp_amd.nlp_solver import build_nlp_data
    from problem_zoo import GOLDEN_ZOO
    prob = GOLDEN_ZOO[name](cp)
    assert prob.is_dnlp()
    if isinstance(prob.objective, cp.Maximize):
        prob = cp.Problem(cp.Minimize(-prob.objective.expr), prob.constraints)
    smooth, _ = Dnlp2Smooth().apply(prob)
    return build_nlp_data(smooth)


def check_oracles_against_golden(g, ev, rtol=1e-12, atol0=1e-12):
    """`ev` has the reference's callback names; compare with the golden record `g`."""
    N, m = int(g["N"]), int(g["m"])
    jr, jc = ev.jacobianstructure()
    hr, hc = ev.hessianstructure()
    # same sparsity pattern as the reference (as sets; our order is row-major sorted).  The
    # reference keeps a structural entry where two constant affine coefficients cancel (the
    # diagonal of theta - theta.T in the power-flow example: +1 and -1 on the same (row, col));
    # the lowering folds those at build time, so the reference may hold EXTRA entries, but only
    # ones whose summed value is exactly zero at every golden point.
    def same_pattern(ours_r, ours_c, ref_r, ref_c, ref_vals, shape):
        ours = set(zip(np.asarray(ours_r).tolist(), np.asarray(ours_c).tolist()))
        ref = set(zip(ref_r.tolist(), ref_c.tolist()))
        assert ours <= ref
        extra = ref - ours
        if extra:
            er, ec = np.array(sorted(extra)).T
            for vals in ref_vals:
                assert np.all(coo_dense(ref_r, ref_c, vals, shape)[er, ec] == 0.0)

    same_pattern(jr, jc, g["jac_rows"], g["jac_cols"],
                 [g["jac_%d" % k] for k in range(K_POINTS)] if m else [], (max(m, 1), N))
    same_pattern(hr, hc, g["hess_rows"], g["hess_cols"],
                 [g["hess_%d" % k] for k in range(K_POINTS)], (N, N))
    for k in range(K_POINTS):
        x, lam, sigma = g["x_%d" % k], g["lam_%d" % k], float(g["sigma_%d" % k])
        # constraint residuals are sums of up to N terms that cancel to ~0: the absolute
        # floor scales with the magnitude that was summed (1e-12 relative to it)
        atol = atol0 * max(1.0, float(np.abs(x).sum()))
        np.testing.assert_allclose(ev.objective(x), float(g["f_%d" % k]), rtol=rtol, atol=atol)
        np.testing.assert_allclose(ev.gradient(x), g["grad_%d" % k], rtol=rtol, atol=atol)
        if m:
            np.testing.assert_allclose(ev.constraints(x), g["g_%d" % k], rtol=rtol, atol=atol)
            J = coo_dense(jr, jc, ev.jacobian(x), (m, N))
            Jg = coo_dense(g["jac_rows"], g["jac_cols"], g["jac_%d" % k], (m, N))
            np.testing.assert_allclose(J, Jg, rtol=rtol, atol=atol)
        H = coo_dense(hr, hc, ev.hessian(x, lam, sigma), (N, N))
        Hg = coo_dense(g["hess_rows"], g["hess_cols"], g["hess_%d" % k], (N, N))
        np.testing.assert_allclose(H, Hg, rtol=rtol, atol=atol)
