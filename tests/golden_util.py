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


def assert_coo_close(shape, r1, c1, v1, r2, c2, v2, rtol, atol):
    """|A - B| <= atol + rtol |B| entrywise for two COO matrices (duplicates summed), without
    densifying: large problems (N in the thousands) would otherwise cost N^2 per comparison."""
    if shape[0] * shape[1] <= 4_000_000:
        np.testing.assert_allclose(coo_dense(r1, c1, v1, shape), coo_dense(r2, c2, v2, shape), rtol=rtol, atol=atol)
        return
    A = sp.coo_matrix((np.asarray(v1, float), (r1, c1)), shape=shape).tocsr()
    B = sp.coo_matrix((np.asarray(v2, float), (r2, c2)), shape=shape).tocsr()
    D = (A - B).tocoo()
    if D.nnz == 0:
        return
    ref = np.asarray(B[D.row, D.col]).ravel()
    bad = np.abs(D.data) > atol + rtol * np.abs(ref)
    assert not bad.any(), "max abs diff %g at %d entries" % (np.abs(D.data[bad]).max(), int(bad.sum()))


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
                M = sp.coo_matrix((np.asarray(vals, float), (ref_r, ref_c)), shape=shape).tocsr()
                assert np.all(np.asarray(M[er, ec]).ravel() == 0.0)

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
            assert_coo_close((m, N), jr, jc, ev.jacobian(x), g["jac_rows"], g["jac_cols"], g["jac_%d" % k], rtol, atol)
        assert_coo_close((N, N), hr, hc, ev.hessian(x, lam, sigma), g["hess_rows"], g["hess_cols"],
                         g["hess_%d" % k], rtol, atol)
