"""The constant maps of the lowering (canonical G, Mg, Mw, MJ / Jc, MH, Jacobian / Hessian patterns, dense-block
positions) are built in C++ behind the C ABI (csrc/lower_maps.h, dnlp_lower_maps — the role of cvxcore's
build_matrix, cvxcore/src/cvxcore.cpp:161-215); dnlp_amd/lowering.py keeps the numpy / scipy construction as the
fallback and as the checker.  Every array of the two must be identical, on every problem of the golden zoo, on
problems whose constraint rows need canonicalising (unsorted / duplicated columns) and on a listed dense block."""
import numpy as np
import pytest
import scipy.sparse as sp

from problem_zoo import GOLDEN_ZOO


def _tape_arrays(name, cxx, monkeypatch):
    import dnlp_amd as cp
    from dnlp_amd.dnlp2smooth import Dnlp2Smooth
    from dnlp_amd.nlp_solver import build_nlp_data
    monkeypatch.setenv("DNLP_LOWER_CXX", "1" if cxx else "0")
    prob = GOLDEN_ZOO[name](cp) if isinstance(name, str) else name(cp)
    if isinstance(prob.objective, cp.Maximize):
        prob = cp.Problem(cp.Minimize(-prob.objective.expr), prob.constraints)
    smooth, _ = Dnlp2Smooth().apply(prob)
    data, _ = build_nlp_data(smooth)
    return data["tape_arrays"]


def _same(a, b):
    assert set(a) == set(b)
    for k in a:
        assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape, k
        assert np.array_equal(a[k], b[k], equal_nan=True), k


def test_library_exports_the_lowering_entry_points():
    from dnlp_amd import _capi
    lib = _capi.load().lib
    for sym in ("dnlp_lower_maps", "dnlp_lowered_sizes", "dnlp_lowered_csr", "dnlp_lowered_pattern",
                "dnlp_lowered_block", "dnlp_lowered_free", "dnlp_lowered_csr_view", "dnlp_lf_const", "dnlp_lf_range",
                "dnlp_lf_select", "dnlp_lf_add", "dnlp_lf_scale", "dnlp_lf_apply_csr", "dnlp_lf_apply_dense", "dnlp_lf_vstack",
                "dnlp_lf_view", "dnlp_lf_export", "dnlp_lf_gather", "dnlp_lf_info", "dnlp_lf_free"):
        assert hasattr(lib, sym), sym


@pytest.mark.parametrize("name", sorted(GOLDEN_ZOO))
def test_cxx_maps_equal_the_numpy_construction(name, monkeypatch):
    _same(_tape_arrays(name, True, monkeypatch), _tape_arrays(name, False, monkeypatch))


def _unsorted_duplicated(cp):
    """Constraint rows whose columns arrive unsorted and duplicated (reversed / permuted variables, the same
    variable twice in a row) around nonlinear atoms of the same variables."""
    rng = np.random.default_rng(3)
    n = 40
    A = rng.standard_normal((n, n)) + 2.0
    perm = rng.permutation(n)
    x, y = cp.Variable(n), cp.Variable(n)
    cons = [A @ x[::-1] + A @ x[perm] + cp.exp(y)[perm] == 1, cp.hstack([y, x])[n // 2:n // 2 + n] + cp.square(x) >= -3,
            cp.multiply(x, y) + x[::-1] <= 5]
    return cp.Problem(cp.Minimize(cp.sum_squares(x) + cp.sum(cp.exp(y)) + cp.sum(cp.multiply(x, y))), cons)


def _dense_block_with_neighbours(cp):
    """A listed dense quad_form block whose rows also receive other Hessian entries (position table, not a run)."""
    rng = np.random.default_rng(4)
    n = 30
    P = rng.standard_normal((n, n))
    P = P @ P.T + np.eye(n)
    x = cp.Variable(n)
    return cp.Problem(cp.Minimize(cp.quad_form(x, P) + cp.sum(cp.exp(x)) + cp.sum(cp.multiply(x[:-1], x[1:]))),
                      [cp.sum(x) == 1])


@pytest.mark.parametrize("builder", [_unsorted_duplicated, _dense_block_with_neighbours])
def test_cxx_maps_on_rows_that_need_canonicalising(builder, monkeypatch):
    a = _tape_arrays(builder, True, monkeypatch)
    b = _tape_arrays(builder, False, monkeypatch)
    _same(a, b)
    G = sp.csr_matrix((a["G_val"], a["G_idx"], a["G_ptr"]))
    assert G.has_canonical_format


def test_affine_form_handles_against_scipy():
    """The C ABI's affine forms (dnlp_lf_*, csrc/linform.h) against the scipy forms of lowering.py on the operations
    the DAG walk uses: dense and sparse left-multiplication (zeros of the constant carry no entry; a non-selection
    operand goes through the general product), sums, scaling, selection with repeats, stacking."""
    from dnlp_amd.lowering import CLinForm, LinForm
    if not CLinForm.available():
        pytest.skip("libdnlp_hip.so not built")
    rng = np.random.default_rng(11)
    ncol, n = 30, 12
    M = rng.standard_normal((7, n))
    M[rng.random(M.shape) < 0.3] = 0.0
    S = sp.random(9, n, density=0.3, random_state=5, format="csr")
    sel = rng.integers(0, n, size=n)
    scale = rng.standard_normal(n)
    scale[3] = 0.0

    def build(LF):
        x = LF.range(ncol, n, 4)
        y = LF.range(ncol, n, 15).select(sel)                      # repeated rows
        mix = LF.add([x.scale(scale), y, LF.const(ncol, rng.standard_normal(n) * 0 + 1.5)])
        return [x.apply_dense(M), mix.apply_dense(M), mix.apply(S), LF.add([x.apply_dense(M), LF.const(ncol, np.arange(7.0))]),
                LF.vstack([x.neg(), mix]), LF.add([x, x.neg()])]

    for a, b in zip(build(CLinForm), build(LinForm)):
        Aa, ba = a.csr()
        Ab, bb = b.csr()
        Aa, Ab = Aa.tocsr(), sp.csr_matrix(Ab)
        Ab.sum_duplicates()
        Ab.sort_indices()
        assert Aa.shape == Ab.shape
        assert np.array_equal(Aa.indptr, Ab.indptr) and np.array_equal(Aa.indices, Ab.indices)
        assert np.array_equal(Aa.data, Ab.data) and np.array_equal(ba, bb)


def test_affine_form_views_share_the_matrix_and_outlive_the_handle():
    """A x - b keeps the arrays of A x (no 1e7-entry copy for BASELINE C3's block), and the exported arrays are
    views of the handle that stay valid after the Python wrapper is dropped."""
    import gc
    from dnlp_amd.lowering import CLinForm
    if not CLinForm.available():
        pytest.skip("libdnlp_hip.so not built")
    M = np.arange(1.0, 13.0).reshape(3, 4)
    Ax = CLinForm.range(10, 4, 2).apply_dense(M)
    res = CLinForm.add([Ax, CLinForm.const(10, [-1.0, -2.0, -3.0])])
    A1, b1 = Ax.csr()
    A2, b2 = res.csr()
    assert A1.data.ctypes.data == A2.data.ctypes.data and A1.indices.ctypes.data == A2.indices.ctypes.data
    assert np.array_equal(b1, np.zeros(3)) and np.array_equal(b2, [-1.0, -2.0, -3.0])
    del Ax, res, A1, b1
    gc.collect()
    assert np.array_equal(A2.tocsr().toarray()[:, 2:6], M) and np.array_equal(A2.indptr, [0, 4, 8, 12])
