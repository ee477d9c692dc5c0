"""GPU parity tests (run with `-m gpu` on the MI355X box).  Everything goes through the C ABI of
libdnlp_hip.so; the CPU oracle (oracle/) and the golden vectors captured from the reference
(tests/golden/) are only the checkers.  Nothing here reads /root/reference."""
import ctypes as C
import os

import numpy as np
import pytest

from golden_util import build_canonical, check_oracles_against_golden, load_golden
from problem_zoo import GOLDEN_ZOO, ZOO

pytestmark = pytest.mark.gpu


class _Ev:
    """Adapter: C-ABI handle -> the reference's callback names."""

    def __init__(self, h):
        self.h = h

    def objective(self, x): return self.h.eval_f(x)
    def gradient(self, x): return self.h.eval_grad_f(x)
    def constraints(self, x): return self.h.eval_g(x)
    def jacobianstructure(self): return self.h.jac_structure()
    def jacobian(self, x): return self.h.eval_jac_g(x)
    def hessianstructure(self): return self.h.hess_structure()
    def hessian(self, x, lam, sigma): return self.h.eval_h(x, lam, sigma)


def _device_problem(name):
    from dnlp_amd import _capi
    from dnlp_amd.tape import serialize
    data, inv = build_canonical(name)
    blob = serialize(data["tape_arrays"])
    return data, blob, _capi.DeviceProblem(blob, data["tape"], device=0)


@pytest.mark.parametrize("name", sorted(GOLDEN_ZOO))
def test_device_oracles_match_reference_golden(name, gpu_required):
    """f, grad f, g, Jacobian, Hessian of the HIP tape kernels vs vectors captured from the
    reference's Oracles (tolerance 1e-12 relative, FP64)."""
    data, blob, dev = _device_problem(name)
    check_oracles_against_golden(load_golden(name), _Ev(dev))


@pytest.mark.parametrize("name", sorted(ZOO))
def test_execution_space_agreement_device_vs_host_build(name, gpu_required):
    """EXECUTION-SPACE AGREEMENT, not parity: the host oracle instantiates the same algorithm text
    (csrc/ipm_core.h) over host loops, so this catches kernel / reduction / memory bugs of the HIP
    space — never an algorithm error shared by both.  Parity evidence is elsewhere: golden oracle
    vectors from the reference (above), reference-held optima (test_appendix_d.py,
    test_paper_examples.py, test_adapter_blobs.py) and closed forms (C3, C4).  Both builds must land
    on the same optimum (1e-6 relative on the objective and the primal point)."""
    from oracle.oracle_capi import OracleProblem
    data, blob, dev = _device_problem(name)
    orc = OracleProblem(blob)
    di = dev.solve(data["x0"])
    oi = orc.solve(data["x0"])
    assert di["status"] == 0, dev.log()
    assert oi["status"] == 0
    scale = max(1.0, abs(oi["obj_val"]))
    assert abs(di["obj_val"] - oi["obj_val"]) <= 1e-6 * scale
    np.testing.assert_allclose(di["x"], oi["x"], rtol=1e-5, atol=1e-6)
    # KKT residual of the device solution evaluated by the CPU oracle (unscaled problem)
    x, lam = di["x"], di["mult_g"]
    N, m = dev.n, dev.m
    grad = orc.eval_grad_f(x)
    if m:
        jr, jc = orc.jac_structure()
        import scipy.sparse as sp
        J = sp.coo_matrix((orc.eval_jac_g(x), (jr, jc)), shape=(m, N)).tocsr()
        grad = grad + J.T @ lam
    r = grad - di["mult_x_L"] + di["mult_x_U"]
    assert np.max(np.abs(r)) <= 1e-5 * max(1.0, np.max(np.abs(lam)) if m else 1.0)


def test_frontend_known_answers(gpu_required):
    """Known optima pinned by the reference's own tests (test_nlp_solvers.py:25-189)."""
    import dnlp_amd as cp
    from problem_zoo import hs071, localization, portfolio_qp, qcp, readme_toy, rosenbrock2, socp
    p = hs071(cp)
    p.solve(nlp=True)
    assert p.status == cp.OPTIMAL
    assert np.allclose(p.variables()[0].value, [0.75450865, 4.63936861, 3.78856881, 1.88513184])
    p = readme_toy(cp)
    p.solve(nlp=True)
    assert abs(p.value - 11.95081085398) <= 1e-6 * 11.95
    p = socp(cp)
    p.solve(nlp=True)
    assert np.allclose(p.value, -13.548638814247532)
    p = rosenbrock2(cp)
    p.solve(nlp=True)
    assert np.allclose(p.variables()[0].value, [1.0, 1.0])
    p = qcp(cp)
    p.solve(nlp=True)
    assert np.allclose(p.value, 0.32699284)
    p = portfolio_qp(cp)
    p.solve(nlp=True)
    assert np.allclose(p.variables()[0].value, [497.045504, 0.0, 502.954496], atol=1e-4)
    p = localization(cp)
    p.solve(nlp=True)
    x = [v for v in p.variables() if v.name() == "x"][0]
    assert np.allclose(x.value, [2.0, -1.5])


def _ldlt(A, pivoted, rhs):
    from dnlp_amd import _capi
    api = _capi.require_device(0)
    n = A.shape[0]
    Af = np.asfortranarray(A.copy())
    ipiv = np.zeros(n, np.int32)
    nneg, nzero = C.c_int(), C.c_int()
    sol = np.zeros(n)
    sec = C.c_double()
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
    rc = api.lib.dnlp_ldlt_host(0, dp(Af), n, n, ipiv.ctypes.data_as(C.POINTER(C.c_int32)), int(pivoted),
                                C.byref(nneg), C.byref(nzero), dp(np.ascontiguousarray(rhs)), dp(sol),
                                C.byref(sec))
    assert rc == 0, api.error()
    return sol, nneg.value, nzero.value, sec.value


@pytest.mark.parametrize("n", [1, 2, 7, 33, 64, 257, 600, 1100])
def test_pivoted_ldlt_indefinite(n, gpu_required):
    """Bunch-Kaufman path on symmetric indefinite KKT-shaped matrices with a zero (2,2) block:
    solve residual and inertia vs numpy."""
    rng = np.random.default_rng(n)
    nh = max(1, (2 * n) // 3)
    H = rng.standard_normal((nh, nh))
    H = H + H.T
    J = rng.standard_normal((n - nh, nh))
    A = np.block([[H, J.T], [J, np.zeros((n - nh, n - nh))]]) if n > nh else H
    b = rng.standard_normal(n)
    if nh > 6:
        # variables without curvature: their columns can only be eliminated by 2x2 pivots, which then land on
        # every position relative to the 32-column blocks of the one-workgroup solve
        H[:nh // 3, :] = 0.0
        H[:, :nh // 3] = 0.0
        A = np.block([[H, J.T], [J, np.zeros((n - nh, n - nh))]])
    sol, nneg, nzero, _ = _ldlt(A, True, b)
    ev = np.linalg.eigvalsh(A)
    assert nzero == 0
    assert nneg == int(np.sum(ev < 0))
    assert np.linalg.norm(A @ sol - b) <= 1e-9 * np.linalg.norm(A, 2) * max(np.linalg.norm(sol), 1.0)


@pytest.mark.parametrize("n", [33, 128, 129, 255, 256, 257, 300, 384, 511, 512, 513, 777, 1500, 2600])
def test_blocked_mfma_ldlt_quasidefinite(n, gpu_required):
    """Blocked unpivoted LDL^T (FP64-MFMA trailing update) on quasi-definite KKT matrices:
    inertia (n1, n2, 0) and solve residual vs numpy; sizes straddle the 128 / 256 tiles."""
    rng = np.random.default_rng(n)
    n1 = (3 * n) // 4
    G = rng.standard_normal((n1, n1))
    H = G @ G.T / n1 + np.eye(n1)
    J = rng.standard_normal((n - n1, n1))
    A = np.block([[H, J.T], [J, -1e-2 * np.eye(n - n1)]])
    b = rng.standard_normal(n)
    sol, nneg, nzero, _ = _ldlt(A, False, b)
    assert (nneg, nzero) == (n - n1, 0)
    ref = np.linalg.solve(A, b)
    assert np.linalg.norm(sol - ref) <= 1e-9 * np.linalg.norm(ref) * np.linalg.cond(A)


@pytest.mark.parametrize("n", [300, 1100, 2600])
def test_one_launch_triangular_sweeps_against_the_step_kernels(n, gpu_required, monkeypatch):
    """csrc/ldlt_blocked.h: the dataflow sweeps (one launch per sweep, 128-blocks exchanged as data-as-flag vectors, two
    workgroups per block row) and the per-step kernels they replace solve the same factored system: both within the
    residual bound, agreeing to rounding, and the sweep result identical on a second run (its sums are order-fixed)."""
    rng = np.random.default_rng(n)
    n1 = (3 * n) // 4
    G = rng.standard_normal((n1, n1))
    H = G @ G.T / n1 + np.eye(n1)
    J = rng.standard_normal((n - n1, n1))
    A = np.block([[H, J.T], [J, -1e-2 * np.eye(n - n1)]])
    b = rng.standard_normal(n)
    sol_sweep, nneg, nzero, _ = _ldlt(A, False, b)
    sol_again, _, _, _ = _ldlt(A, False, b)
    monkeypatch.setenv("DNLP_LDLT_SWEEP", "0")
    sol_steps, nneg2, nzero2, _ = _ldlt(A, False, b)
    assert (nneg, nzero) == (nneg2, nzero2) == (n - n1, 0)
    assert np.array_equal(sol_sweep, sol_again)
    scale = np.linalg.norm(A, 2) * max(np.linalg.norm(sol_steps), 1.0)
    assert np.linalg.norm(A @ sol_sweep - b) <= 1e-9 * scale
    assert np.linalg.norm(sol_sweep - sol_steps) <= 1e-10 * max(np.linalg.norm(sol_steps), 1.0) * np.linalg.cond(A)


def _ldlt_factor(A, top_mfma, env=None):
    """Factor through the C ABI with the chosen top-block kernel (csrc/ldlt_top_mfma.h / ldlt_top128_kernel); returns
    the factor as stored (unit-lower L below the diagonal, D on it), the inertia counts and the solution of A x = 1.
    `env`: further DNLP_LDLT_* switches for this one factorisation."""
    from dnlp_amd import _capi
    api = _capi.require_device(0)
    n = A.shape[0]
    old = os.environ.get("DNLP_LDLT_TOP_MFMA")
    os.environ["DNLP_LDLT_TOP_MFMA"] = "1" if top_mfma else "0"
    saved = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        Af = np.asfortranarray(A.copy())
        ipiv = np.zeros(n, np.int32)
        nneg, nzero, sec = C.c_int(), C.c_int(), C.c_double()
        sol = np.zeros(n)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))  # noqa: E731
        rc = api.lib.dnlp_ldlt_host(0, dp(Af), n, n, ipiv.ctypes.data_as(C.POINTER(C.c_int32)), 0, C.byref(nneg),
                                    C.byref(nzero), dp(np.ones(n)), dp(sol), C.byref(sec))
    finally:
        if old is None:
            del os.environ["DNLP_LDLT_TOP_MFMA"]
        else:
            os.environ["DNLP_LDLT_TOP_MFMA"] = old
        for k, v in saved.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v
    return rc, np.tril(Af), nneg.value, nzero.value, sol


@pytest.mark.parametrize("n", [1100, 2700])
def test_fused_panel_forms_reproduce_the_sub_panel_chain(n, gpu_required):
    """A full 512-column panel three ways (csrc/ldlt_blocked.h, round 5): the sub-panel chain (default); the diagonal
    block on its own rows followed by ldlt_rows512_kernel; the same with the diagonal block in one four-workgroup launch
    (ldlt_diag512_kernel, flags between workgroups).  Same products in the same order: the factors are equal bit for bit."""
    rng = np.random.default_rng(7 + n)
    n1 = (3 * n) // 4
    G = rng.standard_normal((n1, n1))
    H = G @ G.T / n1 + np.eye(n1)
    J = rng.standard_normal((n - n1, n1))
    A = np.block([[H, J.T], [J, -1e-2 * np.eye(n - n1)]])
    rc0, F0, neg0, zero0, x0 = _ldlt_factor(A, True, {"DNLP_LDLT_FUSED_ROWS": "0"})
    assert rc0 == 0 and (neg0, zero0) == (n - n1, 0)
    for diag in ("0", "1"):
        rc1, F1, neg1, zero1, x1 = _ldlt_factor(A, True, {"DNLP_LDLT_FUSED_ROWS": "1", "DNLP_LDLT_DIAG512": diag})
        assert rc1 == 0 and (neg1, zero1) == (neg0, zero0)
        assert np.array_equal(F1, F0), (diag, np.abs(F1 - F0).max())
        assert np.array_equal(x1, x0)


@pytest.mark.parametrize("n", [128, 200, 256, 640, 1100])
def test_top_block_kernels_agree(n, gpu_required):
    """The 128 x 128 top block of a sub-panel on FP64 MFMA blocks against the 4 x 4 tile kernel it replaces: same
    factor to rounding, same inertia, and the factor reproduces the matrix (quasi-definite KKT shape, so the
    unpivoted factorisation exists and is stable)."""
    rng = np.random.default_rng(100 + n)
    n1 = (3 * n) // 4
    G = rng.standard_normal((n1, n1))
    H = G @ G.T / n1 + np.eye(n1)
    J = rng.standard_normal((n - n1, n1))
    A = np.block([[H, J.T], [J, -1e-2 * np.eye(n - n1)]])
    rc0, F0, neg0, zero0, x0 = _ldlt_factor(A, False)
    rc1, F1, neg1, zero1, x1 = _ldlt_factor(A, True)
    assert rc0 == 0 and rc1 == 0
    assert (neg1, zero1) == (neg0, zero0) == (n - n1, 0)
    scale = np.abs(F0).max()
    assert np.abs(F1 - F0).max() <= 1e-10 * scale
    L = np.tril(F1, -1) + np.eye(n)
    d = np.diag(F1)
    assert np.abs((L * d) @ L.T - A).max() <= 1e-11 * np.abs(A).max() * n
    ref = np.linalg.solve(A, np.ones(n))
    assert np.linalg.norm(x1 - ref) <= 1e-9 * np.linalg.norm(ref) * np.linalg.cond(A)
    assert np.linalg.norm(x1 - x0) <= 1e-9 * np.linalg.norm(ref) * np.linalg.cond(A)


def test_top_block_kernels_count_a_zero_pivot_alike(gpu_required):
    """An exactly zero pivot inside a top block (column 37 decoupled from the rest, zero diagonal): both kernels replace it
    by the tiny pivot, count it once and leave the same inertia; a NaN in the block fails the factorisation in both."""
    n = 384
    rng = np.random.default_rng(5)
    G = rng.standard_normal((n, n))
    A = G @ G.T / n + np.eye(n)
    A[37, :] = 0.0
    A[:, 37] = 0.0
    out = [_ldlt_factor(A, m) for m in (False, True)]
    assert out[0][0] == 0 and out[1][0] == 0
    assert (out[0][2], out[0][3]) == (out[1][2], out[1][3]) == (0, 1)
    assert np.abs(out[1][1] - out[0][1]).max() <= 1e-10 * np.abs(out[0][1]).max()
    A[200, 100] = A[100, 200] = np.nan
    assert _ldlt_factor(A, False)[0] != 0
    assert _ldlt_factor(A, True)[0] != 0


def test_device_matrix_sphere_and_symv(gpu_required):
    """BASELINE C4 shape at n = 1500: quad_form on an HBM-resident matrix generated on the
    device; optimum must equal lambda_max of the downloaded matrix (analytic answer)."""
    import dnlp_amd as cp
    from dnlp_amd.device import symmetric_test_matrix
    n = 1500
    A = symmetric_test_matrix(n, seed=7, spike_eig=4.0 * np.sqrt(n), device=0)
    Ah = A.to_host()
    assert np.allclose(Ah, Ah.T)
    xh = np.random.default_rng(0).standard_normal(n)
    np.testing.assert_allclose(A.symv(xh), Ah @ xh, rtol=1e-11, atol=1e-9)
    x = cp.Variable(n)
    x.value = np.ones(n) / np.sqrt(n)
    prob = cp.Problem(cp.Maximize(cp.quad_form(x, cp.Constant(A.handle))), [cp.sum_squares(x) == 1])
    prob.solve(nlp=True, kkt_pivot_max_n=0)      # force the blocked MFMA factorisation
    lam_max = float(np.linalg.eigvalsh(Ah)[-1])
    assert prob.status == cp.OPTIMAL
    assert abs(prob.value - lam_max) <= 1e-6 * lam_max
    v = x.value / np.linalg.norm(x.value)
    assert np.linalg.norm(Ah @ v - lam_max * v) <= 1e-4 * lam_max
    # dual closed form: stationarity of -x'Ax + y (x'x - 1) gives y = lambda_max
    assert abs(float(prob._nlp_last["mult_g"][0]) - lam_max) <= 1e-6 * lam_max


@pytest.mark.parametrize("n", [4096, 4097, 4611, 6000, 9999])
def test_device_symmetric_product_from_the_lower_triangle(gpu_required, n):
    """From order 4096 the symmetric product reads the lower triangle only (gemv_sym_stage1: row sums and, by DPP wave
    reductions, column sums of every block on or below the diagonal).  Against numpy on the downloaded matrix, at
    orders that leave ragged last blocks (odd order: a last lane with one row), and against the full-matrix kernels
    (DNLP_GEMV_FULL is read once per process, so the comparison kernel is the order-1500 path of the test above)."""
    from dnlp_amd.device import symmetric_test_matrix
    A = symmetric_test_matrix(n, seed=n, spike_eig=2.0 * np.sqrt(n), device=0)
    Ah = A.to_host()
    rng = np.random.default_rng(n)
    for _ in range(2):
        xh = rng.standard_normal(n)
        ref = Ah @ xh
        got = A.symv(xh)
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-10 * np.linalg.norm(ref, np.inf) * np.sqrt(n))
        assert np.array_equal(got, A.symv(xh))            # fixed summation order: bitwise reproducible


def test_dense_eq_qp_blocked_kkt_closed_form(gpu_required):
    """BASELINE C3 shape (dense equality-constrained QP) at n=2400, m=240: KKT order 2640 goes
    through the blocked FP64-MFMA LDL^T; one Newton step must reproduce the closed-form KKT
    solution (primal and dual)."""
    import dnlp_amd as cp
    n, m = 2400, 240
    rng = np.random.default_rng(0)
    Gm = rng.standard_normal((n, n))
    Q = Gm.T @ Gm / n + np.eye(n)
    c = rng.standard_normal(n)
    A = rng.standard_normal((m, n))
    b = A @ rng.standard_normal(n)
    x = cp.Variable(n)
    prob = cp.Problem(cp.Minimize(0.5 * cp.quad_form(x, Q) + c @ x), [A @ x == b])
    chain = prob._build_chain(None)
    data, inv = chain.apply(prob)
    info = chain.solver.solve_via_data(data, True, False, {})
    K = np.block([[Q, A.T], [A, np.zeros((m, m))]])
    sol = np.linalg.solve(K, np.concatenate([-c, b]))
    assert info["status"] == 0 and info["iterations"] <= 2
    np.testing.assert_allclose(info["x"], sol[:n], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(info["mult_g"], sol[n:], rtol=1e-6, atol=1e-8)


def _c4_solve_and_check(n, tol_rel):
    """Solve BASELINE C4 at order n and certify the optimum with a size-independent property:
    the optimal value of max x'Ax on the unit sphere is lambda_max(A), obtained independently
    by power iteration with the device symmetric product."""
    import dnlp_amd as cp
    from dnlp_amd.device import symmetric_test_matrix
    A = symmetric_test_matrix(n, seed=0, spike_eig=4.0 * np.sqrt(n), device=0)
    rng = np.random.default_rng(0)
    x = cp.Variable(n)
    x.value = np.ones(n) / np.sqrt(n) + 0.1 * rng.standard_normal(n) / np.sqrt(n)
    prob = cp.Problem(cp.Maximize(cp.quad_form(x, cp.Constant(A.handle))), [cp.sum_squares(x) == 1])
    prob.solve(nlp=True, kkt_pivot_max_n=0)
    v = rng.standard_normal(n)
    v /= np.linalg.norm(v)
    lam = 0.0
    for _ in range(200):
        w = A.symv(v)
        lam_new = float(v @ w)
        v = w / np.linalg.norm(w)
        if abs(lam_new - lam) <= 1e-13 * abs(lam_new):
            break
        lam = lam_new
    lam = lam_new
    assert prob.status == cp.OPTIMAL
    assert abs(prob.value - lam) <= tol_rel * abs(lam)
    xs = x.value / np.linalg.norm(x.value)
    assert abs(np.linalg.norm(x.value) - 1.0) <= 1e-8                     # feasibility
    assert np.linalg.norm(A.symv(xs) - prob.value * xs) <= 1e-3 * abs(lam)   # eigen-residual
    assert abs(float(prob._nlp_last["mult_g"][0]) - lam) <= 1e-6 * abs(lam)  # dual closed form: y = lambda_max
    A.free()
    return prob


def test_c4_medium_size_lambda_max(gpu_required):
    p = _c4_solve_and_check(16384, 1e-6)
    assert p.solver_stats.num_iters <= 30


def test_c4_full_size_n1e5_lambda_max(gpu_required):
    """BASELINE config C4 at its full size (n = 1e5, 80 GB matrix + 80 GB KKT in HBM): the
    reference cannot run this size at all; parity is certified through the
    eigenvalue property, to the 1e-6 relative tolerance the north star states."""
    _c4_solve_and_check(100000, 1e-6)


def test_c2_rosenbrock_chain_full_size_lbfgs(gpu_required):
    """BASELINE config C2 at its full size (n = 1e5, canonical N = 399,997): unconstrained
    Rosenbrock chain, tape f / grad f evaluation + line search only (reduced-space L-BFGS);
    analytic optimum x* = 1, f* = 0."""
    import dnlp_amd as cp
    from problem_zoo import rosenbrock_chain
    n = 100000
    p = rosenbrock_chain(cp, n)
    p.solve(nlp=True, algorithm="lbfgs")
    x = p.variables()[0]
    assert p.status == cp.OPTIMAL
    assert np.max(np.abs(x.value - 1.0)) <= 1e-6
    assert abs(p.value) <= 1e-10


def test_c2_lbfgs_execution_space_agreement(gpu_required):
    """Same text in both spaces (see test_execution_space_agreement_device_vs_host_build); the
    analytic optimum x* = 1 is checked at full size in the test above."""
    from dnlp_amd import _capi
    from dnlp_amd.dnlp2smooth import Dnlp2Smooth
    from dnlp_amd.nlp_solver import build_nlp_data
    from dnlp_amd.tape import serialize
    from oracle.oracle_capi import OracleProblem
    import dnlp_amd as cp
    from problem_zoo import rosenbrock_chain
    p = rosenbrock_chain(cp, 500)
    smooth, _ = Dnlp2Smooth().apply(p)
    data, _ = build_nlp_data(smooth, p.variables())
    blob = serialize(data["tape_arrays"])
    d = _capi.DeviceProblem(blob, data["tape"], device=0).solve_reduced(data["x0"])
    o = OracleProblem(blob).solve_reduced(data["x0"])
    assert d["status"] == 0 and o["status"] == 0
    np.testing.assert_allclose(d["x"], o["x"], atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["zero", "rank_deficient", "neg_diagonal", "arrow", "random_sizes"])
def test_pivoted_ldlt_special_structures(case, gpu_required):
    """The panel-blocked Bunch-Kaufman factorisation on inputs that exercise its rarely taken branches: zero pivot
    columns (counted, replaced, no division by zero), a rank-deficient matrix (inertia with zeros), pure 1x1
    negative pivots, an arrow matrix whose pivot search always lands on the last row, and orders that are not
    multiples of the 16-column panel or of the 32-column solve block."""
    rng = np.random.default_rng(7)
    if case == "zero":
        n = 70
        A = np.zeros((n, n))
        sol, nneg, nzero, _ = _ldlt(A, True, np.zeros(n))
        assert nzero == n and nneg == 0 and np.all(np.isfinite(sol))
        return
    if case == "rank_deficient":
        n, r = 90, 60
        G = rng.standard_normal((n, r))
        A = G @ np.diag(np.where(np.arange(r) % 3 == 0, -1.0, 1.0)) @ G.T       # rank 60, indefinite
        _, nneg, nzero, _ = _ldlt(A, True, np.zeros(n))
        # pivots below the replacement threshold are counted as zeros; rounding leaves O(1e-14) pivots, which
        # are counted by their sign: the non-zero part of the inertia is what must come out exactly
        ev = np.linalg.eigvalsh(A)
        assert nneg >= int(np.sum(ev < -1e-8)) and nneg <= int(np.sum(ev < 1e-8))
        return
    if case == "neg_diagonal":
        n = 100
        A = -np.diag(rng.uniform(1.0, 2.0, n))
        b = rng.standard_normal(n)
        sol, nneg, nzero, _ = _ldlt(A, True, b)
        assert (nneg, nzero) == (n, 0)
        np.testing.assert_allclose(sol, b / np.diag(A), rtol=1e-13)
        return
    if case == "arrow":
        n = 97
        A = np.diag(rng.uniform(0.01, 0.02, n))
        A[-1, :] = A[:, -1] = rng.uniform(1.0, 2.0, n)
        b = rng.standard_normal(n)
        sol, nneg, nzero, _ = _ldlt(A, True, b)
        ev = np.linalg.eigvalsh(A)
        assert nzero == 0 and nneg == int(np.sum(ev < 0))
        assert np.linalg.norm(A @ sol - b) <= 1e-9 * np.linalg.norm(A, 2) * max(np.linalg.norm(sol), 1.0)
        return
    for n in (34, 47, 48, 49, 63, 95, 129, 161):
        nh = (2 * n) // 3
        H = rng.standard_normal((nh, nh))
        H = H + H.T
        H[: nh // 2, : nh // 2] = 0.0                     # a zero diagonal block: 2x2 pivots from the first column on
        J = rng.standard_normal((n - nh, nh))
        A = np.block([[H, J.T], [J, np.zeros((n - nh, n - nh))]])
        b = rng.standard_normal(n)
        sol, nneg, nzero, _ = _ldlt(A, True, b)
        ev = np.linalg.eigvalsh(A)
        assert nzero == 0 and nneg == int(np.sum(ev < 0)), n
        assert np.linalg.norm(A @ sol - b) <= 1e-9 * np.linalg.norm(A, 2) * max(np.linalg.norm(sol), 1.0), n


@pytest.mark.gpu
def test_create_from_arrays_on_the_device():
    """dnlp_create_arrays (no blob: the arrays are uploaded from where the front-end has them) gives the same
    oracles as dnlp_create, and the front-end takes that path for large tapes."""
    from dnlp_amd import _capi
    from dnlp_amd.nlp_solver import HIPNLP
    from dnlp_amd.tape import serialize
    from test_solver_oracle_cpu import _arrays_problem, _same_oracles
    data, z, lam = _arrays_problem()
    arrays = data["tape_arrays"]
    _same_oracles(_capi.DeviceProblem(serialize(arrays), data["tape"]), _capi.DeviceProblem(arrays, data["tape"]), z, lam)
    import dnlp_amd as cp
    from problem_zoo import ZOO
    ref = ZOO["hs071"](cp)
    ref.solve(nlp=True)                    # through the blob
    old = HIPNLP.LARGE_TAPE_BYTES
    HIPNLP.LARGE_TAPE_BYTES = 0            # every tape counts as large
    try:
        prob = ZOO["hs071"](cp)
        prob.solve(nlp=True)
        assert prob.status == ref.status == "optimal" and prob.value == ref.value
        assert np.array_equal(prob.variables()[0].value, ref.variables()[0].value)
    finally:
        HIPNLP.LARGE_TAPE_BYTES = old


def test_device_intermediate_callback_and_user_requested_stop(gpu_required):
    """Row a10 on the DEVICE library: dnlp_set_intermediate_cb on libdnlp_hip.so (Oracles.intermediate,
    nlp_solver.py:423-427) — called at iteration 0 and after every iteration with cyipopt's eleven values; a
    False return ends the solve with status 5 (ipopt_nlpif.py:31-61).  A callback forces the host-driven loop
    (a problem this small would otherwise run inside one kernel)."""
    from dnlp_amd.nlp_solver import HIPNLP
    data, blob, dev = _device_problem("hs071")
    seen = []
    dev.set_intermediate(lambda *a: seen.append(a) or True)
    for k, v in HIPNLP.DEFAULT_OPTIONS.items():
        dev.set_option(k, v)
    info = dev.solve(data["x0"])
    assert info["status"] == 0
    assert [a[1] for a in seen] == list(range(info["iterations"] + 1))       # iter_count 0..K
    assert len(seen[0]) == 11
    assert abs(seen[-1][2] - info["obj_val"]) <= 1e-12 * abs(info["obj_val"])
    assert all(a[5] > 0 for a in seen)                                       # mu
    xs = np.array([0.75450865, 4.63936861, 3.78856881, 1.88513184])          # test_nlp_solvers.py:37
    assert abs(info["obj_val"] - (xs[0] * xs[3] * (xs[0] + xs[1] + xs[2]) + xs[2])) <= 1e-6 * info["obj_val"]
    # stop after the third iteration
    dev.set_intermediate(lambda alg, it, *rest: it < 3)
    info = dev.solve(data["x0"])
    assert info["status"] == 5 and info["iterations"] == 3
    assert HIPNLP.STATUS_MAP[5] == "user_limit"
    dev.set_intermediate(None)
    assert dev.solve(data["x0"])["status"] == 0
    dev.close()


def test_device_intermediate_callback_through_the_front_end(gpu_required):
    """Problem.solve(nlp=True, intermediate_callback=fn): the user's hook sees every iteration of the
    device solve and may stop it (status "user_limit")."""
    import dnlp_amd as cp
    from problem_zoo import hs071
    prob = hs071(cp)
    calls = []
    prob.solve(nlp=True, intermediate_callback=lambda *a: calls.append(a[1]) or True)
    assert prob.status == cp.OPTIMAL
    assert calls == list(range(prob.solver_stats.num_iters + 1))
    full = prob.solver_stats.num_iters
    prob2 = hs071(cp)
    prob2.solve(nlp=True, intermediate_callback=lambda alg, it, *rest: it < 2)
    assert prob2.status == "user_limit" and prob2.solver_stats.num_iters == 2 < full


def test_device_ipm_step_counts_only_iterations_that_were_carried_out(gpu_required):
    """The counters bench.py's `value` is made of, on the device library: dnlp_ipm_step counts an iteration
    only when the iterate advanced (the call that merely detects convergence adds nothing) and the cumulative
    counters survive dnlp_ipm_begin (stats[16..18])."""
    from dnlp_amd.nlp_solver import HIPNLP
    data, blob, dev = _device_problem("hs071")
    for k, v in HIPNLP.DEFAULT_OPTIONS.items():
        dev.set_option(k, v)
    dev.set_intermediate(lambda *a: True)           # host-driven loop, as ipm_begin / ipm_step are
    total = dev.solve(data["x0"])["iterations"]
    dev.set_intermediate(None)
    st0 = dev.stats()
    dev.ipm_begin(data["x0"])
    rc, k = dev.ipm_step(3)
    assert (rc, k) == (99, 3)
    rc, k2 = dev.ipm_step(1000)
    assert rc == 0 and k + k2 == total
    rc, k3 = dev.ipm_step(5)                 # already converged: nothing is carried out, nothing counted
    assert rc == 0 and k3 == 0
    st = dev.stats()
    assert st[16] - st0[16] == total and st[18] - st0[18] == 1
    dev.ipm_begin(data["x0"])
    dev.ipm_step(2)
    st2 = dev.stats()
    assert st2[16] - st0[16] == total + 2 and st2[18] - st0[18] == 2 and st2[17] > st[17]
    assert st2[0] == 2                     # per-solve statistics were reset by begin
    dev.close()
