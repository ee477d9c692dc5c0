"""Generated fused-objective kernels (csrc/fused_codegen.h, BASELINE config C2).  Without a GPU:
the kernel text generated for an objective is (a) cross-compiled for gfx950 by hiprtc inside
libdnlp_hip.so and (b) compiled by g++ behind a thin lane-by-lane emulation shim and run on the host,
where its f / grad f must equal the numpy tree interpreter (oracle/fused_eval.py) — the owner-computes
index logic (halo elements, ownership of f terms, boundary chunks) is exactly what can go wrong.
GPU: the same kernels through the C ABI."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

import dnlp_amd as cp
from dnlp_amd.dnlp2smooth import Dnlp2Smooth
from dnlp_amd.fused import build_fused_spec
from dnlp_amd.nlp_solver import build_nlp_data
from dnlp_amd.tape import serialize
from problem_zoo import rosenbrock_chain

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SHIM = r"""
#include <cmath>
#include <cstring>
using std::exp; using std::log; using std::log1p; using std::pow; using std::sqrt; using std::sin; using std::cos;
using std::tan; using std::sinh; using std::cosh; using std::tanh; using std::asinh; using std::atanh; using std::fabs;
#define DNLP_EMULATE 1
#define __device__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(x)
struct double2 { double x, y; };
static inline double2 make_double2(double a, double b) { double2 v; v.x = a; v.y = b; return v; }
struct idx3 { unsigned x, y, z; };
static idx3 blockIdx, threadIdx, gridDim;
static inline void dnlp_store_f(double facc, double* partial) { partial[blockIdx.x * 4 + (threadIdx.x >> 6)] += facc; }
"""

DRIVER = r"""
extern "C" void emulate(const double* x, const double* consts, double* grad, double* partial, long long nfree,
                        long long nchunks, unsigned blocks) {
  gridDim.x = blocks;
  for (unsigned b = 0; b < blocks; ++b)
    for (unsigned t = 0; t < 256; ++t) {
      blockIdx.x = b; threadIdx.x = t;
      dnlp_fused_eval(x, consts, grad, partial, nfree, nchunks);
    }
}
"""


def _lib():
    path = os.path.join(ROOT, "dnlp_amd", "libdnlp_hip.so")
    if not os.path.exists(path):
        import __graft_entry__ as g
        g.build()
    lib = C.CDLL(path)
    lib.dnlp_fused_codegen_check.restype = C.c_int
    return lib


def _generate(ta, E):
    blob = bytes(serialize(ta))
    src = C.create_string_buffer(1 << 21)
    log = C.create_string_buffer(1 << 16)
    rc = _lib().dnlp_fused_codegen_check(blob, C.c_size_t(len(blob)), E, src, C.c_size_t(len(src)), log,
                                         C.c_size_t(len(log)))
    return rc, src.value.decode(), log.value.decode()


def _emulate(src, ta, z, E, blocks):
    with tempfile.TemporaryDirectory() as d:
        cpp = os.path.join(d, "gen.cpp")
        so = os.path.join(d, "gen.so")
        with open(cpp, "w") as fh:
            fh.write(SHIM + src + DRIVER)
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-w", cpp, "-o", so])
        lib = C.CDLL(so)
        n = z.size
        grad = np.full(n, np.nan)                   # every entry must be WRITTEN (no memset, no accumulation)
        partial = np.zeros(blocks * 4)
        consts = np.ascontiguousarray(ta.get("fz_consts", np.zeros(1)), dtype=float)
        if consts.size == 0:
            consts = np.zeros(1)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))   # noqa: E731
        lib.emulate(dp(z), dp(consts), dp(grad), dp(partial), C.c_longlong(n), C.c_longlong((n + E - 1) // E),
                    C.c_uint(blocks))
        return float(ta["fz_c0"][0]) + partial.sum(), grad


def _data(prob):
    smooth, _ = Dnlp2Smooth().apply(prob)
    data, _ = build_nlp_data(smooth, user_variables=prob.variables(), fused_spec=build_fused_spec(prob))
    return data


def _single_variable_objective(seed, n=301):
    """Random sums of elementwise trees over shifted slices of ONE variable (the windowed form the
    generator handles; offsets 0..3)."""
    rng = np.random.default_rng(seed)
    x = cp.Variable(n)
    x.value = rng.uniform(0.6, 1.4, n)
    m = n - 3

    def leaf():
        o = int(rng.integers(0, 4))
        return x[o:o + m]

    def tree(depth):
        r = rng.random()
        if depth == 0 or r < 0.15:
            return leaf()
        if r < 0.30:
            return cp.multiply(tree(depth - 1), tree(depth - 1))
        if r < 0.40:
            return tree(depth - 1) / (1.5 + cp.square(tree(depth - 1)))
        if r < 0.50:
            return float(rng.uniform(-2, 2)) * tree(depth - 1) + float(rng.uniform(-1, 1))
        if r < 0.60:
            return float(rng.uniform(0.5, 2)) - tree(depth - 1)
        if r < 0.70:
            return tree(depth - 1) - tree(depth - 1)
        if r < 0.78:
            return cp.multiply(rng.uniform(0.5, 1.5, m), tree(depth - 1))
        if r < 0.86:
            return cp.square(tree(depth - 1))
        if r < 0.93:
            return cp.exp(0.3 * tree(depth - 1))
        return cp.sin(tree(depth - 1))

    f = 0
    for _ in range(int(rng.integers(1, 4))):
        f = f + float(rng.uniform(-2, 2)) * cp.sum(tree(3))
    return cp.Problem(cp.Minimize(f + 0.1 * cp.sum_squares(x)), [])


@pytest.mark.parametrize("E", [1, 2, 4, 5, 8])
def test_generated_rosenbrock_kernel_compiles_for_gfx950_and_is_exact_on_the_host(E):
    from oracle.fused_eval import numpy_eval
    for n in (7, 1000, 1027):
        data = _data(rosenbrock_chain(cp, n))
        ta = data["tape_arrays"]
        rc, src, log = _generate(ta, E)
        assert rc == 0, log
        assert "atomic" not in src.split("extern \"C\"")[1]      # owner computes: no atomics in the kernel
        z = np.random.default_rng(n).standard_normal(n)
        f, g = _emulate(src, ta, z, E, blocks=2)
        f1, g1 = numpy_eval(ta, z)
        assert abs(f - f1) <= 1e-12 * max(1.0, abs(f1))
        np.testing.assert_allclose(g, g1, rtol=1e-12, atol=1e-11)


@pytest.mark.parametrize("seed", range(12))
def test_generated_kernels_for_random_windowed_trees(seed):
    from oracle.fused_eval import numpy_eval
    prob = _single_variable_objective(seed)
    if build_fused_spec(prob) is None:
        pytest.skip("tree outside the fused grammar")
    data = _data(prob)
    ta = data["tape_arrays"]
    if not data.get("fused"):
        pytest.skip("program beyond the fused capacities")
    rc, src, log = _generate(ta, 4)
    assert rc == 0, log
    z = np.random.default_rng(seed).uniform(0.6, 1.4, ta["fz_dims"][3])
    f, g = _emulate(src, ta, z, 4, blocks=3)
    f1, g1 = numpy_eval(ta, z)
    assert abs(f - f1) <= 1e-11 * max(1.0, abs(f1))
    np.testing.assert_allclose(g, g1, rtol=1e-11, atol=1e-11)


def test_objectives_without_a_generated_form_say_why():
    rng = np.random.default_rng(5)
    n = 40
    x, y = cp.Variable(n), cp.Variable(n)
    x.value, y.value = rng.uniform(0.5, 1.5, n), rng.uniform(0.5, 1.5, n)
    prob = cp.Problem(cp.Minimize(cp.sum(cp.exp(x)) + cp.sum(cp.multiply(x[:-1], y[1:])) + cp.sum(cp.sin(x[::2]))), [])
    rc, src, log = _generate(_data(prob)["tape_arrays"], 4)
    assert rc == 1 and log                      # interpreter fallback, with the reason


@pytest.mark.gpu
def test_device_generated_kernel_matches_interpreter_and_numpy(gpu_required):
    from dnlp_amd import _capi
    from oracle.fused_eval import numpy_eval
    for n in (5, 5000, 700003):
        data = _data(rosenbrock_chain(cp, n))
        ta = data["tape_arrays"]
        z = np.random.default_rng(n).standard_normal(n)
        out = {}
        for mode in ("yes", "no"):
            dev = _capi.DeviceProblem(serialize(ta), data["tape"], device=0)
            dev.set_option("fused_codegen", mode)
            out[mode] = dev.eval_fused(z)
            dev.close()
        f1, g1 = numpy_eval(ta, z)
        for mode in ("yes", "no"):
            f, g = out[mode]
            assert abs(f - f1) <= 1e-11 * max(1.0, abs(f1))
            np.testing.assert_allclose(g, g1, rtol=1e-11, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_device_generated_kernels_for_random_windowed_trees(gpu_required, seed):
    from dnlp_amd import _capi
    from oracle.fused_eval import numpy_eval
    prob = _single_variable_objective(seed, n=40003)
    if build_fused_spec(prob) is None:
        pytest.skip("tree outside the fused grammar")
    data = _data(prob)
    ta = data["tape_arrays"]
    if not data.get("fused"):
        pytest.skip("program beyond the fused capacities")
    z = np.random.default_rng(seed).uniform(0.6, 1.4, ta["fz_dims"][3])
    dev = _capi.DeviceProblem(serialize(ta), data["tape"], device=0)
    f, g = dev.eval_fused(z)
    dev.close()
    f1, g1 = numpy_eval(ta, z)
    assert abs(f - f1) <= 1e-11 * max(1.0, abs(f1))
    np.testing.assert_allclose(g, g1, rtol=1e-11, atol=1e-11)


# ---- device-resident L-BFGS (csrc/lbfgs_codegen.h): the four slot kernels, emulated on the host ----------
LB_DRIVER = r"""
#include <vector>
template <class K, class... A> static void run_grid(unsigned blocks, unsigned threads, K kern, A... a) {
  gridDim.x = blocks;
  for (unsigned b = 0; b < blocks; ++b)
    for (unsigned t = 0; t < threads; ++t) { blockIdx.x = b; threadIdx.x = t; kern(a...); }
}
extern "C" int emulate_lbfgs(double* x, const double* consts, double c0, long long nf, int M, double tol, int max_iter,
                             int blocks, double* out) {
  LbfgsState S;
  std::memset(&S, 0, sizeof S);
  S.tol = tol; S.max_iter = max_iter; S.M = M; S.nblocks = blocks;
  const int ldp = 64;
  std::vector<double> BV((size_t)2 * M * nf, 0.0), dir(nf, 0.0), g0(nf, 0.0), g1(nf, 0.0), fpart(8 * ldp, 0.0), upart(DNLP_NV * ldp, 0.0);
  const long long nchunks = (nf + DNLP_E - 1) / DNLP_E;
  int slots = 0;
  while (S.done == 0 && slots < 100000) {
    run_grid(blocks, 256, dnlp_lb_eval, (const LbfgsState*)&S, (const double*)x, (const double*)BV.data(), g0.data(), g1.data(), dir.data(), consts, fpart.data(), nf, nchunks, ldp);
    run_grid(1, 64, dnlp_lb_accept, &S, (const double*)fpart.data(), c0, ldp);
    run_grid(blocks, 256, dnlp_lb_update, (const LbfgsState*)&S, x, BV.data(), (const double*)g0.data(), (const double*)g1.data(), (const double*)dir.data(), upart.data(), nf, ldp);
    run_grid(1, 256, dnlp_lb_control, &S, (const double*)upart.data(), ldp);
    ++slots;
  }
  out[0] = S.f; out[1] = S.gn; out[2] = S.iter; out[3] = S.evals; out[4] = S.done; out[5] = slots;
  return S.done;
}
"""


def _emulate_lbfgs(ta, x0, E, M=10, tol=1e-7, max_iter=20000, blocks=2):
    blob = bytes(serialize(ta))
    lib = _lib()
    lib.dnlp_lbfgs_codegen_check.restype = C.c_int
    src = C.create_string_buffer(1 << 21)
    log = C.create_string_buffer(1 << 16)
    rc = lib.dnlp_lbfgs_codegen_check(blob, C.c_size_t(len(blob)), E, src, C.c_size_t(len(src)), log, C.c_size_t(len(log)))
    assert rc == 0, log.value.decode()            # also: the translation unit compiles for gfx950
    with tempfile.TemporaryDirectory() as d:
        cpp, so = os.path.join(d, "lb.cpp"), os.path.join(d, "lb.so")
        with open(cpp, "w") as fh:
            fh.write(SHIM + "#define __shared__ static\n" + src.value.decode() + LB_DRIVER)
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-w", cpp, "-o", so])
        em = C.CDLL(so)
        x = np.array(x0, dtype=float)
        consts = np.ascontiguousarray(ta.get("fz_consts", np.zeros(1)), dtype=float)
        if consts.size == 0:
            consts = np.zeros(1)
        out = np.zeros(8)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))   # noqa: E731
        em.emulate_lbfgs.restype = C.c_int
        done = em.emulate_lbfgs(dp(x), dp(consts), C.c_double(float(ta["fz_c0"][0])), C.c_longlong(x.size), M,
                                C.c_double(tol), max_iter, blocks, dp(out))
        return done, x, out


@pytest.mark.parametrize("n,E", [(60, 4), (1000, 4), (1000, 2)])
def test_device_resident_lbfgs_kernels_agree_with_the_host_driven_loop(n, E):
    """Same algorithm, decisions moved into the slot kernels: the emulated kernels must take the
    host-driven loop's path (iteration and evaluation counts up to summation-order effects) and reach
    the analytic optimum x* = 1."""
    from oracle.oracle_capi import OracleProblem
    data = _data(rosenbrock_chain(cp, n))
    ta = data["tape_arrays"]
    free = np.asarray(ta["free_idx"])
    x0 = np.asarray(data["x0"])[free]
    done, x, out = _emulate_lbfgs(ta, x0, E)
    assert done == 1
    assert np.max(np.abs(x - 1.0)) <= 1e-5
    ref = OracleProblem(serialize(ta)).solve_reduced(data["x0"])
    assert ref["status"] == 0
    assert abs(int(out[2]) - ref["iterations"]) <= max(3, ref["iterations"] // 10)
    assert abs(int(out[3]) - ref["evaluations"]) <= max(4, ref["evaluations"] // 10)
    assert abs(out[0] - ref["obj_val"]) <= 1e-8


def test_device_resident_lbfgs_status_paths():
    data = _data(rosenbrock_chain(cp, 200))
    ta = data["tape_arrays"]
    x0 = np.asarray(data["x0"])[np.asarray(ta["free_idx"])]
    done, x, out = _emulate_lbfgs(ta, x0, 4, max_iter=5)
    assert done == 3 and int(out[2]) == 5                 # iteration limit -> status -1
    bad = x0.copy()
    bad[3] = np.nan
    done, x, out = _emulate_lbfgs(ta, bad, 4)
    assert done == 4                                      # invalid number at the start -> status -13


@pytest.mark.parametrize("env", [{}, {"DNLP_LBFGS_ATOMIC_SUMS": "1"}, {"DNLP_LBFGS_ATOMIC_SUMS": "1", "DNLP_LBFGS_FULL_FENCE": "1"}])
def test_persistent_lbfgs_kernel_compiles_for_gfx950_at_the_stated_size(env, monkeypatch):
    """BASELINE C2 at n = 1e5: the translation unit with the persistent single-launch kernel (a slice of 392
    variables and the whole L-BFGS history per workgroup in LDS) is generated and compiles for gfx950 — in the shipped
    form (sums across workgroups in a fixed order, order-only fences) and in the two earlier forms kept for A/B."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    data = _data(rosenbrock_chain(cp, 100000))
    blob = bytes(serialize(data["tape_arrays"]))
    lib = _lib()
    lib.dnlp_lbfgs_codegen_check.restype = C.c_int
    src = C.create_string_buffer(1 << 21)
    log = C.create_string_buffer(1 << 16)
    rc = lib.dnlp_lbfgs_codegen_check(blob, C.c_size_t(len(blob)), 4, src, C.c_size_t(len(src)), log, C.c_size_t(len(log)))
    assert rc == 0, log.value.decode()
    text = src.value.decode()
    assert "#define DNLP_PER 392" in text and "dnlp_lb_persist" in text
    assert ("#define DNLP_LB_ATOMIC_SUMS 1" in text) == ("DNLP_LBFGS_ATOMIC_SUMS" in env)
    assert "dnlp_grid_barrier_sum" in text


@pytest.mark.parametrize("n,mode", [(300000, 1), (1000000, 2)])
def test_persistent_lbfgs_kernel_beyond_lds_compiles_for_gfx950(n, mode):
    """Above n ~ 2e5 a workgroup's slice no longer fits LDS with its whole history: the history rows (mode 1) or the
    whole slice (mode 2) live in a per-workgroup strip of global memory; the same kernel text, compiled for gfx950."""
    data = _data(rosenbrock_chain(cp, n))
    blob = bytes(serialize(data["tape_arrays"]))
    lib = _lib()
    lib.dnlp_lbfgs_codegen_check.restype = C.c_int
    src = C.create_string_buffer(1 << 21)
    log = C.create_string_buffer(1 << 16)
    rc = lib.dnlp_lbfgs_codegen_check(blob, C.c_size_t(len(blob)), 4, src, C.c_size_t(len(src)), log, C.c_size_t(len(log)))
    assert rc == 0, log.value.decode()
    text = src.value.decode()
    assert ("#define DNLP_PMODE %d" % mode) in text and "dnlp_lb_persist" in text
