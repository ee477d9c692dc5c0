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
