"""The per-template batch kernel (dnlp_amd/csrc/wave_codegen.h): the wavefront solver's own algorithm text (wave_ipm.h)
compiled at run time with every size, table offset and vector place of ONE template as a literal.  Role in the reference:
the serial re-solve loop cvxpy/problems/problem.py:1256-1269 -> ipopt_nlpif.py:140-170.

Without a GPU: the generated translation unit compiles for gfx950 (hiprtc is part of the image) and its constants are those
of the host lane's layout.  On the MI355X: the same launches through the per-template kernel and through the library's own
kernel take the same path (statuses, iteration counts; results to 1e-8), and the per-template kernel repeats its own bits
from launch to launch and from a ragged launch to a full one."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import batch_problems as bp
from dnlp_amd.batch import ParametricBatch
from hiprtc_util import compile_for_gfx950
from wave_oracle import HostBatch

TEMPLATES = {"localization": bp.template_localization, "circle_packing": bp.template_circle_packing,
             "circle_packing10": lambda: bp.template_circle_packing(10)}


def _source(name, nw=0):
    prob, params, sample, _ = TEMPLATES[name]()
    hb = HostBatch(ParametricBatch(prob, params))
    f = hb.lib.orc_wave_spec_source
    f.restype = C.c_longlong
    f.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_longlong]
    buf = C.create_string_buffer(1 << 22)
    n = f(hb.handle.ptr, nw, buf, len(buf))
    assert 0 < n < len(buf), (n, hb.lib.orc_last_error())
    return buf.value.decode()


@pytest.mark.parametrize("name", sorted(TEMPLATES))
def test_per_template_kernel_text_compiles_for_gfx950(name):
    src = _source(name)
    ok, log, seconds, code = compile_for_gfx950(src)
    assert ok, log[:4000]
    assert len(code) > 10000
    # the template is in the text as literals
    consts = dict(re.findall(r"constexpr int (\w+) = (-?\d+);", src))
    prob, params, sample, _ = TEMPLATES[name]()
    pb = ParametricBatch(prob, params)
    assert int(consts["k_N"]) == int(pb.arrays0["dims"][0]) and int(consts["k_m"]) == int(pb.arrays0["dims"][1])
    assert 1 <= int(consts["kNW"]) <= 8
    # plan (16 bit) + kNW shares (record + vectors) fit a compute unit's LDS
    lds = 2 * int(consts["kPlanInts"]) + int(consts["kNW"]) * (int(consts["kRecBytesMax"]) + 8 * int(consts["kStateDoubles"]))
    assert lds <= 160 * 1024


def test_templates_the_wavefront_solver_refuses_have_no_per_template_kernel():
    prob, params, sample, _ = bp.template_power_flow()
    hb = HostBatch(ParametricBatch(prob, params))
    f = hb.lib.orc_wave_spec_source
    f.restype = C.c_longlong
    f.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_longlong]
    assert f(hb.handle.ptr, 0, None, 0) == -3          # (its state, 655 KB, does not fit LDS)


def _both(pb, thetas, **kw):
    out = []
    for mode in ("1", "0"):
        old = os.environ.get("DNLP_WAVE_SPEC")
        os.environ["DNLP_WAVE_SPEC"] = mode
        try:
            out.append(pb.solve(thetas, want_duals=True, **kw))
        finally:
            if old is None:
                os.environ.pop("DNLP_WAVE_SPEC", None)
            else:
                os.environ["DNLP_WAVE_SPEC"] = old
    return out


def _same_path(s, o, frac=0.98):
    """Two compilations of one algorithm text: the compiler contracts / orders a few sums differently (last-bit differences),
    the path is the same — statuses, (almost always) iteration counts, and the optima of the instances that took the same
    number of iterations to 1e-8."""
    assert np.array_equal(s.status, o.status)
    same = s.iterations == o.iterations
    assert same.mean() >= frac, same.mean()
    same &= s.status == 0                     # (a run stopped by an iteration cap is not at a point worth comparing to 1e-8)
    np.testing.assert_allclose(s.obj_val[same], o.obj_val[same], rtol=1e-8, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("name,B,frac", [("localization", 1024, 0.98), ("circle_packing", 512, 0.95), ("circle_packing10", 256, 0.85)])
def test_per_template_kernel_follows_the_library_kernel(gpu_required, name, B, frac):
    """(circle packing has several optima and a flat face: a last-bit difference — the generated phases add every sum in
    storage order, the library kernel's long sums go through a reduction tree — sends some instances through a few more or
    fewer iterations; tests/test_wave_batch.py accepts the same of the library kernel against its host lane)"""
    prob, params, sample, _ = TEMPLATES[name]()
    pb = ParametricBatch(prob, params)
    thetas = np.stack([sample(i) for i in range(B)])
    s, o = _both(pb, thetas)
    assert s.raw["launch"]["wave_spec"] and not o.raw["launch"]["wave_spec"]
    # (state in LDS; the tables in LDS too, or — circle packing n = 10, whose two shares leave no room for them — in global memory)
    assert s.raw["launch"]["wave_form"] % 100 == (10 if name == "circle_packing10" else 11)
    _same_path(s, o, frac)
    assert (s.status == 0).mean() >= 0.9
    pb.close()


@pytest.mark.gpu
def test_per_template_kernel_under_other_options_and_ragged_launches(gpu_required):
    prob, params, sample, _ = bp.template_localization()
    pb = ParametricBatch(prob, params)
    thetas = np.stack([sample(i) for i in range(300)])
    for opts in ({"mu_strategy": "monotone", "tol": 1e-6}, {"max_soc": 0, "max_iter": 25}):
        s, o = _both(pb, thetas, **opts)
        assert s.raw["launch"]["wave_spec"]
        _same_path(s, o)
    full, _ = _both(pb, thetas)
    for n in (1, 3, 65, 257):
        part, _ = _both(pb, thetas[:n])
        assert np.array_equal(part.x, full.x[:n]) and np.array_equal(part.iterations, full.iterations[:n]), n
    pb.close()


# ---- templates whose state exceeds LDS: a workgroup per instance (dnlp_amd/csrc/wave_wg_kernel.h) ---------------------------------
def _wg_source(name, nwg=4):
    prob, params, sample, _ = {"path_planning": bp.template_path_planning, "power_flow": bp.template_power_flow}[name]()
    hb = HostBatch(ParametricBatch(prob, params))
    f = hb.lib.orc_wave_wg_source
    f.restype = C.c_longlong
    f.argtypes = [C.c_void_p, C.c_int, C.c_char_p, C.c_longlong]
    n = f(hb.handle.ptr, nwg, None, 0)
    assert n > 0, (n, hb.lib.orc_last_error())
    buf = C.create_string_buffer(n + 16)
    f(hb.handle.ptr, nwg, buf, len(buf))
    return buf.value.decode()


def test_workgroup_per_instance_kernel_text_compiles_for_gfx950():
    """path planning (KKT order 1 636, 58 elimination-tree levels): the generated phases for 4 x 64 lanes, the lane policy with
    block-wide reductions, every vector in global memory."""
    src = _wg_source("path_planning")
    assert "#define WG_LANES 256" in src and "dnlp_wave_wg_kernel" in src
    # the narrow phases stage the next phase's table, the chain levels' rows and products go through the LDS windows; every
    # placeholder of the generator has been resolved
    assert "@@" not in src and src.count("WG_STAGE_NEXT(") > 100 and src.count("WG_WW(") > 50 and src.count("WG_SW") > 50
    consts = dict(re.findall(r"constexpr int (\w+) = (-?\d+)[,;]", src))
    assert 0 < int(consts["kStageWords"]) <= 2048
    ok, log, seconds, code = compile_for_gfx950(src)
    assert ok, log[:4000]


_FITS_CHILD = r"""
import sys
if sys.argv[1] == "torch":
    import torch                                  # binds torch/lib/libhiprtc.so + libamd_comgr.so (ROCm 7.0) under the soname
sys.path.insert(0, sys.argv[2]); sys.path.insert(0, sys.argv[2] + "/tests")
import hiprtc_util as h
import test_wave_spec as t
for name in ("path_planning", "power_flow"):
    ok, log, seconds, code = h.compile_for_gfx950(t._wg_source(name, 8))
    assert ok, log[:2000]
    print("FITS", name, h.registers_of(code), h.fits_register_file(code, 8), flush=True)
maps = open("/proc/self/maps").read()
print("COMPILER", "torch" if "torch/lib/libamd_comgr" in maps else "rocm", flush=True)
"""


@pytest.mark.parametrize("compiler", ["rocm", "torch"])
def test_workgroup_kernel_fits_its_launch_under_both_compilers_of_the_image(compiler):
    """A process that imported PyTorch compiles with the hiprtc / comgr torch ships (ROCm 7.0), any other with the ROCm install's
    (7.2).  The 7.0 compiler splits this module's budget into 128 ordinary + 128 accumulation registers and let an entry point
    with a body take 140 on top: 268 registers under __launch_bounds__(512), INVALID_ISA at the first launch (bench.py and
    pytest run under torch; the tools do not).  wave_wg_kernel.h's entry point therefore only calls the body: both compilers
    must stay inside 256 registers for the eight-wavefront launch."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _FITS_CHILD, compiler, root], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("FITS")]
    assert len(lines) == 2 and all(l.endswith("True") for l in lines), r.stdout
    if compiler == "rocm":
        assert "COMPILER rocm" in r.stdout, r.stdout


_CHOICE_CHILD = r"""
import sys, ctypes as C
if sys.argv[1] == "torch":
    import torch
lib = C.CDLL(sys.argv[2] + "/dnlp_amd/libdnlp_hip.so")
buf = C.create_string_buffer(1024)
rc = lib.dnlp_rtc_compiler(buf, C.c_size_t(1024))
print("CHOICE", rc, buf.value.decode(), flush=True)
"""


@pytest.mark.parametrize("first,env,want,where", [("rocm", None, 0, "/lib/libhiprtc.so"), ("torch", None, 0, "torch/lib/libhiprtc.so"),
                                                  ("torch", "clang", 1, "/lib/llvm/bin/clang++"), ("rocm", "clang", 1, "/lib/llvm/bin/clang++")])
def test_the_kernel_cache_is_keyed_by_the_compiler_this_process_compiles_with(first, env, want, where):
    """include/dnlp_hip.h dnlp_rtc_compiler / csrc/fused_rtc.h: a process that imported PyTorch first compiles with the hiprtc
    torch ships, any other with the ROCm install's — different compilers behind the same hiprtcVersion, so the identity of the
    bound library is part of the cache key; $DNLP_RTC_COMPILER=clang takes the install's clang++ (a child process) in both."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.pop("DNLP_RTC_COMPILER", None)
    if env:
        e["DNLP_RTC_COMPILER"] = env
    r = subprocess.run([sys.executable, "-c", _CHOICE_CHILD, first, root], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    row = [l.split(None, 2) for l in r.stdout.splitlines() if l.startswith("CHOICE")][0]
    assert int(row[1]) == want and where in row[2], r.stdout


def _wg_and_generic(pb, thetas, **kw):
    out = []
    for mode in ("1", "0"):
        old = os.environ.get("DNLP_WAVE_SPEC")
        os.environ["DNLP_WAVE_SPEC"] = mode
        try:
            out.append(pb.solve(thetas, want_duals=True, **kw))
        finally:
            if old is None:
                os.environ.pop("DNLP_WAVE_SPEC", None)
            else:
                os.environ["DNLP_WAVE_SPEC"] = old
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name,same_iters", [("path_planning", 0.97), ("power_flow", 0.4)])
def test_workgroup_per_instance_kernel_follows_the_generic_kernel(gpu_required, name, same_iters):
    """256 fresh instances through the workgroup-per-instance kernel (eight wavefronts each) and through the generic batch kernel (the library's form for
    these templates before): the same statuses (power flow: all but a handful — its flat start sits next to an infeasibility
    verdict for ~7 % of the loads, and the two kernels add in different orders), mostly the same iteration counts on path
    planning, the same optima where both converge in the same number of iterations."""
    prob, params, sample, _ = {"path_planning": bp.template_path_planning, "power_flow": bp.template_power_flow}[name]()
    pb = ParametricBatch(prob, params)
    thetas = np.stack([sample(i) for i in range(256)])
    w, g = _wg_and_generic(pb, thetas)
    # (wave_form = 100 x wavefronts per instance, state and plan in global memory; eight by default)
    assert w.raw["launch"]["wave_spec"] and w.raw["launch"]["lanes"] in (256, 512) and w.raw["launch"]["wave_form"] == 100 * w.raw["launch"]["lanes"] // 64
    assert g.raw["launch"]["wave_form"] == 0
    assert np.mean(w.status == g.status) >= 0.98
    both = (w.status == 0) & (g.status == 0)
    same = (w.iterations == g.iterations) & both
    assert same.sum() >= same_iters * both.sum(), (same.sum(), both.sum())
    np.testing.assert_allclose(w.obj_val[same], g.obj_val[same], rtol=1e-6, atol=1e-8)
    assert (w.status == 0).mean() >= 0.9
    # bitwise repeatable from launch to launch
    w2, _ = _wg_and_generic(pb, thetas)
    assert np.array_equal(w.iterations, w2.iterations) and np.array_equal(w.x, w2.x) and np.array_equal(w.raw["mult_g"], w2.raw["mult_g"])
    pb.close()


@pytest.mark.gpu
def test_per_template_kernel_takes_a_warm_start(gpu_required):
    """IPOPT's warm_start_init_point through the per-template kernel: 1024 localization instances re-solved for slightly moved
    ranges from the previous result's primal point and multipliers — fewer iterations than cold, the same optima (where the cold
    start stays in the basin), and the same
    statuses as the library's own kernel gives for the warm launch."""
    prob, params, sample, _ = bp.template_localization()
    pb = ParametricBatch(prob, params)
    th0 = np.stack([sample(i) for i in range(1024)])
    th1 = th0.copy()
    th1[:, 20:] *= 1.0 + 0.01 * np.sin(np.arange(th1.shape[0]))[:, None]          # ranges moved by up to 1 %
    os.environ["DNLP_WAVE_SPEC"] = "1"
    try:
        first = pb.solve(th0, want_duals=True)
        cold = pb.solve(th1, want_duals=True)
        warm = pb.solve(th1, warm_from=first, mu_init=1e-6, want_duals=True)
        assert warm.raw["launch"]["wave_spec"] and cold.raw["launch"]["wave_spec"]
        os.environ["DNLP_WAVE_SPEC"] = "0"
        warm_own = pb.solve(th1, warm_from=first, mu_init=1e-6, want_duals=True)
    finally:
        os.environ.pop("DNLP_WAVE_SPEC", None)
    ok = (warm.status == 0) & (cold.status == 0)
    assert ok.mean() >= 0.95
    assert warm.iterations[ok].mean() < 0.7 * cold.iterations[ok].mean()
    # (a non-convex problem: a cold start may leave the basin of the previous solution — tests/test_warm_start.py)
    same = ok & (np.abs(warm.obj_val - cold.obj_val) <= 1e-5 * np.maximum(1.0, np.abs(cold.obj_val)))
    assert same.sum() >= 0.97 * ok.sum()
    assert np.mean(warm.status == warm_own.status) >= 0.99
    pb.close()


_TORCH_FIRST_CHILD = r"""
import sys
import torch                                  # FIRST: the process then compiles with the hiprtc / comgr torch ships
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import os
import numpy as np
import batch_problems as bp
from dnlp_amd.batch import ParametricBatch
os.environ["DNLP_WAVE_SPEC"] = "1"
os.environ["DNLP_KERNEL_CACHE"] = sys.argv[2]          # (a cache of its own: the kernel is compiled here, by this compiler)
prob, params, sample, _ = bp.template_path_planning()
pb = ParametricBatch(prob, params)
r = pb.solve(np.stack([sample(i) for i in range(256)]))
print("RESULT", int(r.raw["launch"]["wave_spec"]), int(r.raw["launch"]["wave_form"]), int((r.status == 0).sum()), int(r.iterations.max()), flush=True)
pb.close()
"""


@pytest.mark.gpu
def test_workgroup_kernel_runs_in_a_process_that_imported_torch_first(gpu_required, tmp_path):
    """bench.py and pytest import torch before this library: the generated kernels are then compiled by the ROCm 7.0 hiprtc
    torch ships.  Its code object of the workgroup kernel once took 268 registers under a bound of 256 and aborted the queue at
    its first launch (INVALID_ISA) — only there, the tools compile with the ROCm install's compiler.  The kernel compiled by
    THAT compiler, from an empty cache, must run and solve what the library's other kernels solve."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cache = tmp_path / "kc"
    cache.mkdir(mode=0o700)
    e = dict(os.environ)
    e.pop("DNLP_RTC_COMPILER", None)
    r = subprocess.run([sys.executable, "-c", _TORCH_FIRST_CHILD, root, str(cache)], capture_output=True, text=True, timeout=600, env=e)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    row = [l.split() for l in r.stdout.splitlines() if l.startswith("RESULT")][0]
    assert row[1] == "1" and row[2] == "800", row                  # the per-template workgroup kernel, eight wavefronts per instance
    assert int(row[3]) >= 245, row                                   # (250 of the first 256 instances are optimal under every kernel)


@pytest.mark.gpu
def test_small_template_with_a_large_state_takes_the_workgroup_kernel_from_six_instances_per_unit(gpu_required):
    """circle packing n = 10 (74 KB of state: two wavefronts per compute unit in the one-wavefront kernels): from 1536 instances
    on a launch takes the workgroup-per-instance kernel in its second form — two wavefronts per instance, four workgroups per
    compute unit, hot arrays in a quarter of the LDS each (csrc/batch.h wave_wg_prepare) — and solves what the one-wavefront
    kernel solves; DNLP_WAVE_WG_SMALL=0 keeps the one-wavefront kernel."""
    prob, params, sample, _ = bp.template_circle_packing(10)
    pb = ParametricBatch(prob, params)
    th = np.stack([sample(i) for i in range(2048)])
    try:
        a = pb.solve(th)
        os.environ["DNLP_WAVE_WG_SMALL"] = "0"
        b = pb.solve(th)
    finally:
        os.environ.pop("DNLP_WAVE_WG_SMALL", None)
    la, lb = a.raw["launch"], b.raw["launch"]
    assert la["wave_wg"] and la["lanes"] == 128 and la["per_cu"] == 4 and la["wave_form"] == 200, la
    assert lb["wave_spec"] and not lb["wave_wg"] and lb["wave_form"] == 210, lb
    assert (a.status == 0).all() and (b.status == 0).all()
    # (a non-convex problem and two summation orders: the same optimum on nearly every instance, another local one on a few)
    rel = np.abs(a.obj_val - b.obj_val) / np.maximum(1.0, np.abs(b.obj_val))
    print("same optimum on", float((rel <= 1e-6).mean()), "mean iterations", float(a.iterations.mean()), float(b.iterations.mean()))
    assert (rel <= 1e-6).mean() >= 0.9
    assert abs(a.iterations.mean() - b.iterations.mean()) <= 0.05 * b.iterations.mean()
    # a launch below six instances per compute unit keeps the one-wavefront kernel
    c = pb.solve(th[:1024])
    assert not c.raw["launch"]["wave_wg"] and c.raw["launch"]["wave_spec"]
    pb.close()
