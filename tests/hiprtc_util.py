"""TEST INFRASTRUCTURE — hiprtc through ctypes: compiles a HIP translation unit for gfx950 without a GPU (the compiler is
part of the ROCm image; nothing is executed)."""
import ctypes as C
import time


def compile_for_gfx950(src: str, opts=("-O3", "-std=c++17", "-munsafe-fp-atomics")):
    """-> (ok, log, seconds, code bytes)."""
    try:
        rtc = C.CDLL("libhiprtc.so")
    except OSError:
        rtc = C.CDLL("/opt/rocm/lib/libhiprtc.so")
    prog = C.c_void_p()
    rtc.hiprtcCreateProgram.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
    rc = rtc.hiprtcCreateProgram(C.byref(prog), src.encode(), b"dnlp_generated.hip", 0, None, None)
    assert rc == 0, rc
    args = [b"--offload-arch=gfx950"] + [o.encode() for o in opts]
    arr = (C.c_char_p * len(args))(*args)
    rtc.hiprtcCompileProgram.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p)]
    t0 = time.time()
    rc = rtc.hiprtcCompileProgram(prog, len(args), arr)
    dt = time.time() - t0
    n = C.c_size_t()
    rtc.hiprtcGetProgramLogSize.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    rtc.hiprtcGetProgramLogSize(prog, C.byref(n))
    log = C.create_string_buffer(max(n.value, 1))
    rtc.hiprtcGetProgramLog.argtypes = [C.c_void_p, C.c_char_p]
    rtc.hiprtcGetProgramLog(prog, log)
    code = b""
    if rc == 0:
        rtc.hiprtcGetCodeSize.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        rtc.hiprtcGetCodeSize(prog, C.byref(n))
        buf = C.create_string_buffer(n.value)
        rtc.hiprtcGetCode.argtypes = [C.c_void_p, C.c_char_p]
        rtc.hiprtcGetCode(prog, buf)
        code = buf.raw
    rtc.hiprtcDestroyProgram.argtypes = [C.POINTER(C.c_void_p)]
    rtc.hiprtcDestroyProgram(C.byref(prog))
    return rc == 0, log.value.decode(errors="replace"), dt, code


def registers_of(code: bytes):
    """-> (vgpr_count, agpr_count, max_flat_workgroup_size) of the first kernel in a code object (its metadata note)."""
    import os
    import re
    import subprocess
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".hsaco", delete=False) as f:
        f.write(code)
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    finally:
        os.unlink(f.name)
    g = lambda key: int(re.search(r"\.%s:\s*(\d+)" % key, out).group(1))
    return g("vgpr_count"), g("agpr_count"), g("max_flat_workgroup_size")


def fits_register_file(code: bytes, waves: int) -> bool:
    """gfx950: 512 unified registers per lane and SIMD, four SIMDs per compute unit — ceil(waves / 4) wavefronts of a workgroup
    share one (the rule of csrc/fused_rtc.h RtcKernel::fits)."""
    v, a, bound = registers_of(code)
    return v * ((waves + 3) // 4) <= 512
