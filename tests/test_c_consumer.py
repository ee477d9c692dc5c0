"""A second, plain-C consumer of the oracle-level C ABI (the reference registers
the same callbacks with Knitro, knitro_nlpif.py:211-309): tests/c_consumer/consumer.c includes
include/dnlp_hip.h, links the library, and reaches the reference-held optima with its OWN
interior-point solver over IPOPT-C-interface-shaped callbacks — dnlp_solve is never called."""
import os
import subprocess
import tempfile
import warnings

import numpy as np
import pytest

from golden_util import build_canonical

HERE = os.path.dirname(os.path.abspath(__file__))
CDIR = os.path.join(HERE, "c_consumer")

# name -> check(x by variable name, objective of the canonical minimisation)
def _check_hs071(vals, obj):
    x = [v for v in vals.values() if v.shape == (4,)][0]
    assert np.allclose(x, [0.75450865, 4.63936861, 3.78856881, 1.88513184], atol=1e-6)   # test_nlp_solvers.py:37


def _check_readme(vals, obj):
    np.random.seed(0)
    A = np.random.randn(3, 3)
    A = A.T @ A
    lam = np.linalg.eigvalsh(A)[-1]
    assert abs(-obj - lam) <= 1e-6 * lam                                                  # README.md:50-52
    x = [v for v in vals.values() if v.shape == (3,)][0]
    assert abs(abs(x @ np.linalg.eigh(A)[1][:, -1]) - 1.0) <= 1e-6


def _check_qcp(vals, obj):
    assert abs(obj + 0.32699284) <= 1e-6                                                  # test_nlp_solvers.py:111


def _check_socp(vals, obj):
    assert abs(obj + 13.548638814247532) <= 1e-6 * 13.5                                   # test_nlp_solvers.py:151


CASES = {"hs071": _check_hs071, "readme_toy": _check_readme, "qcp": _check_qcp, "socp": _check_socp}


def _run(binary, name, device=None):
    from dnlp_amd.tape import serialize
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        data, inv = build_canonical(name)
    with tempfile.NamedTemporaryFile(suffix=".blob", delete=False) as fh:
        fh.write(serialize(data["tape_arrays"]))
        path = fh.name
    try:
        cmd = [binary, path] + ([str(device)] if device is not None else [])
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    finally:
        os.unlink(path)
    assert out.returncode == 0, (out.stdout, out.stderr)
    tok = out.stdout.split()
    status, iters, obj = int(tok[0]), int(tok[1]), float(tok[2])
    x = np.array([float(t) for t in tok[3:]])
    assert status == 0 and x.size == len(data["x0"])
    vals = {}
    for v in data["problem"].variables():
        off = inv.var_offsets[v.id]
        vals[v.name()] = x[off:off + v.size].reshape(v.shape, order="F")
    return vals, obj, iters


def _build(target):
    subprocess.check_call(["make", "-C", CDIR, "-s", target])
    return os.path.join(CDIR, target)


def test_header_compiles_as_c99_and_consumer_links_the_cpu_oracle():
    """Without a GPU: the header is valid C99 (-pedantic) and the consumer's own solver reaches the
    known optima through the same callbacks bound to the CPU oracle's orc_* entry points."""
    from oracle.oracle_capi import build
    build()
    binary = _build("consumer_oracle")
    for name, check in CASES.items():
        vals, obj, iters = _run(binary, name)
        check(vals, obj)
        assert iters < 200


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_c_consumer_drives_the_device_oracles(name, gpu_required):
    binary = _build("consumer")
    vals, obj, iters = _run(binary, name, device=0)
    CASES[name](vals, obj)


@pytest.mark.gpu
def test_c_consumer_streams_batches_through_the_library(gpu_required):
    """dnlp_batch_stream_* from plain C (include/dnlp_hip.h; the role of the serial loop problems/problem.py:1256-1269):
    eight fresh batches of 2048 localization instances, two launches in flight inside the library — no Python thread,
    one handle — give bit for bit the results of dnlp_solve_batch_theta one launch at a time, in clearly less time."""
    import batch_problems as bp
    from dnlp_amd.batch import ParametricBatch
    from dnlp_amd.tape import serialize
    prob, params, sample, _ = bp.template_localization()
    pb = ParametricBatch(prob, params)
    assert pb.affine
    nb, B = 8, 2048
    thetas = np.stack([sample(i) for i in range(nb * B)])
    D = pb.D.tocsr()
    D.sort_indices()
    binary = _build("batch_stream")
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "tape.blob"), "wb") as fh:
            fh.write(serialize(pb.arrays0))
        with open(os.path.join(td, "map.bin"), "wb") as fh:
            fh.write(np.array([pb.d0.size, pb.P], np.int64).tobytes())
            fh.write(np.ascontiguousarray(pb.d0, np.float64).tobytes())
            fh.write(np.ascontiguousarray(pb.theta0, np.float64).tobytes())
            fh.write(np.ascontiguousarray(D.indptr, np.int64).tobytes())
            idx = np.ascontiguousarray(D.indices, np.int32).tobytes()
            fh.write(idx + b"\0" * (-len(idx) % 8))
            fh.write(np.ascontiguousarray(D.data, np.float64).tobytes())
        with open(os.path.join(td, "thetas.bin"), "wb") as fh:
            fh.write(np.ascontiguousarray(thetas, np.float64).tobytes())
        out = subprocess.run([binary, os.path.join(td, "tape.blob"), os.path.join(td, "map.bin"), os.path.join(td, "thetas.bin"),
                              "0", str(nb), str(B), "2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout, out.stderr)
    t_serial, t_stream, same, optimal = out.stdout.split()
    assert int(same) == 1
    assert int(optimal) >= 0.999 * nb * B
    assert float(t_stream) < 0.8 * float(t_serial), out.stdout
    print("one at a time %.1f k problems/s, two in flight %.1f k" % (nb * B / float(t_serial) / 1e3, nb * B / float(t_stream) / 1e3))
