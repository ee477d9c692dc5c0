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
