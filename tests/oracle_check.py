"""TEST INFRASTRUCTURE: the one place outside tests/test_*.py through which measurement tools reach
the CPU oracle (oracle/ may only be used from tests/, smoke() and bench.py's cpu_baseline leg).
tools/run_c5_batch.py --check and the cyipopt stand-in of tools/refshim import THIS module."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from oracle.oracle_capi import OracleProblem  # noqa: E402,F401


def oracle_solve(arrays, options=None):
    """Solve the tape `arrays` (dnlp_amd.tape.tape_arrays layout) on the host build of the solver."""
    from dnlp_amd.nlp_solver import HIPNLP
    from dnlp_amd.tape import serialize
    orc = OracleProblem(serialize(arrays))
    for k, v in dict(HIPNLP.DEFAULT_OPTIONS, **(options or {})).items():
        orc.set_option(k, v)
    return orc.solve(arrays["x0"])
