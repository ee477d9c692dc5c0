import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    try:
        from dnlp_amd import _capi
        return _capi.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu_required():
    if not _gpu_present():
        pytest.fail("libdnlp_hip.so missing or no MI355X visible: GPU tests cannot fall back")
