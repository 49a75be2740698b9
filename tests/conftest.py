import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """The native libraries are built in-tree; build them once if a fresh checkout has none."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "raytracedshadows_amd", "librts.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "raytracedshadows_amd", "csrc"), "-j4"], check=True)
    if not os.path.exists(os.path.join(ROOT, "oracle", "librts_oracle.so")):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    yield
