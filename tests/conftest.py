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
    from raytracedshadows_amd import build
    if not os.path.exists(os.path.join(ROOT, "raytracedshadows_amd", "librts.so")):
        build.build_product()
    if not os.path.exists(os.path.join(ROOT, "oracle", "librts_oracle.so")):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
    yield
