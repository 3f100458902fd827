import os
import sys

import pytest

# torch bundles its own libamdhip64 (SONAME libamdhip64.so.7) but links it by the unversioned name, so it must be loaded
# BEFORE libgingr_hip.so pulls in /opt/rocm's copy -- otherwise the process ends up with two HIP runtimes and torch sees
# no GPU.  Tests that use torch.distributed next to the library rely on this order (bench.py imports torch first, too).
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    import gingr_amd as ga
    c = ga.Context(0)
    yield c
    c.close()
