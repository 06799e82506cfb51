import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _torch_runtime_first():
    """PyTorch bundles its own copy of the HIP runtime (same SONAME as /opt/rocm's, which libx264hip.so links): whichever
    is loaded first serves both.  Tests that hand torch tensors to the library initialise torch first, as bench.py does,
    so the order in which test modules run does not matter."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    yield


@pytest.fixture(scope="session", autouse=True)
def _build_native():
    """Build the checker (oracle) and, when the reference tree is present, oracle/_ref."""
    from oracle import oraclelib
    oraclelib.build()
    if os.path.isdir("/root/reference") and not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libx264ref8.so")):
        import subprocess
        subprocess.check_call(["bash", os.path.join(ROOT, "oracle", "build_ref.sh")])
    yield
