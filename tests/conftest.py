import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_collection_modifyitems(config, items):
    """A GPU test that stops making progress must fail, not hang the box: 300 s per test when pytest-timeout is
    installed (it is on the image)."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(300))


@pytest.fixture(scope="session", autouse=True)
def _torch_cuda_first(request):
    """When GPU tests are selected, let torch initialise its HIP state before the library has created and destroyed
    dozens of contexts in this process: a first torch.cuda use late in such a process has been seen to fail with
    "No HIP GPUs are available" on the GPU boxes (only the tests that borrow device memory from torch need it)."""
    if any(item.get_closest_marker("gpu") for item in request.session.items):
        try:
            import torch

            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:  # the tests that need torch will say so themselves
            pass
    yield
