import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def scene():
    from slam_amd import synth
    return synth.PlaneScene()


@pytest.fixture(scope="session")
def intr640():
    from slam_amd import synth
    return synth.default_intrinsics(640, 480)


@pytest.fixture(scope="session", autouse=True)
def _torch_owns_the_gpu_first(request):
    """Some GPU tests hold device buffers in torch tensors.  torch's lazy HIP initialisation has been seen to report
    "No HIP GPUs are available" when it first runs after the library's own contexts came and went in the same process,
    so initialise it once, before any test, whenever a GPU run is requested.  Nothing happens on the CPU-only tier."""
    if any(item.get_closest_marker("gpu") for item in request.session.items):
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    yield
