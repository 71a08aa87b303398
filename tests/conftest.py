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
