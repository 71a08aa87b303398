"""GPU parity on RANDOM configurations (a fixed seed of tools/fuzz_parity.py): image sizes incl. odd ones, grids, pyramid
depths, FAST thresholds, distorted cameras, half-resolution depth, explicit initial estimates, both entry paths -- every
pair against the oracle as in tests/test_gpu_pair.py.  The sweep found the one-bucket-wide-level bug that the hand-picked
configurations of tests/test_gpu_configs.py had missed; run it longer with `python tools/fuzz_parity.py 300 <seed>`."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_random_configurations_against_the_oracle():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_parity
    bad, pairs, soft = fuzz_parity.run(40, 20261111)
    assert bad == 0 and pairs >= 100
