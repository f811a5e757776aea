import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "mop-truss-marl_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")

SCENARIOS = {
    # fixture name -> (num_x, symmetry variant of truss2D_ENV.py)
    "train0": (6, None), "train3": (6, None), "train_eval": (8, None),
    "small_bridge": (8, "small"), "small_roof": (8, "small"),
    "large_bridge": (16, "large"), "large_roof": (16, "large"),
}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
