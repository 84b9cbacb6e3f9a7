import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def make_clip(seed, n):
    """Seeded synthetic clip + bits; same generator as tools/make_golden.py."""
    import numpy as np
    rng = np.random.default_rng(seed)
    audio = (0.1 * rng.standard_normal(n)).astype(np.float32)
    bits = rng.integers(0, 2, 20).astype(np.int32)
    return audio, bits
