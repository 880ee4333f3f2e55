import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on CPU")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def rel_fro(a, b):
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.sqrt((b * b).sum())
    return float(np.sqrt(((a - b) ** 2).sum()) / max(den, 1e-30))


@pytest.fixture
def tuning():
    """Kernel-choice overrides (pv_debug_set_tuning) that are reset after the test."""
    from vit_prisma_amd import _native
    yield _native.set_tuning
    _native.set_tuning("reset")
