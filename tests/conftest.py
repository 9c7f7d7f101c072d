import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def body_rel_err(a, ref):
    """Per-body ||a - ref|| / ||ref|| over the xyz lanes (float64)."""
    a = np.asarray(a, np.float64)[:, :3]
    ref = np.asarray(ref, np.float64)[:, :3]
    num = np.linalg.norm(a - ref, axis=1)
    den = np.linalg.norm(ref, axis=1)
    return num / np.maximum(den, 1e-300)


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.lib()
    return oracle
