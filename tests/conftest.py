import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False))


def relerr(got, want):
    """max |got - want| / max|want| over the last axis (a multipole/template row at a time)."""
    got, want = np.asarray(got), np.asarray(want)
    scale = np.max(np.abs(want), axis=-1, keepdims=True)
    scale = np.where(scale == 0, 1.0, scale)
    return float(np.max(np.abs(got - want) / scale))


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get
