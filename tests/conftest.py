import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "iterative-closest-point-avmi_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rot_err(Ra, ta, Rb, tb):
    """Frobenius distance between two rigid transforms (R | t)."""
    return float(np.sqrt(np.sum((np.asarray(Ra) - Rb) ** 2) + np.sum((np.asarray(ta) - tb) ** 2)))


class _LibOptions:
    """Switches of libicpmi.so for one test (include/icpmi.h: icpmi_set_option).  The library reads its ICPMI_*
    environment switches once, at first use, so a test changes them through the library, not through os.environ;
    same interface as monkeypatch.setenv / delenv, and everything touched is unset again afterwards."""

    def __init__(self):
        self.touched = set()

    def setenv(self, name, value):
        from icpmi import _lib
        _lib.set_option(name, value)
        self.touched.add(name)

    def delenv(self, name, raising=True):
        from icpmi import _lib
        _lib.set_option(name, None)
        self.touched.add(name)

    def undo(self):
        from icpmi import _lib
        for name in self.touched:
            _lib.set_option(name, None)


@pytest.fixture
def libopt():
    o = _LibOptions()
    yield o
    o.undo()
