import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The suite goes through libkeds_hip.so everywhere (there is no CPU fallback): build it in-tree if a fresh checkout
    has not done so yet (hipcc cross-compiles without a GPU; on the GPU box the prebuilt file travels with the snapshot)."""
    from keds_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()


def golden_path(name):
    return os.path.join(GOLDEN, name)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(golden_path(name)))
    return load
