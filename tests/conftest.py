import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


LIB_PROBLEM = None


def pytest_sessionstart(session):
    """The suite goes through libkeds_hip.so nearly everywhere (there is no CPU fallback): build it in-tree if a fresh
    checkout has not done so yet (hipcc cross-compiles without a GPU; on the GPU box the prebuilt file travels with the
    snapshot).  On a machine without ROCm the build fails: then only the tests that need the library are skipped -- the
    oracle-vs-golden tests are pure torch / numpy and still run."""
    global LIB_PROBLEM
    from keds_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        try:
            _lib.build()
        except Exception as e:                                   # no hipcc / build error: reported per skipped test
            LIB_PROBLEM = f"libkeds_hip.so is missing and could not be built: {str(e)[:200]}"


def pytest_collection_modifyitems(config, items):
    if LIB_PROBLEM is None:
        return
    skip = pytest.mark.skip(reason=LIB_PROBLEM)
    for item in items:
        if "test_oracle_golden" not in item.nodeid:
            item.add_marker(skip)


def golden_path(name):
    return os.path.join(GOLDEN, name)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return dict(np.load(golden_path(name)))
    return load
