import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build them once before any test runs —
    hipcc cross-compiles libteloscan.so without a GPU, gcc builds the oracle.  (The product itself never builds
    or falls back: importing teloscope_amd without the library raises ImportError.)"""
    lib = os.path.join(ROOT, "teloscope_amd", "libteloscan.so")
    ora = os.path.join(ROOT, "oracle", "libteloscope_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(ora)):
        import __graft_entry__ as entry
        entry.build()


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle
