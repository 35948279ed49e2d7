import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement of the reference (test infrastructure; built on demand with gcc)."""
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def hip():
    """The product library through its host mirror; fails loudly when it is not built."""
    from onepiece_amd import _lib
    if not os.path.exists(_lib.SO_PATH):   # fresh checkout: compile in-tree (hipcc cross-compiles without a GPU)
        _lib.build()
    _lib.load()
    return _lib
