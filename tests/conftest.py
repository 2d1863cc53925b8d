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
    """The CPU oracle (test infrastructure; built on demand from oracle/lemon_oracle.c)."""
    from oracle import oracle as o
    o.build()
    return o


@pytest.fixture(scope="session")
def hip():
    """The HIP library, loaded through the C ABI; fails loudly when it has not been built."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no HIP device is visible")
    from lemon_amd import _lib
    _lib.load()
    import lemon_amd
    return lemon_amd
