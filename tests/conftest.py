import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def hip():
    """The HIP backend through its C ABI.  Fails loudly if the library is missing or no GPU is visible."""
    import torch
    from slambench_amd import _capi
    L = _capi.lib()
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    assert L.kfh_init() == 0, L.kfh_last_error()
    return L
