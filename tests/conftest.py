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
    # the -fopenmp build with as many threads as this process may really use: libgomp sized itself when torch loaded it, from the
    # affinity mask alone - 256 threads inside a 16-CPU quota on a one-GPU box made the long oracle runs ~3x slower than need be
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(cores, int(float(q) / float(per))))
    except Exception:
        pass
    _orc.set_omp_threads(cores)
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
