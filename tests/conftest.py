import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "noguard: run without the out-of-bounds canaries of tests/guards.py (timing tests)")


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _oob_canaries(request):
    """Every `gpu` test runs with halo-guarded device allocations (tests/guards.py); the halos are checked when it ends."""
    if request.node.get_closest_marker("gpu") is None or request.node.get_closest_marker("noguard") is not None:
        yield None
        return
    import torch
    if not torch.cuda.is_available():
        yield None
        return
    from guards import GuardedAllocations
    with GuardedAllocations() as g:
        yield g
    n = g.verify()
    assert n >= 0
