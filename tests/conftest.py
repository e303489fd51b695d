import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: minutes of CPU-oracle time (deselect with -m 'gpu and not slow')")


@pytest.fixture(scope="session")
def oracle():
    from oracle.ns_oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def gpu():
    """(ctx, torch device) -- the HIP library on cuda:0.  Fails, never skips, without a GPU."""
    import torch
    from navierstokes_amd import _lib
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    torch.cuda.set_device(0)
    ctx = _lib.Context(0, torch.cuda.current_stream().cuda_stream)
    yield ctx
    ctx.close()
