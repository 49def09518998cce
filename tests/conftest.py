import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "examodels.jl_amd"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def libs():
    """Build (if stale) and load both shared libraries."""
    from exahip import capi
    import oracle
    capi.build()
    oracle.build()
    return capi.lib(), oracle.lib()


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
