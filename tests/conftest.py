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


class RankReplay:
    """Outputs of a G-way sharded model replayed rank by rank on ONE GPU, completed the way a communicator would: a
    callback that leaves OWNER PIECES (exa_shard_layout: complete values in disjoint pieces, the rest untouched) writes into
    one NaN-poisoned buffer shared by all ranks; one that leaves PARTIAL SUMS is added up."""

    def __init__(self, m, names_sizes, device):
        import torch
        self.m = m
        self.layout = {n: m.shard_layout(n) for n, _ in names_sizes}
        self.buf = {n: torch.full((max(1, k),), float("nan"), dtype=torch.float64, device=device) for n, k in names_sizes}
        self.acc = {n: 0.0 for n, _ in names_sizes}
        self.size = dict(names_sizes)

    def add(self, name, call):
        """call(out) runs the callback of the current rank into `out` (a device tensor) and returns it"""
        import torch
        if self.layout[name] == "pieces":
            call(self.buf[name])
        else:
            tmp = torch.empty_like(self.buf[name])
            self.acc[name] = self.acc[name] + call(tmp).cpu().numpy()[:self.size[name]]

    def result(self, name):
        import torch
        torch.cuda.synchronize()
        return self.buf[name].cpu().numpy()[:self.size[name]] if self.layout[name] == "pieces" else self.acc[name]
