"""Multi-GPU host layer: one process per GPU, every pattern's iterator sharded across ranks.

The reference has no multi-device path; this is new.  What needs communication and what does not:
  * jac_coord / hess_coord / structures: COO slots are private to a data point, so each rank fills a DISJOINT slice
    of the global vector — no collective.  `gather_coo` is offered for consumers that want the whole vector on
    every GPU, and is priced honestly in DESIGN.md (it moves ~G x more bytes than the evaluation itself).
  * obj: one double  -> all_reduce(SUM).
  * grad: dense nvar -> all_reduce(SUM) of the per-rank partial sums.
  * cons: base rows are disjoint by data point, augmentation rows are shared -> all_reduce(SUM) (rows outside the
    shard are zero on a rank, libexahip zero-fills when world > 1).
Collectives go through torch.distributed: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations


def shard_range(n: int, rank: int, world: int):
    """Data points [lo, hi) of a pattern with n points owned by `rank` — the same arithmetic as exa_set_shard."""
    return n * rank // world, n * (rank + 1) // world


class ShardedEvaluator:
    """Wraps a local evaluator (ExaModel, or any object with the same methods) whose iterators have been sharded with
    set_shard(rank, world), and completes the callbacks that need a reduction."""

    def __init__(self, model, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.model = model
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        model.set_shard(self.rank, self.world)

    def _allreduce(self, t):
        if self.world > 1:
            if t.is_cuda and self.dist.get_backend(self.group) == "gloo":
                # gloo is the CPU test backend: stage device tensors through the host (RCCL reduces them in place)
                h = t.cpu()
                self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def obj(self, x):
        import torch
        part = self.model.obj(x)
        dev = x.device if hasattr(x, "device") else "cpu"
        t = torch.tensor([part], dtype=torch.float64, device=dev)
        return self._allreduce(t).item()

    def grad(self, x, out=None):
        return self._allreduce(_as_tensor(self.model.grad(x, out=out)))

    def cons(self, x, out=None):
        return self._allreduce(_as_tensor(self.model.cons(x, out=out)))

    # sharded outputs: this rank's slice of the global COO vector is valid, the rest is untouched
    def jac_coord(self, x, out=None):
        return self.model.jac_coord(x, out=out)

    def hess_coord(self, x, y, obj_weight=1.0, out=None):
        return self.model.hess_coord(x, y, obj_weight, out=out)

    def gather_coo(self, buf):
        """Make a sharded COO vector whole on every rank.  `buf` must have been ZERO-filled before the sharded
        evaluation wrote into it; disjoint slices + zeros => all_reduce(SUM) is a gather."""
        return self._allreduce(_as_tensor(buf))


def _as_tensor(a):
    import torch
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(a)
