"""Multi-GPU host layer: one process per GPU, every pattern's iterator sharded across ranks.

The reference has no multi-device path; this is new.  What needs communication and what does not:
  * jac_coord / hess_coord / structures: COO slots are private to a data point, so each rank fills a DISJOINT slice
    of the global vector (or, with `coo_local`, a packed slice-sized buffer) — no collective (exa_allgather_coo for a
    consumer that wants the whole vector);
  * cons / jprod (rows), grad of range-affine objectives (variables), jtprod / hprod by windows: sharded by OWNER — ranks
    hold complete disjoint pieces, all-gather-v makes them whole;
  * obj (1 double), grad with data-indexed patterns, jtprod / hprod by atomics (nvar): all_reduce(SUM).
The collectives live BEHIND THE C ABI (include/exahip.h, exa_comm_*): libexahip enqueues `ncclAllReduce` on the model's
stream right after the kernels, so a Julia host gets the same multi-GPU path.  This module only distributes the
ncclUniqueId (through torch.distributed, whatever its backend) and, where there is no RCCL — the gloo CPU/one-GPU test
path — installs a host reducer through the same ABI hook (exa_comm_hook).
"""
from __future__ import annotations


def shard_range(n: int, rank: int, world: int):
    """Data points [lo, hi) of a pattern with n points owned by `rank` — the same arithmetic as exa_set_shard (part_lo,
    exa_internal.hpp): n // world points each, the last rank the remainder on top — equal pieces, so that an owner-sharded vector is
    completed by one in-place all-gather (exa_collective_plan); fewer than 16 points per rank: floor(n r / world)."""
    if n < 16 * world:          # few items: split evenly (balance before regularity)
        return n * rank // world, n * (rank + 1) // world
    per = n // world
    return per * rank, (n if rank + 1 >= world else per * (rank + 1))


def attach_communicator(model, group=None, transport=None, coo_local=False):
    """Shards `model` (an exahip.ExaModel) over the ranks of the torch.distributed group and gives it a communicator:
    transport "rccl" — exa_comm_init (rank 0's ncclUniqueId is broadcast through the group);
    transport "hook" — exa_comm_hook with a reducer that stages the device buffer through the host and all-reduces it
                       with the group's own backend (gloo): the test path on machines with fewer GPUs than ranks.
    Default: "rccl" when the group's backend is nccl, else "hook"."""
    import ctypes

    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if transport is None:
        transport = "rccl" if dist.get_backend(group) == "nccl" else "hook"
    if coo_local:
        model.set_coo_local(True)
    if transport == "rccl":
        box = [model.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        model.comm_init(rank, world, box[0])
    else:
        # the HIP runtime libexahip.so is bound to is already in the process; dlopen by SONAME returns that copy
        hip = ctypes.CDLL("libamdhip64.so.7")
        hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]

        def reducer(ptr, count, stream):
            # device -> host on the model's stream, host all-reduce, host -> device on the same stream
            host = torch.empty(count, dtype=torch.float64)
            nbytes = 8 * count
            if hip.hipMemcpyAsync(host.data_ptr(), ptr, nbytes, 2, stream):           # hipMemcpyDeviceToHost
                return 1
            if hip.hipStreamSynchronize(stream):
                return 1
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            if hip.hipMemcpyAsync(ptr, host.data_ptr(), nbytes, 1, stream):           # hipMemcpyHostToDevice
                return 1
            return 1 if hip.hipStreamSynchronize(stream) else 0     # `host` dies with this frame

        model.comm_hook(rank, world, reducer)
    return transport


class ShardedEvaluator:
    """A model whose iterators are sharded over the ranks of a torch.distributed group, complete on every rank.

    For an exahip.ExaModel the reductions happen inside libexahip (attach_communicator).  Any other object with the
    same methods and `set_shard` (the test oracle on CPU) is completed here with torch.distributed — the reference
    semantics the ABI path is checked against."""

    def __init__(self, model, group=None, transport=None, coo_local=False):
        import torch.distributed as dist
        self.dist = dist
        self.model = model
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.in_library = hasattr(model, "comm_hook") and dist.is_initialized()
        if self.in_library:
            self.transport = attach_communicator(model, group, transport, coo_local)
        else:
            self.transport = "host"
            model.set_shard(self.rank, self.world)

    def _allreduce(self, t):
        if self.world > 1:
            if t.is_cuda and self.dist.get_backend(self.group) == "gloo":
                h = t.cpu()
                self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
            else:
                self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def obj(self, x):
        import torch
        part = self.model.obj(x)
        if self.in_library:
            return part
        dev = x.device if hasattr(x, "device") else "cpu"
        return self._allreduce(torch.tensor([part], dtype=torch.float64, device=dev)).item()

    def _done(self, a):
        return _as_tensor(a) if self.in_library else self._allreduce(_as_tensor(a))

    def grad(self, x, out=None):
        return self._done(self.model.grad(x, out=out))

    def cons(self, x, out=None):
        return self._done(self.model.cons(x, out=out))

    def jprod(self, x, v, out=None):
        return self._done(self.model.jprod(x, v, out=out))

    def jtprod(self, x, v, out=None):
        return self._done(self.model.jtprod(x, v, out=out))

    def hprod(self, x, y, v, obj_weight=1.0, out=None):
        return self._done(self.model.hprod(x, y, v, obj_weight, out=out))

    # sharded outputs: this rank's slice of the COO vector is valid, the rest is untouched
    def jac_coord(self, x, out=None):
        return self.model.jac_coord(x, out=out)

    def hess_coord(self, x, y, obj_weight=1.0, out=None):
        return self.model.hess_coord(x, y, obj_weight, out=out)

    def gather_coo(self, buf, hess=True):
        """Make a sharded Hessian (hess=True) / Jacobian (hess=False) COO vector whole on every rank.  In the library:
        exa_allgather_coo, an all-gather-v of the ranks' slot ranges (each piece travels once).  The host fallback (the CPU
        oracle path) needs `buf` ZERO-filled before the sharded evaluation wrote into it: disjoint slices + zeros =>
        all_reduce(SUM) is a gather (moves ~world x the data)."""
        t = _as_tensor(buf)
        if self.in_library and t.is_cuda:
            return self.model.allgather_coo(t, hess=hess)
        return self._allreduce(t)

    def sum_buffer(self, buf):
        """all_reduce(SUM) of any device / host buffer of the caller's (exa_allreduce in the library)."""
        t = _as_tensor(buf)
        if self.in_library and t.is_cuda:
            return self.model.allreduce(t)
        return self._allreduce(t)


def _as_tensor(a):
    import torch
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(a)
