"""Host logic of the multi-GPU layer that needs no device (exa_plan_only models): the local-slice COO addressing
(exa_set_coo_local / exa_coo_slices), the stencil footprint of a shard (exa_shard_var_range) and the communicator hook
bookkeeping.  SURVEY §8e: COO slots are private to a data point, so the ranks' pieces must tile the global vector."""
import ctypes

import numpy as np
import pytest

from zoo import ZOO


@pytest.mark.parametrize("name", ["lv20", "rocket50", "acopf30", "mixed", "stepped", "conaug2d"])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_local_slices_tile_the_global_coo(libs, name, world):
    from exahip import ExaModel
    m = ExaModel(ZOO[name](), device=False)
    for hess in (False, True):
        nnz = m.meta.nnzh if hess else m.meta.nnzj
        owner = np.full(nnz, -1)
        for rank in range(world):
            m.set_shard(rank, world)
            m.set_coo_local(True)
            pieces = m.coo_slices(hess)
            local_n = m.local_nnzh if hess else m.local_nnzj
            assert local_n == sum(p[2] for p in pieces)
            pos = 0
            for k, (g0, l0, cnt) in enumerate(pieces):
                info = m.pattern_info(k)
                step = info["o2step"] if hess else info["o1step"]
                if cnt == 0:
                    continue
                assert l0 == pos if world > 1 else l0 == g0          # packed, pattern after pattern
                pos += cnt
                assert cnt % step == 0 and g0 >= (info["o2"] if hess else info["o1"])
                assert np.all(owner[g0:g0 + cnt] == -1), "two ranks own one slot"
                owner[g0:g0 + cnt] = rank
            # global addressing: the same pieces at their global positions, buffers of nnz entries
            m.set_coo_local(False)
            assert (m.local_nnzh if hess else m.local_nnzj) == nnz
            assert [(g, l, c) for g, l, c in m.coo_slices(hess) if c] == [(g, g, c) for g, _, c in pieces if c]
        assert np.all(owner >= 0), "a slot nobody owns"
    m.set_shard(0, 1)


def test_shard_var_range_is_the_stencil_footprint(libs):
    from exahip import ExaModel, models
    from exahip.dist import shard_range
    N, world = 1000, 4
    m = ExaModel(models.luksan_vlcek_model(N), device=False)
    assert m.shard_var_range() == (0, N)
    for rank in range(world):
        m.set_shard(rank, world)
        lo, hi = m.shard_var_range()
        # constraint i = 1..N-2 reads x[i], x[i+1], x[i+2]; objective i = 2..N reads x[i-1], x[i] (luksan.jl:21-23)
        c_lo, c_hi = shard_range(N - 2, rank, world)
        o_lo, o_hi = shard_range(N - 1, rank, world)
        want_lo = min(c_lo, o_lo)                  # 0-based: con point I reads x[I .. I+2], obj point I reads x[I .. I+1]
        want_hi = max(c_hi - 1 + 3, o_hi - 1 + 2)
        # owner-computes callbacks reach one stencil further: grad! of the variables [N*r/w, N*(r+1)/w) and the J'v / Hv
        # windows the rank owns are evaluated from whatever data points touch them
        assert want_lo - 252 <= lo <= want_lo and want_hi <= hi <= want_hi + 252, (rank, lo, hi, want_lo, want_hi)
        assert hi - lo <= N // world + 510
    # a model whose indices come from data columns may read anywhere
    a = ExaModel(ZOO["acopf30"](), device=False)
    a.set_shard(1, 2)
    assert a.shard_var_range() == (0, a.meta.nvar)


def test_communicator_bookkeeping_without_a_device(libs):
    from exahip import ExaModel, capi
    m = ExaModel(ZOO["lv20"](), device=False)
    assert m.comm_info() == (0, 1, "none")
    calls = []
    m.comm_hook(1, 3, lambda ptr, count, stream: calls.append(count) or 0)
    assert m.comm_info() == (1, 3, "hook")
    # the communicator fixes the shard
    with pytest.raises(capi.ExaHipError):
        m.set_shard(0, 2)
    with pytest.raises(capi.ExaHipError):
        m.comm_hook(0, 3, lambda *a: 0)
    m.comm_free()
    assert m.comm_info() == (1, 3, "none")          # the shard stays
    m.set_shard(0, 1)
    # bad arguments
    L = capi.lib()
    assert L.exa_comm_hook(m.id, 3, 3, None, None) == 1
    assert L.exa_comm_init(m.id, 0, 1, None) == 1
    assert L.exa_comm_attach(m.id, None) == 1
    assert L.exa_tune(m.id, 7, None, None) == 1
    assert L.exa_tune(m.id, 1, None, None) == 1     # planned without a device


def test_collective_plan_is_one_allgather_per_regular_vector(libs):
    """the owner-sharded vectors of Luksan-Vlcek (grad!, cons_nln!, the product windows, each pattern's COO
    slots) are completed by ONE in-place ncclAllGather each — equal pieces by construction (exahip.dist.shard_range), the last
    rank's surplus as a broadcast — and only irregular piece sets fall back to one broadcast per piece.  exa_collective_plan is the
    predicate's output; no device needed."""
    import ctypes
    from exahip import ExaModel, models
    from exahip.dist import shard_range

    def plan(m, which):
        buf = (ctypes.c_int64 * (4 * 256))()
        n = m._L.exa_collective_plan(m.id, which, buf, 256)
        assert 0 <= n <= 256
        return [tuple(buf[4 * k:4 * k + 4]) for k in range(n)]

    N = 100203                                        # not divisible: a surplus on the last rank (405 product windows of 248 variables)
    m = ExaModel(models.luksan_vlcek_model(N), device=False)
    assert plan(m, 1) == []                           # world 1: nothing to do
    for world in (2, 8):
        m.set_shard(world - 1, world)
        # grad!: variables in equal pieces + the surplus of the last rank
        c = N // world
        want = [(0, 0, c, -1)] + ([(1, c * world, N - c * world, world - 1)] if N % world else [])
        assert plan(m, 1) == want
        # cons_nln! / jprod: the rows of the one constraint pattern
        nc = N - 2
        c = nc // world
        assert plan(m, 2) == [(0, 0, c, -1)] + ([(1, c * world, nc - c * world, world - 1)] if nc % world else [])
        assert plan(m, 5) == plan(m, 2) == plan(m, 8)
        # Hessian COO: constraint pattern (6 slots per point) then objective pattern (3 per point), one all-gather each
        no = N - 1
        co = no // world
        got = plan(m, 4)
        assert got[0] == (0, 0, 6 * c, -1) and (0, 6 * nc, 3 * co, -1) in got and sum(1 for op in got if op[0] == 0) == 2
        assert all(op[0] in (0, 1) for op in got) and sum(op[2] * (world if op[0] == 0 else 1) for op in got) == m.meta.nnzh
        # the pieces the plan moves are exactly the ranks' shards
        for r in range(world):
            lo, hi = shard_range(nc, r, world)
            assert lo == r * c and (hi == (r + 1) * c or r == world - 1)
        # obj: one double summed
        assert plan(m, 0) == [(2, 0, 1, -1)]
        # J'v / Hv by windows: whole windows per rank -> all-gather of equal window ranges (+ the tail)
        for which in (6, 7):
            ops = plan(m, which)
            assert ops and ops[0][0] == 0 and sum(op[2] * (world if op[0] == 0 else 1) for op in ops) == N
        # (a window count the ranks divide evenly leaves the LAST piece shorter — its last window is clipped by nvar —: not regular,
        # the pieces then travel as broadcasts; the predicate is about what an in-place all-gather can address)
        # ... and partial sums (atomics) are an all-reduce of the whole vector
        m2 = ExaModel(ZOO["acopf30"](), device=False)
        m2.set_shard(0, world)
        assert plan(m2, 6) == [(2, 0, m2.meta.nvar, -1)] and plan(m2, 1) == [(2, 0, m2.meta.nvar, -1)]
    # irregular: a pattern with fewer points than ranks -> broadcasts of the pieces that exist
    m3 = ExaModel(models.luksan_vlcek_model(6), device=False)
    m3.set_shard(0, 8)
    ops = plan(m3, 2)
    assert ops and all(op[0] == 1 for op in ops) and sum(op[2] for op in ops) == 4
