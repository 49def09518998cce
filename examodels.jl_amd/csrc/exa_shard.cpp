// exa_shard.cpp — the sharded model behind the C ABI: who owns what (pieces), the collectives that make a vector whole
// (RCCL / host hook), resharding, and the exa_comm_* / exa_shard_* / exa_coo_* entry points (include/exahip.h).  Split off exa_runtime.cpp
// in round 4; the shared state is Handle (exa_rt.hpp).
#include "exa_rt.hpp"

using namespace exa;
using namespace exa::rt;

namespace exa {
namespace rt {


// Completes a partial result of a sharded model: sum over the ranks, in place, on the model's stream — RCCL
// (exa_comm_init / exa_comm_attach) or the host's reducer (exa_comm_hook).  A model without a communicator returns its
// partial sums (exa_set_shard alone: the host layer reduces).
void allreduce(Handle &h, double *buf, int64_t count) {
    if (!h.reduce || count <= 0) return;
    if (h.nccl) rccl_allreduce_sum_f64(h.nccl, buf, count, h.stream);
    else if (h.hook) {
        const int rc = h.hook(h.hook_ctx, buf, count, (void *)h.stream);
        if (rc != 0) throw std::runtime_error("the host's all-reduce hook returned status " + std::to_string(rc));
    }
}

// Windows [w0, w1) a rank of a sharded model owns, and the pieces of the output they cover (owner computes: complete
// values, nothing to sum).  pieces: (offset, count, owner rank) for EVERY rank — what an all-gather-v needs.
void owned_windows(const Handle &h, const Handle::Window &w, int rank, int64_t *w0, int64_t *w1) {
    *w0 = part_lo(w.nwin, rank, h.world);
    *w1 = part_lo(w.nwin, rank + 1, h.world);
}
std::vector<Piece> window_pieces(const Handle &h, const Handle::Window &w) {
    std::vector<Piece> out;
    for (int r = 0; r < h.world; r++) {
        int64_t w0, w1;
        owned_windows(h, w, r, &w0, &w1);
        for (size_t q = 0; q < w.spaces.size(); q++) {
            const auto &sp = w.spaces[q];
            const int64_t a = std::min(sp.o + w0 * sp.W, sp.end), b = std::min(sp.o + w1 * sp.W, sp.end);
            if (b > a) out.push_back({a, b - a, r, (int)q});
        }
    }
    return out;
}
// Makes a vector whole whose pieces are complete on their owners (in place): RCCL — per set of pieces ONE in-place ncclAllGather
// where they are regular (plan_allgather, exa_comm.cpp), grouped broadcasts otherwise; a host reducer (exa_comm_hook) only
// knows how to sum, so the other ranks' pieces are zeroed and the covering range summed.
void allgatherv(Handle &h, double *buf, const std::vector<Piece> &pieces, bool force) {
    // (world 1: nothing to do — except for a FORCED gather, exa_allgather_coo, over a real communicator: the one-piece plan is issued, so that
    // the call exercises ncclAllGather in place on a single-GPU machine)
    if ((!h.reduce && !force) || (h.world == 1 && !(force && h.nccl)) || pieces.empty()) return;
    if (h.nccl) {
        rccl_run_plan_f64(h.nccl, buf, plan_allgather(pieces, h.world), h.rank, h.stream);
    } else if (h.hook) {
        int64_t lo = INT64_MAX, hi = 0;
        for (const Piece &q : pieces) {
            if (q.root != h.rank) HIPCHK(hipMemsetAsync(buf + q.off, 0, 8 * (size_t)q.count, h.stream));
            lo = std::min(lo, q.off); hi = std::max(hi, q.off + q.count);
        }
        const int rc = h.hook(h.hook_ctx, buf + lo, hi - lo, (void *)h.stream);
        if (rc != 0) throw std::runtime_error("the host's all-reduce hook returned status " + std::to_string(rc));
    }
}
std::vector<Piece> var_pieces(const Handle &h) {
    std::vector<Piece> out;
    for (int r = 0; r < h.world; r++) out.push_back({own_var_lo(h, r), own_var_lo(h, r + 1) - own_var_lo(h, r), r, 0});
    return out;
}
// constraint rows the ranks own: the base rows of their data points, pattern by pattern
std::vector<Piece> row_pieces(const Handle &h) {
    std::vector<Piece> out;
    for (int r = 0; r < h.world; r++)
        for (size_t k = 0; k < h.m->pats.size(); k++) {
            const Pattern &p = h.m->pats[k];
            if (p.kind != EXA_PAT_CON || p.n <= 0) continue;
            const int64_t lo = part_lo(p.n, r, h.world), hi = part_lo(p.n, r + 1, h.world);
            if (hi > lo) out.push_back({p.o0 + lo, hi - lo, r, (int)k});
        }
    return out;
}
// slots of the Jacobian / Hessian COO the ranks own (global positions)
std::vector<Piece> coo_pieces(const Handle &h, bool hess) {
    std::vector<Piece> out;
    for (int r = 0; r < h.world; r++)
        for (size_t k = 0; k < h.m->pats.size(); k++) {
            const Pattern &p = h.m->pats[k];
            const int64_t step = hess ? p.o2step : (p.kind != EXA_PAT_OBJ ? p.o1step : 0);
            if (step <= 0 || p.n <= 0) continue;
            const int64_t lo = part_lo(p.n, r, h.world), hi = part_lo(p.n, r + 1, h.world);
            if (hi > lo) out.push_back({(hess ? p.o2 : p.o1) + step * lo, step * (hi - lo), r, (int)k});
        }
    return out;
}

// new shard and/or COO addressing: the parameter table, and everything derived from the local COO, start over
void reshard(Handle &h, int rank, int world, bool coo_local) {
    if (h.on_device) HIPCHK(hipStreamSynchronize(h.stream));
    if ((h.nccl || h.hook) && (rank != h.rank || world != h.world)) throw BadInput("the model's communicator fixes its shard (exa_comm_free first)");
    h.rank = rank; h.world = world; h.coo_local = coo_local;
    fill_params(h);
    if (h.on_device) {
        drop_sorted(h, false); drop_sorted(h, true);
        for (auto &q : h.pl) { q.idx.release(); q.ready = false; }
        if (h.compressed) {
            h.cj.release(); h.ch.release(); h.compressed = false;
            for (Handle::Window *w : {&h.wj, &h.wh}) { w->ok = false; w->why.clear(); }
            h.sj.ok = h.sh.ok = false;
        }
        if (g_eager_setup) g_eager_setup(h);
    }
}

}  // namespace rt
}  // namespace exa

extern "C" {

int exa_set_shard(int id, int rank, int world) {
    if (world < 1 || rank < 0 || rank >= world) return 1;
    return guard(id, false, [&](Handle &h) {
        reshard(h, rank, world, h.coo_local);
    });
}

// ---- multi-GPU: collectives behind the ABI ----------------------------------------------------------------
int exa_comm_unique_id(void *out128) {
    if (!out128) return 1;
    try { rccl_unique_id(out128); return 0; } catch (const std::exception &e) { g_err = e.what(); return 2; }
}
int exa_comm_init(int id, int rank, int world, const void *unique_id128) {
    if (!unique_id128 || world < 1 || rank < 0 || rank >= world) return 1;
    return guard(id, true, [&](Handle &h) {
        if (h.nccl || h.hook) throw BadInput("the model already has a communicator (exa_comm_free first)");
        // the communicator first: if it cannot be created (librccl missing, init failure) the model stays as it was — not
        // sharded without a communicator, returning partial results
        void *comm = rccl_comm_init(rank, world, unique_id128);    // collective over all ranks; on the current HIP device
        try { reshard(h, rank, world, h.coo_local); } catch (...) { try { rccl_comm_destroy(comm); } catch (...) {} throw; }
        h.nccl = comm;
        h.nccl_owned = true;
    });
}
int exa_comm_attach(int id, void *nccl_comm) {
    if (!nccl_comm) return 1;
    return guard(id, true, [&](Handle &h) {
        if (h.nccl || h.hook) throw BadInput("the model already has a communicator (exa_comm_free first)");
        int rank = 0, world = 1;
        rccl_comm_shape(nccl_comm, &rank, &world);
        reshard(h, rank, world, h.coo_local);
        h.nccl = nccl_comm;
        h.nccl_owned = false;
    });
}
int exa_comm_hook(int id, int rank, int world, exa_allreduce_fn fn, void *ctx) {
    if (!fn || world < 1 || rank < 0 || rank >= world) return 1;
    return guard(id, false, [&](Handle &h) {
        if (h.nccl || h.hook) throw BadInput("the model already has a communicator (exa_comm_free first)");
        reshard(h, rank, world, h.coo_local);
        h.hook = fn; h.hook_ctx = ctx;
    });
}
int exa_comm_free(int id) {
    return guard(id, false, [&](Handle &h) {
        if (h.on_device) HIPCHK(hipStreamSynchronize(h.stream));
        if (h.nccl && h.nccl_owned) rccl_comm_destroy(h.nccl);
        h.nccl = nullptr; h.nccl_owned = false; h.hook = nullptr; h.hook_ctx = nullptr;
    });
}
int exa_comm_info(int id, int *rank, int *world, int *kind) {
    Handle *h = get(id);
    if (!h) return 1;
    if (rank) *rank = h->rank;
    if (world) *world = h->world;
    if (world && h->nccl) { try { *world = rccl_comm_count(h->nccl); } catch (...) {} }      // what RCCL itself says (ncclCommCount): the ranks it saw
    if (kind) *kind = h->nccl ? 1 : (h->hook ? 2 : 0);
    return 0;
}
int exa_set_reduce(int id, int on) { return guard(id, false, [&](Handle &h) { h.reduce = on != 0; }); }
int exa_allreduce(int id, double *dev_buf, int64_t count) {
    if (!dev_buf || count < 0) return 1;
    return guard(id, true, [&](Handle &h) {
        if (!h.nccl && !h.hook) throw BadInput("the model has no communicator");
        const bool r = h.reduce;
        h.reduce = true;
        try { allreduce(h, dev_buf, count); } catch (...) { h.reduce = r; throw; }
        h.reduce = r;
    });
}
int exa_set_coo_local(int id, int on) { return guard(id, false, [&](Handle &h) { reshard(h, h.rank, h.world, on != 0); }); }
int64_t exa_local_nnzj64(int id) { Handle *h = get(id); return h ? h->lnnzj : -1; }
int64_t exa_local_nnzh64(int id) { Handle *h = get(id); return h ? h->lnnzh : -1; }
int exa_coo_slices(int id, int hess, int64_t *out) {
    Handle *h = get(id);
    if (!h || !out) return 1;
    const Model &m = *h->m;
    for (size_t k = 0; k < m.pats.size(); k++) {
        const Pattern &p = m.pats[k];
        const int64_t lo = part_lo(p.n, h->rank, h->world), hi = part_lo(p.n, h->rank + 1, h->world);
        const bool has = hess ? p.o2step > 0 : (p.kind != EXA_PAT_OBJ && p.o1step > 0);
        const int64_t step = hess ? p.o2step : p.o1step, o = hess ? p.o2 : p.o1, cnt = has ? step * (hi - lo) : 0;
        out[3 * k] = o + step * lo;                                                     // first global slot (0-based)
        out[3 * k + 1] = h->coo_local && h->world > 1 ? (hess ? h->lo2[k] : h->lo1[k]) : o + step * lo;   // where it is in the caller's buffer
        out[3 * k + 2] = cnt;
    }
    return 0;
}
int exa_shard_var_range(int id, int64_t *lo_out, int64_t *hi_out) {
    Handle *h = get(id);
    if (!h || !lo_out || !hi_out) return 1;
    const Model &m = *h->m;
    const ParamLayout &L = h->gen.layout;
    int64_t vmin = INT64_MAX, vmax = INT64_MIN;
    bool anywhere = false;
    auto add = [&](const Pattern &p, int64_t lo, int64_t hi) {
        if (hi <= lo || anywhere) return;
        int64_t a = 0, b = 0;
        if (!pattern_var_range(p, lo, hi, &a, &b)) { anywhere = true; return; }      // data-indexed: anywhere
        if (a <= b) { vmin = std::min(vmin, a); vmax = std::max(vmax, b); }
    };
    for (size_t k = 0; k < m.pats.size(); k++) {
        const Pattern &p = m.pats[k];
        if (p.n <= 0) continue;
        const int64_t lo = part_lo(p.n, h->rank, h->world), hi = part_lo(p.n, h->rank + 1, h->world);
        // a shard holding nothing of a pattern still re-reads one point of it (the branch-free loads of the chained
        // kernels clamp there): the last point before the shard, or point 0
        const int64_t lo_ = hi > lo ? lo : (hi > 0 ? hi - 1 : 0), hi_ = hi > lo ? hi : lo_ + 1;
        add(p, lo_, hi_);
        if (h->world == 1) continue;
        // owner-computes callbacks reach beyond the shard's own data points:
        //   cons_nln! / jprod in one launch: a row's owner evaluates the row's augmentation terms wherever they come from;
        if (p.kind == EXA_PAT_CONAUG && (h->cons1 || !h->on_device)) add(p, 0, p.n);
        //   grad!: the points of a gathered objective pattern that touch the variables this rank owns;
        if (std::find(L.pull.begin(), L.pull.end(), (int)k) != L.pull.end()) add(p, h->P[L.pat[k].qlo], h->P[L.pat[k].qhi]);
    }
    //   J'v / Hv by windows: the points that touch the windows this rank owns
    for (int wk : {WK_JTPROD, WK_HPROD}) {
        const Handle::Window &w = h->wp[wk - WK_JTPROD];
        if (h->world == 1 || !w.planned || w.nx || w.has_shared || w.hR.empty()) continue;
        int64_t w0, w1;
        owned_windows(*h, w, h->rank, &w0, &w1);
        const WindowMatrix &wm = h->pspec.mat[wk];
        std::vector<int> pk;            // R is [window][pass] (one space) or [block][pattern] (block-owned)
        for (const auto &wp : wm.pats) if (wm.nspaces == 0 || std::find(pk.begin(), pk.end(), wp.k) == pk.end()) pk.push_back(wp.k);
        for (size_t q = 0; q < pk.size(); q++) {
            int64_t lo = INT64_MAX, hi = INT64_MIN;
            for (int64_t j = w0; j < w1; j++) {
                const int32_t a = w.hR[(j * pk.size() + q) * 2], b = w.hR[(j * pk.size() + q) * 2 + 1];
                if (b > a) { lo = std::min<int64_t>(lo, a); hi = std::max<int64_t>(hi, b); }
            }
            if (hi > lo) add(m.pats[pk[q]], lo, hi);
        }
    }
    if (anywhere) { *lo_out = 0; *hi_out = m.nvar; return 0; }
    if (vmin > vmax) { *lo_out = 0; *hi_out = 0; return 0; }
    *lo_out = vmin - 1; *hi_out = vmax;       // 0-based [lo, hi)
    return 0;
}
/* How a sharded model's rank leaves the output of callback `which` when nothing completes it (no communicator, or
 * exa_set_reduce(id, 0)): 1 = OWNER PIECES — complete values in disjoint pieces (rows of its data points, variables / windows
 * it owns), nothing else written, an all-gather makes the vector whole; 0 = PARTIAL SUMS over the whole vector, an
 * all-reduce(sum) completes it.  which: 0 obj, 1 grad, 2 cons, 5 jprod, 6 jtprod, 7 hprod (3 jac / 4 hess: always pieces),
 * 8 the cons vector of exa_eval_fused / exa_eval_all (differs from 2 for models with non-linear augmentation terms).
 * -1 bad id / argument. */
int exa_shard_layout(int id, int which) {
    Handle *hh = get(id);
    if (!hh) return -1;
    Handle &h = *hh;
    switch (which) {
    case 0: return 0;
    case 1: return h.gen.layout.active[CB_GRAD].empty() && !h.gen.layout.pull.empty() ? 1 : 0;
    case 2: return rows_owner_complete(h) || !h.on_device ? 1 : 0;
    case 3: case 4: return 1;
    case 5: return (h.m->nconaug == 0 || (h.m->aug_linear && (h.cons1 || !h.on_device))) ? 1 : 0;
    case 6: case 7: {
        // owner pieces only when the call really runs the windows: an explicit or a tuned mode 0 / 1 (atomics, sorted gather)
        // leaves partial sums over the whole vector
        const bool hess = which == 7;
        const Handle::Window &w = h.wp[hess ? 1 : 0];
        const bool can = (h.on_device ? w.ok : w.planned) && w.nx == 0 && !w.has_shared;
        return can && product_mode_query(h, hess) == 2 ? 1 : 0;
    }
    // cons as exa_eval_fused / exa_eval_all leave it: rows complete on their owner only when the augmentation terms are linear
    // (added inside the sweep through the row lists); non-linear ones are partial sums there although exa_cons (which = 2)
    // completes the rows itself
    case 8: return h.m->nconaug == 0 || (h.m->aug_linear && (h.cons1 || !h.on_device)) ? 1 : 0;
    }
    return -1;
}
/* How the library completes (or a host layer should complete) the output of callback `which` of a sharded model: out <- up to cap
 * operations of 4 words — kind (0 in-place all-gather: every rank `count` doubles, rank r's at offset + r * count; 1 broadcast of
 * [offset, offset + count) from `root`; 2 all-reduce(sum) of [offset, offset + count)), offset, count, root (-1 unless kind 1).
 * Returns the number of operations of the plan (call again with a larger buffer when > cap), 0 for world 1 / nothing to do, -1 bad
 * argument.  which as exa_shard_layout: 0 obj, 1 grad, 2 cons, 3 jac COO, 4 hess COO (exa_allgather_coo), 5 jprod, 6 jtprod,
 * 7 hprod, 8 the cons vector of the fused sweeps.  Host logic only: works for plan-only handles (tests/test_shard_layout.py). */
static std::vector<CollOp> collective_ops(int id, Handle &h, int which) {
    const int layout = exa_shard_layout(id, which);
    std::vector<CollOp> ops;
    const Model &m = *h.m;
    auto reduce_all = [&](int64_t n) { if (n > 0) ops.push_back({2, 0, n, -1}); };
    switch (which) {
    case 0: reduce_all(1); break;
    case 1: if (layout == 1) ops = plan_allgather(var_pieces(h), h.world); else reduce_all(m.nvar); break;
    case 2: case 5: case 8: if (layout == 1) ops = plan_allgather(row_pieces(h), h.world); else reduce_all(m.ncon); break;
    case 3: case 4: ops = plan_allgather(coo_pieces(h, which == 4), h.world); break;
    case 6: case 7: if (layout == 1) ops = plan_allgather(window_pieces(h, h.wp[which - 6]), h.world); else reduce_all(m.nvar); break;
    }
    return ops;
}
int exa_collective_plan(int id, int which, int64_t *out, int cap) {
    Handle *hh = get(id);
    if (!hh || which < 0 || which > 8 || (cap > 0 && !out)) return -1;
    Handle &h = *hh;
    if (h.world == 1) return 0;
    const std::vector<CollOp> ops = collective_ops(id, h, which);
    for (size_t k = 0; k < ops.size() && (int)k < cap; k++) { out[4 * k] = ops[k].kind; out[4 * k + 1] = ops[k].off; out[4 * k + 2] = ops[k].count; out[4 * k + 3] = ops[k].root; }
    return (int)ops.size();
}
/* Deferred completion: issues, on the model's stream through the attached RCCL communicator, exactly the operations of
 * exa_collective_plan(which) on buf — what the callback does itself at its end unless exa_set_reduce(id, 0) switched that off (a host that
 * wants the collective of one callback to overlap the kernels of the next).  buf: the callback's output vector (global length).
 * With a WORLD-1 communicator the plan of one piece per vector is issued all the same — a real in-place ncclAllGather / ncclAllReduce of
 * the vector onto itself: how a single-GPU machine exercises the transport and the plan (tests/test_gpu_comm.py).  Returns 0, 1 (bad
 * argument, no RCCL communicator), 2. */
int exa_comm_complete(int id, int which, double *buf) {
    if (!buf || which < 0 || which > 8) return 1;
    return guard(id, true, [&](Handle &h) {
        if (!h.nccl) throw BadInput("exa_comm_complete needs an RCCL communicator (exa_comm_init)");
        rccl_run_plan_f64(h.nccl, buf, collective_ops(id, h, which), h.rank, h.stream);
    });
}
/* Makes a sharded Jacobian (hess = 0) / Hessian (hess = 1) COO vector whole on every rank: all-gather-v of the ranks' slot
 * ranges (a piece travels once; nothing is zero-filled or summed — an all-reduce of zero-padded vectors would move world x
 * the data).  `local`: what this rank's exa_jac / exa_hess wrote — the packed local slice (exa_set_coo_local) or
 * the global-length vector with this rank's slots in place; `global` [nnzj | nnzh]: receives everything (may equal `local`
 * when that is the global-length vector).  Needs a communicator; world 1: a device copy. */
int exa_allgather_coo(int id, int hess, const double *local, double *global) {
    if (!local || !global) return 1;
    return guard(id, true, [&](Handle &h) {
        const Model &m = *h.m;
        if (h.world > 1 && !h.nccl && !h.hook) throw BadInput("the model has no communicator");
        const bool packed = h.coo_local && h.world > 1;
        if (h.world == 1 || packed || local != global) {
            for (size_t k = 0; k < m.pats.size(); k++) {
                const Pattern &p = m.pats[k];
                const int64_t step = hess ? p.o2step : (p.kind != EXA_PAT_OBJ ? p.o1step : 0);
                if (step <= 0 || p.n <= 0) continue;
                const int64_t lo = part_lo(p.n, h.rank, h.world), hi = part_lo(p.n, h.rank + 1, h.world);
                const int64_t g0 = (hess ? p.o2 : p.o1) + step * lo, l0 = packed ? (hess ? h.lo2[k] : h.lo1[k]) : g0;
                if (hi > lo && local + l0 != global + g0)
                    HIPCHK(hipMemcpyAsync(global + g0, local + l0, 8 * (size_t)(step * (hi - lo)), hipMemcpyDeviceToDevice, h.stream));
            }
        }
        allgatherv(h, global, coo_pieces(h, hess != 0), true);
    });
}

}  // extern "C"
