// exa_comm.cpp — RCCL behind the C ABI: the collectives of a sharded model run on the model's stream.
//
// A sharded model is sharded BY OWNER: what a callback writes belongs either to data points (COO slots: disjoint
// slices, no collective at all) or to an owner rank that computes it completely (the rows of its data points, a range of
// variables, a range of windows).  What needs a collective when every pattern's iterator is split over G GPUs:
//   obj                 1 double           all-reduce(sum)
//   grad!               owner pieces of nvar (range-affine objective) -> all-gather; partial sums (a data-indexed objective
//                       pattern scatters) -> all-reduce(sum)            (what KA ext :310-336 accumulates on one device)
//   cons_nln! / jprod   rows complete on their owner (exa_cons1 walks a row's augmentation terms itself) -> all-gather; only
//                       the two-stage path (rows collecting > 512 terms) leaves partial sums -> all-reduce (KA ext :273-308)
//   jtprod / hprod      owner-computes windows -> all-gather; atomics / sorted gather -> all-reduce(sum) of nvar
//   jac_coord! / hess_coord! / structures: NONE (exa_allgather_coo makes a sharded COO vector whole where a consumer wants it).
// "all-gather" = plan_allgather below: the ranks' pieces follow part_lo (equal pieces, the remainder on the last rank), so a
// vector of one piece per rank is ONE in-place ncclAllGather — the tuned primitive; xGMI is a point-to-point mesh and RCCL picks
// its schedule for it — plus a broadcast of the last rank's surplus; only irregular sets (gaps, empty pieces) fall back to one
// grouped ncclBroadcast per piece.  Everything of a call sits in one ncclGroupStart / End.
// librccl is loaded lazily with dlopen (by SONAME: a process that hosts PyTorch gets the copy PyTorch already loaded), so a
// single-GPU consumer needs no RCCL at all.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <mutex>
#include <stdexcept>
#include <string>

#include "exa_comm.hpp"

namespace exa {
namespace {

struct Rccl {
    void *lib = nullptr;
    std::string why_not;
    ncclResult_t (*get_unique_id)(ncclUniqueId *) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*all_reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*all_gather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*group_start)() = nullptr;
    ncclResult_t (*group_end)() = nullptr;
    ncclResult_t (*comm_count)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*comm_user_rank)(const ncclComm_t, int *) = nullptr;
    const char *(*error_string)(ncclResult_t) = nullptr;
};
Rccl &rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (!r.lib) { r.why_not = std::string("librccl could not be loaded: ") + (dlerror() ? dlerror() : "?"); return; }
        auto sym = [&](const char *n) { void *s = dlsym(r.lib, n); if (!s && r.why_not.empty()) r.why_not = std::string("librccl lacks ") + n; return s; };
        r.get_unique_id = (decltype(r.get_unique_id))sym("ncclGetUniqueId");
        r.comm_init_rank = (decltype(r.comm_init_rank))sym("ncclCommInitRank");
        r.comm_destroy = (decltype(r.comm_destroy))sym("ncclCommDestroy");
        r.all_reduce = (decltype(r.all_reduce))sym("ncclAllReduce");
        r.broadcast = (decltype(r.broadcast))sym("ncclBroadcast");
        r.all_gather = (decltype(r.all_gather))sym("ncclAllGather");
        r.group_start = (decltype(r.group_start))sym("ncclGroupStart");
        r.group_end = (decltype(r.group_end))sym("ncclGroupEnd");
        r.comm_count = (decltype(r.comm_count))sym("ncclCommCount");
        r.comm_user_rank = (decltype(r.comm_user_rank))sym("ncclCommUserRank");
        r.error_string = (decltype(r.error_string))sym("ncclGetErrorString");
        if (!r.why_not.empty()) { dlclose(r.lib); r.lib = nullptr; }
    });
    if (!r.lib) throw std::runtime_error(r.why_not);
    return r;
}
void chk(Rccl &r, ncclResult_t e, const char *what) {
    if (e != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + r.error_string(e));
}

}  // namespace

static_assert(EXA_UNIQUE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "exa_comm_unique_id hands out an ncclUniqueId");

void rccl_unique_id(void *out128) {
    Rccl &r = rccl();
    ncclUniqueId id;
    chk(r, r.get_unique_id(&id), "ncclGetUniqueId");
    memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
}
void *rccl_comm_init(int rank, int world, const void *uid128) {
    Rccl &r = rccl();
    ncclUniqueId id;
    memcpy(id.internal, uid128, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t c = nullptr;
    chk(r, r.comm_init_rank(&c, world, id, rank), "ncclCommInitRank");
    return c;
}
void rccl_comm_destroy(void *comm) {
    if (!comm) return;
    Rccl &r = rccl();
    (void)r.comm_destroy((ncclComm_t)comm);
}
void rccl_comm_shape(void *comm, int *rank, int *world) {
    Rccl &r = rccl();
    chk(r, r.comm_user_rank((ncclComm_t)comm, rank), "ncclCommUserRank");
    chk(r, r.comm_count((ncclComm_t)comm, world), "ncclCommCount");
}
void rccl_allreduce_sum_f64(void *comm, double *buf, int64_t count, hipStream_t stream) {
    if (count <= 0) return;
    Rccl &r = rccl();
    chk(r, r.all_reduce(buf, buf, (size_t)count, ncclFloat64, ncclSum, (ncclComm_t)comm, stream), "ncclAllReduce");
}
// The predicate: a set's pieces can go as ONE in-place all-gather when there is exactly one per rank, in rank order, back to
// back, and all but the last have the same positive length c (the last >= c).  Anything else: one broadcast per piece.
std::vector<CollOp> plan_allgather(const std::vector<Piece> &pieces, int world) {
    std::vector<CollOp> ops;
    std::vector<int> sets;
    for (const Piece &q : pieces) if (std::find(sets.begin(), sets.end(), q.set) == sets.end()) sets.push_back(q.set);
    for (int s : sets) {
        std::vector<Piece> ps;
        for (const Piece &q : pieces) if (q.set == s && q.count > 0) ps.push_back(q);
        bool regular = (int)ps.size() == world;
        for (int r = 0; regular && r < world; r++) {
            regular = ps[r].root == r && (r == 0 || ps[r].off == ps[r - 1].off + ps[r - 1].count);
            if (regular && r + 1 < world) regular = ps[r].count == ps[0].count;
            if (regular && r + 1 == world) regular = ps[r].count >= ps[0].count;
        }
        if (regular) {
            const int64_t c = ps[0].count;
            ops.push_back({0, ps[0].off, c, -1});
            if (ps[world - 1].count > c) ops.push_back({1, ps[world - 1].off + c, ps[world - 1].count - c, world - 1});
        } else {
            for (const Piece &q : ps) ops.push_back({1, q.off, q.count, q.root});
        }
    }
    return ops;
}
void rccl_run_plan_f64(void *comm, double *buf, const std::vector<CollOp> &ops, int rank, hipStream_t stream) {
    if (ops.empty()) return;
    Rccl &r = rccl();
    chk(r, r.group_start(), "ncclGroupStart");
    for (const CollOp &o : ops) {
        if (o.count <= 0) continue;
        if (o.kind == 0) chk(r, r.all_gather(buf + o.off + (int64_t)rank * o.count, buf + o.off, (size_t)o.count, ncclFloat64, (ncclComm_t)comm, stream), "ncclAllGather");
        else if (o.kind == 1) chk(r, r.broadcast(buf + o.off, buf + o.off, (size_t)o.count, ncclFloat64, o.root, (ncclComm_t)comm, stream), "ncclBroadcast");
        else chk(r, r.all_reduce(buf + o.off, buf + o.off, (size_t)o.count, ncclFloat64, ncclSum, (ncclComm_t)comm, stream), "ncclAllReduce");
    }
    chk(r, r.group_end(), "ncclGroupEnd");
}
int rccl_comm_count(void *comm) {
    Rccl &r = rccl();
    int n = 0;
    chk(r, r.comm_count((ncclComm_t)comm, &n), "ncclCommCount");
    return n;
}

}  // namespace exa
