// exa_comm.cpp — RCCL behind the C ABI: the collectives of a sharded model run on the model's stream.
//
// What needs a collective when every pattern's iterator is split over G GPUs (one process per GPU):
//   obj        1 double            all-reduce(sum)
//   grad!      nvar doubles        all-reduce(sum)   (what KA ext :310-336 accumulates on one device)
//   cons_nln!  base rows are private to a data point (a slice per rank, all-gather-v to make them whole); only the rows
//              that augmentations add to collect terms from every rank: all-reduce(sum) of those row ranges (KA ext :273-308)
//   jprod / jtprod / hprod         all-reduce(sum) of the product vector — or, owner-computes windows (range-affine models):
//                                  every rank evaluates complete values for the variables it owns, all-gather-v
//   jac_coord! / hess_coord! / structures: NONE — COO slots are private to a data point, ranks own disjoint slices
//              (exa_allgather_coo makes a sharded COO vector whole where a consumer wants that).
// librccl is loaded lazily with dlopen (by SONAME: a process that hosts PyTorch gets the copy PyTorch already loaded),
// so a single-GPU consumer needs no RCCL at all.  xGMI is point-to-point; message sizes here are one dense vector per
// call, so the library issues ONE collective per callback on the whole vector and lets RCCL pick ring/tree.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

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
// All-gather of pieces of UNEQUAL length held in place: piece q of `buf` — count[q] doubles at offset off[q] — is owned by
// rank root[q] and ends up on every rank.  One grouped set of broadcasts (the all-gather-v idiom: a piece travels once over
// each link RCCL routes it through; nothing is zero-filled, nothing is summed).
void rccl_allgatherv_f64(void *comm, double *buf, const int64_t *off, const int64_t *count, const int *root, int npieces, hipStream_t stream) {
    Rccl &r = rccl();
    chk(r, r.group_start(), "ncclGroupStart");
    for (int q = 0; q < npieces; q++)
        if (count[q] > 0) chk(r, r.broadcast(buf + off[q], buf + off[q], (size_t)count[q], ncclFloat64, root[q], (ncclComm_t)comm, stream), "ncclBroadcast");
    chk(r, r.group_end(), "ncclGroupEnd");
}

}  // namespace exa
