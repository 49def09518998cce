// exa_comm.cpp — RCCL behind the C ABI: the collectives of a sharded model run on the model's stream.
//
// What needs a collective when every pattern's iterator is split over G GPUs (one process per GPU):
//   obj        1 double            all-reduce(sum)
//   grad!      nvar doubles        all-reduce(sum)   (what KA ext :310-336 accumulates on one device)
//   cons_nln!  ncon doubles        all-reduce(sum)   (base rows are disjoint — zero outside the shard —, augmentation rows
//                                                      collect terms from every rank: KA ext :273-308)
//   jprod / jtprod / hprod         all-reduce(sum) of the product vector
//   jac_coord! / hess_coord! / structures: NONE — COO slots are private to a data point, ranks own disjoint slices.
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
    ncclResult_t (*all_gather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
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
        r.all_gather = (decltype(r.all_gather))sym("ncclAllGather");
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
void rccl_allgather_f64(void *comm, const double *send, double *recv, int64_t count_per_rank, hipStream_t stream) {
    if (count_per_rank <= 0) return;
    Rccl &r = rccl();
    chk(r, r.all_gather(send, recv, (size_t)count_per_rank, ncclFloat64, (ncclComm_t)comm, stream), "ncclAllGather");
}

}  // namespace exa
