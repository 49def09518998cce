// exa_comm.hpp — RCCL entry points used by the runtime (exa_comm.cpp); librccl is dlopen'ed at first use
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/exahip.h"

namespace exa {

void rccl_unique_id(void *out128);                                   // throws std::runtime_error
void *rccl_comm_init(int rank, int world, const void *uid128);       // on the current HIP device
void rccl_comm_destroy(void *comm);
void rccl_comm_shape(void *comm, int *rank, int *world);
void rccl_allreduce_sum_f64(void *comm, double *buf, int64_t count, hipStream_t stream);     // in place

// A piece of an owner-sharded vector: `count` doubles at `off`, complete on rank `root`; pieces of one `set` (a pattern's rows or
// slots, a window space, the variable ranges) are what the ranks hold of ONE contiguous stretch.
struct Piece { int64_t off, count; int root, set; };
// One step of making such a vector whole on every rank, in place:
//   kind 0  in-place all-gather: every rank contributes `count` doubles, rank r's at off + r * count (ncclAllGather, the tuned
//           full-mesh primitive: chosen when a set's pieces are contiguous in rank order and equal — all but the last rank's,
//           which may be longer: its surplus travels as a broadcast);
//   kind 1  broadcast of [off, off + count) from `root` (pieces of unequal length, gaps, fewer pieces than ranks);
//   kind 2  all-reduce(sum) of [off, off + count) (vectors left as partial sums: what exa_collective_plan reports for them).
struct CollOp { int kind; int64_t off, count; int root; };
// the predicate + the plan: pure host logic (unit-tested through exa_collective_plan without a device)
std::vector<CollOp> plan_allgather(const std::vector<Piece> &pieces, int world);
void rccl_run_plan_f64(void *comm, double *buf, const std::vector<CollOp> &ops, int rank, hipStream_t stream);
int rccl_comm_count(void *comm);

}  // namespace exa
