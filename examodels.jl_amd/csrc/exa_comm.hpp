// exa_comm.hpp — RCCL entry points used by the runtime (exa_comm.cpp); librccl is dlopen'ed at first use
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "../../include/exahip.h"

namespace exa {

void rccl_unique_id(void *out128);                                   // throws std::runtime_error
void *rccl_comm_init(int rank, int world, const void *uid128);       // on the current HIP device
void rccl_comm_destroy(void *comm);
void rccl_comm_shape(void *comm, int *rank, int *world);
void rccl_allreduce_sum_f64(void *comm, double *buf, int64_t count, hipStream_t stream);     // in place
void rccl_allgatherv_f64(void *comm, double *buf, const int64_t *off, const int64_t *count, const int *root, int npieces, hipStream_t stream);

}  // namespace exa
