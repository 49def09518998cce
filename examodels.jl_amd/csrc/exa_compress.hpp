// exa_compress.hpp — compressed (duplicate-summed) COO support, see exa_compress.cpp
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <stdexcept>
#include <string>

namespace exa {

struct CompressedCOO {
    int64_t nnz = 0, cnnz = 0;
    void *perm = nullptr;   // uint32[nnz]   original slot of the j-th entry in (col,row)-sorted order
    void *ptr = nullptr;    // int64[cnnz+1] first sorted position of every distinct (row,col)
    void *rows = nullptr;   // int64[cnnz]
    void *cols = nullptr;   // int64[cnnz]
    void release();
};

void build_compressed(CompressedCOO &c, const int64_t *rows, const int64_t *cols, int64_t nnz, int64_t nrowdim, int64_t ncoldim,
                      hipStream_t stream);
void compress_values(const CompressedCOO &c, const double *buf, double *V, hipStream_t stream);
void compressed_structure(const CompressedCOO &c, void *rows, void *cols, bool wide, hipStream_t stream);

}  // namespace exa
