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
    // entries with more than 512 duplicates (a variable shared by every data point: the rocket's step length puts 1e6
    // contributions into ONE Hessian entry) are summed cooperatively, in a fixed order: per-chunk partial sums, then a fold
    void *long_list = nullptr;   // uint32[nlong]
    void *partial = nullptr;     // double[nlong * chunks]
    int64_t nlong = 0, maxlen = 0;
    void release();
    void release_gather();   // drops perm / ptr / long-group lists (the windowed sweep never reads them); rows / cols stay
};

void build_compressed(CompressedCOO &c, const int64_t *rows, const int64_t *cols, int64_t nnz, int64_t nrowdim, int64_t ncoldim,
                      hipStream_t stream);
void compress_values(const CompressedCOO &c, const double *buf, double *V, hipStream_t stream);
// permuted-store path: pos[slot] = position of the slot in the sorted order (the inverse of perm), and the reduction over
// a buffer that is ALREADY in sorted order: V[k] = sum of sorted[ptr[k] .. ptr[k+1]), ascending (= ascending original slot:
// the sort is stable), so the additions are the gather's, in the gather's order
void build_positions(const CompressedCOO &c, uint32_t *pos, hipStream_t stream);
void compress_sorted(const CompressedCOO &c, const double *sorted, double *V, hipStream_t stream);
// windowed fast path (exa_windows.cpp): cmap[e] = compressed entry of original slot e (int32, device)
void build_slot_map(const CompressedCOO &c, int32_t *cmap, hipStream_t stream);
// how many points I of [0, n) have a slot s with cmap[o + S*I + s] != a[s] + b[s]*I; e_lo = one past the last such point
// below mid (0 if none), e_hi = the first such point at or above mid (n if none).  Synchronises the stream.
void affine_exceptions(const int32_t *cmap, int64_t o, int S, int64_t n, const int64_t *a_host, const int64_t *b_host, int64_t mid, int64_t *count,
                       int64_t *e_lo, int64_t *e_hi, hipStream_t stream);
// CSC view of the compressed entries (they are sorted by column, then row): colptr[ncol+1] and rowval[cnnz], 1-based
void compressed_csc(const CompressedCOO &c, int64_t ncol, int64_t *colptr, int64_t *rowval, hipStream_t stream);
void compressed_structure(const CompressedCOO &c, void *rows, void *cols, bool wide, hipStream_t stream);

// COO entries grouped by one of their coordinates: entries perm[ptr[k] .. ptr[k+1]) have key k (0-based), in ascending
// slot order.  The reference's prod helper (`ExaModel(c; prod = true)`, KA ext :56-178) keeps the same two lists.
struct SortedIndex {
    int64_t nnz = 0, ndim = 0;
    void *perm = nullptr;   // uint32[nnz]
    void *ptr = nullptr;    // int64[ndim+1]
    // groups longer than kLongRow entries (a variable shared by many data points, e.g. the rocket's step length) are
    // reduced cooperatively: one workgroup per 8192-entry chunk, one atomic per chunk
    void *long_rows = nullptr;   // uint32[nlong]
    int64_t nlong = 0, maxlen = 0;
    // per sorted position: 0-based index of the OTHER coordinate of that entry (what v is gathered with), or
    // 0xffffffff for an entry this product skips — read sequentially instead of three random int64 loads per entry
    void *oth = nullptr;         // uint32[nnz], optional (attach_other)
    void *partial = nullptr;     // double[nlong * chunks]: per-chunk sums of the long groups, folded in chunk order (deterministic)
    void release();
};
// fills s.oth from the structure arrays (1-based int64, device); entries with rows[e] == cols[e] are marked skipped when
// skip_diag.  No-op (oth stays null) when an index does not fit 32 bits.
void attach_other(SortedIndex &s, const int64_t *other, const int64_t *rows, const int64_t *cols, bool skip_diag, int64_t other_dim,
                  hipStream_t stream);
void build_sorted_index(SortedIndex &s, const int64_t *keys1, int64_t nnz, int64_t ndim, hipStream_t stream);
// every entry is gathered with v[0] (a plain sum when v = {1.0}: x * 1.0 == x exactly): s.oth = zeros
void attach_unit(SortedIndex &s, hipStream_t stream);
// out[k] (+)= sum_{e in group k, (skip_diag ? rows[e] != cols[e] : true)} vals[e] * v[other[e] - 1]
void spmv_gather(const SortedIndex &s, const double *vals, const int64_t *other, const int64_t *rows, const int64_t *cols, bool skip_diag,
                 const double *v, double *out, bool accumulate, hipStream_t stream);

}  // namespace exa
