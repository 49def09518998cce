// exa_compress.cpp — duplicate-summing ("compressed") COO for the Jacobian and the Hessian.
//
// Replaces CompressedNLPModel (src/utils.jl:425-579) and its device helpers (ext/ExaModelsKernelAbstractions.jl:
// 1290-1319): at set-up the (col,row) pairs of the partially compressed COO are sorted (stable LSD radix sort on the
// device, rocPRIM — a plain library sort, not a hot kernel), runs of equal pairs become one entry, and every
// evaluation is the ordinary fused kernel into an internal buffer followed by ONE gather
//     V[k] = sum_{j in ptr[k] .. ptr[k+1]-1} buffer[perm[j]]            (utils.jl:555-562)
// Order contract: entries are sorted by (col, row) — get_compressed_sparsity builds ((j, i), k) and sorts on the first
// component (utils.jl:476-478, 519-520) — and duplicates are added in ascending original slot order (stable sort).
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_run_length_encode.hpp>
#include <rocprim/device/device_scan.hpp>

#include "exa_compress.hpp"

namespace exa {

namespace {

#define HIPCHK_C(expr)                                                                                        \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if (_e != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e));     \
    } while (0)
// FP64 add to memory as ONE hardware instruction (global_atomic_add_f64), by builtin: what unsafeAtomicAdd compiles to
// depends on the compiler's flags (without -munsafe-fp-atomics: a compare-and-swap loop)
static __device__ __forceinline__ void add_f64(double *p, double v) {
    (void)__builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double *)p, v);
}

__global__ void __launch_bounds__(256) k_make_keys(const int64_t *__restrict__ rows, const int64_t *__restrict__ cols, int64_t nrowdim,
                                                   uint64_t *__restrict__ keys, uint32_t *__restrict__ idx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    keys[i] = (uint64_t)(cols[i] - 1) * (uint64_t)nrowdim + (uint64_t)(rows[i] - 1);   // column-major order
    idx[i] = (uint32_t)i;
}

__global__ void __launch_bounds__(256) k_decode(const uint64_t *__restrict__ ukeys, int64_t nrowdim, int64_t *__restrict__ rows,
                                                int64_t *__restrict__ cols, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    rows[i] = (int64_t)(ukeys[i] % (uint64_t)nrowdim) + 1;
    cols[i] = (int64_t)(ukeys[i] / (uint64_t)nrowdim) + 1;
}

// one thread per compressed entry; duplicates of an entry are few (LV Hessian: <= 5) and their slots are close
__global__ void __launch_bounds__(256) k_compress(double *__restrict__ V, const double *__restrict__ buf, const int64_t *__restrict__ ptr,
                                                  const uint32_t *__restrict__ perm, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int64_t b = ptr[k], e = ptr[k + 1];
    if (e - b > 512) return;                      // kLongRow: summed by k_compress_long / k_compress_fold
    // same additions in the same (slot) order as a plain loop — bit-exact against the reference — but eight gathers in
    // flight instead of one dependent round trip per duplicate (ACOPF bus diagonals collect 20-200 entries)
    double s = 0.0;
    int64_t j = b;
    for (; j + 8 <= e; j += 8) {
        uint32_t q[8];
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) q[u] = perm[j + u];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = buf[q[u]];
#pragma unroll
        for (int u = 0; u < 8; u++) s += a[u];
    }
    for (; j < e; j++) s += buf[perm[j]];
    V[k] = s;
}
// a matrix without duplicates (cnnz == nnz, ptr = identity): a permutation — four independent gathers per thread, no ptr reads
__global__ void __launch_bounds__(256) k_permute(double *__restrict__ V, const double *__restrict__ buf, const uint32_t *__restrict__ perm, int64_t n) {
    const int64_t k0 = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    uint32_t q[4];
    double a[4];
#pragma unroll
    for (int u = 0; u < 4; u++) q[u] = k0 + 256 * u < n ? perm[k0 + 256 * u] : 0u;
#pragma unroll
    for (int u = 0; u < 4; u++) a[u] = buf[q[u]];
#pragma unroll
    for (int u = 0; u < 4; u++) if (k0 + 256 * u < n) V[k0 + 256 * u] = a[u];
}
__global__ void __launch_bounds__(256) k_positions(const uint32_t *__restrict__ perm, uint32_t *__restrict__ pos, int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < n) pos[perm[j]] = (uint32_t)j;
}
// the duplicates of entry k are CONTIGUOUS in `sorted`: sequential reads, same order of additions as k_compress
__global__ void __launch_bounds__(256) k_compress_sorted(double *__restrict__ V, const double *__restrict__ sorted, const int64_t *__restrict__ ptr, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int64_t b = ptr[k], e = ptr[k + 1];
    double s = 0.0;
    int64_t j = b;
    for (; j + 8 <= e; j += 8) {
        double a[8];
#pragma unroll
        for (int u = 0; u < 8; u++) a[u] = sorted[j + u];
#pragma unroll
        for (int u = 0; u < 8; u++) s += a[u];
    }
    for (; j < e; j++) s += sorted[j];
    V[k] = s;
}
// entries with very many duplicates: blockIdx.x = long entry, blockIdx.y = chunk of 8192 sorted positions -> partial sum
__global__ void __launch_bounds__(256) k_compress_long(const uint32_t *__restrict__ list, const int64_t *__restrict__ ptr,
                                                       const uint32_t *__restrict__ perm, const double *__restrict__ buf,
                                                       double *__restrict__ partial, int chunks) {
    __shared__ double red[4];
    const int64_t k = list[blockIdx.x];
    const int64_t beg = ptr[k] + (int64_t)blockIdx.y * 8192;
    const int64_t end = beg + 8192 < ptr[k + 1] ? beg + 8192 : ptr[k + 1];
    double s = 0.0;
    for (int64_t j = beg + threadIdx.x; j < end; j += 256) s += buf[perm[j]];
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * chunks + blockIdx.y] = red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(256) k_compress_fold(const uint32_t *__restrict__ list, const double *__restrict__ partial, int chunks,
                                                       double *__restrict__ V, int64_t nlong) {
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= nlong) return;
    double s = 0.0;
    for (int c = 0; c < chunks; c++) s += partial[l * chunks + c];     // chunks past the entry's end hold 0
    V[list[l]] = s;
}

// ---- windowed fast path support (exa_windows.cpp: window_setup) ----------------------------------------------------
// cmap[e] = compressed entry of original slot e
__global__ void __launch_bounds__(256) k_slot_map(const int64_t *__restrict__ ptr, const uint32_t *__restrict__ perm, int32_t *__restrict__ cmap,
                                                  int64_t cnnz) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= cnnz) return;
    for (int64_t j = ptr[g]; j < ptr[g + 1]; j++) cmap[perm[j]] = (int32_t)g;
}
// points of a pattern whose slots are NOT at a[s] + b*I: count, smallest index >= mid and largest index < mid
__global__ void __launch_bounds__(256) k_affine_check(const int32_t *__restrict__ cmap, int64_t o, int S, int64_t n, const int64_t *__restrict__ a,
                                                      const int64_t *__restrict__ b, int64_t mid, unsigned long long *__restrict__ res) {
    const int64_t I = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (I >= n) return;
    bool bad = false;
    for (int s = 0; s < S; s++) bad = bad || (int64_t)cmap[o + (int64_t)S * I + s] != a[s] + b[s] * I;
    if (!bad) return;
    atomicAdd(&res[0], 1ull);
    if (I < mid) atomicMax(&res[1], (unsigned long long)(I + 1));      // e_lo candidate: one past the last bad point of the head
    else atomicMin(&res[2], (unsigned long long)I);                    // e_hi candidate: first bad point of the tail
}

// colptr (1-based, like SparseMatrixCSC) of the (col,row)-sorted compressed entries
__global__ void __launch_bounds__(256) k_colptr(const int64_t *__restrict__ cols, int64_t n, int64_t ncol, int64_t *__restrict__ colptr) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j > n) return;
    const int64_t prev = j == 0 ? -1 : cols[j - 1] - 1;
    const int64_t cur = j == n ? ncol : cols[j] - 1;
    for (int64_t q = prev + 1; q <= cur; q++) colptr[q] = j + 1;
}
template <class T>
__global__ void __launch_bounds__(256) k_narrow(const int64_t *__restrict__ src, T *__restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (T)src[i];
}

__global__ void __launch_bounds__(256) k_keys0(const int64_t *__restrict__ keys1, uint32_t *__restrict__ k0, uint32_t *__restrict__ idx, int64_t n, int64_t ndim,
                                               unsigned long long *__restrict__ bad) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t key = keys1[i];
    if (key < 1 || key > ndim) { atomicAdd(bad, 1ull); k0[i] = 0; }        // (counted; the caller refuses the list)
    else k0[i] = (uint32_t)(key - 1);
    idx[i] = (uint32_t)i;
}

// group starts from the SORTED keys, no atomics: position j opens every group in (sorted[j-1], sorted[j]]
// (a histogram with atomicAdd took 180 ms for the 9e7 entries of the LV Hessian: neighbouring entries share keys)
__global__ void __launch_bounds__(256) k_ptr_from_sorted(const uint32_t *__restrict__ sk, int64_t n, int64_t ndim, int64_t *__restrict__ ptr) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j > n) return;
    const int64_t prev = j == 0 ? -1 : (int64_t)sk[j - 1];
    const int64_t cur = j == n ? ndim : (int64_t)sk[j];
    for (int64_t q = prev + 1; q <= cur; q++) ptr[q] = j;
}

constexpr int64_t kLongRow = 512, kChunk = 8192, kMaxLong = 1 << 20;

__global__ void __launch_bounds__(256) k_find_long(const int64_t *__restrict__ ptr, int64_t n, uint32_t *__restrict__ list,
                                                   unsigned long long *__restrict__ meta /* [0]=count [1]=maxlen */) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    const int64_t len = ptr[k + 1] - ptr[k];
    if (len > kLongRow) {
        const unsigned long long pos = atomicAdd(&meta[0], 1ull);
        if (pos < (unsigned long long)kMaxLong) list[pos] = (uint32_t)k;
        atomicMax(&meta[1], (unsigned long long)len);
    }
}

// long groups: blockIdx.x = long row, blockIdx.y = chunk; workgroup-reduce the chunk, one FP64 atomic per chunk
__global__ void __launch_bounds__(256) k_spmv_long(const uint32_t *__restrict__ list, const int64_t *__restrict__ ptr,
                                                   const uint32_t *__restrict__ perm, const double *__restrict__ vals,
                                                   const int64_t *__restrict__ other, const int64_t *__restrict__ rows,
                                                   const int64_t *__restrict__ cols, int skip_diag, const double *__restrict__ v,
                                                   double *__restrict__ partial) {
    __shared__ double red[4];
    const int64_t k = list[blockIdx.x];
    const int64_t beg = ptr[k] + (int64_t)blockIdx.y * kChunk;
    const int64_t end = beg + kChunk < ptr[k + 1] ? beg + kChunk : ptr[k + 1];
    double s = 0.0;   // (a chunk past the group's end leaves a zero)
    for (int64_t j = beg + threadIdx.x; j < end; j += 256) {
        const uint32_t e = perm[j];
        if (skip_diag && rows[e] == cols[e]) continue;
        s += vals[e] * v[other[e] - 1];
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * gridDim.y + blockIdx.y] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256) k_other(const uint32_t *__restrict__ perm, const int64_t *__restrict__ other,
                                               const int64_t *__restrict__ rows, const int64_t *__restrict__ cols, int skip_diag,
                                               uint32_t *__restrict__ oth, int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const uint32_t e = perm[j];
    oth[j] = (skip_diag && rows[e] == cols[e]) ? 0xffffffffu : (uint32_t)(other[e] - 1);
}
// same sums as below with the "other" index pre-gathered in sorted order: per entry one sequential 8-byte read
// (perm, oth) and two gathers (value, v) instead of up to four random 8-byte loads
// G lanes per group (G = power of two <= 16, chosen from the mean group length): a thread walking a 60-entry group alone
// is 60 dependent round trips to HBM/L2 — ACOPF 78k J'v spent 118 us here for 5.7e6 entries, latency-bound on the bus
// rows.  Lane g adds entries g, g+G, ...; the partial sums are combined by a fixed xor tree: deterministic.
template <int G>
__global__ void __launch_bounds__(256) k_spmv_gather2(const int64_t *__restrict__ ptr, const uint32_t *__restrict__ perm,
                                                      const uint32_t *__restrict__ oth, const double *__restrict__ vals,
                                                      const double *__restrict__ v, double *__restrict__ out, int accumulate, int64_t n) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t k = t / G;
    const int g = (int)(t % G);
    const bool live = k < n;
    const int64_t b = live ? ptr[k] : 0, e1 = live ? ptr[k + 1] : 0;
    double s = 0.0;
    if (e1 - b <= kLongRow) {
        // four entries per lane in flight: index loads first, then the two dependent gathers of each
        for (int64_t j = b + g; j < e1; j += 4 * G) {
            uint32_t o[4], q[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int64_t ju = j + (int64_t)u * G;
                const bool in = ju < e1;
                o[u] = in ? oth[ju] : 0xffffffffu;
                q[u] = in ? perm[ju] : 0u;
            }
            double a[4], c[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const bool use = o[u] != 0xffffffffu;
                a[u] = use ? vals[q[u]] : 0.0;
                c[u] = use ? v[o[u]] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; u++) s += a[u] * c[u];
        }
    }
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if (live && g == 0) out[k] = (accumulate ? out[k] : 0.0) + s;
}
__global__ void __launch_bounds__(256) k_spmv_long2(const uint32_t *__restrict__ list, const int64_t *__restrict__ ptr,
                                                    const uint32_t *__restrict__ perm, const uint32_t *__restrict__ oth,
                                                    const double *__restrict__ vals, const double *__restrict__ v, double *__restrict__ partial) {
    __shared__ double red[4];
    const int64_t k = list[blockIdx.x];
    const int64_t beg = ptr[k] + (int64_t)blockIdx.y * kChunk;
    const int64_t end = beg + kChunk < ptr[k + 1] ? beg + kChunk : ptr[k + 1];
    double s = 0.0;   // (a chunk past the group's end leaves a zero)
    for (int64_t j = beg + threadIdx.x; j < end; j += 256) {
        const uint32_t o = oth[j];
        if (o != 0xffffffffu) s += vals[perm[j]] * v[o];
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * gridDim.y + blockIdx.y] = red[0] + red[1] + red[2] + red[3];
}

// one thread per group (variable / row): contributions added in ascending slot order -> deterministic
__global__ void __launch_bounds__(256) k_spmv_gather(const int64_t *__restrict__ ptr, const uint32_t *__restrict__ perm,
                                                     const double *__restrict__ vals, const int64_t *__restrict__ other,
                                                     const int64_t *__restrict__ rows, const int64_t *__restrict__ cols, int skip_diag,
                                                     const double *__restrict__ v, double *__restrict__ out, int accumulate, int64_t n) {
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n) return;
    double s = accumulate ? out[k] : 0.0;
    const int64_t b = ptr[k], e1 = ptr[k + 1];
    if (e1 - b <= kLongRow) {     // long groups are added by k_spmv_long
        for (int64_t j = b; j < e1; j++) {
            const uint32_t e = perm[j];
            if (skip_diag && rows[e] == cols[e]) continue;
            s += vals[e] * v[other[e] - 1];
        }
    }
    out[k] = s;
}

struct Tmp {
    void *p = nullptr;
    explicit Tmp(size_t n) { if (hipMalloc(&p, n ? n : 8) != hipSuccess) throw std::runtime_error("hipMalloc failed in exa_compress"); }
    ~Tmp() { if (p) (void)hipFree(p); }
    Tmp(const Tmp &) = delete;
    Tmp &operator=(const Tmp &) = delete;
};

unsigned grid_for(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

void CompressedCOO::release_gather() {
    for (void **q : {&perm, &ptr, &long_list, &partial}) { if (*q) (void)hipFree(*q); *q = nullptr; }
    nlong = maxlen = 0;
}
void CompressedCOO::release() {
    for (void **q : {&perm, &ptr, &rows, &cols, &long_list, &partial}) { if (*q) (void)hipFree(*q); *q = nullptr; }
    cnnz = nnz = nlong = maxlen = 0;
}

// rows/cols: device int64[nnz] (1-based).  nrowdim: number of rows of the matrix (ncon for J, nvar for H).
void build_compressed(CompressedCOO &c, const int64_t *rows, const int64_t *cols, int64_t nnz, int64_t nrowdim, int64_t ncoldim,
                      hipStream_t stream) {
    c.release();
    c.nnz = nnz;
    if (nnz == 0) return;
    if (nnz > 0xffffffffLL) throw std::runtime_error("compressed COO supports up to 2^32-1 entries per matrix");
    Tmp keys(8 * (size_t)nnz), keys2(8 * (size_t)nnz), idx(4 * (size_t)nnz), counts(8 * (size_t)nnz), nruns(8);
    HIPCHK_C(hipMalloc(&c.perm, 4 * (size_t)nnz));
    hipLaunchKernelGGL(k_make_keys, dim3(grid_for(nnz)), dim3(256), 0, stream, rows, cols, nrowdim, (uint64_t *)keys.p, (uint32_t *)idx.p, nnz);
    unsigned bits = 1;
    while (bits < 64 && ((unsigned __int128)1 << bits) < (unsigned __int128)nrowdim * (unsigned __int128)ncoldim) bits++;
    size_t tb = 0;
    HIPCHK_C(rocprim::radix_sort_pairs(nullptr, tb, (uint64_t *)keys.p, (uint64_t *)keys2.p, (uint32_t *)idx.p, (uint32_t *)c.perm,
                                       (size_t)nnz, 0, bits, stream));
    {
        Tmp t(tb);
        HIPCHK_C(rocprim::radix_sort_pairs(t.p, tb, (uint64_t *)keys.p, (uint64_t *)keys2.p, (uint32_t *)idx.p, (uint32_t *)c.perm,
                                           (size_t)nnz, 0, bits, stream));
    }
    // runs of equal keys -> unique keys (reuse keys.p) + run lengths
    tb = 0;
    HIPCHK_C(rocprim::run_length_encode(nullptr, tb, (uint64_t *)keys2.p, (size_t)nnz, (uint64_t *)keys.p, (int64_t *)counts.p,
                                        (uint64_t *)nruns.p, stream));
    {
        Tmp t(tb);
        HIPCHK_C(rocprim::run_length_encode(t.p, tb, (uint64_t *)keys2.p, (size_t)nnz, (uint64_t *)keys.p, (int64_t *)counts.p,
                                            (uint64_t *)nruns.p, stream));
    }
    uint64_t nr = 0;
    HIPCHK_C(hipMemcpyAsync(&nr, nruns.p, 8, hipMemcpyDeviceToHost, stream));
    HIPCHK_C(hipStreamSynchronize(stream));
    c.cnnz = (int64_t)nr;
    HIPCHK_C(hipMalloc(&c.ptr, 8 * (size_t)(c.cnnz + 1)));
    HIPCHK_C(hipMalloc(&c.rows, 8 * (size_t)c.cnnz));
    HIPCHK_C(hipMalloc(&c.cols, 8 * (size_t)c.cnnz));
    // ptr = exclusive scan of the run lengths, plus the end sentinel
    tb = 0;
    const int64_t *in = (const int64_t *)counts.p;
    HIPCHK_C(rocprim::exclusive_scan(nullptr, tb, in, (int64_t *)c.ptr, (int64_t)0, (size_t)c.cnnz, rocprim::plus<int64_t>(), stream));
    {
        Tmp t(tb);
        HIPCHK_C(rocprim::exclusive_scan(t.p, tb, in, (int64_t *)c.ptr, (int64_t)0, (size_t)c.cnnz, rocprim::plus<int64_t>(), stream));
    }
    HIPCHK_C(hipMemcpyAsync((int64_t *)c.ptr + c.cnnz, &nnz, 8, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_decode, dim3(grid_for(c.cnnz)), dim3(256), 0, stream, (const uint64_t *)keys.p, nrowdim, (int64_t *)c.rows,
                       (int64_t *)c.cols, c.cnnz);
    {   // entries with very many duplicates
        Tmp meta(16), list(4 * (size_t)kMaxLong);
        HIPCHK_C(hipMemsetAsync(meta.p, 0, 16, stream));
        if (c.cnnz) hipLaunchKernelGGL(k_find_long, dim3(grid_for(c.cnnz)), dim3(256), 0, stream, (const int64_t *)c.ptr, c.cnnz,
                                       (uint32_t *)list.p, (unsigned long long *)meta.p);
        unsigned long long hm[2] = {0, 0};
        HIPCHK_C(hipMemcpyAsync(hm, meta.p, 16, hipMemcpyDeviceToHost, stream));
        HIPCHK_C(hipStreamSynchronize(stream));
        if (hm[0] > (unsigned long long)kMaxLong) throw std::runtime_error("too many long groups for the compressed COO");
        c.nlong = (int64_t)hm[0]; c.maxlen = (int64_t)hm[1];
        if (c.nlong) {
            const int64_t chunks = (c.maxlen + kChunk - 1) / kChunk;
            HIPCHK_C(hipMalloc(&c.long_list, 4 * (size_t)c.nlong));
            HIPCHK_C(hipMalloc(&c.partial, 8 * (size_t)(c.nlong * chunks)));
            HIPCHK_C(hipMemcpyAsync(c.long_list, list.p, 4 * (size_t)c.nlong, hipMemcpyDeviceToDevice, stream));
            HIPCHK_C(hipStreamSynchronize(stream));
        }
    }
    HIPCHK_C(hipStreamSynchronize(stream));
}

void SortedIndex::release() {
    for (void **q : {&perm, &ptr, &long_rows, &oth, &partial}) { if (*q) (void)hipFree(*q); *q = nullptr; }
    nnz = ndim = nlong = maxlen = 0;
}

void build_sorted_index(SortedIndex &s, const int64_t *keys1, int64_t nnz, int64_t ndim, hipStream_t stream) {
    s.release();
    s.nnz = nnz; s.ndim = ndim;
    if (nnz > 0xffffffffLL || ndim > 0xffffffffLL) throw std::runtime_error("sorted index supports up to 2^32-1 entries");
    HIPCHK_C(hipMalloc(&s.ptr, 8 * (size_t)(ndim + 1)));
    HIPCHK_C(hipMalloc(&s.perm, 4 * (size_t)(nnz ? nnz : 1)));
    Tmp k0(4 * (size_t)nnz), k1(4 * (size_t)nnz), idx(4 * (size_t)nnz);
    if (nnz) {
        // keys outside [1, ndim] (a key kernel that skipped entries, an index column pointing outside the variables) would be
        // mis-sorted by the ndim-bit radix sort and send k_ptr_from_sorted past its array: counted, and refused
        Tmp bad(8);
        HIPCHK_C(hipMemsetAsync(bad.p, 0, 8, stream));
        hipLaunchKernelGGL(k_keys0, dim3(grid_for(nnz)), dim3(256), 0, stream, keys1, (uint32_t *)k0.p, (uint32_t *)idx.p, nnz, ndim, (unsigned long long *)bad.p);
        unsigned long long nbad = 0;
        HIPCHK_C(hipMemcpyAsync(&nbad, bad.p, 8, hipMemcpyDeviceToHost, stream));
        HIPCHK_C(hipStreamSynchronize(stream));
        if (nbad) { s.release(); throw std::runtime_error("sorted index: " + std::to_string(nbad) + " of " + std::to_string(nnz) + " keys lie outside 1.." + std::to_string(ndim)); }
        unsigned bits = 1;
        while (bits < 32 && (1ull << bits) < (unsigned long long)ndim) bits++;
        size_t tb = 0;
        HIPCHK_C(rocprim::radix_sort_pairs(nullptr, tb, (uint32_t *)k0.p, (uint32_t *)k1.p, (uint32_t *)idx.p, (uint32_t *)s.perm, (size_t)nnz, 0,
                                           bits, stream));
        Tmp t(tb);
        HIPCHK_C(rocprim::radix_sort_pairs(t.p, tb, (uint32_t *)k0.p, (uint32_t *)k1.p, (uint32_t *)idx.p, (uint32_t *)s.perm, (size_t)nnz, 0,
                                           bits, stream));
    }
    hipLaunchKernelGGL(k_ptr_from_sorted, dim3(grid_for(nnz + 1)), dim3(256), 0, stream, (const uint32_t *)k1.p, nnz, ndim, (int64_t *)s.ptr);
    HIPCHK_C(hipStreamSynchronize(stream));
    // long groups
    Tmp meta(16);
    HIPCHK_C(hipMemsetAsync(meta.p, 0, 16, stream));
    Tmp list(4 * (size_t)kMaxLong);
    if (ndim) hipLaunchKernelGGL(k_find_long, dim3(grid_for(ndim)), dim3(256), 0, stream, (const int64_t *)s.ptr, ndim, (uint32_t *)list.p,
                                 (unsigned long long *)meta.p);
    unsigned long long hm[2] = {0, 0};
    HIPCHK_C(hipMemcpyAsync(hm, meta.p, 16, hipMemcpyDeviceToHost, stream));
    HIPCHK_C(hipStreamSynchronize(stream));
    if (hm[0] > (unsigned long long)kMaxLong) throw std::runtime_error("too many long groups for the sorted products");
    s.nlong = (int64_t)hm[0]; s.maxlen = (int64_t)hm[1];
    if (s.nlong) {
        // (the per-chunk partial sums of the long groups are allocated HERE, at set-up: a callback never allocates)
        HIPCHK_C(hipMalloc(&s.partial, 8 * (size_t)s.nlong * (size_t)((s.maxlen + kChunk - 1) / kChunk)));
        HIPCHK_C(hipMalloc(&s.long_rows, 4 * (size_t)s.nlong));
        HIPCHK_C(hipMemcpyAsync(s.long_rows, list.p, 4 * (size_t)s.nlong, hipMemcpyDeviceToDevice, stream));
        HIPCHK_C(hipStreamSynchronize(stream));
    }
}

// the chunk sums of a long group, added to out[group] in chunk order by one thread: the same bits every time
__global__ void __launch_bounds__(256) k_spmv_fold(const uint32_t *__restrict__ list, const double *__restrict__ partial, int chunks,
                                                   double *__restrict__ out, int64_t nlong) {
    const int64_t l = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (l >= nlong) return;
    double s = out[list[l]];
    for (int c = 0; c < chunks; c++) s += partial[l * chunks + c];
    out[list[l]] = s;
}

void spmv_gather(const SortedIndex &s, const double *vals, const int64_t *other, const int64_t *rows, const int64_t *cols, bool skip_diag,
                 const double *v, double *out, bool accumulate, hipStream_t stream) {
    if (s.ndim == 0) return;
    if (s.oth) {
        {
            const double mean = (double)s.nnz / (double)std::max<int64_t>(s.ndim, 1);
            int G = mean <= 1.5 ? 1 : (mean <= 3.0 ? 2 : (mean <= 6.0 ? 4 : (mean <= 12.0 ? 8 : 16)));
            auto k = G == 1 ? k_spmv_gather2<1> : G == 2 ? k_spmv_gather2<2> : G == 4 ? k_spmv_gather2<4> : G == 8 ? k_spmv_gather2<8> : k_spmv_gather2<16>;
            hipLaunchKernelGGL(k, dim3(grid_for(s.ndim * G)), dim3(256), 0, stream, (const int64_t *)s.ptr, (const uint32_t *)s.perm,
                               (const uint32_t *)s.oth, vals, v, out, accumulate ? 1 : 0, s.ndim);
        }
        if (s.nlong) {
            const unsigned chunks = (unsigned)((s.maxlen + kChunk - 1) / kChunk);
            hipLaunchKernelGGL(k_spmv_long2, dim3((unsigned)s.nlong, chunks), dim3(256), 0, stream, (const uint32_t *)s.long_rows,
                               (const int64_t *)s.ptr, (const uint32_t *)s.perm, (const uint32_t *)s.oth, vals, v, (double *)s.partial);
            hipLaunchKernelGGL(k_spmv_fold, dim3(grid_for(s.nlong)), dim3(256), 0, stream, (const uint32_t *)s.long_rows, (const double *)s.partial,
                               (int)chunks, out, s.nlong);
        }
        return;
    }
    hipLaunchKernelGGL(k_spmv_gather, dim3(grid_for(s.ndim)), dim3(256), 0, stream, (const int64_t *)s.ptr, (const uint32_t *)s.perm, vals,
                       other, rows, cols, skip_diag ? 1 : 0, v, out, accumulate ? 1 : 0, s.ndim);
    if (s.nlong) {
        const unsigned chunks = (unsigned)((s.maxlen + kChunk - 1) / kChunk);
        if (!s.partial) HIPCHK_C(hipMalloc(&const_cast<SortedIndex &>(s).partial, 8 * (size_t)s.nlong * chunks));
        hipLaunchKernelGGL(k_spmv_long, dim3((unsigned)s.nlong, chunks), dim3(256), 0, stream, (const uint32_t *)s.long_rows,
                           (const int64_t *)s.ptr, (const uint32_t *)s.perm, vals, other, rows, cols, skip_diag ? 1 : 0, v, (double *)s.partial);
        hipLaunchKernelGGL(k_spmv_fold, dim3(grid_for(s.nlong)), dim3(256), 0, stream, (const uint32_t *)s.long_rows, (const double *)s.partial,
                           (int)chunks, out, s.nlong);
    }
}

void attach_other(SortedIndex &s, const int64_t *other, const int64_t *rows, const int64_t *cols, bool skip_diag, int64_t other_dim,
                  hipStream_t stream) {
    if (s.nnz == 0 || other_dim >= 0xffffffffLL) return;
    if (!s.oth) HIPCHK_C(hipMalloc(&s.oth, 4 * (size_t)s.nnz));
    hipLaunchKernelGGL(k_other, dim3(grid_for(s.nnz)), dim3(256), 0, stream, (const uint32_t *)s.perm, other, rows, cols, skip_diag ? 1 : 0,
                       (uint32_t *)s.oth, s.nnz);
}

void attach_unit(SortedIndex &s, hipStream_t stream) {
    if (s.nnz == 0) return;
    if (!s.oth) HIPCHK_C(hipMalloc(&s.oth, 4 * (size_t)s.nnz));
    HIPCHK_C(hipMemsetAsync(s.oth, 0, 4 * (size_t)s.nnz, stream));
}

void compress_values(const CompressedCOO &c, const double *buf, double *V, hipStream_t stream) {
    if (c.cnnz == 0) return;
    if (c.cnnz == c.nnz) {
        hipLaunchKernelGGL(k_permute, dim3((unsigned)((c.cnnz + 1023) / 1024)), dim3(256), 0, stream, V, buf, (const uint32_t *)c.perm, c.cnnz);
        return;
    }
    hipLaunchKernelGGL(k_compress, dim3(grid_for(c.cnnz)), dim3(256), 0, stream, V, buf, (const int64_t *)c.ptr, (const uint32_t *)c.perm,
                       c.cnnz);
    if (c.nlong) {
        const int chunks = (int)((c.maxlen + kChunk - 1) / kChunk);
        hipLaunchKernelGGL(k_compress_long, dim3((unsigned)c.nlong, (unsigned)chunks), dim3(256), 0, stream, (const uint32_t *)c.long_list,
                           (const int64_t *)c.ptr, (const uint32_t *)c.perm, buf, (double *)c.partial, chunks);
        hipLaunchKernelGGL(k_compress_fold, dim3(grid_for(c.nlong)), dim3(256), 0, stream, (const uint32_t *)c.long_list,
                           (const double *)c.partial, chunks, V, c.nlong);
    }
}

void build_positions(const CompressedCOO &c, uint32_t *pos, hipStream_t stream) {
    if (c.nnz == 0) return;
    hipLaunchKernelGGL(k_positions, dim3(grid_for(c.nnz)), dim3(256), 0, stream, (const uint32_t *)c.perm, pos, c.nnz);
}
void compress_sorted(const CompressedCOO &c, const double *sorted, double *V, hipStream_t stream) {
    if (c.cnnz == 0) return;
    hipLaunchKernelGGL(k_compress_sorted, dim3(grid_for(c.cnnz)), dim3(256), 0, stream, V, sorted, (const int64_t *)c.ptr, c.cnnz);
}

void compressed_structure(const CompressedCOO &c, void *rows, void *cols, bool wide, hipStream_t stream) {
    if (c.cnnz == 0) return;
    if (wide) {
        HIPCHK_C(hipMemcpyAsync(rows, c.rows, 8 * (size_t)c.cnnz, hipMemcpyDeviceToDevice, stream));
        HIPCHK_C(hipMemcpyAsync(cols, c.cols, 8 * (size_t)c.cnnz, hipMemcpyDeviceToDevice, stream));
    } else {
        hipLaunchKernelGGL(k_narrow<int32_t>, dim3(grid_for(c.cnnz)), dim3(256), 0, stream, (const int64_t *)c.rows, (int32_t *)rows, c.cnnz);
        hipLaunchKernelGGL(k_narrow<int32_t>, dim3(grid_for(c.cnnz)), dim3(256), 0, stream, (const int64_t *)c.cols, (int32_t *)cols, c.cnnz);
    }
}

void build_slot_map(const CompressedCOO &c, int32_t *cmap, hipStream_t stream) {
    if (c.cnnz == 0) return;
    hipLaunchKernelGGL(k_slot_map, dim3(grid_for(c.cnnz)), dim3(256), 0, stream, (const int64_t *)c.ptr, (const uint32_t *)c.perm, cmap, c.cnnz);
}

void affine_exceptions(const int32_t *cmap, int64_t o, int S, int64_t n, const int64_t *a_host, const int64_t *b_host, int64_t mid, int64_t *count,
                       int64_t *e_lo, int64_t *e_hi, hipStream_t stream) {
    void *da = nullptr, *dres = nullptr;
    HIPCHK_C(hipMalloc(&da, 16 * (size_t)S));
    HIPCHK_C(hipMalloc(&dres, 24));
    unsigned long long init[3] = {0ull, 0ull, (unsigned long long)n};
    HIPCHK_C(hipMemcpyAsync(da, a_host, 8 * (size_t)S, hipMemcpyHostToDevice, stream));
    HIPCHK_C(hipMemcpyAsync((int64_t *)da + S, b_host, 8 * (size_t)S, hipMemcpyHostToDevice, stream));
    HIPCHK_C(hipMemcpyAsync(dres, init, 24, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_affine_check, dim3(grid_for(n)), dim3(256), 0, stream, cmap, o, S, n, (const int64_t *)da, (const int64_t *)da + S, mid,
                       (unsigned long long *)dres);
    unsigned long long res[3];
    HIPCHK_C(hipMemcpyAsync(res, dres, 24, hipMemcpyDeviceToHost, stream));
    HIPCHK_C(hipStreamSynchronize(stream));
    (void)hipFree(da); (void)hipFree(dres);
    *count = (int64_t)res[0]; *e_lo = (int64_t)res[1]; *e_hi = (int64_t)res[2];
}

void compressed_csc(const CompressedCOO &c, int64_t ncol, int64_t *colptr, int64_t *rowval, hipStream_t stream) {
    hipLaunchKernelGGL(k_colptr, dim3(grid_for(c.cnnz + 1)), dim3(256), 0, stream, (const int64_t *)c.cols, c.cnnz, ncol, colptr);
    if (c.cnnz) HIPCHK_C(hipMemcpyAsync(rowval, c.rows, 8 * (size_t)c.cnnz, hipMemcpyDeviceToDevice, stream));
}

}  // namespace exa
