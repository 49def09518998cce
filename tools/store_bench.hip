// store_bench.hip — calibrates the write-side ceiling for the hess_coord! output pattern on MI355X:
// each lane owns S consecutive doubles (S = o2step), a wavefront owns 64*S contiguous doubles.
//   v0: ideal   — every lane writes 16 B, fully coalesced streaming (lane-interleaved)
//   v1: direct  — lane I writes out[S*I .. S*I+S) with dwordx4 stores (what the generated kernel does today)
//   v2: direct + nontemporal
//   v3: LDS transpose — stage the wavefront's 64*S doubles in LDS, then store them lane-interleaved (coalesced)
//   v4: v3 + nontemporal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int S = 6;

__global__ void __launch_bounds__(256) k_ideal(double* __restrict__ out, long n2) {   // n2 = number of double2
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long stride = (long)gridDim.x * 256;
    for (; i < n2; i += stride) { double2 v = make_double2((double)i, 1.0); reinterpret_cast<double2*>(out)[i] = v; }
}
template <bool NT>
__global__ void __launch_bounds__(256) k_direct(double* __restrict__ out, const double* __restrict__ x, long n) {
    const long I = (long)blockIdx.x * 256 + threadIdx.x;
    if (I >= n) return;
    const double a = x[I];
    double v[S];
#pragma unroll
    for (int s = 0; s < S; s++) v[s] = a * (s + 1);
    double* o = out + S * I;
#pragma unroll
    for (int s = 0; s < S; s++) { if (NT) __builtin_nontemporal_store(v[s], o + s); else o[s] = v[s]; }
}
template <bool NT>
__global__ void __launch_bounds__(256) k_lds(double* __restrict__ out, const double* __restrict__ x, long n) {
    __shared__ double tile[4][64 * S + 64];
    const long I = (long)blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const double a = I < n ? x[I] : 0.0;
    double v[S];
#pragma unroll
    for (int s = 0; s < S; s++) v[s] = a * (s + 1);
    // slot-major staging: element (lane, s) at s*65 + lane  (conflict-free writes)
#pragma unroll
    for (int s = 0; s < S; s++) tile[w][s * 65 + lane] = v[s];
    __builtin_amdgcn_wave_barrier();
    const long base = ((long)blockIdx.x * 256 + w * 64) * S;     // first output double of this wavefront
    const long lim = n * S;
#pragma unroll
    for (int k = 0; k < S; k++) {
        const int j = k * 64 + lane;            // position inside the wavefront's contiguous block
        const int l2 = j / S, s2 = j - l2 * S;
        const double val = tile[w][s2 * 65 + l2];
        if (base + j < lim) { if (NT) __builtin_nontemporal_store(val, out + base + j); else out[base + j] = val; }
    }
}
// read-side calibration for the FETCH_SIZE counter: the access pattern of exa_hess on Luksan-Vlcek
// (three overlapping 8-B/lane loads of x, one of y), nothing written but one double per workgroup
__global__ void __launch_bounds__(256) k_read8(const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ part, long n) {
    const long I = (long)blockIdx.x * 256 + threadIdx.x;
    double v = 0.0;
    if (I + 2 < n) v = x[I] + x[I + 1] * 0.5 + x[I + 2] * 0.25 + y[I];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) part[blockIdx.x * 4 + (threadIdx.x >> 6)] = v;
}
int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 15000000;    // points; 15e6*6 doubles = 720 MB
    double *out, *x;
    CHECK(hipMalloc(&out, sizeof(double) * n * S));
    CHECK(hipMalloc(&x, sizeof(double) * n));
    CHECK(hipMemset(x, 0, sizeof(double) * n));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int reps = 50;
    auto timeit = [&](const char* name, auto launch) {
        for (int i = 0; i < 5; i++) launch();
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < reps; i++) launch();
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-28s %8.3f ms  %8.1f GB/s (write only)\n", name, ms, sizeof(double) * n * S / ms / 1e6);
    };
    const unsigned grid = (unsigned)((n + 255) / 256);
    timeit("v0 ideal coalesced x16B", [&] { k_ideal<<<256 * 8, 256>>>(out, n * S / 2); });
    timeit("v0b ideal, 1 elem/thread", [&] { k_ideal<<<(unsigned)((n * S / 2 + 255) / 256), 256>>>(out, n * S / 2); });
    timeit("v1 direct", [&] { k_direct<false><<<grid, 256>>>(out, x, n); });
    timeit("v2 direct nt", [&] { k_direct<true><<<grid, 256>>>(out, x, n); });
    timeit("v3 lds transpose", [&] { k_lds<false><<<grid, 256>>>(out, x, n); });
    timeit("v4 lds transpose nt", [&] { k_lds<true><<<grid, 256>>>(out, x, n); });
    {
        const long nr = 10000000;
        double *xr, *yr, *pr;
        CHECK(hipMalloc(&xr, 8 * nr)); CHECK(hipMalloc(&yr, 8 * nr)); CHECK(hipMalloc(&pr, 8 * (nr / 64 + 8)));
        CHECK(hipMemset(xr, 0, 8 * nr)); CHECK(hipMemset(yr, 0, 8 * nr));
        const unsigned g = (unsigned)((nr + 255) / 256);
        for (int i = 0; i < 5; i++) k_read8<<<g, 256>>>(xr, yr, pr, nr);
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < reps; i++) k_read8<<<g, 256>>>(xr, yr, pr, nr);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
        printf("%-28s %8.3f ms  %8.1f GB/s (reads 2 x 80 MB = 160 MB known bytes; FETCH_SIZE calibration)\n", "k_read8", ms, 16.0 * nr / ms / 1e6);
    }
    timeit("memset 720MB", [&] { CHECK(hipMemsetAsync(out, 0, sizeof(double) * n * S, 0)); });
    return 0;
}
