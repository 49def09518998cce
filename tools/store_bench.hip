// store_bench.hip — calibrates the write-side ceiling for the hess_coord! output pattern on MI355X:
// each lane owns S consecutive doubles (S = o2step), a wavefront owns 64*S contiguous doubles.
//   v0: ideal   — every lane writes 16 B, fully coalesced streaming (lane-interleaved)
//   v1: direct  — lane I writes out[S*I .. S*I+S) with dwordx4 stores (what the generated kernel does today)
//   v2: direct + nontemporal
//   v3: LDS transpose — stage the wavefront's 64*S doubles in LDS, then store them lane-interleaved (coalesced)
//   v4: v3 + nontemporal
//   v5-v11: what the write path itself prefers (no generated kernel uses these shapes — see DESIGN.md §5):
//     v8  K stores of 8 or 16 B per lane per wavefront: 1 KB per wavefront (4 KB per workgroup) 6.9 TB/s, >= 1.5 KB 5.5-5.9
//     v9  a workgroup writing 3 CONSECUTIVE 4 KB units 5.6-5.9;  v10 the same three units 8 apart (= blockIdx mod 8) 6.5
//     v11 v10 with loads + LDS staging (whole 4 KB-aligned units per workgroup, 16-B nt stores) 6.3-7.0 at 720 MB
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
// v5/v6: no loads, no LDS — only the store shape.  CHUNK: a wavefront writes its own 64*S contiguous doubles with S
// 512-B bursts (the shape of the generated flush).  !CHUNK: burst k of every wavefront lands in region k (all wavefronts
// advance one contiguous front per step).
template <bool NT, bool CHUNK>
__global__ void __launch_bounds__(256) k_shape(double* __restrict__ out, long n) {
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const long nw = (n + 63) / 64;
    if (wave >= nw) return;
#pragma unroll
    for (int k = 0; k < S; k++) {
        const long j = CHUNK ? (wave * S + k) * 64 + lane : ((long)k * nw + wave) * 64 + lane;
        if (j < n * S) { if (NT) __builtin_nontemporal_store((double)j, out + j); else out[j] = (double)j; }
    }
}
// v8: K stores of W doubles per lane per wavefront, to its own contiguous chunk (K * 64 * W doubles)
template <int W, int K>
__global__ void __launch_bounds__(256) k_multi(double* __restrict__ out, long total) {
    const long wave = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const long j = ((wave * K + k) * 64 + lane) * W;
        if (j + W <= total) {
            if (W == 2) reinterpret_cast<double2*>(out)[j / 2] = make_double2((double)j, 1.0);
            else out[j] = (double)j;
        }
    }
}
// v9: a workgroup owns 256*S contiguous doubles and writes them in steps of 4 KB: at step k wavefront w writes the
// 1 KB piece k*4 + w with one 16-B-per-lane store (no loads, no LDS)
__global__ void __launch_bounds__(256) k_wg1k(double* __restrict__ out, long total) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long base = (long)blockIdx.x * 256 * S;
#pragma unroll
    for (int k = 0; k < S / 2; k++) {
        const long j = base + ((k * 4 + w) * 64 + lane) * 2;
        if (j + 2 <= total) reinterpret_cast<double2*>(out)[j / 2] = make_double2((double)j, 1.0);
    }
}
// v10: as v9, but the three 4 KB units of workgroup b are 8 units apart and congruent to b mod 8: if the hardware
// places workgroup b on XCD b % 8 and interleaves 4 KB units over the HBM stacks, every XCD then writes to "its" stack
template <int UNIT_DOUBLES>
__global__ void __launch_bounds__(256) k_wg_affine(double* __restrict__ out, long total, int shift) {
    const int t = threadIdx.x;
    const long b = blockIdx.x, grp = b >> 3, r = (b + shift) & 7;
    constexpr int PER_WG = 256 * S / UNIT_DOUBLES;          // units per workgroup
#pragma unroll
    for (int k = 0; k < PER_WG; k++) {
        const long unit = (grp * PER_WG + k) * 8 + r;
#pragma unroll
        for (int q = 0; q < UNIT_DOUBLES / 512; q++) {
            const long j = unit * UNIT_DOUBLES + (q * 256 + t) * 2;
            if (j + 2 <= total) reinterpret_cast<double2*>(out)[j / 2] = make_double2((double)j, 1.0);
        }
    }
}
// v11: the full shape of a unit-affine flush: loads + LDS staging + whole 4 KB units.  A workgroup of 448 threads
// owns 5 units (= blockIdx mod 8, 8 units apart); thread t computes point q = t % 87 of unit t / 87 (a unit touches
// 86-87 points of S = 6 doubles; points straddling a unit boundary are computed by both neighbours), stages its S values
// at their final positions inside the unit, then the unit is streamed out with one 16-B-per-lane store per wavefront.
template <bool NT, int G = 5, int T = 87, int BLK = 448>
__global__ void __launch_bounds__(BLK) k_unit(double* __restrict__ out, const double* __restrict__ x, long n) {
    __shared__ double tile[G][512];
    const int t = threadIdx.x, k = t / T, q = t - k * T;
    const long grp = blockIdx.x >> 3, r = blockIdx.x & 7;
    if (k < G) {
        const long u = (grp * G + k) * 8 + r;
        const long p = (512 * u) / S + q;
        const double a = p < n ? x[p] : 0.0;
#pragma unroll
        for (int s = 0; s < S; s++) {
            const long e = p * S + s - 512 * u;
            if (e >= 0 && e < 512) tile[k][e] = a * (s + 1);
        }
    }
    __syncthreads();
    const long lim = n * S;
    typedef double v2d __attribute__((ext_vector_type(2)));
    for (int f = t; f < G * 256; f += BLK) {
        const int kk = f >> 8, e = (f & 255) * 2;
        const long j = 512 * ((grp * G + kk) * 8 + r) + e;
        if (j + 2 <= lim) {
            const v2d v = *(const v2d*)&tile[kk][e];
            if (NT) __builtin_nontemporal_store(v, (v2d*)(out + j)); else *(v2d*)(out + j) = v;
        }
    }
}
// v7: the LDS-transposed flush, but the 4 wavefronts of a workgroup share one 12 KB tile and store it interleaved:
// at step k wavefront w writes burst 4k + w, so the workgroup's concurrent stores are 2 KB contiguous
template <bool NT>
__global__ void __launch_bounds__(256) k_lds_wg(double* __restrict__ out, const double* __restrict__ x, long n) {
    __shared__ double tile[S * 257];
    const long I = (long)blockIdx.x * 256 + threadIdx.x;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const double a = I < n ? x[I] : 0.0;
#pragma unroll
    for (int s = 0; s < S; s++) tile[s * 257 + t] = a * (s + 1);
    __syncthreads();
    const long base = (long)blockIdx.x * 256 * S;
    const long lim = n * S;
#pragma unroll
    for (int k = 0; k < S; k++) {
        const int j = (k * 4 + w) * 64 + lane;
        const int l2 = j / S, s2 = j - l2 * S;
        const double val = tile[s2 * 257 + l2];
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
    timeit("v5 shape: chunk/wave", [&] { k_shape<false, true><<<grid, 256>>>(out, n); });
    timeit("v5 shape: chunk/wave nt", [&] { k_shape<true, true><<<grid, 256>>>(out, n); });
    timeit("v6 shape: front", [&] { k_shape<false, false><<<grid, 256>>>(out, n); });
    timeit("v6 shape: front nt", [&] { k_shape<true, false><<<grid, 256>>>(out, n); });
    {
        const long total = n * S;
        auto g = [&](int w, int k) { return (unsigned)((total / ((long)w * k) + 255) / 256 + 1); };
        timeit("v8 8B/lane x1 store/wave", [&] { k_multi<1, 1><<<g(1, 1), 256>>>(out, total); });
        timeit("v8 8B/lane x2", [&] { k_multi<1, 2><<<g(1, 2), 256>>>(out, total); });
        timeit("v8 8B/lane x3", [&] { k_multi<1, 3><<<g(1, 3), 256>>>(out, total); });
        timeit("v8 8B/lane x6", [&] { k_multi<1, 6><<<g(1, 6), 256>>>(out, total); });
        timeit("v8 8B/lane x12", [&] { k_multi<1, 12><<<g(1, 12), 256>>>(out, total); });
        timeit("v8 8B/lane x4", [&] { k_multi<1, 4><<<g(1, 4), 256>>>(out, total); });
        timeit("v8 8B/lane x8", [&] { k_multi<1, 8><<<g(1, 8), 256>>>(out, total); });
        timeit("v8 16B/lane x2", [&] { k_multi<2, 2><<<g(2, 2), 256>>>(out, total); });
        timeit("v8 16B/lane x4", [&] { k_multi<2, 4><<<g(2, 4), 256>>>(out, total); });
        timeit("v9 wg 12KB in 3 x 4KB steps", [&] { k_wg1k<<<grid, 256>>>(out, total); });
        for (int c = 0; c < 8; c++) {
            char nm[64]; snprintf(nm, sizeof nm, "v10 3 x 4KB units = b+%d mod 8", c);
            timeit(nm, [&] { k_wg_affine<512><<<grid, 256>>>(out, total, c); });
        }
        timeit("v10 3 x 4KB units, +512B misaligned", [&] { k_wg_affine<512><<<grid, 256>>>(out + 64, total - 64, 0); });
        // (round 4 also listed k_wg_affine<1024> here: 256 * S / 1024 = 1.5 units per workgroup truncates to 1, the kernel wrote 2/3 of the
        // bytes it was credited with — "9 283 GB/s" in profiles/r4_store_bench_7GB.txt is 6 189 GB/s of real stores.  Row removed.)
        {
            const unsigned gu = (unsigned)(((total + 511) / 512 + 4) / 5 + 8);
            timeit("v11 unit-affine flush (loads+LDS)", [&] { k_unit<false><<<gu, 448>>>(out, x, n); });
            timeit("v11 unit-affine flush nt", [&] { k_unit<true><<<gu, 448>>>(out, x, n); });
            const unsigned g2 = (unsigned)(((total + 511) / 512 + 1) / 2 + 8);
            timeit("v11 nt, 192 threads / 2 units", [&] { k_unit<true, 2, 96, 192><<<g2, 192>>>(out, x, n); });
            const unsigned g4 = (unsigned)(((total + 511) / 512 + 3) / 4 + 8);
            timeit("v11 nt, 384 threads / 4 units", [&] { k_unit<true, 4, 96, 384><<<g4, 384>>>(out, x, n); });
            const unsigned g1 = (unsigned)((total + 511) / 512 + 8);
            timeit("v11 nt, 128 threads / 1 unit", [&] { k_unit<true, 1, 128, 128><<<g1, 128>>>(out, x, n); });
        }
        timeit("v8 16B/lane x1", [&] { k_multi<2, 1><<<g(2, 1), 256>>>(out, total); });
        timeit("v8 16B/lane x3", [&] { k_multi<2, 3><<<g(2, 3), 256>>>(out, total); });
        timeit("v8 16B/lane x6", [&] { k_multi<2, 6><<<g(2, 6), 256>>>(out, total); });
    }
    timeit("v7 lds, workgroup-interleaved", [&] { k_lds_wg<false><<<grid, 256>>>(out, x, n); });
    timeit("v7 lds, wg-interleaved nt", [&] { k_lds_wg<true><<<grid, 256>>>(out, x, n); });
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
