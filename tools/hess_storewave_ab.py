#!/usr/bin/env python
"""A/B of the one structure of hess_coord! rounds 1-3 never measured: WAVE-SPECIALISED workgroups.  On gfx9
loads and stores of a wavefront retire through one in-order counter (vmcnt), so a wavefront that evaluates AND stores waits for
the previous tile's stores before it can touch the next tile's loads.  Here NC compute wavefronts evaluate 64-point tiles and
leave the slots slot-major in double-buffered LDS (the tile layout of the shipped LDS-transposed epilogue); ONE store wavefront
per workgroup streams finished tiles to HBM with the shipped non-temporal flush and never waits for a load; hand-over by an LDS
flag per (compute wave, buffer) + s_sleep, s_setprio on the store wave.  Reference counterpart: kerh2,
ext/ExaModelsKernelAbstractions.jl:631-653.

Both kernels live in ONE module: the generated source of the Luksan-Vlcek model + a hand-written `exa_hess_sw` that reuses the
generated load / evaluation stages of exa_hesscl (p0_hessclL / p0_hessclE with the flush cut out: same arithmetic, same tile
layout) — same parameter table, same output buffer, A/B/A/B rounds; outputs compared bit for bit with the library's exa_hess.
Luksan-Vlcek only: a number for the idea before it is built into the generator.

usage (GPU box): python tools/hess_storewave_ab.py [N=1e8] [NC=3] [R=4] [NBUF=2]"""
import ctypes
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examodels.jl_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from exahip import ExaModel, models  # noqa: E402


def function_text(src, name):
    i = src.index("static __device__ __forceinline__ void " + name + "(")
    j = src.index("\n}\n", i) + 3
    return src[i:j]


def variant_source(src, NC, R, NBUF, WIDE=0):
    """exa_hess_sw for the LV module `src`; returns (text, flush template arguments of the two patterns)"""
    out, targs = "", []
    for k in (0, 1):
        f = function_text(src, f"p{k}_hessclE")
        m = re.search(r"exa_flush_points_nb<([^>]*)>\([^;]*;\n", f)
        targs.append(m.group(1))
        out += f.replace(f"p{k}_hessclE(", f"p{k}_hessclW(").replace(m.group(0), "")
    S0, PP0, LD0 = (int(t) for t in targs[0].split(","))
    S1, PP1, LD1 = (int(t) for t in targs[1].split(","))
    assert PP0 == 64 and PP1 == 64
    T0, T1 = S0 * LD0, S1 * LD1
    out += f"""
// LDS flag operations by address, in asm: a `volatile` access through a generic pointer makes the compiler drain EVERY outstanding
// memory operation first (s_waitcnt vmcnt(0) — the stores of the store wave, the prefetched loads of the compute waves)
static __device__ __forceinline__ unsigned sw_lds_addr(const void* p) {{ return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p; }}
static __device__ __forceinline__ int sw_flag_read(unsigned a) {{
    int v;
    asm volatile("ds_read_b32 %0, %1\\n\\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}}
static __device__ __forceinline__ void sw_flag_write(unsigned a, int v) {{
    asm volatile("s_waitcnt lgkmcnt(0)\\n\\tds_write_b32 %0, %1" :: "v"(a), "v"(v) : "memory");
}}
#define SW_NC {NC}
#define SW_R {R}
#define SW_NB {NBUF}
#define SW_TILE {T0 + T1}
#define SW_WIDE {WIDE}
// WIDE store role: the store wave waits for ALL compute waves of a round and streams the round's contiguous output — NC * 64 * S doubles
// of each pattern: NC = 4: 12 288 B of constraint slots = three 4 KB-aligned 4 KB units — with 16-byte stores (1 KB per instruction)
template <int S, int LD>
static __device__ __forceinline__ void sw_flush_round(double* __restrict__ out, double* __restrict__ sink, long obase, long npts, const double* tiles, int tile_stride, int lane) {{
    constexpr int PER = 64 * S;                 // doubles per compute wave's tile
    constexpr int CNT = SW_NC * PER;
    typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int k = 0; k * 128 < CNT; k++) {{
        const int e = k * 128 + 2 * lane;       // two consecutive doubles of the round's run (PER is even: never across two tiles)
        const int w = e / PER, j = e - w * PER;
        const int l0 = j / S, s0 = j - l0 * S, l1 = (j + 1) / S, s1 = (j + 1) - l1 * S;
        const double* t = tiles + w * tile_stride;
        d2 v; v.x = t[s0 * LD + l0]; v.y = t[s1 * LD + l1];
        const bool ok = e < CNT && (long)(w * 64 + l1) < npts;
        const bool ok0 = e < CNT && (long)(w * 64 + l0) < npts;
        if (ok) __builtin_nontemporal_store(v, (d2*)(out + obase + e));
        else if (ok0) __builtin_nontemporal_store(v.x, out + obase + e);
    }}
}}
extern "C" __global__ void __launch_bounds__((SW_NC + 1) * 64) exa_hess_sw(const long* __restrict__ P, const double* __restrict__ x,
        const double* __restrict__ y, const double* __restrict__ th, double* __restrict__ out, double sigma, double* __restrict__ sink) {{
    __shared__ double tile_all[SW_NC * SW_NB * SW_TILE];
    __shared__ double xs_all[SW_NC * 80];
    __shared__ int flag_all[SW_NC * SW_NB];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x < SW_NC * SW_NB) flag_all[threadIdx.x] = 0;
    __syncthreads();
    const long base0 = (long)blockIdx.x * SW_R * (SW_NC * 64);
    if (wave < SW_NC) {{
        double* xs = xs_all + wave * 80;
        const long B0_ = P[9] + P[0] + (0L) - 1L;
        const long B1_ = P[19] + P[10] + (-1L) - 1L;
        const long B_ = B0_ < B1_ ? B0_ : B1_;
        const int d0_ = (int)(B0_ - B_), d1_ = (int)(B1_ - B_);
        const int halo_ = d0_ + 2 > d1_ + 1 ? d0_ + 2 : d1_ + 1;
        long xlast_ = P[9] + P[1] - 1L + (2L) - 1L;
        {{ const long l_ = P[19] + P[11] - 1L + (0L) - 1L; xlast_ = l_ > xlast_ ? l_ : xlast_; }}
        double in0[1], in1[1], in0n[1]; long ik0[1], ik1[1], ik0n[1];
        double g0_, g1_;
        long tidw = base0 + wave * 64;
        {{ const long a_ = B_ + tidw; long a0_ = a_ + lane; a0_ = a0_ < xlast_ ? a0_ : xlast_; long a1_ = a_ + 64 + lane; a1_ = a1_ < xlast_ ? a1_ : xlast_; g0_ = x[a0_]; g1_ = x[lane < halo_ ? a1_ : a0_]; }}
        p0_hessclL(P, y, th, tidw + lane, in0, ik0);
        asm volatile("" : "+v"(g0_)); asm volatile("" : "+v"(g1_)); asm volatile("" : "+v"(in0[0]));
#pragma unroll 1
        for (int r = 0; r < SW_R; r++) {{
            const long tid = tidw + lane;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
            xs[lane] = g0_;
            if (lane < halo_) xs[64 + lane] = g1_;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            p1_hessclL(P, y, th, tid, in1, ik1);
            const long next = r + 1 < SW_R ? tidw + SW_NC * 64 : tidw;
            {{ const long a_ = B_ + next; long a0_ = a_ + lane; a0_ = a0_ < xlast_ ? a0_ : xlast_; long a1_ = a_ + 64 + lane; a1_ = a1_ < xlast_ ? a1_ : xlast_; g0_ = x[a0_]; g1_ = x[lane < halo_ ? a1_ : a0_]; }}
            p0_hessclL(P, y, th, next + lane, in0n, ik0n);
            const unsigned f = sw_lds_addr(&flag_all[wave * SW_NB + (r % SW_NB)]);
            while (sw_flag_read(f) != 0) __builtin_amdgcn_s_sleep(1);          // the store wave has drained this buffer
            double* tile = tile_all + (wave * SW_NB + (r % SW_NB)) * SW_TILE;
            p0_hessclW(P, in0, ik0, xs, lane + d0_, out, sink, sigma, tid, tile);
            p1_hessclW(P, in1, ik1, xs, lane + d1_, out, sink, sigma, tid, tile + {T0});
            sw_flag_write(f, 1);
            asm volatile("" : "+v"(g0_)); asm volatile("" : "+v"(g1_));
            asm volatile("" : "+v"(in0n[0])); in0[0] = in0n[0];
            tidw = next;
        }}
    }} else {{
        __builtin_amdgcn_s_setprio(3);
#if SW_WIDE
#pragma unroll 1
        for (int r = 0; r < SW_R; r++) {{
            for (int w = 0; w < SW_NC; w++) {{ const unsigned f = sw_lds_addr(&flag_all[w * SW_NB + (r % SW_NB)]); while (sw_flag_read(f) == 0) __builtin_amdgcn_s_sleep(1); }}
            const double* tiles = tile_all + (r % SW_NB) * SW_TILE;            // wave w's buffer: + w * SW_NB * SW_TILE
            const long tid0 = base0 + (long)r * (SW_NC * 64);
            {{ const long I0 = P[0] + tid0; sw_flush_round<{S0}, {LD0}>(out, sink, P[4] + {S0}L * I0, P[1] - I0, tiles, SW_NB * SW_TILE, lane); }}
            {{ const long I0 = P[10] + tid0; sw_flush_round<{S1}, {LD1}>(out, sink, P[14] + {S1}L * I0, P[11] - I0, tiles + {T0}, SW_NB * SW_TILE, lane); }}
            for (int w = 0; w < SW_NC; w++) sw_flag_write(sw_lds_addr(&flag_all[w * SW_NB + (r % SW_NB)]), 0);
        }}
#else
#pragma unroll 1
        for (int r = 0; r < SW_R; r++)
#pragma unroll 1
            for (int w = 0; w < SW_NC; w++) {{
                const unsigned f = sw_lds_addr(&flag_all[w * SW_NB + (r % SW_NB)]);
                while (sw_flag_read(f) == 0) __builtin_amdgcn_s_sleep(1);
                const double* tile = tile_all + (w * SW_NB + (r % SW_NB)) * SW_TILE;
                const long tidw = base0 + (long)r * (SW_NC * 64) + w * 64;
                {{ const long I0 = P[0] + tidw; exa_flush_points_nb<{targs[0]}>(out, sink, P[4] + {S0}L * I0, P[1] - I0, tile, lane, 0); }}
                {{ const long I0 = P[10] + tidw; exa_flush_points_nb<{targs[1]}>(out, sink, P[14] + {S1}L * I0, P[11] - I0, tile + {T0}, lane, 0); }}
                sw_flag_write(f, 0);
            }}
#endif
    }}
}}
"""
    return out, targs


def main():
    N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    combos = [(3, 4, 2), (3, 8, 2), (5, 4, 2), (3, 4, 3), (2, 4, 2)]
    if len(sys.argv) > 2:
        combos = [tuple(int(a) for a in c.split(",")) for c in sys.argv[2:]]
    dev = torch.device("cuda:0")
    m = ExaModel(models.luksan_vlcek_model(N))
    src = m.kernel_source()
    assert "P[32])[blockIdx.x]" in src and "P[33]" in src and "p0_hessclE" in src, "parameter layout of the LV module changed: adapt the indices below"
    hiprt = ctypes.CDLL("libamdhip64.so.7")
    tiles = (N - 1 + 255) // 256
    nblk = (tiles + 3) // 4
    bmap = torch.arange(nblk, dtype=torch.int64, device=dev)
    P = np.zeros(64, dtype=np.int64)
    P[0], P[1], P[2], P[4], P[9] = 0, N - 2, 0, 0, 1
    P[10], P[11], P[14], P[19] = 0, N - 1, 6 * (N - 2), 2
    P[32], P[33] = bmap.data_ptr(), tiles
    Pd = torch.from_numpy(P).to(dev)
    x = torch.from_numpy(m.meta.x0 + 0.1 * np.random.default_rng(0).uniform(-1, 1, N)).to(dev)
    y = torch.from_numpy(np.random.default_rng(1).standard_normal(N - 2)).to(dev)
    out = torch.empty(m.meta.nnzh, dtype=torch.float64, device=dev)
    sink = torch.zeros(64, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ref = m.hess_coord(x, y, 0.5).clone()
    alg = 8.0 * (m.meta.nnzh + 2 * N - 2)
    print(f"LV N={N}: hess_coord!, one output buffer, A/B rounds against the shipped exa_hesscl of the same module; algorithmic bytes {alg / 1e9:.2f} GB", flush=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20 if N >= 5e7 else 60
    for combo in combos:
        NC, R, NBUF = combo[:3]
        WIDE = combo[3] if len(combo) > 3 else 0
        text, _ = variant_source(src, NC, R, NBUF, WIDE)
        with tempfile.TemporaryDirectory() as td:
            hip = os.path.join(td, "ab.hip")
            with open(hip, "w") as fh:
                fh.write(src + text)
            co = os.path.join(td, "ab.hsaco")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--genco", "--offload-arch=gfx950", "-O3", "-w", "-Rpass-analysis=kernel-resource-usage", "-o", co, hip],
                                  stderr=open(os.path.join(td, "log"), "w"))
            log = open(os.path.join(td, "log")).read()
            image = open(co, "rb").read()
        regs = {}
        for blk in log.split("Function Name: ")[1:]:
            nm = blk.split()[0]
            if nm in ("exa_hesscl", "exa_hess_sw"):
                regs[nm] = (re.search(r"VGPRs: (\d+)", blk).group(1), re.search(r"Occupancy \[waves/SIMD\]: (\d+)", blk).group(1), re.search(r"LDS Size \[bytes/block\]: (\d+)", blk).group(1))
        mod = ctypes.c_void_p()
        assert hiprt.hipModuleLoadData(ctypes.byref(mod), image) == 0
        fns = {}
        for name in ("exa_hesscl", "exa_hess_sw"):
            f = ctypes.c_void_p()
            assert hiprt.hipModuleGetFunction(ctypes.byref(f), mod, name.encode()) == 0, name
            fns[name] = f
        grid_sw = (N - 1 + NC * 64 * R - 1) // (NC * 64 * R)

        def launch(name):
            args = [ctypes.c_void_p(Pd.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(0),
                    ctypes.c_void_p(out.data_ptr()), ctypes.c_double(0.5), ctypes.c_void_p(sink.data_ptr())]
            arr = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
            if name == "exa_hess_sw":
                rc = hiprt.hipModuleLaunchKernel(fns[name], grid_sw, 1, 1, (NC + 1) * 64, 1, 1, 0, ctypes.c_void_p(stream), arr, None)
            else:
                rc = hiprt.hipModuleLaunchKernel(fns[name], nblk, 1, 1, 256, 1, 1, 0, ctypes.c_void_p(stream), arr, None)
            assert rc == 0, rc

        res = {}
        for name in fns:
            out.fill_(float("nan"))
            launch(name)
            torch.cuda.synchronize()
            res[name] = "bitwise equal to the library's hess_coord!" if torch.equal(out, ref) else f"DIFFERS (max |d| {float((out - ref).abs().nan_to_num(1e300).max()):.3e}, nan {int(torch.isnan(out).sum())})"
        times = {n: [] for n in fns}
        for _ in range(3):
            launch("exa_hesscl")
        for rnd in range(5):
            for name in fns:
                e0.record()
                for _ in range(reps):
                    launch(name)
                e1.record()
                torch.cuda.synchronize()
                times[name].append(e0.elapsed_time(e1) / reps)
        print(f" NC={NC} compute waves + 1 store wave, R={R} rounds per workgroup, {NBUF} LDS buffers per compute wave{', WIDE stores (16 B per lane over the whole round)' if WIDE else ''}:", flush=True)
        for name in fns:
            t = min(times[name])
            print(f"  {name:12s} min {t:.4f} ms  med {float(np.median(times[name])):.4f} ms  {alg / t / 1e6 / 8000:.3f} of 8 TB/s  VGPRs/occupancy/LDS {regs.get(name)}  {res[name]}", flush=True)
        hiprt.hipModuleUnload(mod)


if __name__ == "__main__":
    main()
