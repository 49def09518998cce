#!/usr/bin/env python
"""A/B of the one variant of hess_coord! that north_star names and rounds 1-2 never measured: the workgroup's stretch of x
(+ halo) staged ONCE through LDS — one coalesced 8-byte load per lane and a two-lane halo load instead of the 2 + 1
overlapping wide loads of the constraint pattern and the 1 of the objective pattern; the three / two stencil operands are then
ds_read — against the shipped exa_hessc (chained, grouped, software-pipelined; operands loaded straight from global memory
and left to L1 / L2).  Reference counterpart: the gathered reads of kerh2, ext/ExaModelsKernelAbstractions.jl:631-653.

Both kernels live in ONE module: the generated source of the Luksan-Vlcek model + a hand-written `exa_hessc_lds` that reuses
the generated evaluation stages (p0_hesscE / p1_hesscE: same arithmetic, same LDS-transposed store epilogue) and replaces
only the LOAD stage.  Same parameter table, same block map, same output buffer, A/B/A/B rounds; outputs compared bit for bit
with each other and with the library's own exa_hess.  Luksan-Vlcek only (the stretch arithmetic is written out for its two
patterns); the point is a number for the idea before it is built into the generator.

usage (GPU box): python tools/hessc_lds_ab.py [N=1e8] > profiles/r3_hessc_lds_ab.txt"""
import ctypes
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examodels.jl_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from exahip import ExaModel, models  # noqa: E402

VARIANT = r"""
// ---- hand-written variant: inputs staged through LDS (one stretch of x per wavefront and tile) -------------------------------
extern "C" __global__ void __launch_bounds__(EXA_BLOCK) exa_hessc_lds(const long* __restrict__ P, const double* __restrict__ x,
        const double* __restrict__ y, const double* __restrict__ th, double* __restrict__ out, double sigma, double* __restrict__ sink) {
    __shared__ double lds_all[(EXA_BLOCK / 64) * 402];
    double* lds = lds_all + (threadIdx.x >> 6) * 402;
    __shared__ double xs_all[(EXA_BLOCK / 64) * 72];
    double* xs = xs_all + (threadIdx.x >> 6) * 72;              // this wavefront's stretch: 64 points + 2 of halo
    const int lane = threadIdx.x & 63;
    const long e_ = ((const long*)P[32])[blockIdx.x];
    const long t0_ = (e_ & ((1L << 40) - 1)) * 4;
    const long tend_ = t0_ + 4 < P[33] ? t0_ + 4 : P[33];
    // constraint point I reads x[P9 - 1 + I + {0, 1, 2}] (0-based), objective point I reads x[P19 - 2 + I + {0, 1}], and
    // P19 - 2 == P9 - 1 for this model: one stretch serves both patterns
    const long last = P[9] - 1 + (P[1] - 1) + 2;                // the last variable any point reads
    double g0, g1, gy;
    auto G = [&](long t) {
        const long Iw = t * EXA_BLOCK + (threadIdx.x & ~63);
        const long base = P[9] - 1 + Iw;
        long a0 = base + lane;        a0 = a0 < last ? a0 : last;
        long a1 = base + 64 + lane;   a1 = a1 < last ? a1 : last;
        g0 = x[a0];
        g1 = x[lane < 2 ? a1 : a0];                                // (lanes >= 2 repeat their own address: same line, no extra traffic)
        long Ic = Iw + lane;          Ic = Ic < P[1] ? Ic : P[1] - 1;
        gy = y[P[2] + Ic];
    };
    G(t0_);
    asm volatile("" : "+v"(g0)); asm volatile("" : "+v"(g1)); asm volatile("" : "+v"(gy));
#pragma unroll 1
    for (long t = t0_; t < tend_; t++) {
        const long tid = t * EXA_BLOCK + threadIdx.x;
        double in0[4], in1[2]; long ik[1];
        xs[lane] = g0;
        if (lane < 2) xs[64 + lane] = g1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        in0[0] = xs[lane + 1]; in0[1] = xs[lane + 2]; in0[2] = xs[lane]; in0[3] = gy;
        in1[0] = xs[lane]; in1[1] = xs[lane + 1];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier();
        G(t + 1 < tend_ ? t + 1 : t);                               // the next tile's loads are in flight during this tile's evaluation and stores
        p0_hesscE(P, in0, ik, out, sink, sigma, tid, lds);
        p1_hesscE(P, in1, ik, out, sink, sigma, tid, lds);
        asm volatile("" : "+v"(g0)); asm volatile("" : "+v"(g1)); asm volatile("" : "+v"(gy));
    }
}
"""


def main():
    N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
    dev = torch.device("cuda:0")
    m = ExaModel(models.luksan_vlcek_model(N))
    src = m.kernel_source()
    assert "P[32])[blockIdx.x]" in src and "P[33]" in src and "p0_hesscE" in src, "parameter layout of the LV module changed: adapt the indices below"
    with tempfile.TemporaryDirectory() as td:
        hip = os.path.join(td, "ab.hip")
        with open(hip, "w") as fh:
            fh.write(src + VARIANT)
        co = os.path.join(td, "ab.hsaco")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--genco", "--offload-arch=gfx950", "-O3", "-w", "-o", co, hip])
        image = open(co, "rb").read()
    hiprt = ctypes.CDLL("libamdhip64.so.7")
    mod = ctypes.c_void_p()
    assert hiprt.hipModuleLoadData(ctypes.byref(mod), image) == 0
    fns = {}
    for name in ("exa_hessc", "exa_hessc_lds"):
        f = ctypes.c_void_p()
        assert hiprt.hipModuleGetFunction(ctypes.byref(f), mod, name.encode()) == 0, name
        fns[name] = f
    # the parameter table of the generated module (exa_gen_module.cpp: 9 words + columns per pattern, then one block-map word
    # per callback, then the chained groups' tile counts), for the unsharded LV model: pattern 0 = constraint, 1 = objective
    tiles = (N - 1 + 255) // 256
    nblk = (tiles + 3) // 4
    bmap = torch.arange(nblk, dtype=torch.int64, device=dev)          # (group 0 << 40) | first tile / 4
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

    def launch(name):
        args = [ctypes.c_void_p(Pd.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(0),
                ctypes.c_void_p(out.data_ptr()), ctypes.c_double(0.5), ctypes.c_void_p(sink.data_ptr())]
        arr = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
        rc = hiprt.hipModuleLaunchKernel(fns[name], nblk, 1, 1, 256, 1, 1, 0, ctypes.c_void_p(stream), arr, None)
        assert rc == 0, rc

    ref = m.hess_coord(x, y, 0.5).clone()
    res = {}
    for name in fns:
        out.fill_(float("nan"))
        launch(name)
        torch.cuda.synchronize()
        res[name] = "bitwise equal to the library's hess_coord!" if torch.equal(out, ref) else f"DIFFERS (max |d| {float((out - ref).abs().max()):.3e}, nan {int(torch.isnan(out).sum())})"
    del ref
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20 if N >= 5e7 else 100
    times = {n: [] for n in fns}
    for _ in range(3):
        launch("exa_hessc")
    for rnd in range(6):
        for name in fns:
            e0.record()
            for _ in range(reps):
                launch(name)
            e1.record()
            torch.cuda.synchronize()
            times[name].append(e0.elapsed_time(e1) / reps)
    alg = 8.0 * (m.meta.nnzh + 2 * N - 2)
    print(f"LV N={N}: hess_coord! chained kernels, one module, one output buffer, {reps} launches x 6 A/B rounds; algorithmic bytes {alg / 1e9:.2f} GB")
    for name in fns:
        t = min(times[name])
        print(f"  {name:14s} min {t:.4f} ms  med {float(np.median(times[name])):.4f} ms  {alg / t / 1e6:7.0f} GB/s = {alg / t / 1e6 / 8000:.3f} of 8 TB/s   {res[name]}")


if __name__ == "__main__":
    main()
