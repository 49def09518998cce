#!/usr/bin/env python
"""A/B for the headline configuration (LV N = 1e7, hess_coord!): is ONE tile of BOTH co-indexed patterns per workgroup — x read
once, with or without the LDS staging of exa_hesscl, no loop and no software pipelining — faster than what ships?  Shipped: exa_hess
(one (pattern, tile) per workgroup, block order chosen by exa_tune) and exa_hesscl (4 tiles per workgroup, pipelined, staged).

The two candidates are hand-written around the GENERATED pattern functions of the module (same arithmetic, same store epilogue):
  exa_hess_g1   p0/p1_hesscL + p0/p1_hesscE of one tile (grouped, loads straight from global memory)
  exa_hess_l1   p0/p1_hessclL + the stretch of x staged through LDS + p0/p1_hessclE (grouped + staged)
One process, ONE output buffer, A/B rounds, outputs compared bit for bit with the library's.  Luksan-Vlcek only.

usage (GPU box): python tools/hess_grouped_ab.py [N=1e7] > profiles/r3_hess_grouped_ab.txt"""
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

VARIANTS = r"""
extern "C" __global__ void __launch_bounds__(EXA_BLOCK) exa_hess_g1(const long* __restrict__ P, const double* __restrict__ x,
        const double* __restrict__ y, const double* __restrict__ th, double* __restrict__ out, double sigma, double* __restrict__ sink) {
    __shared__ double lds_all[(EXA_BLOCK / 64) * 402];
    double* lds = lds_all + (threadIdx.x >> 6) * 402;
    const long tid = (long)blockIdx.x * EXA_BLOCK + threadIdx.x;
    double in0[8], in1[8]; long ik0[2], ik1[2];
    p0_hesscL(P, x, y, th, tid, in0, ik0);
    p1_hesscL(P, x, y, th, tid, in1, ik1);
    p0_hesscE(P, in0, ik0, out, sink, sigma, tid, lds);
    p1_hesscE(P, in1, ik1, out, sink, sigma, tid, lds);
}
extern "C" __global__ void __launch_bounds__(EXA_BLOCK) exa_hess_l1(const long* __restrict__ P, const double* __restrict__ x,
        const double* __restrict__ y, const double* __restrict__ th, double* __restrict__ out, double sigma, double* __restrict__ sink) {
    __shared__ double lds_all[(EXA_BLOCK / 64) * 402];
    double* lds = lds_all + (threadIdx.x >> 6) * 402;
    __shared__ double xs_all[(EXA_BLOCK / 64) * 80];
    double* xs = xs_all + (threadIdx.x >> 6) * 80;
    const int lane = threadIdx.x & 63;
    const long tid = (long)blockIdx.x * EXA_BLOCK + threadIdx.x;
    // stretch geometry exactly as the generated exa_hesscl computes it for this model
    const long B0_ = P[@C0@] + P[@LO0@] + (0L) - 1L, B1_ = P[@C1@] + P[@LO1@] + (-1L) - 1L;
    const long B_ = B0_ < B1_ ? B0_ : B1_;
    const int d0_ = (int)(B0_ - B_), d1_ = (int)(B1_ - B_);
    int halo_ = d0_ + 2; halo_ = d1_ + 1 > halo_ ? d1_ + 1 : halo_;
    long xlast_ = P[@C0@] + P[@HI0@] - 1L + 2L - 1L; { const long l_ = P[@C1@] + P[@HI1@] - 1L + 0L - 1L; xlast_ = l_ > xlast_ ? l_ : xlast_; }
    const long a_ = B_ + (long)blockIdx.x * EXA_BLOCK + (threadIdx.x & ~63);
    long a0_ = a_ + lane; a0_ = a0_ < xlast_ ? a0_ : xlast_;
    long a1_ = a_ + 64 + lane; a1_ = a1_ < xlast_ ? a1_ : xlast_;
    const double g0_ = x[a0_], g1_ = x[lane < halo_ ? a1_ : a0_];
    double in0[8], in1[8]; long ik0[2], ik1[2];
    p0_hessclL(P, y, th, tid, in0, ik0);
    p1_hessclL(P, y, th, tid, in1, ik1);
    xs[lane] = g0_;
    if (lane < halo_) xs[64 + lane] = g1_;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    p0_hessclE(P, in0, ik0, xs, lane + d0_, out, sink, sigma, tid, lds);
    p1_hessclE(P, in1, ik1, xs, lane + d1_, out, sink, sigma, tid, lds);
}
"""


def words(src, k):
    """parameter words of pattern k read off the generated staged load / evaluation stage: (lo, hi, column, y offset, o2)"""
    L = src[src.index(f"void p{k}_hessclL("):]
    L = L[:L.index("\n}\n")]
    E = src[src.index(f"void p{k}_hessclE("):]
    E = E[:E.index("\n}\n")]
    lo, hi = re.search(r"const long I0 = P\[(\d+)\] \+ tid;\s+const long hi = P\[(\d+)\];", L).groups()
    col = re.search(r"const long k0 = P\[(\d+)\] \+ I;", L).group(1)
    yo = re.search(r"y\[P\[(\d+)\] \+ I\]", L)
    o2 = re.search(r"const long obase = P\[(\d+)\]", E).group(1)
    return int(lo), int(hi), int(col), int(yo.group(1)) if yo else -1, int(o2)


def main():
    N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    dev = torch.device("cuda:0")
    m = ExaModel(models.luksan_vlcek_model(N))
    src = m.kernel_source()
    assert "exa_hesscl(" in src, "the staged kernel is not generated for this model"
    w0, w1 = words(src, 0), words(src, 1)
    var = VARIANTS
    for tag, val in (("@LO0@", w0[0]), ("@HI0@", w0[1]), ("@C0@", w0[2]), ("@LO1@", w1[0]), ("@HI1@", w1[1]), ("@C1@", w1[2])):
        var = var.replace(tag, str(val))
    with tempfile.TemporaryDirectory() as td:
        hip = os.path.join(td, "ab.hip")
        with open(hip, "w") as fh:
            fh.write(src + var)
        co = os.path.join(td, "ab.hsaco")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--genco", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-w", "-o", co, hip])
        image = open(co, "rb").read()
    hiprt = ctypes.CDLL("libamdhip64.so.7")
    mod = ctypes.c_void_p()
    assert hiprt.hipModuleLoadData(ctypes.byref(mod), image) == 0
    fns = {}
    for name in ("exa_hess_g1", "exa_hess_l1"):
        f = ctypes.c_void_p()
        assert hiprt.hipModuleGetFunction(ctypes.byref(f), mod, name.encode()) == 0, name
        fns[name] = f
    # the words the pattern functions read (unsharded LV: constraint = pattern 0 over 1:N-2, objective = pattern 1 over 2:N)
    P = np.zeros(256, dtype=np.int64)
    P[w0[0]], P[w0[1]], P[w0[2]], P[w0[4]] = 0, N - 2, 1, 0
    if w0[3] >= 0:
        P[w0[3]] = 0
    P[w1[0]], P[w1[1]], P[w1[2]], P[w1[4]] = 0, N - 1, 2, 6 * (N - 2)
    Pd = torch.from_numpy(P).to(dev)
    x = torch.from_numpy(m.meta.x0 + 0.1 * np.random.default_rng(0).uniform(-1, 1, N)).to(dev)
    y = torch.from_numpy(np.random.default_rng(1).standard_normal(N - 2)).to(dev)
    out = torch.empty(m.meta.nnzh, dtype=torch.float64, device=dev)
    sink = torch.zeros(64, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    nblk = (N - 1 + 255) // 256

    def launch(name):
        args = [ctypes.c_void_p(Pd.data_ptr()), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(0),
                ctypes.c_void_p(out.data_ptr()), ctypes.c_double(0.5), ctypes.c_void_p(sink.data_ptr())]
        arr = (ctypes.c_void_p * len(args))(*[ctypes.cast(ctypes.pointer(a), ctypes.c_void_p) for a in args])
        rc = hiprt.hipModuleLaunchKernel(fns[name], nblk, 1, 1, 256, 1, 1, 0, ctypes.c_void_p(stream), arr, None)
        assert rc == 0, rc

    # the shipped kernels, each as exa_tune leaves it
    lib = {}
    for label, variant in (("exa_hess (shipped, tuned order)", "0"), ("exa_hesscl (shipped)", "1")):
        os.environ["EXAHIP_HESS_VARIANT"] = variant
        mv = ExaModel(models.luksan_vlcek_model(N))
        mv.tune(1, x, y)
        lib[label] = mv
    os.environ.pop("EXAHIP_HESS_VARIANT")
    ref = lib["exa_hess (shipped, tuned order)"].hess_coord(x, y, 0.5).clone()
    same = {}
    for name in fns:
        out.fill_(float("nan"))
        launch(name)
        torch.cuda.synchronize()
        same[name] = "bitwise equal to the library's hess_coord!" if torch.equal(out, ref) else f"DIFFERS (max |d| {float((out - ref).abs().max()):.3e}, nan {int(torch.isnan(out).sum())})"
    del ref
    runs = {**{k: (lambda mv=mv: mv.time_callback("hess", reps, x, y, 0.5, out=out)) for k, mv in lib.items()}}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200 if N <= 2e7 else 30

    def timed(name):
        e0.record()
        for _ in range(reps):
            launch(name)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    for name in fns:
        runs[name] = (lambda name=name: timed(name))
    times = {k: [] for k in runs}
    for k, f in runs.items():
        f()
    for rnd in range(6):
        for k, f in runs.items():
            times[k].append(f())
    alg = 8.0 * (m.meta.nnzh + 2 * N - 2)
    print(f"LV N={N}: hess_coord!, one process, one output buffer, {reps} launches x 6 A/B rounds; algorithmic bytes {alg / 1e9:.3f} GB")
    for k in runs:
        t = min(times[k])
        print(f"  {k:34s} min {t:.4f} ms  med {float(np.median(times[k])):.4f} ms  {alg / t / 1e6:6.0f} GB/s = {alg / t / 1e6 / 8000:.3f} of 8 TB/s   {same.get(k, '')}")


if __name__ == "__main__":
    main()
