#!/usr/bin/env python
"""Accuracy of the SpecialFunctions device routines (kSpecialPrelude of examodels.jl_amd/csrc/exa_gen_prelude.cpp) without a GPU: the
prelude text is compiled for the HOST (ROCm's clang, -O3 -ffp-contract=fast -mfma: contraction on, as hipcc's default) and every routine is
compared with 40-digit mpmath over its argument list, cross-over points of the algorithms included.  Output: profiles/r4_special_functions.txt.

    python tools/special_accuracy.py            # prints one line per routine: worst relative error and where
"""
import os
import re
import subprocess
import sys
import tempfile

import mpmath as mp
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = r'''
#include <cmath>
#include <cstdio>
#include <cstring>
#define __device__
#define __forceinline__ inline
#define EXA_PI 3.14159265358979323846
static double sinpi(double x){ return sin(M_PI*(x - 2.0*floor(x/2.0))); }
static double cospi(double x){ return cos(M_PI*(x - 2.0*floor(x/2.0))); }
#include "special.inc"
static double ev(const char* f,double x,double y){ double r=NAN;
  if(!strcmp(f,"psi0")) r=exa_polygamma<0>(x); else if(!strcmp(f,"psi1")) r=exa_polygamma<1>(x);
  else if(!strcmp(f,"psi2")) r=exa_polygamma<2>(x); else if(!strcmp(f,"psi3")) r=exa_polygamma<3>(x);
  else if(!strcmp(f,"invpsi")) r=exa_invdigamma(x); else if(!strcmp(f,"erfi")) r=exa_erfi(x); else if(!strcmp(f,"dawson")) r=exa_dawson(x);
  else if(!strcmp(f,"beta")) r=exa_beta(x,y); else if(!strcmp(f,"logbeta")) r=exa_logbeta(x,y);
  else if(!strcmp(f,"ai")) r=exa_airy<0>(x); else if(!strcmp(f,"aip")) r=exa_airy<1>(x); else if(!strcmp(f,"bi")) r=exa_airy<2>(x); else if(!strcmp(f,"bip")) r=exa_airy<3>(x);
  return r; }
int main(){ char f[64]; double x,y; while(scanf("%63s %lf %lf",f,&x,&y)==3) printf("%.17g\n", ev(f,x,y)); return 0; }
'''


def main():
    mp.mp.dps = 40
    src = open(os.path.join(ROOT, "examodels.jl_amd", "csrc", "exa_gen_prelude.cpp")).read()
    text = re.search(r'kSpecialPrelude = R"HIP\((.*?)\)HIP";', src, re.S).group(1)
    td = tempfile.mkdtemp()
    open(os.path.join(td, "special.inc"), "w").write(text)
    open(os.path.join(td, "host.cpp"), "w").write(HOST)
    exe = os.path.join(td, "host")
    subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-O3", "-std=c++17", "-ffp-contract=fast", "-mfma", "-w", "-o", exe, os.path.join(td, "host.cpp")])
    rng = np.random.default_rng(0)
    F = {"psi0": lambda x, y: mp.digamma(x), "psi1": lambda x, y: mp.polygamma(1, x), "psi2": lambda x, y: mp.polygamma(2, x),
         "psi3": lambda x, y: mp.polygamma(3, x), "invpsi": None, "erfi": lambda x, y: mp.erfi(x),
         "dawson": lambda x, y: mp.sqrt(mp.pi) / 2 * mp.exp(-x * x) * mp.erfi(x), "beta": lambda x, y: mp.beta(x, y),
         "logbeta": lambda x, y: mp.log(abs(mp.beta(x, y))), "ai": lambda x, y: mp.airyai(x), "aip": lambda x, y: mp.airyai(x, derivative=1),
         "bi": lambda x, y: mp.airybi(x), "bip": lambda x, y: mp.airybi(x, derivative=1)}
    pts = {}
    pts["psi0"] = pts["psi1"] = pts["psi2"] = pts["psi3"] = [(float(v), 0.0) for v in np.concatenate([rng.uniform(0.01, 30, 40), rng.uniform(-8, 0, 30), [1e-3, 100.0, 1e4, 0.5, 1.0, 2.0]])]
    pts["invpsi"] = [(float(v), 0.0) for v in rng.uniform(-10, 5, 40)]
    pts["erfi"] = pts["dawson"] = [(float(v), 0.0) for v in np.concatenate([rng.uniform(-8, 8, 60), [0.0, 1e-8, 5.99, 6.0, 6.01, 12.0, 20.0, -20.0]])]
    pts["beta"] = pts["logbeta"] = [(float(a), float(b)) for a, b in zip(np.concatenate([rng.uniform(0.05, 20, 40), rng.uniform(-3, 3, 20)]),
                                                                         np.concatenate([rng.uniform(0.05, 20, 40), rng.uniform(0.1, 3, 20)]))]
    pts["ai"] = pts["aip"] = pts["bi"] = pts["bip"] = [(float(v), 0.0) for v in np.concatenate([rng.uniform(-25, 25, 150), [0.0, 8.99, 8.999, 9.0, 9.01, -8.99, -9.0, -9.01, 1e-9, -1e-9, 40.0, -60.0]])]
    worst_all = 0.0
    for name, P in pts.items():
        inp = "".join(f"{name} {x!r} {y!r}\n" for x, y in P)
        out = subprocess.run([exe], input=inp, capture_output=True, text=True).stdout.split()
        worst, wa = 0.0, None
        for (x, y), o in zip(P, out):
            o = float(o)
            if name == "invpsi":
                ref = mp.findroot(lambda t: mp.digamma(t) - x, max(float(np.exp(x) + 0.5), 1e-3) if x >= -2.22 else -1 / (x + 0.5772))
            else:
                ref = F[name](mp.mpf(x), mp.mpf(y))
            scale = abs(ref)
            if name in ("ai", "bi") and x < 0:          # oscillating side: relative to the amplitude
                scale = max(scale, 0.56 / abs(x) ** 0.25 * 0.3)
            if name in ("aip", "bip") and x < 0:
                scale = max(scale, 0.56 * abs(x) ** 0.25 * 0.3)
            e = float(abs(o - ref) / scale) if scale != 0 else abs(o)
            if e > worst:
                worst, wa = e, (x, y)
        worst_all = max(worst_all, worst)
        print(f"{name:8s} worst relative error {worst:.2e} at {wa}  ({len(P)} arguments)")
    return 0 if worst_all < 1e-12 else 1


if __name__ == "__main__":
    sys.exit(main())
