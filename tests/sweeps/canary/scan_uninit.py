#!/usr/bin/env python
"""Which register, which lanes and which stretch of the ISA: localises the read of a never-written register lane in the DEFAULT
build of the canary kernel (hprodw_canary.hip, exa_hprodw; profiles/NOTES.md rounds 3-5).  TEST INFRASTRUCTURE; needs hipcc and a
gfx950 GPU, not libexahip.

Method (everything on the compiler's own assembly, `hipcc -save-temps`, re-assembled with clang / ld.lld):
 0. baseline: the kernel's entry zeroes every VGPR (v1..) and AGPR it owns.  Deterministic output B.
 1. REGISTERS: one variant per register R: as baseline, then R := 0xFFFFFFFF in all lanes (NaN as the high word of a double).  Output
    != B  <=>  the kernel reads R in a lane before writing it.
 2. LANES: for every such R, the marker only in a subset of the lanes (8 groups of 8, then single lanes of the groups that matter).
 3. WHERE: for every such R a bisection over the text position p: a "sanitizer" (marker -> 0, all lanes, no flags touched) at the
    head of every basic block BEHIND p.  Control flow is forward but for a few short loops, so the marker survives exactly until
    execution passes p: output != B  <=>  the offending read sits before p.
Writes a report (stdout) that names the registers, lanes and the block holding the first offending read, with the ISA around it.

    python scan_uninit.py [WORKDIR] [--flags "-mllvm ..."]"""
import os
import re
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LLVM = "/opt/rocm/lib/llvm/bin"
BASE = "--genco --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -w".split()
KERNEL = "exa_hprodw"


def sh(cmd, **kw):
    return subprocess.run(cmd, check=True, capture_output=True, text=True, **kw)


class Scan:
    def __init__(self, work, extra):
        self.work = work
        os.makedirs(work, exist_ok=True)
        sh(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-o", f"{work}/canary_run", f"{HERE}/canary_run.cpp"])
        sh(["/opt/rocm/bin/hipcc", *BASE, *extra, "-save-temps", "-o", f"{work}/k.hsaco", f"{HERE}/hprodw_canary.hip"], cwd=work)
        asm = [n for n in os.listdir(work) if n.endswith("gfx950.s")]
        self.lines = open(f"{work}/{asm[0]}").read().split("\n")
        L = self.lines
        self.start = next(i for i, ln in enumerate(L) if ln.startswith(f"{KERNEL}:")) + 2      # the line after "; %bb.0:"
        assert L[self.start - 1].startswith("; %bb.0")
        self.end = next(i for i in range(self.start, len(L)) if L[i].strip().startswith("s_endpgm"))
        d = next(i for i in range(self.end, len(L)) if L[i].strip() == f".amdhsa_kernel {KERNEL}")
        self.desc = d
        get = lambda key: next((i, int(L[i].split()[-1])) for i in range(d, d + 60) if L[i].strip().startswith(key))   # noqa: E731
        self.i_nv, nfree = get(".amdhsa_next_free_vgpr")
        self.i_ns, _ = get(".amdhsa_next_free_sgpr")
        _, acc = get(".amdhsa_accum_offset")
        self.nv, self.na = min(acc, nfree), max(0, nfree - acc)
        # heads of basic blocks: labels and fall-through block comments
        self.heads = [i + 1 for i in range(self.start, self.end) if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", L[i])]
        self.n = 0
        print(f"{KERNEL}: {self.end - self.start} lines, {len(self.heads)} block heads, {self.nv} VGPRs + {self.na} AGPRs", flush=True)

    def regs(self):
        return [f"v{k}" for k in range(1, self.nv)] + [f"a{k}" for k in range(self.na)]

    def entry(self, reg=None, mask=(1 << 64) - 1):
        t = [f"\tv_mov_b32 v{k}, 0" for k in range(1, self.nv)] + [f"\tv_accvgpr_write_b32 a{k}, v1" for k in range(self.na)]
        if reg:
            t += [f"\ts_mov_b32 exec_lo, 0x{mask & 0xffffffff:x}", f"\ts_mov_b32 exec_hi, 0x{mask >> 32:x}", "\ts_nop 4"]
            t += [f"\tv_mov_b32 {reg}, -1"] if reg[0] == "v" else ["\tv_mov_b32 v1, -1", f"\tv_accvgpr_write_b32 {reg}, v1", "\ts_nop 1", "\tv_mov_b32 v1, 0"]
            t += ["\ts_mov_b64 exec, -1", "\ts_nop 4"]
        return t

    def sanitizer(self, reg):
        t = ["\ts_mov_b64 s[98:99], exec", "\ts_mov_b64 exec, -1", "\ts_nop 4"]
        if reg[0] == "v":
            t += [f"\tv_cmp_ne_u32_e64 s[100:101], -1, {reg}", "\ts_nop 4", f"\tv_cndmask_b32_e64 {reg}, 0, {reg}, s[100:101]"]
        else:      # through v1, parked in the first AGPR past the kernel's own
            t += [f"\tv_accvgpr_write_b32 a{self.na}, v1", "\ts_nop 1", f"\tv_accvgpr_read_b32 v1, {reg}", "\ts_nop 1",
                  "\tv_cmp_ne_u32_e64 s[100:101], -1, v1", "\ts_nop 4", "\tv_cndmask_b32_e64 v1, 0, v1, s[100:101]", "\ts_nop 1",
                  f"\tv_accvgpr_write_b32 {reg}, v1", "\ts_nop 1", f"\tv_accvgpr_read_b32 v1, a{self.na}", "\ts_nop 1"]
        return t + ["\ts_nop 4", "\ts_mov_b64 exec, s[98:99]", "\ts_nop 4"]

    def run(self, reg=None, mask=(1 << 64) - 1, behind=None):
        """Output of the variant: marker in `reg` (lanes `mask`) at the entry; sanitizers at every block head at text position >= behind."""
        L = list(self.lines)
        L[self.i_nv] = f"\t\t.amdhsa_next_free_vgpr {self.nv + self.na + 8}"
        L[self.i_ns] = "\t\t.amdhsa_next_free_sgpr 102"
        ins = {self.start: self.entry(reg, mask)}
        if behind is not None:
            for h in self.heads:
                if h >= behind:
                    ins[h] = ins.get(h, []) + self.sanitizer(reg)
        out = []
        for i, ln in enumerate(L):
            if i in ins:
                out += ins[i]
            out.append(ln)
        self.n += 1
        stem = f"{self.work}/v"
        open(stem + ".s", "w").write("\n".join(out))
        sh([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", stem + ".s", "-o", stem + ".o"])
        sh([f"{LLVM}/ld.lld", "-shared", stem + ".o", "-o", stem + ".hsaco"])
        r = subprocess.run(["timeout", "60", f"{self.work}/canary_run", f"{HERE}/hprodw_canary.dump", stem + ".hsaco", stem + ".bin"], capture_output=True, text=True)
        if r.returncode != 0:
            return None, None
        a = np.fromfile(stem + ".bin", dtype=np.uint64)
        return a[: len(a) // 2], a[len(a) // 2:]


def diff(a, b):
    return int((a != b).sum())


def main():
    argv = sys.argv[1:]
    extra = []
    if "--flags" in argv:
        k = argv.index("--flags")
        extra = argv[k + 1].split()
        del argv[k:k + 2]
    s = Scan(argv[0] if argv else "/tmp/canary_scan", extra)
    exp, plain = None, None
    # the untouched assembly, re-assembled: must show the fault as the compiler's own code object does
    L0 = s.lines
    open(f"{s.work}/p.s", "w").write("\n".join(L0))
    sh([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", f"{s.work}/p.s", "-o", f"{s.work}/p.o"])
    sh([f"{LLVM}/ld.lld", "-shared", f"{s.work}/p.o", "-o", f"{s.work}/p.hsaco"])
    for name in ("k", "p"):
        subprocess.run([f"{s.work}/canary_run", f"{HERE}/hprodw_canary.dump", f"{s.work}/{name}.hsaco", f"{s.work}/{name}.bin"], check=True)
        a = np.fromfile(f"{s.work}/{name}.bin", dtype=np.uint64)
        exp, plain = a[: len(a) // 2], a[len(a) // 2:]
        print(f"{'compiler code object' if name == 'k' else 're-assembled .s       '}: {diff(exp, plain)} of {len(exp)} entries differ bitwise from the recorded output", flush=True)
    _, B = s.run()
    print(f"baseline (registers zeroed at the entry): {diff(exp, B)} entries differ from the recorded output", flush=True)
    _, B2 = s.run()
    print(f"baseline again: {diff(B, B2)} entries differ from the first baseline run (0 = deterministic)", flush=True)
    # 1. registers
    culprits = []
    for r in s.regs():
        _, o = s.run(r)
        if o is None:
            print(f"  {r}: the run FAILED (fault / timeout)", flush=True)
            culprits.append((r, -1))
        elif diff(o, B):
            nanc = int(np.isnan(o.view(np.float64)).sum())
            print(f"  {r}: read before written — {diff(o, B)} entries change, {nanc} NaN", flush=True)
            culprits.append((r, diff(o, B)))
    print("registers read before written:", culprits, flush=True)
    # 2. lanes
    lanes_of = {}
    for r, _ in culprits[:16]:
        lanes = []
        for g in range(8):
            _, o = s.run(r, 0xFF << (8 * g))
            if o is None or diff(o, B):
                for ln in range(8 * g, 8 * g + 8):
                    _, o1 = s.run(r, 1 << ln)
                    if o1 is None or diff(o1, B):
                        lanes.append(ln)
        lanes_of[r] = lanes
        print(f"  {r}: lanes {lanes}", flush=True)
    # 3. where
    for r, _ in culprits[:16]:
        lo, hi = 0, len(s.heads)           # invariant: sanitizers from heads[lo] on -> still differs (read before heads[lo]) ... find the smallest
        # f(k) = output differs from B with sanitizers at heads[k:].  f(0) should be False (sanitized at the first head), f(len) True.
        _, o = s.run(r, behind=s.heads[0])
        f0 = o is None or diff(o, B) > 0
        if f0:
            print(f"  {r}: read inside the entry block (before line {s.heads[0] - s.start})", flush=True)
            continue
        while hi - lo > 1:
            mid = (lo + hi) // 2
            _, o = s.run(r, behind=s.heads[mid])
            if o is None or diff(o, B) > 0:
                hi = mid          # the read happens before heads[mid] is reached
            else:
                lo = mid
        a, b = s.heads[lo], (s.heads[hi] if hi < len(s.heads) else s.end)
        print(f"  {r}: first offending read between kernel lines {a - s.start} and {b - s.start} (block head index {lo})", flush=True)
        reg = re.compile(r"\b" + (r"[va]\[?(\d+)(:(\d+))?\]?"))
        k = int(r[1:])
        for i in range(max(s.start, a - 3), min(s.end, b + 1)):
            ln = s.lines[i]
            hit = False
            for m in reg.finditer(ln):
                if ln[m.start()] != r[0]:
                    continue
                x, y = int(m.group(1)), int(m.group(3) or m.group(1))
                hit |= x <= k <= y
            print(("   >> " if hit else "      ") + f"{i - s.start:6d} " + ln, flush=True)
    print(f"{s.n} variants assembled and run", flush=True)


if __name__ == "__main__":
    main()
