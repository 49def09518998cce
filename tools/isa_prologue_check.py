#!/usr/bin/env python
"""Looks for the ONE code-generation fault behind every wrong result this project has seen from hipcc (profiles/NOTES.md round 5,
tests/sweeps/canary/REPORT.md): vector instructions placed at the head of a control-flow JOIN block, IN FRONT OF the instruction
that re-enables the lanes (`s_or_b64 exec, exec, s[..]` = SI_END_CF, or the `s_or_saveexec_b64` of an else-flow block).

    .LBB5_858:                              ; join of `if (lane == 0) ds_write ...`: EXEC still holds the narrowed mask
        v_writelane_b32 v255, s76, 16       ;   SGPR spill (ignores EXEC: harmless)
        v_accvgpr_write_b32 a50, v108       ;   live-range split copy of the VGPR allocator: EXEC-masked -> only lane 0 is copied
        s_mov_b64 s[2:3], s[86:87]          ;   live-range split copy of the SGPR allocator  <- what lets the VGPR copies in
        s_or_b64 exec, exec, s[4:5]         ;   lanes re-enabled HERE: too late
        ...
        v_accvgpr_read_b32 v30, a50         ;   lanes 1..63 read what was never written

LLVM's SIInstrInfo::isBasicBlockPrologue keeps spill code behind the exec restore, and counts SGPR SPILLS in front of it as part of
the prologue, but not SGPR live-range-split COPIES ("FIXME: Copies inserted in the block prolog for live-range split should also
be included"): once the greedy SGPR allocator has split a live range at such a block, the VGPR allocator's own insertion point
stops in front of the copy — in front of the exec restore.  Needs: SGPR pressure (region splitting of SGPRs) and anything the VGPR
allocator inserts at that block (split copies, AGPR / scratch spills).

TEST / BUILD-MACHINE INFRASTRUCTURE (needs llvm-objdump); the library itself avoids the fault by flags (exa_build.cpp) and does not
call this.  Usage:  isa_prologue_check.py FILE.hsaco...   |   --cache DIR   (every code object of a kernel cache)
Exit status 1 when a fault site is found."""
import os
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
_LABEL = re.compile(r"^[0-9a-f]+ <(.+)>:$")
_EXEC_FREE_V = ("v_readlane_b32", "v_writelane_b32")
_VMEM = ("ds_", "global_", "flat_", "scratch_", "buffer_", "tbuffer_", "image_")


def disassemble(path):
    """-> {kernel: [(mnemonic, operands, head)]}: head = the label that opens a basic block at this instruction ("" for a block that only
    follows a branch, None inside a block)."""
    out = subprocess.run([OBJDUMP, "-d", "--symbolize-operands", "--mcpu=gfx950", path], capture_output=True, text=True, check=True).stdout
    kernels, cur, head = {}, None, ""
    for ln in out.split("\n"):
        m = _LABEL.match(ln.strip())
        if m:
            name = m.group(1)
            if not re.fullmatch(r"L\d+", name):
                cur = kernels.setdefault(name, [])
            head = name
            continue
        if cur is None or not ln.startswith("\t"):
            continue
        text = ln.split("//")[0].strip()
        if not text:
            continue
        mn, _, ops = text.partition(" ")
        cur.append((mn, ops.strip(), head))
        head = "" if mn.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm", "s_swappc")) else None
    return kernels


def _writes_exec(mn, ops):
    if "saveexec" in mn or mn.startswith("v_cmpx"):
        return True
    dst = ops.split(",")[0].strip()
    return mn.startswith("s_") and dst in ("exec", "exec_lo", "exec_hi")


def _exec_dependent(mn):
    if mn.startswith("v_"):
        return not mn.startswith(_EXEC_FREE_V)
    return mn.startswith(_VMEM)


def fault_sites(insts):
    """[(index of the exec restore, [indices of EXEC-dependent instructions in front of it])] for one kernel, and the number of exec
    restores that could not be judged.  Judged: the restore sits in a block whose label is the target of an `s_cbranch_execz` — the
    join block of a divergent region, entered with the region's narrowed EXEC (or EXEC = 0): everything between the label and the
    restore is the block's prologue.  (A body that runs straight into its join — branch removed by the skip threshold, label gone —
    cannot be told from its prologue in the text; recompile with -mllvm -amdgpu-skip-threshold=0 to see those.)"""
    joins = {ops.strip() for mn, ops, _ in insts if mn == "s_cbranch_execz"}
    sites, unjudged = [], 0
    for i, (mn, ops, head) in enumerate(insts):
        restore = (mn == "s_or_b64" and re.match(r"exec,\s*exec,", ops)) or (mn == "s_or_saveexec_b64" and not ops.endswith("-1"))
        if not restore or head is not None:         # (a restore that opens its block has nothing in front of it)
            continue
        bad, j, judged = [], i - 1, False
        while j >= 0:
            pmn, pops, phead = insts[j]
            if _writes_exec(pmn, pops):
                break
            if _exec_dependent(pmn):
                bad.append(j)
            if phead is not None:
                judged = phead in joins
                break
            j -= 1
        if judged and bad:
            sites.append((i, sorted(bad)))
        elif not judged:
            unjudged += 1
    return sites, unjudged


def check_file(path):
    res = {}
    for k, insts in disassemble(path).items():
        sites, unjudged = fault_sites(insts)
        res[k] = (sites, unjudged, insts)
    return res


def main(argv):
    files = []
    if argv and argv[0] == "--cache":
        files = sorted(os.path.join(argv[1], f) for f in os.listdir(argv[1]) if f.endswith(".hsaco"))
    else:
        files = argv
    nk = nbad = 0
    for f in files:
        for k, (sites, unjudged, insts) in check_file(f).items():
            nk += 1
            if sites:
                nbad += 1
                print(f"{os.path.basename(f)} {k}: {len(sites)} exec restores with vector instructions in front of them")
                for i, bad in sites[:3]:
                    for j in range(min(bad), i + 1):
                        print(("   >> " if j in bad else "      ") + insts[j][0] + " " + insts[j][1])
    print(f"{len(files)} code objects, {nk} kernels, {nbad} with the fault pattern")
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
