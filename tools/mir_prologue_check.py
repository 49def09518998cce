#!/usr/bin/env python
"""The compiler fault of tests/sweeps/canary/REPORT.md, looked for in the compiler's OWN intermediate code: every basic block that holds an exec
restore (`$exec = S_OR_B64 $exec, ...` = SI_END_CF, or `S_OR_SAVEEXEC_B64`) is searched for instructions IN FRONT OF the restore that define a
vector register (VGPR / AGPR / AV classes: COPYs, V_* instructions, vector spills and reloads).  Unlike tools/isa_prologue_check.py (which reads the final
ISA and cannot see a join block whose `execz` branch — and label — was removed) this has no blind spot: the MIR still has every block.

BUILD-MACHINE AUDIT TOOL (hipcc; no GPU): compiles each source with -mllvm -print-after=greedy and reads the LAST dump of every function (the
allocator runs three times: SGPR, WWM, VGPR).  Slow and verbose (tens of MB of dump per module): for audits, not for the test suite.

    mir_prologue_check.py [--flags "-mllvm -sgpr-regalloc=basic"] SOURCE.hip...        exit 1 when a site is found"""
import re
import subprocess
import sys

BASE = "--genco --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -w".split()
_VEC_DEF = re.compile(r"^\s*(?:\d+B\s+)?(?:(?:early-clobber|dead|renamable|undef|internal)\s+)*(?:%\d+(?:\.\w+)?:(?:vgpr|vreg|av|agpr|areg)_\w+|\$(?:vgpr|agpr)\d+\w*)")
_SGPR_SPILL = re.compile(r"SI_SPILL_S\d+_TO_VGPR|SI_RESTORE_S\d+_FROM_VGPR|SI_SPILL_S\d+_(SAVE|RESTORE)")


def dumps(text):
    """{function: [lines of its LAST 'after greedy' dump]}"""
    out, cur, name = {}, None, None
    for ln in text.split("\n"):
        if ln.startswith("# *** IR Dump After Greedy Register Allocator"):
            cur, name = [], None
            continue
        if cur is None:
            continue
        m = re.match(r"# Machine code for function (\S+):", ln)
        if m:
            name = m.group(1)
            out[name] = cur          # a later dump of the same function replaces the earlier one
            continue
        if ln.startswith("# End machine code"):
            cur = None
            continue
        cur.append(ln)
    return out


def sites(lines):
    """[(block header, [instructions defining a vector register in front of the block's exec restore])]"""
    res, block, head = [], [], None
    def close():
        if head is None:
            return
        for i, ln in enumerate(block):
            if re.search(r"\$exec = S_OR_B64 (?:killed )?\$exec,|S_OR_SAVEEXEC_B64", ln) and "-1," not in ln:
                bad = [q.strip() for q in block[:i] if _VEC_DEF.match(q) and not _SGPR_SPILL.search(q)]
                if bad:
                    res.append((head.strip(), bad))
                break
    for ln in lines:
        if re.match(r"^\s*(?:\d+B\s+)?bb\.\d+", ln):
            close()
            block, head = [], ln
        elif head is not None and ln.strip() and not ln.strip().startswith((";", "successors", "liveins", "predecessors")):
            block.append(ln)
    close()
    return res


def check_source(path, extra):
    r = subprocess.run(["/opt/rocm/bin/hipcc", *BASE, *extra, "-mllvm", "-print-after=greedy", "-o", "/dev/null", path], capture_output=True, text=True)
    if r.returncode != 0 and "IR Dump" not in r.stderr:
        raise SystemExit(f"{path}: does not compile\n{r.stderr[-2000:]}")
    found = {}
    fns = dumps(r.stderr)
    for fn, lines in fns.items():
        s = sites(lines)
        if s:
            found[fn] = s
    return len(fns), found


def main(argv):
    extra = []
    if argv and argv[0] == "--flags":
        extra, argv = argv[1].split(), argv[2:]
    nfn = nbad = 0
    for p in argv:
        n, found = check_source(p, extra)
        nfn += n
        nbad += len(found)
        for fn, s in found.items():
            print(f"{p} {fn}: {len(s)} block(s) with vector definitions in front of the exec restore")
            for head, bad in s[:2]:
                print("   ", head[:100])
                for b in bad[:6]:
                    print("       ", b[:160])
    print(f"{len(argv)} sources, {nfn} functions, {nbad} with the fault pattern")
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
