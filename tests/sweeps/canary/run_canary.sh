#!/bin/bash
# Replays the recorded exa_hprodw launch (hprodw_canary.dump) against code objects compiled from hprodw_canary.hip with the
# default flags and with the library's conservative allocator flags.  Needs only hipcc and a gfx950 GPU — not libexahip.
#   default DIFFERENT, safe equal : the allocator fault is present and the library's fallback (exa_build.cpp safe_flags) avoids it
#   both equal                    : this compiler no longer shows the fault on this kernel
#   safe DIFFERENT                : the fallback is NOT sufficient — exit status 1
#   run_canary.sh [DIR=this directory] [VARIANTS="default safe sgpr-fast sgpr-basic O1"]
here="$(cd "$(dirname "$0")" && pwd)"
dir="${1:-$here}"
want="${2:-default safe sgpr-fast sgpr-basic O1}"
tmp="$(mktemp -d)"
base="--genco --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -w"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o "$tmp/canary_host" "$here/canary_host.cpp" || exit 2
status=0
for variant in "default|" "safe|-mllvm -grow-region-complexity-budget=0" "sgpr-fast|-mllvm -sgpr-regalloc=fast" "sgpr-basic|-mllvm -sgpr-regalloc=basic" "O1|-O1"; do
    name="${variant%%|*}"; flags="${variant#*|}"
    case " $want " in *" $name "*) ;; *) continue ;; esac
    /opt/rocm/bin/hipcc $base $flags -o "$tmp/$name.hsaco" "$dir/hprodw_canary.hip" || { echo "$name: does not compile"; continue; }
    for p in "" poison; do
        printf "%-10s %-6s " "$name" "$p"
        "$tmp/canary_host" "$dir/hprodw_canary.dump" "$tmp/$name.hsaco" $p
        rc=$?
        if { [ "$name" = safe ] || [ "$name" = sgpr-basic ]; } && [ $rc -ne 0 ]; then status=1; fi
    done
done
rm -rf "$tmp"
exit $status
