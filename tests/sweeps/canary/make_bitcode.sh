#!/bin/bash
# How hprodw_canary.bc (the single-kernel LLVM bitcode of the reproducer, for `llc` alone) is made from hprodw_canary.hip — ROCm 7.2.0:
#   the optimized device bitcode of the module (hipcc -save-temps), once more through opt -O3 (as hipcc's backend step does with an IR input),
#   every kernel but exa_hprodw internalised and dropped.
set -e
here="$(cd "$(dirname "$0")" && pwd)"; tmp="$(mktemp -d)"; cd "$tmp"
LLVM=/opt/rocm/lib/llvm/bin
/opt/rocm/bin/hipcc --genco --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -w -save-temps -o k.hsaco "$here/hprodw_canary.hip"
$LLVM/opt -O3 -mtriple=amdgcn-amd-amdhsa -mcpu=gfx950 hprodw_canary-hip-amdgcn-amd-amdhsa-gfx950.bc -o o3.bc
$LLVM/opt -passes='internalize,globaldce' -internalize-public-api-list=exa_hprodw o3.bc -o "$here/hprodw_canary.bc"
rm -rf "$tmp"
