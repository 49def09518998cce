#!/bin/bash
# Round 5, on the GPU box: what the guard on the cause of the compiler fault costs (tests/sweeps/canary/REPORT.md).
#   basic  = the library's base flags (-mllvm -sgpr-regalloc=basic, csrc/exa_build.cpp)
#   greedy = the compiler's default SGPR allocator (what rounds 1-4 shipped)
#   grow0  = greedy + round 4's fallback flags on everything (-grow-region-complexity-budget=0)
# A/B/A/B in ONE process per callback, one output buffer, bitwise comparison of the outputs (tools/ab_variants.py).
O=${1:-gpurun_out/r5b}; mkdir -p $O
V="basic: greedy:EXAHIP_SGPR_REGALLOC=greedy grow0:EXAHIP_SGPR_REGALLOC=greedy,EXAHIP_HIPCC_FLAGS=-mllvm+-grow-region-complexity-budget=0"
{
for cb in hess jac cons grad; do SWEEP_MODEL=lv SWEEP_N=1e7 python tools/ab_variants.py --cb $cb $V; done
SWEEP_MODEL=lv SWEEP_N=1e8 python tools/ab_variants.py --cb hess $V
for cb in hess jac cons; do SWEEP_MODEL=rocket python tools/ab_variants.py --cb $cb $V; done
for cb in hess jac cons grad; do SWEEP_MODEL=acopf python tools/ab_variants.py --cb $cb $V; done
} 2>&1 | grep -v "amdgpu.ids" | tee $O/sgpr_regalloc_ab.txt
