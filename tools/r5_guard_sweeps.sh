#!/bin/bash
# Round 5: the guard on the CAUSE of the compiler fault, ALONE (csrc/exa_build.cpp base_flags: -mllvm -sgpr-regalloc=basic) — the second fence
# (modules with over-sized kernels rebuilt with -grow-region-complexity-budget=0) switched OFF and the first, over-sized plan of the product
# windows kept (EXAHIP_WINDOW_REPLAN=0: the canary's kernel shape): every entry point of random range models and deep data-indexed models
# after a NaN poisoning of the register files, one process per chunk / seed.
#   tools/r5_guard_sweeps.sh prebuild     on the build machine: compile the models' modules into the kernel cache (no GPU)
#   tools/r5_guard_sweeps.sh [OUTDIR]     on the GPU box
export EXAHIP_SAFE_FLAGS=none EXAHIP_WINDOW_REPLAN=0
NR=${NR:-12}; ND=${ND:-12}
if [ "$1" = prebuild ]; then
  for fl in blocks unit mixed; do python tests/sweeps/range_model_check.py 0 $NR $fl --prebuild > /dev/null 2>&1 & done
  (for s in $(seq 2000 $((2000 + ND - 1))); do echo "$s 12 6"; done; for s in $(seq 3000 $((3000 + ND - 1))); do echo "$s 8 4"; done) | PREBUILD=1 xargs -P 5 -L 1 python tests/sweeps/random_model_check.py > /dev/null 2>&1
  wait; exit 0
fi
O=${1:-gpurun_out/r5b}; mkdir -p $O
for fl in blocks unit mixed; do
  timeout 1500 python tests/sweeps/range_model_check.py 0 $NR $fl --poison 2>&1 | grep -E "^seed|Error|error" > $O/guard_range_$fl.log || echo "chunk $fl: timeout / crash" >> $O/guard_range_$fl.log
done
bash tests/sweeps/sweep_deep_poison.sh 2000 $ND 12 6 > $O/guard_deep_12x6.txt 2>&1
bash tests/sweeps/sweep_deep_poison.sh 3000 $ND 8 4 > $O/guard_deep_8x4.txt 2>&1
echo "ok lines:"; grep -c " ok" $O/guard_range_*.log $O/guard_deep_*.txt; echo "BAD / CRASH:"; grep -l "BAD\|CRASH\|timeout" $O/guard_* || echo none
