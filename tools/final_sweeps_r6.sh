#!/bin/bash
# Round-6 closing sweeps on the FINAL tree, library AS SHIPPED (guard flag + round-4 fence + re-planning), FRESH seeds, register poison on,
# one process per chunk / seed.   tools/final_sweeps_r5.sh prebuild  (build machine)   |   tools/final_sweeps_r5.sh [OUTDIR]  (GPU box)
R0=${R0:-100}; NR=${NR:-30}; D0=${D0:-8000}; ND=${ND:-24}; E0=${E0:-9000}
if [ "$1" = prebuild ]; then
  for fl in blocks unit mixed; do python tests/sweeps/range_model_check.py $R0 $NR $fl --prebuild > /dev/null 2>&1 & done
  (for s in $(seq $D0 $((D0 + ND - 1))); do echo "$s 12 6"; done; for s in $(seq $E0 $((E0 + ND - 1))); do echo "$s 8 4"; done) | PREBUILD=1 xargs -P 5 -L 1 python tests/sweeps/random_model_check.py > /dev/null 2>&1
  wait; exit 0
fi
O=${1:-gpurun_out/r6e}; mkdir -p $O
for fl in blocks unit mixed; do
  for s in $(seq $R0 10 $((R0 + NR - 1))); do
    timeout 900 python tests/sweeps/range_model_check.py $s 10 $fl --poison 2>&1 | grep -E "^seed|Error|error" >> $O/final_range_$fl.log || echo "chunk $s $fl: timeout / crash" >> $O/final_range_$fl.log
  done
done
bash tests/sweeps/sweep_deep_poison.sh $D0 $ND 12 6 > $O/final_deep_12x6.txt 2>&1
bash tests/sweeps/sweep_deep_poison.sh $E0 $ND 8 4 > $O/final_deep_8x4.txt 2>&1
echo "ok lines:"; grep -c " ok" $O/final_range_*.log $O/final_deep_*.txt; echo "BAD / CRASH:"; grep -l "BAD\|CRASH\|timeout" $O/final_* || echo none
