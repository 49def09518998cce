set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5h; mkdir -p $O
timeout 1400 bash tests/sweeps/sweep_deep_poison.sh 8100 40 12 6 > $O/final_deep_12x6.txt 2>&1
timeout 900 bash tests/sweeps/sweep_deep_poison.sh 9100 40 8 4 > $O/final_deep_8x4.txt 2>&1
for s in 240 250; do timeout 500 python tests/sweeps/range_model_check.py $s 10 mixed --poison 2>&1 | grep -E "^seed|Error|error" >> $O/final_range_mixed_tail.log || echo "chunk $s mixed: timeout / crash" >> $O/final_range_mixed_tail.log; done
grep -c " ok" $O/final_deep_*.txt $O/final_range_mixed_tail.log; grep -l "BAD\|CRASH\|timeout" $O/final_deep_*.txt $O/final_range_mixed_tail.log || echo none
