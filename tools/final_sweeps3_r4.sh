# After the fix of the owner-pull key kernels: the five seeds that failed, the two deep sweeps again (crashes now reported), then the
# whole GPU suite and smoke() on the final tree.
O=gpurun_out/r4s3; mkdir -p $O
for a in "3021 8 4" "3039 8 4" "3003 8 4" "3005 8 4" "2031 12 6"; do set -- $a
  POISON=1 CHECK_ALL=1 timeout 200 python tests/sweeps/random_model_check.py $1 $2 $3 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-600 >> $O/failed_seeds_again.txt
done
cat $O/failed_seeds_again.txt
timeout 800 bash tests/sweeps/sweep_deep_poison.sh 2000 48 12 6 > $O/deep_2000_48.txt 2>&1
timeout 600 bash tests/sweeps/sweep_deep_poison.sh 3000 40 8 4 > $O/deep_3000_40.txt 2>&1
grep -c " ok" $O/deep_*.txt; grep -h "BAD\|CRASH" $O/deep_*.txt | cut -c1-300 | head
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_full.log 2>&1; tail -14 $O/pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
