set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ad; mkdir -p $O
OLD=tools/_old_r5b/examodels.jl_amd; NEW=examodels.jl_amd
for rep in 1 2; do for w in "1e7 lv" "1e6 rocket" "0 acopf"; do for p in $OLD $NEW; do timeout 300 python tools/lv_callbacks_ab.py $p $w 2>/dev/null | tail -1 >> $O/callbacks_ab.txt; done; done; done
cat $O/callbacks_ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -m gpu -x -q 2>&1 | tail -3
