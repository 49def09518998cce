set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5x; mkdir -p $O
OLD=tools/_old_r5/examodels.jl_amd; NEW=examodels.jl_amd
for N in 1e8 1e7; do for v in 2 1 0; do for p in $OLD $NEW $OLD $NEW; do timeout 200 python tools/lv_hess_ab.py $p $N $v 2>/dev/null | tail -1 >> $O/lv_hess_ab.txt; done; done; done
EXAHIP_FAST_EXP=0 timeout 200 python tools/lv_hess_ab.py $NEW 1e8 2 2>/dev/null | tail -1 | sed 's/^/FAST_EXP=0 /' >> $O/lv_hess_ab.txt
cat $O/lv_hess_ab.txt
timeout 600 python -m pytest tests/test_registered_functions.py -m gpu -x -q 2>&1 | tail -3
