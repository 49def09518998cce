set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ac; mkdir -p $O
OLD=tools/_old_r5/examodels.jl_amd; NEW=examodels.jl_amd
for rep in 1 2 3; do for p in $OLD $NEW; do timeout 200 python tools/lv_callbacks_ab.py $p 1e7 2>/dev/null | tail -1 >> $O/lv_callbacks_ab.txt; done; done
EXAHIP_KTAB=1 timeout 200 python tools/lv_callbacks_ab.py $NEW 1e7 2>/dev/null | tail -1 | sed 's/^/KTAB=1 /' >> $O/lv_callbacks_ab.txt
cat $O/lv_callbacks_ab.txt
