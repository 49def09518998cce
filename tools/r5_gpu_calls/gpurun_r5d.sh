set -x
O=gpurun_out/r5d; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite.txt 2>&1; grep -n "passed\|failed\|Fatal" $O/gpu_suite.txt | tail -3
timeout 3000 bash tools/refresh_profiles_r5.sh 2 3 4 > $O/refresh.log 2>&1; tail -5 $O/refresh.log
