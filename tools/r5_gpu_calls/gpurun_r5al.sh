# final tree of round 5: the whole GPU suite, smoke, then the measurement set (tools/refresh_profiles_r5.sh)
set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5al
(cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r5al/gpu_suite.txt 2>&1; tail -2 gpurun_out/r5al/gpu_suite.txt; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1)
timeout 900 bash tools/refresh_profiles_r5.sh > gpurun_out/r5p_refresh.log 2>&1
tail -3 gpurun_out/r5p_refresh.log
