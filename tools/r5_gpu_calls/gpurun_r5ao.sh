# kernel table of config 2 again (the looped kernels' names were missing from tools/run_callbacks.py's callback -> kernel map)
set -x
cd $GRAFT_REPO_ROOT
SKIP_BENCH=1 timeout 300 bash tools/refresh_profiles_r5.sh 2 > gpurun_out/r5ao_refresh.log 2>&1
grep -E "^\| (cons|jac) " gpurun_out/r5p/r5_kernels_config2.md | cut -c1-200
