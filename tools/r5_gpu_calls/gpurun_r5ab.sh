# final-tree measurement set of round 5 (third session): tools/refresh_profiles_r5.sh = every callback of configs 2-4 under rocprofv3 stats + PMC passes,
# bench lines of configs 2-5 with stats and the traffic JSONs of the hess kernel, then the default bench line with those JSONs in place
set -x
cd $GRAFT_REPO_ROOT
timeout 1700 bash tools/refresh_profiles_r5.sh > gpurun_out/r5p_refresh.log 2>&1
tail -5 gpurun_out/r5p_refresh.log
ls gpurun_out/r5p | head -50
