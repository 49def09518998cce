mkdir -p gpurun_out/r2p
bash tools/refresh_profiles_r2.sh > gpurun_out/r2p/refresh.log 2>&1
for M in acopf rocket lv; do python tools/bench_configs.py $M > gpurun_out/r2p/cfg_$M.json 2>/dev/null; done
tail -3 gpurun_out/r2p/refresh.log
