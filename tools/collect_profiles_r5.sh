# After `bash tools/refresh_profiles_r5.sh` ran on the GPU box (its outputs come back under gpurun_out/r5p/): copies the files
# the documents cite into profiles/ and rewrites the number rows of DESIGN.md §5 from them.  Run from the repo root.
set -e
O=gpurun_out/r5p
for c in 2 3 4; do cp $O/callbacks_config$c.json profiles/r5_callbacks_config$c.json; cp $O/r5_kernels_config$c.md profiles/; done
if [ -f $O/r5_bench_default.json ]; then
for c in 2 3 4 5; do cp $O/r5_stats_config$c.txt profiles/; cp $O/r5_traffic_config$c.json profiles/; done
tail -1 $O/r5_bench_default.json > profiles/r5_bench_default.json
fi
python tools/measurements_md.py
