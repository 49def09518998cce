# After `bash tools/refresh_profiles_r3.sh` ran on the GPU box (its outputs come back under gpurun_out/r3p/): copies the files
# the documents cite into profiles/ and rewrites the number rows of DESIGN.md §5 from them.  Run from the repo root.
set -e
O=gpurun_out/r3p
for c in 2 3 4; do cp $O/callbacks_config$c.json profiles/r3_callbacks_config$c.json; cp $O/r3_kernels_config$c.md profiles/; done
if [ -f $O/r3_bench_default.json ]; then
for c in 2 3 4 5; do cp $O/r3_stats_config$c.txt profiles/; cp $O/r3_traffic_config$c.json profiles/; done
tail -1 $O/r3_bench_default.json > profiles/r3_bench_default.json
fi
python tools/design_tables.py
