set -x
export R=$PWD O=$PWD/gpurun_out/r5p
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
c=4
RC="python $R/tools/run_callbacks.py $c"
timeout 600 $RC --reps 200 > $O/callbacks_config$c.json 2> $O/callbacks_config$c.err
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof/cb_stats_c$c -o r5 -- $RC --reps 20 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof/cb_fetch_c$c -o r5 -- $RC --reps 3 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof/cb_write_c$c -o r5 -- $RC --reps 3 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $O/prof/cb_sq_c$c -o r5 -- $RC --reps 3 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/prof/cb_grbm_c$c -o r5 -- $RC --reps 3 > /dev/null 2>&1
python $R/tools/roofline_table.py $O/callbacks_config$c.json $O/prof/cb_stats_c$c $O/prof/cb_fetch_c$c $O/prof/cb_write_c$c $O/prof/cb_sq_c$c $O/prof/cb_grbm_c$c > $O/r5_kernels_config$c.md
B="--no-cpu --no-config5-n1 --no-extra-configs"
python $R/bench.py --config $c $B > $O/bench_pre_config$c.json 2> $O/bench_pre_config$c.err
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof/stats_c$c -o r5 -- python $R/bench.py --config $c $B > $O/bench_config${c}_under_rocprof.json 2>/dev/null
X="--config $c $B --steps 20 --warmup 5 --preheat-ms 0"
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof/fetch_c$c -o r5 -- python $R/bench.py $X > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof/write_c$c -o r5 -- python $R/bench.py $X > /dev/null 2>&1
python $R/tools/prof_summary.py $O/prof/stats_c$c > $O/r5_stats_config$c.txt
python $R/tools/prof_summary.py $O/prof/fetch_c$c > $O/pmc_fetch_config$c.txt
python $R/tools/prof_summary.py $O/prof/write_c$c > $O/pmc_write_config$c.txt
python $R/tools/make_traffic_json.py $c $O/bench_pre_config$c.json $O/pmc_fetch_config$c.txt $O/pmc_write_config$c.txt > $O/r5_traffic_config$c.json
cp $O/r5_traffic_config$c.json $R/profiles/
cd $R
python bench.py > $O/r5_bench_default.json 2> $O/r5_bench_default.err
rm -rf $O/prof
tail -c 400 $O/r5_bench_default.json
