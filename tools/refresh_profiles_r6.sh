# Round-6 measurement set (run on the GPU box from the repo root: bash tools/refresh_profiles_r6.sh [configs...]).
#  1. per config 2-4: tools/run_callbacks.py (EVERY callback: the five of benchmark/runbenchmark.jl:79-101, the products,
#     the fused sweeps, the compressed COO) under rocprofv3 --kernel-trace --stats, then under separate --pmc passes
#     (FETCH_SIZE | WRITE_SIZE | SQ_* | GRBM_GUI_ACTIVE: never together with the trace domains gpurun refuses), joined by
#     tools/roofline_table.py into profiles/r6_kernels_config<k>.md;
#  2. bench.py lines of configs 2-5 with the rocprofv3 stats of the same command and the traffic JSONs of the hess kernel.
set -x
export R=$PWD O=$PWD/gpurun_out/r6p
CONFIGS="${@:-2 3 4}"
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in $CONFIGS; do
  RC="python $R/tools/run_callbacks.py $c"
  # the event-bracketed ms per call: WITHOUT a profiler attached (it adds microseconds to every launch: ACOPF grad! 0.009 -> 0.019)
  timeout 600 $RC --reps 200 > $O/callbacks_config$c.json 2> $O/callbacks_config$c.err
  timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof/cb_stats_c$c -o r6 -- $RC --reps 20 > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof/cb_fetch_c$c -o r6 -- $RC --reps 3 > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof/cb_write_c$c -o r6 -- $RC --reps 3 > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $O/prof/cb_sq_c$c -o r6 -- $RC --reps 3 > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/prof/cb_grbm_c$c -o r6 -- $RC --reps 3 > /dev/null 2>&1
  python $R/tools/roofline_table.py $O/callbacks_config$c.json $O/prof/cb_stats_c$c $O/prof/cb_fetch_c$c $O/prof/cb_write_c$c $O/prof/cb_sq_c$c $O/prof/cb_grbm_c$c > $O/r6_kernels_config$c.md
done
if [ -z "$SKIP_BENCH" ]; then
B="--no-cpu --no-config5-n1 --no-extra-configs"
for c in 2 3 4 5; do
  S=""; [ $c = 5 ] && S="--steps 100 --warmup 10"
  python $R/bench.py --config $c $B $S > $O/bench_pre_config$c.json 2> $O/bench_pre_config$c.err
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof/stats_c$c -o r6 -- python $R/bench.py --config $c $B $S > $O/bench_config${c}_under_rocprof.json 2>/dev/null
  X="--config $c $B --steps 20 --warmup 5 --preheat-ms 0"
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof/fetch_c$c -o r6 -- python $R/bench.py $X > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof/write_c$c -o r6 -- python $R/bench.py $X > /dev/null 2>&1
  python $R/tools/prof_summary.py $O/prof/stats_c$c > $O/r6_stats_config$c.txt
  python $R/tools/prof_summary.py $O/prof/fetch_c$c > $O/pmc_fetch_config$c.txt
  python $R/tools/prof_summary.py $O/prof/write_c$c > $O/pmc_write_config$c.txt
  python $R/tools/make_traffic_json.py $c $O/bench_pre_config$c.json $O/pmc_fetch_config$c.txt $O/pmc_write_config$c.txt > $O/r6_traffic_config$c.json
  cp $O/r6_traffic_config$c.json $R/profiles/
done
cd $R
# the bench lines that are kept: with the traffic JSONs of THIS code in place; and the line of a host that never tunes (--no-tune: the
# plan-time defaults)
python bench.py > $O/r6_bench_default.json 2> $O/r6_bench_default.err
find / -name "*.tune" -not -path "/proc/*" -delete 2>/dev/null
python bench.py --no-tune --no-cpu > $O/r6_bench_no_tune.json 2> $O/r6_bench_no_tune.err
fi
rm -rf $O/prof
ls -la $O
