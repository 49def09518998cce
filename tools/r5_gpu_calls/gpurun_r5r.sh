set -x
export R=$PWD O=$PWD/gpurun_out/r5p
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in 3; do
RC="python $R/tools/run_callbacks.py $c"
timeout 600 $RC --reps 200 > $O/callbacks_config$c.json 2> $O/callbacks_config$c.err
timeout 900 rocprofv3 --kernel-trace --stats -d $O/prof/cb_stats_c$c -o r5 -- $RC --reps 20 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof/cb_fetch_c$c -o r5 -- $RC --reps 3 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof/cb_write_c$c -o r5 -- $RC --reps 3 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $O/prof/cb_sq_c$c -o r5 -- $RC --reps 3 > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d $O/prof/cb_grbm_c$c -o r5 -- $RC --reps 3 > /dev/null 2>&1
python $R/tools/roofline_table.py $O/callbacks_config$c.json $O/prof/cb_stats_c$c $O/prof/cb_fetch_c$c $O/prof/cb_write_c$c $O/prof/cb_sq_c$c $O/prof/cb_grbm_c$c > $O/r5_kernels_config$c.md
done
rm -rf $O/prof
grep "^| cons" $O/r5_kernels_config3.md
