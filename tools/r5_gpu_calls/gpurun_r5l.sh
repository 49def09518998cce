set -x
export R=$PWD O=$PWD/gpurun_out/r5l; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r5 -- python $R/tools/run_callbacks.py 3 --only jtprod,hprod,chess,cjac --reps 50 > /dev/null 2>&1
python $R/tools/prof_summary.py $O/prof > $O/rocket_tail_stats.txt; rm -rf $O/prof; cat $O/rocket_tail_stats.txt | head -30
