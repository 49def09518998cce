# Round-2 measurement set: bench lines for configs 2-4 (+ the N=1 point of config 5), rocprofv3 kernel stats of the same
# commands, PMC passes (separate runs: --pmc never together with the trace domains gpurun refuses), traffic JSONs.
set -x
export R=$PWD O=$PWD/gpurun_out/r2p
mkdir -p $O
# a first pair of lines only names module / kernel / points for the traffic JSONs; the lines that are kept run at the end
python bench.py --no-cpu --no-config5-n1 > $O/bench_config2.json 2> $O/bench_config2.err
python bench.py --config 5 --steps 100 --warmup 10 --no-cpu > $O/bench_config5_n1.json 2> $O/bench_config5_n1.err
cd /tmp; export TMPDIR=/tmp
for c in 2 3 4; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof/stats_c$c -o r2 -- python $R/bench.py --config $c --no-cpu --no-config5-n1 > $O/bench_config${c}_under_rocprof.json 2>/dev/null
done
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof/stats_c5 -o r2 -- python $R/bench.py --config 5 --steps 100 --warmup 10 --no-cpu > $O/bench_config5_n1_under_rocprof.json 2>/dev/null
for c in 2 5; do
  X="--config $c --no-cpu --no-config5-n1 --steps 20 --warmup 5 --preheat-ms 0"
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof/fetch_c$c -o r2 -- python $R/bench.py $X > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof/write_c$c -o r2 -- python $R/bench.py $X > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $O/prof/sq_c$c -o r2 -- python $R/bench.py $X > /dev/null 2>&1
done
cd $R
for c in 2 3 4 5; do python tools/prof_summary.py $O/prof/stats_c$c > $O/stats_config$c.txt; done
for c in 2 5; do
  python tools/prof_summary.py $O/prof/fetch_c$c > $O/pmc_fetch_config$c.txt
  python tools/prof_summary.py $O/prof/write_c$c > $O/pmc_write_config$c.txt
  python tools/prof_summary.py $O/prof/sq_c$c > $O/pmc_sq_config$c.txt
done
python tools/make_traffic_json.py 2 $O/bench_config2.json $O/pmc_fetch_config2.txt $O/pmc_write_config2.txt > $O/r2_traffic_config2.json
python tools/make_traffic_json.py 5 $O/bench_config5_n1.json $O/pmc_fetch_config5.txt $O/pmc_write_config5.txt > $O/r2_traffic_config5.json
rm -rf $O/prof
# the bench lines that are kept: with the traffic JSONs of THIS code in place (roofline.traffic is attached only on a
# module / kernel / size match)
cp $O/r2_traffic_config2.json $O/r2_traffic_config5.json profiles/
python bench.py > $O/bench_config2.json 2> $O/bench_config2.err
python bench.py --config 3 > $O/bench_config3.json 2> $O/bench_config3.err
python bench.py --config 4 > $O/bench_config4.json 2> $O/bench_config4.err
python bench.py --config 5 --steps 200 --warmup 20 > $O/bench_config5_n1.json 2> $O/bench_config5_n1.err
head -c 1200 $O/bench_config2.json; echo; grep -v rocclr $O/stats_config2.txt | head -12; cat $O/r2_traffic_config2.json
