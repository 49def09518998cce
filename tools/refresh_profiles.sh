set -x
export R=$PWD
python bench.py > gpurun_out/r1f_bench.json 2> gpurun_out/r1f_bench.err
python tools/bench_configs.py > gpurun_out/r1f_bench_configs.jsonl 2>/dev/null
python tools/bench_products.py > gpurun_out/r1f_bench_products.txt 2>/dev/null
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof11/stats -o r1 -- python $R/bench.py --no-cpu > $R/gpurun_out/r1f_bench_under_rocprof.json 2>/dev/null
export EXAHIP_INTERLEAVE=128
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/prof11/fetch_il -o r1 -- python $R/bench.py --no-cpu --steps 20 --warmup 5 --preheat-ms 0 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/prof11/write_il -o r1 -- python $R/bench.py --no-cpu --steps 20 --warmup 5 --preheat-ms 0 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/prof11/sq_il -o r1 -- python $R/bench.py --no-cpu --steps 20 --warmup 5 --preheat-ms 0 > /dev/null 2>&1
cd $R
python tools/prof_summary.py gpurun_out/prof11 > gpurun_out/r1f_prof_summary.txt
cat gpurun_out/r1f_bench.json | head -c 1500
grep -v rocclr gpurun_out/r1f_prof_summary.txt | head -40
