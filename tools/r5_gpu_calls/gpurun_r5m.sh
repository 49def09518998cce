set -x
export R=$PWD O=$PWD/gpurun_out/r5m; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; cd $R
timeout 900 python -m pytest tests/test_gpu_product_windows.py tests/test_gpu_compressed.py tests/test_gpu_products.py tests/test_gpu_owner_sharding.py tests/test_gpu_poison.py -x -q -p no:cacheprovider > $O/window_tests.txt 2>&1; tail -3 $O/window_tests.txt
python tools/run_callbacks.py 3 --only jtprod,hprod,chess,cjac --reps 300 > $O/rocket_tail.json 2> $O/rocket_tail.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5m/rocket_tail.json"))
print({c: round(v["ms"], 5) for c, v in d["callbacks"].items()})
PY
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o r5 -- python $R/tools/run_callbacks.py 3 --only jtprod,hprod,chess,cjac --reps 50 > /dev/null 2>&1
python $R/tools/prof_summary.py $O/prof > $O/rocket_tail_stats.txt; rm -rf $O/prof; grep "prodx\|chessx\|cjacx\|prodw\|chessw\|cjacw" $O/rocket_tail_stats.txt
