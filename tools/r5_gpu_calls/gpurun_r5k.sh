set -x
O=gpurun_out/r5k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_product_windows.py tests/test_gpu_compressed.py tests/test_gpu_products.py tests/test_gpu_owner_sharding.py -x -q -p no:cacheprovider > $O/window_tests.txt 2>&1; tail -3 $O/window_tests.txt
python tools/run_callbacks.py 3 --only jtprod,hprod,chess,cjac --reps 300 > $O/rocket_tail.json 2> $O/rocket_tail.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5k/rocket_tail.json"))
print({c: round(v["ms"], 5) for c, v in d["callbacks"].items()})
PY
