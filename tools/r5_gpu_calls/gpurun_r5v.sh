set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5v; mkdir -p $O
# 1. accuracy of the lean exp / sincos on the device
timeout 300 python tools/exp_accuracy.py > $O/exp_accuracy.txt 2>&1
EXAHIP_FAST_EXP=0 timeout 300 python tools/exp_accuracy.py >> $O/exp_accuracy.txt 2>&1
timeout 300 python tools/trig_accuracy.py > $O/trig_accuracy.txt 2>&1
# 2. parity subset
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_special_values.py tests/test_gpu_golden.py tests/test_gpu_knobs.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/parity_subset.txt 2>&1
tail -3 $O/parity_subset.txt
# 3. A/B on LV 1e7
timeout 600 python tools/math_ab.py 1e7 > $O/math_ab.txt 2>&1
cat $O/math_ab.txt $O/exp_accuracy.txt $O/trig_accuracy.txt
# 4. callbacks of configs 2-4 on the new modules
for c in 2 3 4; do timeout 300 python tools/run_callbacks.py $c --reps 200 > $O/callbacks_config$c.json 2> $O/callbacks_config$c.err; done
python - <<'PY'
import json
for c in (2, 3, 4):
    try:
        d = json.loads(open(f"gpurun_out/r5v/callbacks_config{c}.json").read().strip().splitlines()[-1])
        print(c, {k: round(v["ms"], 4) for k, v in d["callbacks"].items()})
    except Exception as e:
        print(c, "failed", e)
PY
