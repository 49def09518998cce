set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5w; mkdir -p $O
# 1. the default bench line on the lean-math modules (config 2 + the riders: 3, 4, 5 at N = 1)
timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5w/bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"]["traffic"], d["scaling"])
for k in ("config3", "config4", "config5_n1"):
    print(k, d[k]["ms_per_step"], d[k]["roofline"]["frac"], d[k]["roofline"]["kernel"], d[k]["roofline"]["traffic"])
PY
# 2. every callback of configs 2-4 with the Horner coefficients from constant memory
for c in 2 3 4; do EXAHIP_KTAB=1 timeout 300 python tools/run_callbacks.py $c --reps 200 > $O/callbacks_ktab1_config$c.json 2> $O/callbacks_ktab1_config$c.err; done
for c in 2 3 4; do timeout 300 python tools/run_callbacks.py $c --reps 200 > $O/callbacks_ktab0_config$c.json 2> $O/callbacks_ktab0_config$c.err; done
python - <<'PY'
import json
for c in (2, 3, 4):
    for kt in (0, 1):
        try:
            d = json.loads(open(f"gpurun_out/r5w/callbacks_ktab{kt}_config{c}.json").read().strip().splitlines()[-1])
            print(c, kt, {k: round(v["ms"], 4) for k, v in d["callbacks"].items()})
        except Exception as e:
            print(c, kt, "failed", e)
PY
# 3. the whole GPU suite
timeout 1200 python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1
tail -3 $O/gpu_suite.txt
