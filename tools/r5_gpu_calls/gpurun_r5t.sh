set -x
O=gpurun_out/r5t; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite.txt 2>&1; grep -n "passed\|failed\|Fatal" $O/gpu_suite.txt | tail -3
for topo in random bus; do python tools/run_callbacks.py 4 --only obj,grad,cons,fused,eval_all --reps 300 --topology $topo > $O/acopf_$topo.json 2> $O/acopf_$topo.err; done
python - <<'PY' | tee $O/summary.txt
import json
for f in ("acopf_random", "acopf_bus"):
    d = json.load(open(f"gpurun_out/r5t/{f}.json"))
    print(f"{f:14s}", {c: round(v["ms"], 5) for c, v in d["callbacks"].items()})
PY
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -1
