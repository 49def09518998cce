set -x
O=gpurun_out/r5j; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite.txt 2>&1; grep -n "passed\|failed\|Fatal" $O/gpu_suite.txt | tail -3
timeout 300 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5j/bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"]["traffic"], d["scaling"])
for k in ("config3", "config4", "config5_n1"):
    print(k, d[k]["ms_per_step"], d[k]["roofline"]["frac"], d[k]["roofline"]["kernel"], d[k]["roofline"]["traffic"])
PY
