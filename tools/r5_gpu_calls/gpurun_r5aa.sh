set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5aa; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_knobs.py -m gpu -x -q 2>&1 | tail -3
EXAHIP_VERBOSE=1 timeout 600 python bench.py > $O/bench_default.json 2> $O/bench_default.err
grep "tune hess_coord" $O/bench_default.err | cut -c1-400
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5aa/bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"]["throttle_lds_bytes"])
for k in ("config3", "config4", "config5_n1"):
    print(k, d[k]["ms_per_step"], d[k]["roofline"]["frac"], d[k]["roofline"]["kernel"], d[k]["roofline"].get("throttle_lds_bytes"))
PY
