set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5p
python bench.py > gpurun_out/r5p/r5_bench_default.json 2> gpurun_out/r5p/r5_bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5p/r5_bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"], d["roofline"]["traffic"], d["scaling"])
for k in ("config3", "config4", "config5_n1"):
    print(k, d[k]["ms_per_step"], d[k]["roofline"]["frac"], d[k]["roofline"]["kernel"], d[k]["roofline"]["traffic"])
PY
