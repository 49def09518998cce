# Runs the default bench line several times in fresh processes with EXAHIP_VERBOSE=1 and collects which hess_coord! kernel
# exa_tune chose each time (the decision must not depend on when in the process it was measured).  On the GPU box.
set -x
O=gpurun_out/r4t; mkdir -p $O
for i in 1 2 3 4 5; do
  EXAHIP_VERBOSE=1 python bench.py > $O/bench_$i.json 2> $O/bench_$i.err
  rm -f examodels.jl_amd/kernel_cache/*.tune
done
python - <<'PY' > $O/r4_tune_stability.txt
import json, glob, re
print("default bench line, 5 fresh processes (tune files removed in between): kernel chosen by exa_tune, ms_per_step, roofline.frac")
for f in sorted(glob.glob("gpurun_out/r4t/bench_?.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    e = open(f.replace(".json", ".err")).read()
    t = [l for l in e.splitlines() if "tune hess_coord kernels" in l]
    print(f, d["roofline"]["kernel"], "%.4f" % d["ms_per_step"], "%.3f" % d["roofline"]["frac"], "scale_base", d["scale_base"]["kernel"], "%.3f" % d["scale_base"]["roofline_frac"])
    for l in t: print("   ", l)
PY
python bench.py > $O/r4_bench_default.json 2> $O/r4_bench_default.err
cat $O/r4_tune_stability.txt
