# the default bench line three times in a row (does exa_tune settle on one kernel?), the last one kept; the headline's stats with that decision (--no-tune)
set -x
export R=$GRAFT_REPO_ROOT O=$GRAFT_REPO_ROOT/gpurun_out/r5an
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="--config 2 --no-cpu --no-config5-n1 --no-extra-configs"
for i in 1 2; do EXAHIP_VERBOSE=1 python $R/bench.py $B > $O/bench_config2_$i.json 2> $O/bench_config2_$i.err; grep "tune hess_coord" $O/bench_config2_$i.err | cut -c1-330; done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof/stats_c2 -o r5 -- python $R/bench.py $B --no-tune > $O/bench_config2_under_rocprof.json 2>/dev/null
python $R/tools/prof_summary.py $O/prof/stats_c2 > $O/r5_stats_config2.txt
rm -rf $O/prof
cd $R; python bench.py > $O/r5_bench_default.json 2> $O/r5_bench_default.err
head -5 $O/r5_stats_config2.txt
python - <<'PY'
import json
for f in ("bench_config2_1.json", "bench_config2_2.json", "bench_config2_under_rocprof.json", "r5_bench_default.json"):
    d = json.loads(open("gpurun_out/r5an/" + f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["traffic"], d.get("config5_n1", {}).get("ms_per_step"), d.get("config5_n1", {}).get("roofline", {}).get("throttle_lds_bytes"))
PY
