# the headline's rocprofv3 stats with the kernel choice of the unprofiled line: bench.py tunes and persists, the profiled run reads the decision (--no-tune)
set -x
export R=$GRAFT_REPO_ROOT O=$GRAFT_REPO_ROOT/gpurun_out/r5am
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="--config 2 --no-cpu --no-config5-n1 --no-extra-configs"
python $R/bench.py $B > $O/bench_pre_config2.json 2> $O/bench_pre_config2.err
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof/stats_c2 -o r5 -- python $R/bench.py $B --no-tune > $O/bench_config2_under_rocprof.json 2>/dev/null
python $R/tools/prof_summary.py $O/prof/stats_c2 > $O/r5_stats_config2.txt
rm -rf $O/prof
cd $R; python bench.py > $O/r5_bench_default.json 2> $O/r5_bench_default.err
head -6 $O/r5_stats_config2.txt
python - <<'PY'
import json
for f in ("bench_pre_config2.json", "bench_config2_under_rocprof.json", "r5_bench_default.json"):
    d = json.loads(open("gpurun_out/r5am/" + f).read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["roofline"]["traffic"])
PY
