set -x
O=gpurun_out/r5b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite.txt 2>&1; tail -3 $O/gpu_suite.txt
timeout 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.json
EXAHIP_COMPILER=hipcc timeout 1500 tools/sgpr_regalloc_ab.sh $O > /dev/null 2>&1; cat $O/sgpr_regalloc_ab.txt | tail -60
timeout 2400 tools/r5_guard_sweeps.sh $O 2>&1 | tail -15
