# GPU evidence on the closing tree of round 4 (after the registered-function commit): whole GPU suite, smoke(), the default bench line,
# rocprofv3 stats of the same command.
O=gpurun_out/r4head; mkdir -p $O
timeout 420 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_full.log 2>&1; tail -14 $O/pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 200 python bench.py 2>$O/bench.err | tee $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/$O/bench_prof.json 2>$GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT; python tools/prof_summary.py $O/prof > $O/stats.txt 2>&1 || true; head -12 $O/stats.txt
