# GPU evidence on the closing tree of round 4: whole GPU suite, smoke(), the default bench line, rocprofv3 stats of the config-2 command.
O=gpurun_out/r4head2; mkdir -p $O
timeout 480 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest_full.log 2>&1; tail -14 $O/pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 200 python bench.py 2>$O/bench.err | tee $O/bench.json | cut -c1-400
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $R/$O/prof -o bench -- python $R/bench.py --no-config5-n1 --no-extra-configs --no-cpu > $R/$O/bench_prof.json 2>$R/$O/prof.err
cd $R; python tools/prof_summary.py $O/prof > $O/stats.txt 2>&1 || true; head -8 $O/stats.txt; rm -rf $O/prof
