set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ae; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_suite.txt 2>&1
tail -3 $O/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
