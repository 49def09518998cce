set -x
O=gpurun_out/r5c; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_round5.py tests/test_gpu_products.py tests/test_gpu_owner_sharding.py -x -q -p no:cacheprovider > $O/locality_tests.txt 2>&1; tail -5 $O/locality_tests.txt
timeout 1500 tools/r5c_measure.sh $O 2>&1 | tail -30
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite.txt 2>&1; grep -n "passed\|failed" $O/gpu_suite.txt | tail -2
