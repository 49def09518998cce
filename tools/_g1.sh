set -x
mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r2a/pytest.log
python bench.py > gpurun_out/r2a/bench_default.json 2> gpurun_out/r2a/bench_default.err
EXAHIP_COMPILER=hipcc python bench.py --no-cpu --no-config5-n1 > gpurun_out/r2a/bench_hipcc.json 2> gpurun_out/r2a/bench_hipcc.err
python bench.py --config 3 > gpurun_out/r2a/bench_c3.json 2> gpurun_out/r2a/bench_c3.err
python bench.py --config 4 > gpurun_out/r2a/bench_c4.json 2> gpurun_out/r2a/bench_c4.err
python bench.py --config 2 --points 3e7 --steps 200 --no-cpu --no-config5-n1 > gpurun_out/r2a/bench_3e7.json 2> gpurun_out/r2a/bench_3e7.err
tail -3 gpurun_out/r2a/pytest.log
cat gpurun_out/r2a/*.json | cut -c1-600
