set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5aj
timeout 600 python tools/tile_loop_ab.py 1e7 2>&1 | grep -v amdgpu.ids > gpurun_out/r5aj/tile_loop_ab.txt
timeout 600 python tools/tile_loop_ab.py 1e8 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5aj/tile_loop_ab.txt
cat gpurun_out/r5aj/tile_loop_ab.txt
timeout 600 python -m pytest tests/test_gpu_knobs.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
