set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5ak
timeout 600 python tools/tile_loop_ab.py 1e7 2>&1 | grep -v amdgpu.ids > gpurun_out/r5ak/tile_loop_ab.txt
timeout 600 python tools/tile_loop_ab.py 1e8 2>&1 | grep -v amdgpu.ids >> gpurun_out/r5ak/tile_loop_ab.txt
cat gpurun_out/r5ak/tile_loop_ab.txt
