set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5af
timeout 600 python tools/cons_skeleton.py 1e7 2>&1 | grep -v amdgpu.ids > gpurun_out/r5af/cons_skeleton.txt
cat gpurun_out/r5af/cons_skeleton.txt
