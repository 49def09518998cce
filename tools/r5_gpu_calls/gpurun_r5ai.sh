set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5ai
timeout 600 python tools/ppt_ab.py 1e7 2>&1 | grep -v amdgpu.ids > gpurun_out/r5ai/ppt_ab.txt
cat gpurun_out/r5ai/ppt_ab.txt
