set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 5000 tools/final_sweeps_r5.sh gpurun_out/r5e 2>&1 | tail -20
