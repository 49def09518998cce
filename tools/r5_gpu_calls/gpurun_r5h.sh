set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R0=200 NR=60 D0=8100 ND=40 E0=9100 timeout 9000 tools/final_sweeps_r5.sh gpurun_out/r5h 2>&1 | tail -20
