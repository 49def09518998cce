set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R0=300 NR=20 D0=8200 ND=0 E0=9200 timeout 3000 tools/final_sweeps_r5.sh gpurun_out/r5s 2>&1 | tail -12
