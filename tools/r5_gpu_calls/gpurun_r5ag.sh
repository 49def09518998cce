# closing sweep of the third session of round 5: fresh seeds, library as shipped (lean math, occupancy throttle), register poison on
set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R0=300 NR=10 D0=8300 ND=6 E0=9300 timeout 1500 bash tools/final_sweeps_r5.sh gpurun_out/r5ag 2>&1 | tail -12
