# instruction mix of LV's first-order kernels by hardware counters (vector / scalar / scalar-memory instructions per launch), default and EXAHIP_KTAB=1
set -x
export R=$GRAFT_REPO_ROOT O=$GRAFT_REPO_ROOT/gpurun_out/r5ah
mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $O/sq_counters_available.txt
for kt in 0 1; do
  RC="python $R/tools/run_callbacks.py 2 --only cons,jac,jprod,hprod --reps 3"
  EXAHIP_KTAB=$kt timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES -d $O/prof/mix_kt$kt -o r5 -- $RC > /dev/null 2>&1
  EXAHIP_KTAB=$kt timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $O/prof/act_kt$kt -o r5 -- $RC > /dev/null 2>&1
  python $R/tools/prof_summary.py $O/prof/mix_kt$kt > $O/mix_ktab$kt.txt 2>&1
  python $R/tools/prof_summary.py $O/prof/act_kt$kt > $O/active_ktab$kt.txt 2>&1
done
rm -rf $O/prof
grep -E "exa_(cons|jac|jprod1|hprodw) " $O/*.txt | cut -c1-300
