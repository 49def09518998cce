set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5z; mkdir -p $O
NEW=examodels.jl_amd
# hesscl: 15.4 KB static -> dyn 0: 4 WG/CU (VGPR-bound), 26000: 3, 40000: 2;   hess: 12.9 KB static, 8 WG/CU -> dyn 8000: 7, 14000: 5, 20000: 4, 28000: 3, 41000: 2
for N in 1e8 1e7; do
  for d in 0 26000 40000 0 26000; do EXAHIP_HESS_DYN_LDS=$d timeout 200 python tools/lv_hess_ab.py $NEW $N 1 2>/dev/null | tail -1 | sed "s/^/dyn $d /" >> $O/hess_occupancy.txt; done
  for d in 0 26000 40000; do EXAHIP_HESS_DYN_LDS=$d timeout 200 python tools/lv_hess_ab.py $NEW $N 2 2>/dev/null | tail -1 | sed "s/^/dyn $d /" >> $O/hess_occupancy.txt; done
  for d in 0 8000 14000 20000 28000 41000; do EXAHIP_HESS_DYN_LDS=$d timeout 200 python tools/lv_hess_ab.py $NEW $N 0 2>/dev/null | tail -1 | sed "s/^/dyn $d /" >> $O/hess_occupancy.txt; done
done
cat $O/hess_occupancy.txt | cut -c1-150
