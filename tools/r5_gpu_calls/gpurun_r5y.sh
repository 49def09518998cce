set -x
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5y; mkdir -p $O
for rep in 1 2; do for b in 0 40960; do
  EXAHIP_TILE_LD_BUDGET=$b timeout 300 python tools/run_callbacks.py 3 --reps 200 --only jac,hess,fused,eval_all 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('budget $b nh=1e6', d['module'], {k: round(v['ms'],4) for k,v in d['callbacks'].items()})" >> $O/rocket_ld_ab.txt
done; done
for b in 0 40960; do
  EXAHIP_TILE_LD_BUDGET=$b timeout 300 python tools/run_callbacks.py 3 --points 2e7 --reps 40 --only jac,hess,fused 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('budget $b nh=2e7', d['module'], {k: round(v['ms'],4) for k,v in d['callbacks'].items()})" >> $O/rocket_ld_ab.txt
done
cat $O/rocket_ld_ab.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "rocket or fullsize or zoo" 2>&1 | tail -3
