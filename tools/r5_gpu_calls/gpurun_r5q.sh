set -x
O=gpurun_out/r5q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
{
SWEEP_MODEL=lv SWEEP_N=1e7 python tools/ab_variants.py --cb cons plain:EXAHIP_CONS_FUSED=0 fused:EXAHIP_CONS_FUSED=1
SWEEP_MODEL=lv SWEEP_N=1e8 python tools/ab_variants.py --cb cons plain:EXAHIP_CONS_FUSED=0 fused:EXAHIP_CONS_FUSED=1
} 2>&1 | grep -v amdgpu.ids | tee $O/cons_fused_lv_ab.txt
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/gpu_suite.txt 2>&1; grep -n "passed\|failed\|Fatal" $O/gpu_suite.txt | tail -3
python tools/run_callbacks.py 3 --reps 200 > $O/callbacks_config3.json 2> $O/callbacks_config3.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r5q/callbacks_config3.json"))
print({c: round(v["ms"], 5) for c, v in d["callbacks"].items()})
PY
