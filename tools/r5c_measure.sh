#!/bin/bash
# Round 5, second GPU call: (1) locality-ordered table copies of the order-free kernels, in / out, both ACOPF topologies; (2) exa_eval_all with the
# gradient tiles in the interleaved block order (default now) against exa_fused alone; (3) the rocket's LDS-staged chained kernel kept although it
# outgrows 256 VGPRs, against exa_hessc and exa_hess, at nh = 1e6 and 2e7.
O=${1:-gpurun_out/r5c}; mkdir -p $O
for topo in random bus; do
  for loc in 1 0; do
    EXAHIP_LOCALITY=$loc python tools/run_callbacks.py 4 --only grad,jtprod,hprod,cons,jprod,eval_all,fused --reps 300 --topology $topo > $O/acopf_${topo}_locality$loc.json 2> $O/acopf_${topo}_locality$loc.err
  done
done
python tools/run_callbacks.py 2 --only fused,eval_all,grad,hess --reps 200 > $O/lv_evalall.json 2> $O/lv_evalall.err
python tools/run_callbacks.py 3 --only fused,eval_all,grad,hess --reps 200 > $O/rocket_evalall.json 2> $O/rocket_evalall.err
python - <<'PY' | tee $O/summary.txt
import json
for f in ("acopf_random_locality1", "acopf_random_locality0", "acopf_bus_locality1", "acopf_bus_locality0", "lv_evalall", "rocket_evalall"):
    try:
        d = json.load(open(f"gpurun_out/r5c/{f}.json"))
        print(f"{f:26s}", {c: round(v["ms"], 5) for c, v in d["callbacks"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
{
for n in 1e6 2e7; do
  EXAHIP_KEEP_STAGE=1 EXAHIP_SAFE_FLAGS=none SWEEP_MODEL=rocket SWEEP_N=$n python tools/ab_variants.py --cb hess plain:EXAHIP_HESS_VARIANT=0 chained:EXAHIP_HESS_VARIANT=2 staged:EXAHIP_HESS_VARIANT=1
done
} 2>&1 | grep -v amdgpu.ids | tee $O/rocket_staging_ab.txt
