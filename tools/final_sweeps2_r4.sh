# Second part of the closing sweeps (the first call spent its limit on the range models): compressed-COO A/B on ACOPF, then the deep
# data-indexed models (owner pull included) under register poison for as long as the budget allows.
O=gpurun_out/r4s; mkdir -p $O
for topo in random bus; do
  EXAHIP_CSCATTER=0 timeout 300 python tools/run_callbacks.py 4 --only cjac,chess,jac,hess --reps 200 --topology $topo > $O/cscatter0_$topo.json 2> $O/cscatter0_$topo.err
  timeout 300 python tools/run_callbacks.py 4 --only cjac,chess,jac,hess --reps 200 --topology $topo > $O/cscatter1_$topo.json 2> $O/cscatter1_$topo.err
done
python - <<'PY' | tee $O/cscatter_ab.txt
import json
for t in ("random", "bus"):
    for k in (0, 1):
        try:
            d = json.load(open(f"gpurun_out/r4s/cscatter{k}_{t}.json"))
            print(t, "EXAHIP_CSCATTER", k, {c: round(v["ms"], 5) for c, v in d["callbacks"].items()})
        except Exception as e:
            print(t, k, "failed", e)
PY
timeout 700 bash tests/sweeps/sweep_deep_poison.sh 2000 48 12 6 > $O/deep_2000_48.txt 2>&1
timeout 500 bash tests/sweeps/sweep_deep_poison.sh 3000 40 8 4 > $O/deep_3000_40.txt 2>&1
grep -c " ok" $O/deep_*.txt; grep -h "BAD\|CRASH" $O/deep_*.txt | head
