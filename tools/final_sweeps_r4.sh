# Round-4 closing sweeps on the FINAL generator (multi-stretch LDS staging, owner pull, interleaved tuning), register poison on,
# one process at a time; the models' modules are prebuilt into the kernel cache on the build machine
# (range_model_check.py --prebuild, PREBUILD=1 random_model_check.py).  On the GPU box; logs under gpurun_out/r4s/.
O=gpurun_out/r4s; mkdir -p $O
for fl in blocks unit mixed; do
  for s in 0 10 20 30 40; do
    timeout 900 python tests/sweeps/range_model_check.py $s 10 $fl --poison 2>&1 | grep -E "^seed|Error|error" >> $O/range_$fl.log || echo "chunk $s $fl: timeout / crash" >> $O/range_$fl.log
  done
done
bash tests/sweeps/sweep_deep_poison.sh 2000 48 12 6 > $O/deep_2000_48.txt 2>&1
bash tests/sweeps/sweep_deep_poison.sh 3000 40 8 4 > $O/deep_3000_40.txt 2>&1
# compressed COO on ACOPF: permuted store (default) against uncompressed sweep + gather (EXAHIP_CSCATTER=0)
for topo in random bus; do
  EXAHIP_CSCATTER=0 python tools/run_callbacks.py 4 --only cjac,chess,jac,hess --reps 200 --topology $topo > $O/cscatter0_$topo.json 2> $O/cscatter0_$topo.err
  python tools/run_callbacks.py 4 --only cjac,chess,jac,hess --reps 200 --topology $topo > $O/cscatter1_$topo.json 2> $O/cscatter1_$topo.err
done
grep -c " ok" $O/range_*.log $O/deep_*.txt; grep -l "BAD\|CRASH\|timeout" $O/* ; python - <<'PY'
import json
for t in ("random", "bus"):
    for k in (0, 1):
        try:
            d = json.load(open(f"gpurun_out/r4s/cscatter{k}_{t}.json"))
            print(t, "CSCATTER", k, {c: round(v["ms"], 5) for c, v in d["callbacks"].items()})
        except Exception as e:
            print(t, k, "failed", e)
PY
