mkdir -p gpurun_out/r2h
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r2h/pytest.log; tail -4 gpurun_out/r2h/pytest.log
for M in acopf rocket lv; do python tools/bench_configs.py $M > gpurun_out/r2h/$M.json 2> gpurun_out/r2h/$M.err; done
python - <<'PY'
import json
for f in ("acopf","rocket","lv"):
    try:
        d=json.loads(open(f"gpurun_out/r2h/{f}.json").read().strip().splitlines()[-1])
        print(f, {k:round(v["ms"],4) for k,v in d["callbacks"].items()}, {k:round(v,4) for k,v in d["products"].items()}, {k:(round(v,4) if isinstance(v,float) else v) for k,v in d["compressed"].items()}, {k:round(v,4) for k,v in d.get("fused_obj_cons_jac_hess").items() if k.endswith("ms")})
    except Exception as e:
        print(f,"ERR",e, open(f"gpurun_out/r2h/{f}.err").read()[-1500:])
PY
