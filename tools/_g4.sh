mkdir -p gpurun_out/r2d
python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/r2d/pytest.log
tail -5 gpurun_out/r2d/pytest.log
python tools/bench_configs.py acopf > gpurun_out/r2d/acopf.json 2> gpurun_out/r2d/acopf.err
python tools/bench_configs.py lv > gpurun_out/r2d/lv.json 2> gpurun_out/r2d/lv.err
python - <<'PY'
import json
for f in ("acopf","lv"):
    try:
        d=json.loads(open(f"gpurun_out/r2d/{f}.json").read().strip().splitlines()[-1])
        print(f, {k:round(v["ms"],4) for k,v in d["callbacks"].items()}, {k:round(v,4) for k,v in d["products"].items()}, {k:(round(v,4) if isinstance(v,float) else v) for k,v in d["compressed"].items()}, d.get("fused_obj_cons_jac_hess"))
    except Exception as e:
        print(f,"ERR",e, open(f"gpurun_out/r2d/{f}.err").read()[-800:])
PY
