mkdir -p gpurun_out/r2c
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2c/pytest.log
tail -5 gpurun_out/r2c/pytest.log
python tools/bench_configs.py acopf > gpurun_out/r2c/acopf.json 2> gpurun_out/r2c/acopf.err
EXAHIP_GROUP_SCATTER=0 EXAHIP_CONS1=0 python tools/bench_configs.py acopf > gpurun_out/r2c/acopf_old.json 2> gpurun_out/r2c/acopf_old.err
python tools/bench_configs.py rocket > gpurun_out/r2c/rocket.json 2> gpurun_out/r2c/rocket.err
python - <<'PY'
import json
for f in ("acopf","acopf_old","rocket"):
    try:
        d=json.loads(open(f"gpurun_out/r2c/{f}.json").read().strip().splitlines()[-1])
        print(f, {k:round(v["ms"],4) for k,v in d["callbacks"].items()}, {k:round(v,4) for k,v in d["products"].items()}, {k:(round(v,4) if isinstance(v,float) else v) for k,v in d["compressed"].items()})
    except Exception as e:
        print(f,"ERR",e, open(f"gpurun_out/r2c/{f}.err").read()[-800:])
PY
