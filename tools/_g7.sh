mkdir -p gpurun_out/r2g
python -m pytest tests/test_gpu_compressed.py tests/test_gpu_dist.py tests/test_gpu_comm.py -m gpu -q -x 2>&1 | tail -12 > gpurun_out/r2g/pytest.log; tail -6 gpurun_out/r2g/pytest.log
python tools/bench_configs.py acopf > gpurun_out/r2g/acopf.json 2> gpurun_out/r2g/acopf.err
python - <<'PY'
import json
for f in ("acopf",):
    try:
        d=json.loads(open(f"gpurun_out/r2g/{f}.json").read().strip().splitlines()[-1])
        print(f, {k:round(v["ms"],4) for k,v in d["callbacks"].items()}, {k:round(v,4) for k,v in d["products"].items()}, {k:(round(v,4) if isinstance(v,float) else v) for k,v in d["compressed"].items()}, d.get("fused_obj_cons_jac_hess"))
    except Exception as e:
        print(f,"ERR",e, open(f"gpurun_out/r2g/{f}.err").read()[-1500:])
PY
