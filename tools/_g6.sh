mkdir -p gpurun_out/r2f
V="grouped: ungrouped:EXAHIP_GROUP_COO=0"
for M in acopf rocket; do for cb in hess jac; do SWEEP_MODEL=$M python tools/ab_variants.py --cb $cb $V; done; done > gpurun_out/r2f/ab_group_coo.txt 2>&1
grep -v "^+\|amdgpu.ids" gpurun_out/r2f/ab_group_coo.txt
python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r2f/pytest.log; tail -4 gpurun_out/r2f/pytest.log
