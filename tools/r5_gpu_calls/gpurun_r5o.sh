set -x
O=gpurun_out/r5o; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
{
SWEEP_MODEL=rocket python tools/ab_variants.py --cb cons plain:EXAHIP_CONS_FUSED=0 fused:EXAHIP_CONS_FUSED=1
SWEEP_MODEL=rocket SWEEP_N=2e7 python tools/ab_variants.py --cb cons plain:EXAHIP_CONS_FUSED=0 fused:EXAHIP_CONS_FUSED=1
SWEEP_MODEL=chain SWEEP_N=2e6 python tools/ab_variants.py --cb cons plain:EXAHIP_CONS_FUSED=0 fused:EXAHIP_CONS_FUSED=1
SWEEP_MODEL=chain SWEEP_N=2e7 python tools/ab_variants.py --cb cons plain:EXAHIP_CONS_FUSED=0 fused:EXAHIP_CONS_FUSED=1
} 2>&1 | grep -v amdgpu.ids | tee $O/cons_fused_ab.txt
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_owner_sharding.py tests/test_gpu_edge_cases.py -x -q -p no:cacheprovider 2>&1 | tail -3
