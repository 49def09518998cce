set -x
mkdir -p gpurun_out/r2b
L="-mllvm -disable-machine-licm"
V="base0:EXAHIP_CHAIN=0,EXAHIP_INTERLEAVE=0 base128:EXAHIP_CHAIN=0,EXAHIP_INTERLEAVE=128 c4: c4nl:EXAHIP_HIPCC_FLAGS=-mllvm+-disable-machine-licm c4nlg0:EXAHIP_HIPCC_FLAGS=-mllvm+-disable-machine-licm,EXAHIP_GROUP=0 c4g0:EXAHIP_GROUP=0 c4late:EXAHIP_CHAIN_EARLY=0 c2:EXAHIP_CHAIN=2 c8:EXAHIP_CHAIN=8 c1:EXAHIP_CHAIN=1"
for N in 1e7 3e7 1e8; do SWEEP_MODEL=lv SWEEP_N=$N python tools/ab_variants.py $V; done > gpurun_out/r2b/ab_lv.txt 2>&1
SWEEP_MODEL=rocket python tools/ab_variants.py $V > gpurun_out/r2b/ab_rocket.txt 2>&1
SWEEP_MODEL=acopf python tools/ab_variants.py $V > gpurun_out/r2b/ab_acopf.txt 2>&1
SWEEP_MODEL=lv SWEEP_N=3e7 python tools/ab_variants.py --cb jac $V > gpurun_out/r2b/ab_lv_jac.txt 2>&1
cat gpurun_out/r2b/*.txt | grep -v "^+"
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
