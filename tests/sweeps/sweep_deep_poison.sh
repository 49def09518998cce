# sequential sweep of deep random data-indexed models (one process per seed: a GPU fault kills only that seed), every callback
# after a NaN poisoning of the register files.  usage: bash tests/sweeps/sweep_deep_poison.sh FIRST COUNT [NPAT=12] [DEPTH=6]
# USERFN=1 in the environment: registered twins of table entries in every tree (random_model_check.py).
# A seed without a result line (GPU fault, exception, timeout) is reported as CRASH with the last line of its output.
F=${1:-2000}; C=${2:-20}; NP=${3:-12}; D=${4:-6}
for s in $(seq $F $((F + C - 1))); do
  out=$(POISON=1 CHECK_ALL=1 timeout 300 python tests/sweeps/random_model_check.py $s $NP $D 2>&1)
  line=$(printf '%s\n' "$out" | grep "^seed" | cut -c1-400)
  if [ -n "$line" ]; then echo "$line"; else echo "seed $s $NP $D CRASH: $(printf '%s\n' "$out" | grep -v amdgpu.ids | tail -1 | cut -c1-200)"; fi
done
