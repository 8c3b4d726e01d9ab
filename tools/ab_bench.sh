# same-box A/B of whole bench steps: bash tools/ab_bench.sh libA.so libB.so "rows samples" ...
set -e
A=$1; B=$2; shift 2
for cfg in "$@"; do
  set -- $cfg
  for rep in 1 2; do
    for L in $A $B; do
      CHICDIFF_HIP_LIB=$PWD/$L python bench.py --rows $1 --samples $2 --steps 50 --warmup 5 --no-cpu-baseline --no-hbm-kernels 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$L', j['config']['rows_per_gpu'],'x',j['config']['samples'],'ms_per_step',j['ms_per_step'])"
    done
  done
done
