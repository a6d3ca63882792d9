#!/bin/bash
# round-4 GPU call J: 256-pixel forward kernel with pinned consumers (variant build) vs default, per layer and per step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_j
mkdir -p $O
cd $R
export REPS=3 MIMO_LAYER_BENCH_WGRAD=0
V=$R/build/variants
bash scripts/layer_ab.sh r04_j/pin "MIMO_CONV_WIDE=0" "MIMO_CONV_WIDE=0 MIMO_HIP_LIB=$V/libmimo_pinfwd.so" "MIMO_CONV_WIDE=0" "MIMO_CONV_WIDE=0 MIMO_HIP_LIB=$V/libmimo_pinfwd.so"
python3 scripts/layer_ab_table.py $O/pin default pinned default pinned > $O/pin_ab.txt 2>&1
for i in 1 2 3; do
  for v in default pinned; do
    if [ $v = pinned ]; then export MIMO_HIP_LIB=$V/libmimo_pinfwd.so; else unset MIMO_HIP_LIB; fi
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in l['roofline']['kernels'].items()})" >> $O/step_ab.txt
  done
done
unset MIMO_HIP_LIB
grep -E "^fwd|^#" $O/pin_ab.txt; cat $O/step_ab.txt
