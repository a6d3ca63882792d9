#!/bin/bash
# round-4 GPU call D: weight-gradient consumers with the read pipeline really one tap ahead (prologue pinned) vs before
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_d
mkdir -p $O
cd $R
python -m pytest tests/test_ops_gpu.py -x -q 2>&1 | tail -5 > $O/pytest_ops.txt
export REPS=3
V=$R/build/variants
bash scripts/layer_ab.sh r04_d/wg "MIMO_HIP_LIB=$V/libmimo_wg_nopin.so" "-" "MIMO_HIP_LIB=$V/libmimo_wg_nopin.so" "-"
python3 scripts/layer_ab_table.py $O/wg before pinned before pinned > $O/wg_ab.txt 2>&1
# whole step, alternating runs in one box
for i in 1 2 3 4; do
  for v in nopin pin; do
    if [ $v = nopin ]; then export MIMO_HIP_LIB=$V/libmimo_wg_nopin.so; else unset MIMO_HIP_LIB; fi
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in l['roofline']['kernels'].items()})" >> $O/step_ab.txt
  done
done
unset MIMO_HIP_LIB
cat $O/pytest_ops.txt; grep -E "^wgrad|^#" $O/wg_ab.txt; cat $O/step_ab.txt
