#!/bin/bash
# round-4 GPU call F: weight gradient with two consumer waves per SIMD (MIMO_WGRAD_CW=2, default) vs one (=1)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_f
mkdir -p $O
cd $R
python -m pytest tests/test_ops_gpu.py tests/test_network_gpu.py -x -q 2>&1 | tail -5 > $O/pytest.txt
export REPS=3
bash scripts/layer_ab.sh r04_f/wg "MIMO_WGRAD_CW=1" "MIMO_WGRAD_CW=2" "MIMO_WGRAD_CW=1" "MIMO_WGRAD_CW=2"
python3 scripts/layer_ab_table.py $O/wg cw1 cw2 cw1 cw2 > $O/wg_ab.txt 2>&1
for i in 1 2 3 4; do
  for v in 1 2; do
    MIMO_WGRAD_CW=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cw$v', l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in l['roofline']['kernels'].items()})" >> $O/step_ab.txt
  done
done
cat $O/pytest.txt; grep -E "^wgrad|^#" $O/wg_ab.txt; cat $O/step_ab.txt
