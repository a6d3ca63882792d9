#!/bin/bash
# round-4 GPU call P: weight-gradient consumers on v_mfma_f32_32x32x16 (MIMO_WGRAD_M32=1) vs 16x16x32 (=0): parity, per layer, step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_p
mkdir -p $O
cd $R
python -m pytest tests/test_ops_gpu.py -x -q --tb=short -k "wgrad or conv3x3" 2>&1 | tail -6 > $O/pytest_ops.txt
tail -3 $O/pytest_ops.txt
export REPS=3
bash scripts/layer_ab.sh r04_p/wg "MIMO_WGRAD_M32=0" "MIMO_WGRAD_M32=1" "MIMO_WGRAD_M32=0" "MIMO_WGRAD_M32=1"
python3 scripts/layer_ab_table.py $O/wg m16 m32 m16 m32 > $O/wg_ab.txt 2>&1
grep -E "^wgrad|^#" $O/wg_ab.txt | cut -c1-120
for i in 1 2 3; do
  for v in 0 1; do
    MIMO_WGRAD_M32=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('m32=$v', l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in l['roofline']['kernels'].items()})" >> $O/step_ab.txt
  done
done
cat $O/step_ab.txt
python -m pytest tests/test_network_gpu.py -x -q --tb=short 2>&1 | tail -4
