#!/bin/bash
# round-4 GPU call X: weight gradient of 90->45 (cin_p 96) on 32-input-channel tiles (96 padded) instead of 64 (128 padded)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_x
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv3x3_forward and split16" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for i in 1 2 3; do
  for v in 0 1; do
    MIMO_WGRAD_CI32=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print('ci32=$v', l['value'], l['ms_per_step'], 'wgrad', r['kernels']['conv3x3_wgrad']['ms_per_step'], r['tiers']['256x256']['kernels_ms']['conv3x3_wgrad'])" >> $O/step_ab.txt
  done
done
cat $O/step_ab.txt
