#!/bin/bash
# round-4 GPU call R3: LDS-staged plain-FMA kernels for the image convolution: full parity (ops, network, configs, variants), step A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_r
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_network_gpu.py tests/test_configs_gpu.py tests/test_variants_gpu.py tests/test_mixed_precision_gpu.py -x -q -m gpu > $O/pytest3.txt 2>&1
tail -4 $O/pytest3.txt
rm -f $O/step_ab3.txt
for i in 1 2 3 4; do
  for v in 1 0; do
    MIMO_CONV_THIN=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; k=r['tiers']['256x256']['kernels_ms']; print('thin=$v', l['value'], l['ms_per_step'], 'fwd256', k['conv3x3_fwd'], 'wgrad256', k['conv3x3_wgrad'], 'other256', k['other'])" >> $O/step_ab3.txt
  done
done
cat $O/step_ab3.txt
