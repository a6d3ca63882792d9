#!/bin/bash
# round-4 GPU call R: the image convolution on plain-FMA kernels (MIMO_CONV_THIN): op parity, network goldens, step A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_r
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_network_gpu.py -x -q -m gpu -k "conv3x3_forward or golden or odd or accumulation or fgsm or input_grad or dx" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for i in 1 2 3; do
  for v in 1 0; do
    MIMO_CONV_THIN=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print('thin=$v', l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in r['kernels'].items()}, r['tiers']['256x256']['kernels_ms'])" >> $O/step_ab.txt
  done
done
cat $O/step_ab.txt
