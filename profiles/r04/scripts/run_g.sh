#!/bin/bash
# round-4 GPU call G: BatchNorm + ReLU applied in the second convolution's loaders — bit-identity, parity suite, step A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_g
mkdir -p $O
cd $R
python -m pytest tests/test_network_gpu.py -x -q -k "bit_identical or golden or cfg3_shape or cfg2_shape" --tb=short 2>&1 | tail -30 > $O/pytest_a.txt
tail -5 $O/pytest_a.txt
python -m pytest tests -m gpu -x -q --tb=short 2>&1 | tail -15 > $O/pytest_all.txt
for i in 1 2 3 4; do
  for v in 0 1; do
    MIMO_FUSE_BN_IN=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print('fuse$v', l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in r['kernels'].items()}, 'bw', r['bandwidth_kernels']['ms_per_step'], 'bnfwd', r['bandwidth_kernels']['kernels']['bn_relu_fwd']['ms_per_step'])" >> $O/step_ab.txt
  done
done
tail -6 $O/pytest_all.txt; cat $O/step_ab.txt
