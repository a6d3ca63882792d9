#!/bin/bash
# round-4 GPU call L: backward twin — dz formed in the gradient kernels' loaders: bit-identity + parity + step A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_l
mkdir -p $O
cd $R
python -m pytest tests/test_network_gpu.py tests/test_ops_gpu.py -x -q --tb=short 2>&1 | tail -12 > $O/pytest.txt
tail -4 $O/pytest.txt
for i in 1 2 3; do
  for v in 0 1; do
    MIMO_FUSE_BN_DZ=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print('dz$v', l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in r['kernels'].items()}, 'bw', r['bandwidth_kernels']['ms_per_step'], 'apply', r['bandwidth_kernels']['kernels']['bn_bwd_apply']['ms_per_step'])" >> $O/step_ab.txt
  done
done
cat $O/step_ab.txt
