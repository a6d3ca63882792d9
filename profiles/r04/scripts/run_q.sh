#!/bin/bash
# round-4 GPU call Q: pool / head gradients formed by the BatchNorm backward (MIMO_FUSE_BWD_SRC): parity, then step A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_q
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_network_gpu.py -x -q -m gpu -k "pool_and_head or golden or bit_identical or fgsm or accumulation" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for i in 1 2 3 4; do
  for v in 1 0; do
    MIMO_FUSE_BWD_SRC=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print('fuse_bwd_src=$v', l['value'], l['ms_per_step'], 'bw', r['bandwidth_kernels']['ms_per_step'], {k:v['ms_per_step'] for k,v in r['bandwidth_kernels']['kernels'].items()})" >> $O/step_ab.txt
  done
done
cat $O/step_ab.txt
