#!/bin/bash
# round-4 GPU call I: Up-block outputs read through BatchNorm + ReLU by upcat_fwd / the head kernels — suite + step A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_i
mkdir -p $O
cd $R
python -m pytest tests -m gpu -x -q --tb=short 2>&1 | tail -15 > $O/pytest_all.txt
tail -4 $O/pytest_all.txt
for i in 1 2 3; do
  for v in 0 1; do
    MIMO_FUSE_BN_IN=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print('fuse$v', l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in r['kernels'].items()}, 'bw', r['bandwidth_kernels']['ms_per_step'], {k:v['ms_per_step'] for k,v in r['bandwidth_kernels']['kernels'].items()})" >> $O/step_ab.txt
  done
done
cat $O/step_ab.txt
