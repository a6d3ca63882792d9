#!/bin/bash
# round-4 GPU call S: fused gradient sources in the 16-bit storage modes: parity tests, step A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_s
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_network_gpu.py tests/test_mixed_precision_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt; grep "fused gradient" $O/pytest.txt
for i in 1 2 3; do
  for v in 1 0; do
    MIMO_PRECISION=16-mixed MIMO_FUSE_BWD_SRC=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print('16-mixed fuse_bwd_src=$v', l['value'], l['ms_per_step'], 'bw', r['bandwidth_kernels']['ms_per_step'])" >> $O/step_ab.txt
  done
done
cat $O/step_ab.txt
