#!/bin/bash
# round-5 GPU call L: max |dz| per workgroup slot: batch-4 and batch-32 A/B of the weight-gradient arithmetic, parity subset
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_l
mkdir -p $O
cd "$R"
for i in 1 2 3; do for v in 3 2; do
  MIMO_WGRAD_NP=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --batch 4 --profile-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('b4 np=$v', l['value'], l['ms_per_step'])" >> $O/b4_ab.txt
done; done
cat $O/b4_ab.txt
for i in 1 2; do for v in 3 2; do
  MIMO_WGRAD_NP=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; b=r['bandwidth_kernels']['kernels']; print('b32 np=$v', l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in r['kernels'].items()}, 'apply', b['bn_bwd_apply']['ms_per_step'], 'bw', r['bandwidth_kernels']['ms_per_step'])" >> $O/b32_ab.txt
done; done
cat $O/b32_ab.txt
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_network_gpu.py tests/test_configs_gpu.py -q -m gpu -x -k "wgrad or golden or two_mfma or cfg3 or cfg4 or accumulation" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
