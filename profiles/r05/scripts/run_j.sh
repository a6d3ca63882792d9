#!/bin/bash
# round-5 GPU call J: up-sampling kernels by 2 x 2 blocks (forward and backward): parity subset, step A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_j
mkdir -p $O
cd "$R"
MIMO_PARITY_LOG=$O/parity_errors.txt timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_network_gpu.py tests/test_configs_gpu.py -q -m gpu > $O/pytest.txt 2>&1
tail -6 $O/pytest.txt
for i in 1 2 3; do
  for v in 1 0; do
    MIMO_UPCAT_2X2=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; b=r['bandwidth_kernels']['kernels']; print('2x2=$v', l['value'], l['ms_per_step'], 'upcat', b['upcat_fwd']['ms_per_step'], b['upcat_fwd']['hbm_frac'], 'up_bwd', b['up_bwd']['ms_per_step'], b['up_bwd']['hbm_frac'], 'bw', r['bandwidth_kernels']['ms_per_step'])" >> $O/step_ab.txt
  done
done
cat $O/step_ab.txt
MIMO_UPCAT_2X2=1 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --batch 4 --profile-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('b4 2x2=1', l['value'], l['ms_per_step'])"
MIMO_UPCAT_2X2=0 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --batch 4 --profile-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('b4 2x2=0', l['value'], l['ms_per_step'])"
