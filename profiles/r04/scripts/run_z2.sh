#!/bin/bash
# round-4 GPU call Z2: batch-4 step after pricing both convolution families by the CUs their persistent grids occupy
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_z
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_configs_gpu.py tests/test_network_gpu.py -x -q -m gpu -k "golden or batch or cfg3" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for i in 1 2 3; do
  python bench.py --batch 4 --steps 40 --warmup 10 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b4', l['value'], l['ms_per_step'])" >> $O/b4_step.txt
done
python bench.py --steps 30 --warmup 8 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b32', l['value'], l['ms_per_step'])" >> $O/b4_step.txt
cat $O/b4_step.txt
