#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_z
mkdir -p $O
cd $R
for rep in 1 2 3; do
for v in 1 4 16; do
  echo -n "fan_small=$v batch 4: " | tee -a $O/ab.txt
  MIMO_EXP_REDUCE_FAN_SMALL=$v timeout 300 python bench.py --batch 4 --steps 400 --warmup 40 --no-strict --no-cpu-baseline --profile-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
done
done
for v in 1 16; do
  echo -n "fan_small=$v batch 32: " | tee -a $O/ab.txt
  MIMO_EXP_REDUCE_FAN_SMALL=$v timeout 300 python bench.py --steps 60 --warmup 10 --no-strict --no-cpu-baseline --profile-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
done
