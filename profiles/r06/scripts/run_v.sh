#!/bin/bash
# round 6, call V: BatchNorm-backward kernels shaped to co-reside with the weight-gradient workgroups (variant libraries in build_exp/)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_v
mkdir -p $O
cd $R
VARS=${VARS:-"v0 v1"}
for rep in 1 2; do
for v in $VARS; do
  echo -n "$v batch 32: " | tee -a $O/ab.txt
  MIMO_HIP_LIB=$R/build_exp/libmimo_$v.so timeout 300 python bench.py --steps 60 --warmup 10 --no-strict --no-cpu-baseline --profile-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
done
done
for v in $VARS; do
  echo -n "$v batch 4: " | tee -a $O/ab.txt
  MIMO_HIP_LIB=$R/build_exp/libmimo_$v.so timeout 300 python bench.py --batch 4 --steps 300 --warmup 30 --no-strict --no-cpu-baseline --profile-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
done
