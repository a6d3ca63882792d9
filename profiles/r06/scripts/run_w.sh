#!/bin/bash
# round 6, call W: CU share of the weight-gradient launches at batch 32, fine steps below the whole chip (anatomy of the overlap:
# profiles/r06/exp/overlap_anatomy.txt — do a few free CUs let the BatchNorm passes run beside the weight gradient?)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_w
mkdir -p $O
cd $R
for rep in 1 2; do
for c in 256 248 240 224 208; do
  echo -n "MIMO_WGRAD_CUS=$c batch 32: " | tee -a $O/ab.txt
  MIMO_WGRAD_CUS=$c timeout 300 python bench.py --steps 60 --warmup 10 --no-strict --no-cpu-baseline --profile-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" | tee -a $O/ab.txt
done
done
