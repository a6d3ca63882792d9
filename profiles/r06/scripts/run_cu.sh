#!/bin/bash
# round 6, call CU: side stream with a CU mask (hipExtStreamCreateWithCUMask) — n CUs kept free for the main stream's kernels while a
# weight gradient holds the rest (profiles/r06/exp/overlap_anatomy.txt).  The step runs on a non-default stream here: a CU-mask stream
# is a BLOCKING stream (implicit synchronisation with the null stream).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_cu
mkdir -p $O
cd $R
export MIMO_EXP_BENCH_OWN_STREAM=1
run() { timeout 300 python bench.py "$@" --no-strict --no-cpu-baseline --profile-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for rep in 1 2; do
for f in 0 16 32 48 64 96; do
  echo -n "free=$f batch 32: " | tee -a $O/ab.txt
  MIMO_EXP_WGRAD_CU_FREE=$f run --steps 60 --warmup 10 | tee -a $O/ab.txt
done
done
for f in 32 64; do
  echo -n "free=$f layout 1 batch 32: " | tee -a $O/ab.txt
  MIMO_EXP_WGRAD_CU_FREE=$f MIMO_EXP_WGRAD_CU_LAYOUT=1 run --steps 60 --warmup 10 | tee -a $O/ab.txt
done
for f in 0 32 64 128; do
  echo -n "free=$f batch 4: " | tee -a $O/ab.txt
  MIMO_EXP_WGRAD_CU_FREE=$f MIMO_WGRAD_CUS=256 run --batch 4 --steps 300 --warmup 30 | tee -a $O/ab.txt
done
echo -n "free=0 batch 4 (default share): " | tee -a $O/ab.txt
run --batch 4 --steps 300 --warmup 30 | tee -a $O/ab.txt
