#!/bin/bash
# round 6: the per-GPU batches of a 2- and 4-GPU strong-scaling run (16 and 8 images), plain and on the one-rank RCCL route
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_mid
mkdir -p $O
cd $R
run() { timeout 300 python bench.py "$@" --no-strict --no-cpu-baseline --profile-steps 0 2>/dev/null | grep "^{" | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"; }
for b in 16 8 4; do
  echo -n "batch $b plain: " | tee -a $O/mid.txt; run --batch $b --steps 150 --warmup 20 | tee -a $O/mid.txt
  echo -n "batch $b one-rank RCCL: " | tee -a $O/mid.txt
  RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 MIMO_BENCH_FORCE_DIST=1 run --batch $b --steps 150 --warmup 20 | tee -a $O/mid.txt
done
