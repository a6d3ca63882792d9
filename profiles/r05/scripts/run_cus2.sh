#!/bin/bash
# MIMO_WGRAD_CUS at 8 and 16 images per GPU (the per-GPU batches of 4- and 2-way strong scaling), more values at 4, three pairs at 32
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_cus2
mkdir -p $O
cd $R
one() { MIMO_WGRAD_CUS=$2 python3 bench.py --batch $1 --steps $3 --warmup 10 --profile-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('b$1 cus=$2', l['value'], l['ms_per_step'])" >> $O/ab.txt; }
for i in 1 2; do for c in 256 160 128 112; do one 4 $c 60; done; done
for i in 1 2; do for c in 256 192 128; do one 8 $c 50; done; done
for i in 1 2; do for c in 256 192 128; do one 16 $c 40; done; done
for i in 1 2 3; do for c in 256 128; do one 32 $c 25; done; done
cat $O/ab.txt
