#!/bin/bash
# dz ring of 2 (default), 3, 4 buffers between the main stream and the side stream's weight gradients: step at 4 / 8 / 32 images
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_dzbufs
mkdir -p $O
cd $R
one() { if [ $2 = 2 ]; then unset MIMO_HIP_LIB; else export MIMO_HIP_LIB=$R/build/variants/libmimo_dz$2.so; fi; python3 bench.py --batch $1 --steps $3 --warmup 10 --profile-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('b$1 dz buffers $2', l['value'], l['ms_per_step'])" >> $O/ab.txt; }
for i in 1 2; do for d in 2 3 4; do one 4 $d 60; one 8 $d 50; one 32 $d 25; done; done
unset MIMO_HIP_LIB
cat $O/ab.txt
