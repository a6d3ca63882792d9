#!/bin/bash
# the CU-share rule on the other configurations: cfg4 (S = 4, batch 16 -> 192 CUs by the rule) and cfg2 (batch 64 -> 256, unchanged) against MIMO_WGRAD_CUS=256
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_cus4
mkdir -p $O
cd $R
one() { if [ $2 = default ]; then unset MIMO_WGRAD_CUS; else export MIMO_WGRAD_CUS=$2; fi; python3 bench.py --config $1 --steps 12 --warmup 4 --profile-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('$1 cus=$2', l['value'], l['ms_per_step'])" >> $O/ab.txt; }
for i in 1 2 3; do for c in 256 default 128; do one cfg4 $c; done; done
for i in 1 2; do for c in 256 default; do one cfg2 $c; done; done
unset MIMO_WGRAD_CUS
cat $O/ab.txt
