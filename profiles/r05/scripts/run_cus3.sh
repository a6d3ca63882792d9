#!/bin/bash
# the CU-share rule as the default against MIMO_WGRAD_CUS=256 at 4 / 8 / 16 / 32 images; GPU suite
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_cus3
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -3 > $O/pytest.txt
one() { if [ $2 = default ]; then unset MIMO_WGRAD_CUS; else export MIMO_WGRAD_CUS=$2; fi; python3 bench.py --batch $1 --steps $3 --warmup 10 --profile-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('b$1 cus=$2', l['value'], l['ms_per_step'])" >> $O/ab.txt; }
for i in 1 2; do for c in 256 default; do one 4 $c 60; one 8 $c 50; one 16 $c 40; one 32 $c 25; done; done
unset MIMO_WGRAD_CUS
cat $O/pytest.txt $O/ab.txt
