#!/bin/bash
# side-stream weight-gradient launches sized for part of the chip (MIMO_WGRAD_CUS): step time at 4 and at 32 images
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_cus
mkdir -p $O
cd $R
for i in 1 2; do for c in 256 192 128 96 64; do
  MIMO_WGRAD_CUS=$c python3 bench.py --batch 4 --steps 60 --warmup 10 --profile-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('b4 cus=$c', l['value'], l['ms_per_step'])" >> $O/ab.txt
done; done
for c in 256 192 128; do
  MIMO_WGRAD_CUS=$c python3 bench.py --steps 25 --warmup 8 --profile-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('b32 cus=$c', l['value'], l['ms_per_step'])" >> $O/ab.txt
done
cat $O/ab.txt
