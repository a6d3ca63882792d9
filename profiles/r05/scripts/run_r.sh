#!/bin/bash
# batch 4: overlapped (default) kernel trace -> idle time of the chip inside a step; side stream on / off A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_r
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/t -o t --output-format csv -- python3 $R/bench.py --batch 4 --steps 12 --warmup 5 --profile-steps 0 --no-cpu-baseline > $O/bench_traced.json 2> $O/err.txt
python3 $R/scripts/trace_overlap.py $O/t 0 0 > $O/overlap_b4.txt 2>&1
python3 $R/scripts/trace_overlap.py $O/t > $O/overlap_b4_full.txt 2>&1
rm -rf $O/t
cd $R
for i in 1 2 3; do for w in 1 0; do
  MIMO_WGRAD_STREAM=$w python3 bench.py --batch 4 --steps 60 --warmup 10 --profile-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('b4 wgrad_stream=$w', l['value'], l['ms_per_step'])" >> $O/ab.txt
done; done
head -30 $O/overlap_b4.txt; cat $O/ab.txt
