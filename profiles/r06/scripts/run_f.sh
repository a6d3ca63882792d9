#!/bin/bash
# round 6, call F: overlapped timeline of one step at 4 images per GPU (both streams, default configuration, eager launches)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_f
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MIMO_TRAIN_GRAPH=0 rocprofv3 --kernel-trace -d $O/t -o b4 --output-format csv -- python3 $R/bench.py --batch 4 --steps 12 --warmup 5 --profile-steps 0 --no-cpu-baseline --no-strict > $O/bench_under_rocprof.json 2> $O/err.txt
cd $R
python3 scripts/trace_overlap.py $O/t > $O/overlap_b4.txt 2>&1
tail -12 $O/overlap_b4.txt
rm -rf $O/t
