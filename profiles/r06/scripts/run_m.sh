#!/bin/bash
# round 6, call M: serialised timeline of one step at 4 images per GPU with / without the K split (which layers split, what they and the reduction cost)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in 0 -1; do
  MIMO_CONV_KSPLIT=$v MIMO_WGRAD_STREAM=0 rocprofv3 --kernel-trace -d $O/t$v -o b4 --output-format csv -- python3 $R/bench.py --batch 4 --steps 8 --warmup 4 --profile-steps 0 --no-cpu-baseline --no-strict > $O/bench_$v.json 2> $O/err_$v.txt
  python3 $R/scripts/trace_step.py $O/t$v > $O/step_serial_ksplit_$v.txt 2>&1
  rm -rf $O/t$v
done
grep -c "conv_ksplit_reduce" $O/step_serial_ksplit_-1.txt
tail -3 $O/step_serial_ksplit_-1.txt
