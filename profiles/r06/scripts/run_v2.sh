#!/bin/bash
# round 6, call V2: kernel traces (overlapped streams) of the variant libraries of call V
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_v
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in ${VARS:-v0 v1}; do
  export MIMO_HIP_LIB=$R/build_exp/libmimo_$v.so
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/tr_$v -o t --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-strict --no-cpu-baseline --profile-steps 0 > /dev/null 2>&1
  f=$(find $O/tr_$v -name "*kernel_stats.csv" | head -1)
  cp $f $O/kernel_stats_$v.csv
  find $O/tr_$v -name "*kernel_trace.csv" -delete; find $O/tr_$v -name "*.db" -delete
done
