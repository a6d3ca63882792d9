#!/bin/bash
# round 6, call Y: one-launch BatchNorm backward of small tensors (bn_bwd_small_kernel): suite files that cover it, then the
# step at 4 images per GPU with and without (MIMO_BN_BWD_SMALL=0), and a kernel trace of each
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_y
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_network_gpu.py tests/test_streams_gpu.py tests/test_variants_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.txt
for rep in 1 2; do
for v in 1 0; do
  echo "== MIMO_BN_BWD_SMALL=$v" | tee -a $O/b4.txt
  MIMO_BN_BWD_SMALL=$v timeout 300 python bench.py --batch 4 --steps 300 --warmup 30 --no-strict --no-cpu-baseline --profile-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" | tee -a $O/b4.txt
done
done
for v in 1 0; do
  echo "== batch 32 MIMO_BN_BWD_SMALL=$v" | tee -a $O/b32.txt
  MIMO_BN_BWD_SMALL=$v timeout 300 python bench.py --steps 60 --warmup 10 --no-strict --no-cpu-baseline --profile-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" | tee -a $O/b32.txt
done
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  MIMO_BN_BWD_SMALL=$v MIMO_WGRAD_ASYNC=0 timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_small$v -o t -- python3 $R/bench.py --batch 4 --steps 20 --warmup 5 --no-strict --no-cpu-baseline --profile-steps 0 > /dev/null 2>&1
  f=$(ls $O/trace_small$v/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && grep -i "bn_bwd\|bnrelu_bwd" $f | cut -c1-200 | tee -a $O/kernels_small$v.txt
  rm -rf $O/trace_small$v/*/*.db $O/trace_small$v/*/*kernel_trace.csv
done
