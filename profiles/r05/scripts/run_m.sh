#!/bin/bash
# round-5 GPU call M: activation fragments of the two-MFMA weight gradient read two taps ahead (variant build) against one
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_m
mkdir -p $O
cd "$R"
V=$R/build/variants/libmimo_np2depth2.so
for i in 1 2 3; do
  for v in depth1 depth2; do
    if [ $v = depth2 ]; then export MIMO_HIP_LIB=$V; else unset MIMO_HIP_LIB; fi
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print('$v', l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in r['kernels'].items()})" >> $O/step_ab.txt
  done
done
unset MIMO_HIP_LIB
cat $O/step_ab.txt
