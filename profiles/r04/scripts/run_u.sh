#!/bin/bash
# round-4 GPU call U: BatchNorm-backward passes with 4 pixels in flight per thread: grid scan, step A/B against HEAD~
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_u
mkdir -p $O
cd $R
# (parity of the 4-pixel build was checked in the previous call, run_s.sh's suite on the same sources; this call only times)
one() {
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; b=r['bandwidth_kernels']['kernels']; print('$1', l['value'], l['ms_per_step'], 'bw', r['bandwidth_kernels']['ms_per_step'], 'reduce', b['bn_bwd_reduce']['ms_per_step'], 'apply', b['bn_bwd_apply']['ms_per_step'])" >> $O/step_ab.txt
}
for i in 1 2; do
  MIMO_BN_BLOCKS_R=448 MIMO_BN_BLOCKS_A=1792 one "un2-R448-A1792"
  MIMO_BN_BLOCKS_R=384 MIMO_BN_BLOCKS_A=1536 one "un2-R384-A1536"
  MIMO_BN_BLOCKS_R=256 MIMO_BN_BLOCKS_A=1280 one "un2-R256-A1280"
  MIMO_HIP_LIB=$R/build/variants/libmimo_prev.so one "previous-build"
done
cat $O/step_ab.txt
