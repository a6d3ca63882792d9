#!/bin/bash
# non-temporal loads / stores in the NHWC bandwidth kernels (variant builds -DMIMO_EW_NT=1|2|3) against the default, alternating
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_v
mkdir -p $O
cd $R
for i in 1 2; do for v in default ewnt1 ewnt2 ewnt3; do
  if [ $v = default ]; then unset MIMO_HIP_LIB; else export MIMO_HIP_LIB=$R/build/variants/libmimo_$v.so; fi
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=l['roofline']['bandwidth_kernels']; print('$v', l['value'], l['ms_per_step'], 'bw', b['ms_per_step'], {k:(v['ms_per_step'], v['hbm_frac']) for k,v in b['kernels'].items()})" >> $O/step_ab.txt
done; done
unset MIMO_HIP_LIB
cat $O/step_ab.txt
