#!/bin/bash
# bn_bwd_apply walking its tensors back to front (what the statistics pass read last is what the last-level cache holds):
# variant build against the default, alternating; the parity subset that covers the kernel
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_x
mkdir -p $O
cd $R
V=$R/build/variants/libmimo_applyrev.so
MIMO_HIP_LIB=$V python -m pytest tests/test_network_gpu.py -m gpu -q -x 2>&1 | tail -3 > $O/pytest_subset.txt
for i in 1 2 3; do for v in default applyrev; do
  if [ $v = default ]; then unset MIMO_HIP_LIB; else export MIMO_HIP_LIB=$V; fi
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=l['roofline']['bandwidth_kernels']; t=l['roofline']['tiers']; print('$v', l['value'], l['ms_per_step'], 'bw', b['ms_per_step'], {k:(v['ms_per_step'], v['hbm_frac']) for k,v in b['kernels'].items() if k.startswith('bn_bwd')}, {k: v['kernels_ms'].get('bn_bwd_apply') for k,v in t.items()})" >> $O/step_ab.txt
done; done
unset MIMO_HIP_LIB
cat $O/pytest_subset.txt $O/step_ab.txt
