#!/bin/bash
# non-temporal loads in the single-pass bandwidth kernels (default) against plain loads (variant build), alternating; parity subset
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_w
mkdir -p $O
cd $R
python -m pytest tests/test_ops_gpu.py tests/test_network_gpu.py -m gpu -q -x 2>&1 | tail -3 > $O/pytest_subset.txt
for i in 1 2 3; do for v in default nostream; do
  if [ $v = default ]; then unset MIMO_HIP_LIB; else export MIMO_HIP_LIB=$R/build/variants/libmimo_$v.so; fi
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=l['roofline']['bandwidth_kernels']; print('$v', l['value'], l['ms_per_step'], 'bw', b['ms_per_step'], {k:(v['ms_per_step'], v['hbm_frac']) for k,v in b['kernels'].items()})" >> $O/step_ab.txt
  python bench.py --batch 4 --steps 60 --warmup 10 --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v b4', l['value'], l['ms_per_step'])" >> $O/step_ab.txt
done; done
unset MIMO_HIP_LIB
cat $O/pytest_subset.txt $O/step_ab.txt
