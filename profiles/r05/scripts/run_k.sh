#!/bin/bash
# round-5 GPU call K: bn_bwd_apply with / without the max |dz| tracking and the 7-waves launch bound (variant builds), one box
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_k
mkdir -p $O
cd "$R"
one() {
  python bench.py --steps 20 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; b=r['bandwidth_kernels']['kernels']; print('$1', l['value'], l['ms_per_step'], 'apply', b['bn_bwd_apply']['ms_per_step'], b['bn_bwd_apply']['hbm_frac'], 'reduce', b['bn_bwd_reduce']['ms_per_step'], 'bw', r['bandwidth_kernels']['ms_per_step'], 'fwd', r['kernels']['conv3x3_fwd']['ms_per_step'])" >> $O/apply_ab.txt
}
for i in 1 2; do
  one default
  MIMO_HIP_LIB=$R/build/variants/libmimo_apply_lb1.so one lb1
  MIMO_HIP_LIB=$R/build/variants/libmimo_apply_noabs.so MIMO_WGRAD_NP=3 one noabs_np3
  MIMO_HIP_LIB=$R/build/variants/libmimo_apply_lb1_noabs.so MIMO_WGRAD_NP=3 one lb1_noabs_np3
  MIMO_WGRAD_NP=3 one default_np3
done
cat $O/apply_ab.txt
for i in 1 2 3; do for v in 3 2; do
  MIMO_WGRAD_NP=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --batch 4 --profile-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('b4 np=$v', l['value'], l['ms_per_step'])" >> $O/b4_ab.txt
done; done
cat $O/b4_ab.txt
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_network_gpu.py -q -m gpu -x -k "wgrad or golden or two_mfma or cfg3_shape or accumulation" > $O/pytest.txt 2>&1
tail -4 $O/pytest.txt
