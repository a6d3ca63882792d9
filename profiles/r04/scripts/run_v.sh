#!/bin/bash
# round-4 GPU call V: BatchNorm-backward sums in the data-gradient epilogue (MIMO_FUSE_BST): parity, step A/B, threshold scan
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_v
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_network_gpu.py -x -q -m gpu -k "golden or pool_and_head or bit_identical or fgsm or odd or accumulation" > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt
one() {
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; b=r['bandwidth_kernels']['kernels']; print('$1', l['value'], l['ms_per_step'], 'dgrad', r['kernels']['conv3x3_dgrad']['ms_per_step'], 'reduce', b['bn_bwd_reduce']['ms_per_step'], b['bn_bwd_reduce']['launches_per_step'], 'bw', r['bandwidth_kernels']['ms_per_step'])" >> $O/step_ab.txt
}
for i in 1 2; do
  MIMO_FUSE_BST=0 one "bst=0"
  one "bst=all"
  MIMO_FUSE_BST_MINPIX=4096 one "bst>=64x64"
  MIMO_FUSE_BST_MINPIX=16384 one "bst>=128x128"
done
cat $O/step_ab.txt
