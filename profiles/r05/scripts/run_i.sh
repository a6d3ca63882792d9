#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_i
mkdir -p $O
cd "$R"
python profiles/r05/scripts/debug_status.py 2>&1 | tail -8 | tee $O/debug_status.txt
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_network_gpu.py -q -m gpu -x -k "upsample or golden or odd or bit_identical or fgsm" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
for i in 1 2 3; do
  for v in 1 0; do
    MIMO_UPCAT_2X2=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; b=r['bandwidth_kernels']['kernels']; print('upcat2x2=$v', l['value'], l['ms_per_step'], 'upcat', b['upcat_fwd']['ms_per_step'], b['upcat_fwd']['hbm_frac'], 'bw', r['bandwidth_kernels']['ms_per_step'])" >> $O/step_ab.txt
  done
done
cat $O/step_ab.txt
