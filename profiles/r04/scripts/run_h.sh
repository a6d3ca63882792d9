#!/bin/bash
# round-4 GPU call H: bilinear backward with the border as independent loads — parity, batch-4 and batch-32 step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_h
mkdir -p $O
cd $R
python -m pytest tests/test_ops_gpu.py tests/test_network_gpu.py -x -q --tb=short 2>&1 | tail -8 > $O/pytest.txt
tail -3 $O/pytest.txt
for i in 1 2; do
python bench.py --batch 4 --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print('b4', l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in r['bandwidth_kernels']['kernels'].items()})" >> $O/steps.txt
python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print('b32', l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in r['bandwidth_kernels']['kernels'].items()})" >> $O/steps.txt
done
cat $O/steps.txt
