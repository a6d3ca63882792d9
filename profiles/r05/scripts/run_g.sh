#!/bin/bash
# round-5 GPU call G: full GPU suite, batch-4 A/B of the weight-gradient arithmetic after the cheaper max |dz| hand-over,
# batch-32 lines, ordered kernel timeline of a batch-4 step
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_g
mkdir -p $O
cd "$R"
MIMO_PARITY_LOG=$O/parity_errors.txt timeout 2000 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1
tail -8 $O/pytest.txt
for i in 1 2 3; do
  for v in 3 2; do
    MIMO_WGRAD_NP=$v python bench.py --steps 60 --warmup 10 --no-cpu-baseline --batch 4 --profile-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('b4 np=$v', l['value'], l['ms_per_step'])" >> $O/b4_ab.txt
  done
done
cat $O/b4_ab.txt
for i in 1 2; do
  for v in 3 2; do
    MIMO_WGRAD_NP=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print('b32 np=$v', l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in r['kernels'].items()}, r['bandwidth_kernels']['ms_per_step'])" >> $O/b32_ab.txt
  done
done
cat $O/b32_ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/b4trace -o t --output-format csv -- python3 $R/bench.py --batch 4 --steps 12 --warmup 5 --profile-steps 0 --no-cpu-baseline > $O/b4trace.json 2> $O/b4trace.err
python3 $R/scripts/trace_step.py $O/b4trace > $O/b4_step.txt 2>&1
rm -rf $O/b4trace
tail -5 $O/b4_step.txt
