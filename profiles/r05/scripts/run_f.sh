#!/bin/bash
# round-5 GPU call F: full GPU suite after the fixes, batch-4 step A/B of the two-MFMA weight gradient, host-fed bench lines
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_f
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
for m in pinned pageable; do
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --host-batches $m 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('b32 $m', l['value'], l['ms_per_step'], l['config']['inputs'])" >> $O/host_fed.txt
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline --batch 4 --host-batches $m 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('b4 $m', l['value'], l['ms_per_step'], l['config']['inputs'])" >> $O/host_fed.txt
done
python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('b32 resident', l['value'], l['ms_per_step'])" >> $O/host_fed.txt
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --batch 4 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('b4 resident', l['value'], l['ms_per_step'])" >> $O/host_fed.txt
cat $O/host_fed.txt
