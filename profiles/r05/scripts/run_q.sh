#!/bin/bash
# data-path tests and host-fed bench lines with the host-side hand-off as the default
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_q
mkdir -p $O
cd $R
MIMO_PARITY_LOG=$O/parity.txt python -m pytest tests/test_data_gpu.py -m gpu -q 2>&1 | tail -5 > $O/pytest.txt
MIMO_PREFETCH_HANDOFF=gpu python -m pytest tests/test_data_gpu.py -m gpu -q 2>&1 | tail -5 > $O/pytest_gpu_handoff.txt
for b in 32 4; do
  for m in pinned pageable; do
    python3 bench.py --batch $b --steps $((b==4?60:30)) --warmup 8 --no-cpu-baseline --host-batches $m 2>/dev/null | tail -1 > $O/bench_b${b}_host_$m.json
  done
  python3 bench.py --batch $b --steps $((b==4?60:30)) --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_b${b}_resident.json
done
cat $O/pytest.txt $O/pytest_gpu_handoff.txt; grep -h "host" $O/parity.txt | tail -8
for f in $O/bench_*.json; do python3 -c "import json,sys; l=json.loads(open('$f').read()); print('$f'.split('/')[-1], l['value'], l['ms_per_step'])"; done
