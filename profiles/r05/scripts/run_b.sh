#!/bin/bash
# round-5 GPU call B: rest of the GPU suite, data-path probe, host-fed bench lines
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_b
mkdir -p $O
cd "$R"
python profiles/r05/scripts/data_path_probe.py > $O/data_path_probe.txt 2>&1
cat $O/data_path_probe.txt
MIMO_PARITY_LOG=$O/parity_errors.txt timeout 1800 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1
tail -25 $O/pytest.txt
for i in 1 2; do
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_resident_$i.json
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --host-batches pinned 2>/dev/null | tail -1 > $O/bench_pinned_$i.json
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --host-batches pageable 2>/dev/null | tail -1 > $O/bench_pageable_$i.json
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --batch 4 2>/dev/null | tail -1 > $O/bench_b4_$i.json
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --batch 4 --host-batches pageable 2>/dev/null | tail -1 > $O/bench_b4_pageable_$i.json
  python bench.py --steps 50 --warmup 10 --no-cpu-baseline --batch 4 --host-batches pinned 2>/dev/null | tail -1 > $O/bench_b4_pinned_$i.json
done
for f in $O/bench_*.json; do python -c "import json,sys; l=json.load(open('$f')); print('$f'.split('/')[-1], l['value'], l['ms_per_step'], l['config'].get('inputs'), l['config']['host_enqueue_ms_per_step'])"; done
