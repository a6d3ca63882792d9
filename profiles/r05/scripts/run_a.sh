#!/bin/bash
# round-5 GPU call A: full GPU suite with the new tests, bench baseline, host-fed bench, batch-4 line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_a
mkdir -p $O
cd "$R"
MIMO_PARITY_LOG=$O/parity_errors.txt timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
for i in 1 2; do
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_resident_$i.json
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --host-batches pinned 2>/dev/null | tail -1 > $O/bench_pinned_$i.json
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline --host-batches pageable 2>/dev/null | tail -1 > $O/bench_pageable_$i.json
done
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --batch 4 2>/dev/null | tail -1 > $O/bench_b4.json
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --batch 4 --host-batches pageable 2>/dev/null | tail -1 > $O/bench_b4_pageable.json
for f in $O/bench_*.json; do python -c "import json,sys; l=json.load(open('$f')); print('$f'.split('/')[-1], l['value'], l['ms_per_step'], l['config'].get('inputs'), l['config']['host_enqueue_ms_per_step'])"; done
