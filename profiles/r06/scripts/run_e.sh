#!/bin/bash
# round 6, call E: asynchronous backward stages (no per-stage join of the two streams) on the data-parallel route, eager / graph;
# MIMO_EW_MIN_ITERS scan of the bandwidth kernels' grids at 4 and 32 images
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_e
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_streams_gpu.py tests/test_network_gpu.py tests/test_ddp_gpu.py tests/test_data_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.txt
for g in 0 1; do
  echo "== MIMO_TRAIN_GRAPH=$g, batch 4" | tee -a $O/ddp_overhead_b4.txt
  MIMO_TRAIN_GRAPH=$g timeout 600 python scripts/ddp_overhead.py 4 2>/dev/null | grep -v "version\|Hostname\|path" | tee -a $O/ddp_overhead_b4.txt
done
for rep in 1 2; do
  for it in 1 2 4 8; do
    MIMO_EW_MIN_ITERS=$it timeout 300 python bench.py --batch 4 --steps 60 --warmup 15 --no-cpu-baseline --no-strict --profile-steps 3 > $O/b4_it${it}_$rep.json 2> $O/b4_it${it}_$rep.err
    python - <<PY
import json
try:
    d = json.load(open("$O/b4_it${it}_$rep.json"))
    bw = d["roofline"]["bandwidth_kernels"]
    print("b4 min_iters=$it rep=$rep ms/step", d["ms_per_step"], "bandwidth classes", bw["ms_per_step"], {k: v["ms_per_step"] for k, v in bw["kernels"].items()})
except Exception as e:
    print("b4 it=$it FAILED", e)
PY
  done
done
for it in 1 2 4; do
  MIMO_EW_MIN_ITERS=$it timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-strict --profile-steps 3 > $O/b32_it${it}.json 2> $O/b32_it${it}.err
  python - <<PY
import json
try:
    d = json.load(open("$O/b32_it${it}.json"))
    bw = d["roofline"]["bandwidth_kernels"]
    print("b32 min_iters=$it ms/step", d["ms_per_step"], "bandwidth classes", bw["ms_per_step"], {k: v["ms_per_step"] for k, v in bw["kernels"].items()})
except Exception as e:
    print("b32 it=$it FAILED", e)
PY
done
