#!/bin/bash
# round 6, call B: where the data-parallel route's time goes at 4 images per GPU (one-rank RCCL group): plain / staged backward /
# + reducer / + collectives, eager and as graph replays; plus the new tests of this round's host-side changes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_b
mkdir -p $O
cd $R
for g in 0 1; do
  echo "== MIMO_TRAIN_GRAPH=$g, batch 4" | tee -a $O/ddp_overhead_b4.txt
  MIMO_TRAIN_GRAPH=$g timeout 600 python scripts/ddp_overhead.py 4 2>/dev/null | tee -a $O/ddp_overhead_b4.txt
done
timeout 900 python -m pytest tests/test_data_gpu.py tests/test_ddp_gpu.py -m gpu -q -x 2>&1 | tail -6 | tee $O/pytest_data_ddp.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 2 > $O/bench_default.json 2> $O/bench_default.err
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("bench default: value", d["value"], "strict", d["value_strict"], "labels", d["config"]["labels"])
PY
