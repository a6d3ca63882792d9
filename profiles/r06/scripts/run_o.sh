#!/bin/bash
# round 6, call O: why does bench.py's one-rank RCCL line (5.76 ms) differ from scripts/ddp_overhead.py (4.45 ms) at 4 images per GPU?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_o
mkdir -p $O
cd $R
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
b() {  # name, extra env / args
  local name=$1; shift
  env MIMO_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 "$@" timeout 300 python bench.py --batch 4 --steps 80 --warmup 15 --no-cpu-baseline --no-strict --profile-steps 0 2>/dev/null | grep "^{" > $O/$name.json
  python - <<PY
import json
try:
    d = json.load(open("$O/$name.json")); print("$name", d["ms_per_step"], "ms/step")
except Exception as e:
    print("$name FAILED", e)
PY
}
b bench_default
b bench_async0 MIMO_DDP_ASYNC_STAGES=0
MIMO_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --batch 4 --steps 80 --warmup 15 --no-cpu-baseline --no-strict --profile-steps 0 --labels learnable 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"bench_learnable_labels\", d[\"ms_per_step\"], \"ms/step\")"
echo "ddp_overhead.py under the bench's environment (RANK / WORLD_SIZE set):"
RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 600 python scripts/ddp_overhead.py 4 2>/dev/null | grep "async stages) + 1-rank\|^plain  " | head -3
