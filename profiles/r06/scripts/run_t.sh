#!/bin/bash
# round 6, call T: side stream on a low-priority hardware queue — bench.py's one-rank RCCL route and the plain route, priority 0 / low
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_t
mkdir -p $O
cd $R
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
b() {  # name, batch, steps, env...
  local name=$1 batch=$2 steps=$3; shift 3
  env "$@" timeout 300 python bench.py --batch $batch --steps $steps --warmup 12 --no-cpu-baseline --no-strict --profile-steps 0 2>/dev/null | grep "^{" > $O/$name.json
  python - <<PY
import json
try:
    d = json.load(open("$O/$name.json")); print("$name", d["ms_per_step"], "ms/step", d["value"], "images/s")
except Exception as e:
    print("$name FAILED", e)
PY
}
for rep in 1 2; do
  b ddp_b4_prio0_$rep 4 80 MIMO_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MIMO_WGRAD_STREAM_PRIORITY=0
  b ddp_b4_low_$rep 4 80 MIMO_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
  b plain_b4_prio0_$rep 4 80 MIMO_WGRAD_STREAM_PRIORITY=0
  b plain_b4_low_$rep 4 80 MIMO_DUMMY=1
  b plain_b32_prio0_$rep 32 30 MIMO_WGRAD_STREAM_PRIORITY=0
  b plain_b32_low_$rep 32 30 MIMO_DUMMY=1
done
b ddp_b32_prio0 32 30 MIMO_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MIMO_WGRAD_STREAM_PRIORITY=0
b ddp_b32_low 32 30 MIMO_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
timeout 600 python -m pytest tests/test_streams_gpu.py tests/test_data_gpu.py tests/test_ddp_gpu.py -m gpu -q -x 2>&1 | tail -3
