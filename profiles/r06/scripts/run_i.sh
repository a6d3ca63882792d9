#!/bin/bash
# round 6, call I: fused loss-buffer step (MIMO_LOSS_STEP_FUSED) A/B at 4 and 32 images, then the whole GPU suite with the parity log
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_i
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_network_gpu.py tests/test_streams_gpu.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest_net.txt
run() {  # name, batch, steps, env...
  local name=$1 batch=$2 steps=$3; shift 3
  env "$@" timeout 300 python bench.py --batch $batch --steps $steps --warmup 10 --no-cpu-baseline --no-strict --profile-steps 0 2>/dev/null | tail -1 > $O/$name.json
  python - <<PY
import json
try:
    d = json.load(open("$O/$name.json")); print("$name", d["ms_per_step"], "ms/step", d["value"], "images/s", "host", d["config"]["host_enqueue_ms_per_step"])
except Exception as e:
    print("$name FAILED", e)
PY
}
for rep in 1 2 3; do
  run b4_torch_$rep 4 80 MIMO_LOSS_STEP_FUSED=0
  run b4_fused_$rep 4 80 MIMO_LOSS_STEP_FUSED=1
done
for rep in 1 2; do
  run b32_torch_$rep 32 30 MIMO_LOSS_STEP_FUSED=0
  run b32_fused_$rep 32 30 MIMO_LOSS_STEP_FUSED=1
done
MIMO_PARITY_LOG=$O/parity_errors.txt timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/pytest_gpu_tail.txt
