#!/bin/bash
# round 6, call H: hand-off events attached to the launches (MIMO_EVENT_ON_LAUNCH) and dz buffers released in pairs (4 buffers):
# bit-identity tests, then alternating A/B at 4 and 32 images: [records + 2 buffers] | [records + 4 paired] | [attached + 2] | [attached + 4 paired]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_h
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_streams_gpu.py tests/test_network_gpu.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest.txt
run() {  # name, batch, steps, env...
  local name=$1 batch=$2 steps=$3; shift 3
  env "$@" timeout 300 python bench.py --batch $batch --steps $steps --warmup 10 --no-cpu-baseline --no-strict --profile-steps 0 2>/dev/null | tail -1 > $O/$name.json
  python - <<PY
import json
try:
    d = json.load(open("$O/$name.json")); print("$name", d["ms_per_step"], "ms/step", d["value"], "images/s")
except Exception as e:
    print("$name FAILED", e)
PY
}
V=$R/build/variants/libmimo_dz2.so
for rep in 1 2 3; do
  run b4_rec_dz2_$rep 4 80 MIMO_EVENT_ON_LAUNCH=0 MIMO_HIP_LIB=$V
  run b4_rec_dz4_$rep 4 80 MIMO_EVENT_ON_LAUNCH=0
  run b4_att_dz2_$rep 4 80 MIMO_HIP_LIB=$V
  run b4_att_dz4_$rep 4 80 MIMO_EVENT_ON_LAUNCH=1
done
for rep in 1 2 3; do
  run b32_rec_dz2_$rep 32 30 MIMO_EVENT_ON_LAUNCH=0 MIMO_HIP_LIB=$V
  run b32_rec_dz4_$rep 32 30 MIMO_EVENT_ON_LAUNCH=0
  run b32_att_dz2_$rep 32 30 MIMO_HIP_LIB=$V
  run b32_att_dz4_$rep 32 30 MIMO_EVENT_ON_LAUNCH=1
done
