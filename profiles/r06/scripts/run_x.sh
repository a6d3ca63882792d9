#!/bin/bash
# round 6, call X: depth of the dz ring (2 with a wait per layer | 4 in pairs | 8 in pairs), 4 and 32 images
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_x
mkdir -p $O
cd $R
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
for rep in 1 2 3; do
  run b4_dz2_$rep 4 80 MIMO_HIP_LIB=$R/build/variants/libmimo_dz2.so
  run b4_dz4_$rep 4 80 MIMO_DUMMY=1
  run b4_dz8_$rep 4 80 MIMO_HIP_LIB=$R/build/variants/libmimo_dz8.so
done
for rep in 1 2; do
  run b32_dz2_$rep 32 30 MIMO_HIP_LIB=$R/build/variants/libmimo_dz2.so
  run b32_dz4_$rep 32 30 MIMO_DUMMY=1
  run b32_dz8_$rep 32 30 MIMO_HIP_LIB=$R/build/variants/libmimo_dz8.so
done
