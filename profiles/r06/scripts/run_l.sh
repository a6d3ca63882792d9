#!/bin/bash
# round 6, call L: K split of the 256-pixel persistent convolution — operator parity (cost rule + forced), network tests, A/B at 4 images
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_l
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "conv3x3_forward_dgrad_wgrad and split16" 2>&1 | tail -5 | tee $O/pytest_ops.txt
MIMO_CONV_KSPLIT=3 timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "conv3x3_forward_dgrad_wgrad and split16" 2>&1 | tail -5 | tee $O/pytest_ops_forced3.txt
MIMO_CONV_KSPLIT=2 timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_network_gpu.py tests/test_streams_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest_forced2.txt
timeout 900 python -m pytest tests/test_network_gpu.py tests/test_streams_gpu.py tests/test_configs_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest_default.txt
run() {  # name, batch, steps, env...
  local name=$1 batch=$2 steps=$3; shift 3
  env "$@" timeout 300 python bench.py --batch $batch --steps $steps --warmup 10 --no-cpu-baseline --no-strict --profile-steps 3 2>/dev/null | tail -1 > $O/$name.json
  python - <<PY
import json
try:
    d = json.load(open("$O/$name.json")); k = d["roofline"]["kernels"]
    print("$name", d["ms_per_step"], "ms/step", d["value"], "images/s; fwd", k["conv3x3_fwd"]["ms_per_step"], "dgrad", k["conv3x3_dgrad"]["ms_per_step"], "ms (second pass)")
except Exception as e:
    print("$name FAILED", e)
PY
}
for rep in 1 2 3; do
  run b4_nosplit_$rep 4 80 MIMO_CONV_KSPLIT=0
  run b4_ksplit_$rep 4 80 MIMO_DUMMY=1
done
run b32_nosplit 32 30 MIMO_CONV_KSPLIT=0
run b32_ksplit 32 30 MIMO_DUMMY=1
run b8_nosplit 8 60 MIMO_CONV_KSPLIT=0
run b8_ksplit 8 60 MIMO_DUMMY=1
