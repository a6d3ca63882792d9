#!/bin/bash
# Per-layer convolution table (rocprofv3 kernel trace, streams serialised) for two settings of one environment switch:
#   bash scripts/ab_conv_layers.sh <tag> <VAR> <value A> <value B>
set -u
TAG=$1; VAR=$2; A=$3; B=$4
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MIMO_WGRAD_STREAM=0
for V in "$A" "$B"; do
  export $VAR=$V
  L=$(basename "$V" .so)  # label of the setting in the file names (values may be paths: MIMO_HIP_LIB)
  rocprofv3 --kernel-trace -d "$OUT/trace_$L" -o t --output-format csv -- python3 "$R/bench.py" --steps 3 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-strict > "$OUT/bench_$L.json" 2> "$OUT/trace_$L.err"
  python3 "$R/scripts/trace_convs.py" "$OUT/trace_$L" > "$OUT/conv_layers_$L.txt" 2>&1
  rm -rf "$OUT/trace_$L"
done
