#!/bin/bash
# Per-shape conv timings (scripts/conv_layer_bench.py) for several settings in one GPU call:
#   bash scripts/layer_ab.sh <tag> "<ENV=VAL ...>" "<ENV=VAL ...>" ...
# each argument after the tag is one setting (a space-separated list of environment assignments; "-" = none)
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
I=0
for SETTING in "$@"; do
  D=$OUT/t$I
  mkdir -p "$D"
  (
    if [ "$SETTING" != "-" ]; then export $SETTING; fi
    export MIMO_LAYER_BENCH_LABELS=$D/labels.txt
    timeout ${RUN_TIMEOUT:-120} rocprofv3 --kernel-trace -d "$D" -o t --output-format csv -- python3 "$R/scripts/conv_layer_bench.py" run ${REPS:-2} > "$D/run.out" 2> "$D/run.err"
  )
  echo "== $SETTING" > "$OUT/layers_$I.txt"
  python3 "$R/scripts/conv_layer_bench.py" report "$D" >> "$OUT/layers_$I.txt" 2>&1
  rm -rf "$D"
  I=$((I+1))
done
