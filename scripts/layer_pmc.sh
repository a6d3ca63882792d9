#!/bin/bash
# PMC pass of the per-shape conv bench (scripts/conv_layer_bench.py) for several settings in one GPU call:
#   bash scripts/layer_pmc.sh <tag> "<ENV=VAL ...>" ...
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
    timeout ${RUN_TIMEOUT:-180} rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d "$D" -o t --output-format csv -- python3 "$R/scripts/conv_layer_bench.py" run ${REPS:-2} > "$D/run.out" 2> "$D/run.err"
  )
  echo "== $SETTING" > "$OUT/pmc_$I.txt"
  python3 "$R/scripts/layer_pmc.py" "$D" >> "$OUT/pmc_$I.txt" 2>&1
  rm -rf "$D"
  I=$((I+1))
done
