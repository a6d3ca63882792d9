#!/bin/bash
# Effective shader clock per kernel class inside the training step, 256-pixel kernels only against the default dispatch
# (same box): bash scripts/step_clock_ab.sh <tag>
set -u
TAG=${1:-clock}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MIMO_WGRAD_STREAM=0
for W in 0 1 0 1; do
  D=$OUT/t
  rm -rf "$D"; mkdir -p "$D"
  MIMO_CONV_WIDE=$W timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d "$D" -o s --output-format csv -- \
    python3 "$R/bench.py" --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline > /dev/null 2> "$OUT/run_$W.err"
  echo "== MIMO_CONV_WIDE=$W" >> "$OUT/step_clock.txt"
  python3 "$R/scripts/step_clock.py" "$D" 1000 >> "$OUT/step_clock.txt" 2>&1
  rm -rf "$D"
done
cat "$OUT/step_clock.txt"
