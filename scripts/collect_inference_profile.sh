#!/bin/bash
# rocprofv3 summary of the cfg5 inference harness (MC-dropout ensemble, 16 passes) at batch 1 and 8:
#   bash scripts/collect_inference_profile.sh <tag>   ->  gpurun_out/<tag>/{b1,b8}_kernel_stats.csv, b1.json, b8.json
set -u
TAG=${1:-cfg5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for B in 1 8; do
  rocprofv3 --kernel-trace --stats -d "$OUT/trace_b$B" -o cfg5 --output-format csv -- python3 "$R/scripts/measure_inference_speed.py" --batch $B --repetitions 50 > "$OUT/b${B}_under_rocprof.json" 2> "$OUT/trace_b$B.err"
  cp "$OUT/trace_b$B/cfg5_kernel_stats.csv" "$OUT/b${B}_kernel_stats.csv"
  rm -rf "$OUT/trace_b$B"
  python3 "$R/scripts/measure_inference_speed.py" --batch $B > "$OUT/b$B.json" 2> /dev/null
done
cat "$OUT/b1.json" "$OUT/b8.json"
