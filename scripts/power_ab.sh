#!/bin/bash
# Is the training step power-bound?  One gpurun call:
#   bash scripts/power_ab.sh <tag>
# (1) bench.py (cfg3, 60 steps) with the 256-pixel kernels only (MIMO_CONV_WIDE=0) and with the default dispatch,
#     alternating, power and shader clock sampled beside each run (scripts/power_sample.py);
# (2) the largest layer (960->480 at 32x32) back to back for ~3 s on either kernel: burst (min) against sustained
#     (mean of the last quarter) kernel durations from the rocprofv3 trace, again with power / clock samples.
set -u
TAG=${1:-power}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
for I in 0 1; do
  for W in 0 1; do
    MIMO_CONV_WIDE=$W python3 scripts/power_sample.py "$OUT/bench_wide${W}_$I.power.txt" -- \
      python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline > "$OUT/bench_wide${W}_$I.json" 2> "$OUT/bench_wide${W}_$I.err"
    head -4 "$OUT/bench_wide${W}_$I.power.txt"
    python3 - "$OUT/bench_wide${W}_$I.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
k = d["roofline"]["kernels"]
print("images/s", d["value"], "ms/step", d["ms_per_step"], {n: v["ms_per_step"] for n, v in k.items()})
PY
  done
done
cd /tmp && export TMPDIR=/tmp
for W in 0 2; do
  D=$OUT/layer_wide$W
  mkdir -p "$D"
  export MIMO_CONV_WIDE=$W MIMO_LAYER_BENCH_ONLY=8 MIMO_LAYER_BENCH_WGRAD=0 MIMO_LAYER_BENCH_LABELS=$D/labels.txt
  timeout 300 python3 "$R/scripts/power_sample.py" "$OUT/layer_wide$W.power.txt" -- \
    rocprofv3 --kernel-trace -d "$D" -o t --output-format csv -- python3 "$R/scripts/conv_layer_bench.py" run 2000 > "$D/run.out" 2> "$D/run.err"
  for S in min median tail; do
    echo "== MIMO_CONV_WIDE=$W stat=$S" >> "$OUT/layer_sustained.txt"
    MIMO_LAYER_BENCH_STAT=$S python3 "$R/scripts/conv_layer_bench.py" report "$D" | head -2 >> "$OUT/layer_sustained.txt"
  done
  head -4 "$OUT/layer_wide$W.power.txt"
  rm -rf "$D"
done
cat "$OUT/layer_sustained.txt"
