#!/bin/bash
# The non-headline workloads (run through gpurun from the repo root): writes gpurun_out/other_configs.raw
# with one line per command; scripts/other_configs.py turns it into profiles/<round>/final/other_configs.jsonl
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
O=gpurun_out/other_configs.raw
: > $O
run() { echo "CMD $*" >> $O; "$@" 2>/dev/null | tail -n 1 >> $O; }
runp() { P=$1; shift; echo "CMD MIMO_PRECISION=$P $*" >> $O; MIMO_PRECISION=$P "$@" 2>/dev/null | tail -n 1 >> $O; }
run python scripts/measure_inference_speed.py
run python scripts/measure_inference_speed.py --batch 8
run python scripts/measure_inference_speed.py --monte_carlo_steps 0 --dropout 0 --height 128 --width 160
for P in bf16-mixed 16-mixed; do  # cfg5 in the 16-bit storage modes (16-mixed = the reference's production precision)
  runp $P python scripts/measure_inference_speed.py
  runp $P python scripts/measure_inference_speed.py --batch 8
done
run python bench.py --config cfg2 --steps 10 --warmup 3 --no-cpu-baseline
run python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline
for P in bf16 bf16-mixed 16-mixed; do
  runp $P python bench.py --config cfg4 --steps 10 --warmup 3 --no-cpu-baseline
  runp $P python bench.py --steps 20 --warmup 5 --no-cpu-baseline
done
runp fp32 python bench.py --steps 10 --warmup 3 --no-cpu-baseline
run python bench.py --batch 4 --steps 30 --warmup 8 --no-cpu-baseline --profile-steps 0
cat $O | cut -c1-200
