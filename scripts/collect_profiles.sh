#!/bin/bash
# Collect the rocprofv3 evidence of one round on the GPU box (run through gpurun from the repo root):
#   bash scripts/collect_profiles.sh <tag>
# writes gpurun_out/<tag>/{kernel_stats.csv, bench.json, pmc_*.csv}; copy what should be judged into profiles/.
# Counter passes are separate runs with --kernel-trace only (no sys/hip/hsa tracing next to --pmc).
# Optional environment: BENCH_ARGS (e.g. "--config cfg4"), MIMO_PRECISION, TRACE_CONV_ARGS ("S f N H W Ci" for
# scripts/trace_convs.py when the workload is not cfg3 at batch 32).
set -u
TAG=${1:-prof}
BENCH_ARGS=${BENCH_ARGS:-}
TRACE_CONV_ARGS=${TRACE_CONV_ARGS:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# default run first (weight gradients on the side stream: kernel durations in this trace are OVERLAPPED durations)
rocprofv3 --kernel-trace --stats -d "$OUT/trace_default" -o cfg3 --output-format csv -- python3 "$R/bench.py" $BENCH_ARGS --steps 5 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-strict > "$OUT/bench_under_rocprof_default.json" 2> "$OUT/trace_default.err"
cp "$OUT/trace_default/cfg3_kernel_stats.csv" "$OUT/kernel_stats_default_overlapped.csv"
python3 "$R/scripts/trace_overlap.py" "$OUT/trace_default" 0 0 > "$OUT/overlap_default.txt" 2>&1
rm -rf "$OUT/trace_default"
# per-kernel evidence: the same command with the streams serialised (MIMO_WGRAD_STREAM=0), which is also how
# bench.py's own HIP-event pass measures the kernels (the plan serialises while its profiler is armed)
export MIMO_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o cfg3 --output-format csv -- python3 "$R/bench.py" $BENCH_ARGS --steps 5 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-strict > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
cp "$OUT/trace/cfg3_kernel_stats.csv" "$OUT/kernel_stats.csv"
python3 "$R/scripts/trace_step.py" "$OUT/trace" > "$OUT/step_timeline.txt" 2>&1
python3 "$R/scripts/trace_convs.py" "$OUT/trace" $TRACE_CONV_ARGS > "$OUT/conv_layers.txt" 2>&1
for SET in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS"; do
  NAME=$(echo "$SET" | cut -d' ' -f1)
  rocprofv3 --kernel-trace --pmc $SET -d "$OUT/pmc_$NAME" -o cfg3 --output-format csv -- python3 "$R/bench.py" $BENCH_ARGS --steps 2 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-strict > /dev/null 2> "$OUT/pmc_$NAME.err"
  python3 "$R/scripts/pmc_summary.py" "$OUT/pmc_$NAME" 8 --csv "$OUT/pmc_$NAME.csv" > "$OUT/pmc_$NAME.txt"
  rm -rf "$OUT/pmc_$NAME"
done
rm -rf "$OUT/trace"
python3 "$R/scripts/pmc_traffic.py" "$OUT" 3 > /dev/null 2>&1
unset MIMO_WGRAD_STREAM
# PMC_INSTALL_DIR (repo-relative, e.g. profiles/r04/final): put the fresh counter summary where bench.py looks for it (on the
# box's scratch copy of the repo), so that the final bench line below carries the traffic of THIS build
if [ -n "${PMC_INSTALL_DIR:-}" ] && [ -f "$OUT/pmc_traffic.json" ]; then
  mkdir -p "$R/$PMC_INSTALL_DIR" && cp "$OUT/pmc_traffic.json" "$R/$PMC_INSTALL_DIR/pmc_traffic.json"
fi
cd "$R" && python3 bench.py $BENCH_ARGS --steps ${FINAL_STEPS:-20} --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
tail -c 600 "$OUT/bench.json"
