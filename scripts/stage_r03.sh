#!/bin/bash
# Copy the round-3 evidence from gpurun_out/ (scratch) into profiles/r03/ (tracked).  Run from the repo root after the
# three `gpurun ... scripts/collect_r03.sh {A,B,C}` calls (and scripts/power_ab.sh / scripts/step_clock_ab.sh).
set -eu
G=gpurun_out
P=profiles/r03
mkdir -p $P/final $P/cfg4_bf16mixed $P/cfg5
F=$G/r03_final
cp $F/bench.json $P/final/bench_cfg3_n1.json
cp $F/kernel_stats.csv $P/final/bench_cfg3_n1_kernel_stats.csv
cp $F/kernel_stats_default_overlapped.csv $P/final/bench_cfg3_n1_kernel_stats_default_overlapped.csv
cp $F/bench_under_rocprof.json $P/final/bench_cfg3_n1_under_rocprof.json
cp $F/bench_under_rocprof_default.json $P/final/bench_cfg3_n1_under_rocprof_default.json
cp $F/conv_layers.txt $F/step_timeline.txt $F/overlap_default.txt $F/pmc_traffic.json $P/final/
cp $F/pmc_FETCH_SIZE.csv $F/pmc_WRITE_SIZE.csv $F/pmc_SQ_WAVE_CYCLES.csv $F/pmc_SQ_INSTS_LDS.csv $P/final/
python3 scripts/other_configs.py $G/other_configs.raw $P/final/other_configs.jsonl
M=$G/r03_cfg4_bf16mixed
cp $M/bench.json $P/cfg4_bf16mixed/bench_cfg4_bf16mixed_n1.json
cp $M/kernel_stats.csv $P/cfg4_bf16mixed/kernel_stats.csv
cp $M/conv_layers.txt $M/step_timeline.txt $M/pmc_traffic.json $M/pmc_FETCH_SIZE.csv $M/pmc_WRITE_SIZE.csv $P/cfg4_bf16mixed/
for B in 1 8; do
  cp $G/r03_cfg5/b${B}_kernel_stats.csv $P/cfg5/b${B}_kernel_stats.csv
  cp $G/r03_cfg5/b$B.json $P/cfg5/b$B.json
done
{
  echo "# Per-layer convolution timings of cfg3 at batch 32 (scripts/conv_layer_bench.py under rocprofv3 --kernel-trace, fastest of the"
  echo "# repetitions, one launch per distinct layer shape): 256-pixel kernels only | default dispatch (sched::wide_config) | wide"
  echo "# kernel wherever supported.  Run-to-run spread of a layer inside one box is +-3...7 % (clock state left by the kernels before it)."
  echo
  echo "## split16"
  python3 scripts/layer_ab_table.py $G/r03_ab_split16 256-pixel default wide-forced
  echo
  echo "## bf16-mixed (16-bit storage, one MFMA per product)"
  python3 scripts/layer_ab_table.py $G/r03_ab_bf16mixed 256-pixel default wide-forced
} > $P/conv_layers_ab.txt
{
  echo "# Effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and matrix-pipe busy fraction (SQ_VALU_MFMA_BUSY_CYCLES / 1024"
  echo "# SIMDs / cycles) per layer, rocprofv3 --kernel-trace --pmc (scripts/layer_pmc.sh); first block split16, second bf16-mixed."
  cat $G/r03_pmc/pmc_0.txt $G/r03_pmc/pmc_1.txt
  echo
  echo "## bf16-mixed"
  cat $G/r03_pmc_bf16mixed/pmc_0.txt $G/r03_pmc_bf16mixed/pmc_1.txt
} > $P/conv_clock_busy.txt
{
  echo "# 30->30 at 256x256 against the batch size (images per launch 8..128), 256-pixel kernel | wide kernel: time is linear in the"
  echo "# batch (no fixed cost); 504 MB of fp32 activations per 32 images in ~150 us = 3.4 TB/s next to ~95 us of MFMA time."
  python3 scripts/layer_ab_table.py $G/r03_nscale 256-pixel wide
} > $P/conv_batch_scaling.txt
cp $G/r03_clock/step_clock.txt $P/step_clock_ab.txt
ls -R $P | head -60
