#!/bin/bash
# Copy the round-5 evidence from gpurun_out/ (scratch) into profiles/r05/ (tracked).  Run from the repo root after the
# `gpurun ... scripts/collect_r05.sh {A,B}` calls.
set -eu
G=gpurun_out
P=profiles/r05
mkdir -p $P/final $P/cfg5 $P/b4
F=$G/r05_final
if [ -d $F ]; then
  cp $F/bench.json $P/final/bench_cfg3_n1.json
  cp $F/kernel_stats.csv $P/final/bench_cfg3_n1_kernel_stats.csv
  cp $F/kernel_stats_default_overlapped.csv $P/final/bench_cfg3_n1_kernel_stats_default_overlapped.csv
  cp $F/bench_under_rocprof.json $P/final/bench_cfg3_n1_under_rocprof.json
  cp $F/conv_layers.txt $F/step_timeline.txt $F/overlap_default.txt $F/pmc_traffic.json $P/final/
  cp $F/pmc_FETCH_SIZE.csv $F/pmc_WRITE_SIZE.csv $F/pmc_SQ_WAVE_CYCLES.csv $F/pmc_SQ_INSTS_LDS.csv $P/final/
  cp $F/parity_errors.txt $P/parity_errors.txt
  cp $F/pytest.txt $P/final/pytest_gpu_tail.txt
fi
if [ -f $G/other_configs.raw ]; then python3 scripts/other_configs.py $G/other_configs.raw $P/final/other_configs.jsonl; fi
for B in 1 8; do
  if [ -f $G/r05_cfg5/b$B.json ]; then
    cp $G/r05_cfg5/b${B}_kernel_stats.csv $P/cfg5/b${B}_kernel_stats.csv
    cp $G/r05_cfg5/b$B.json $P/cfg5/b$B.json
  fi
done
if [ -d $G/r05_b4 ]; then
  cp $G/r05_b4/kernel_stats.csv $P/b4/kernel_stats.csv
  cp $G/r05_b4/bench.json $P/b4/bench_cfg3_b4.json
  cp $G/r05_b4/bench_under_rocprof.json $P/b4/bench_cfg3_b4_under_rocprof.json
  cp $G/r05_b4/b4_np_ab.txt $P/b4/wgrad_np_ab.txt
  for m in pinned pageable; do for b in b32 b4; do cp $G/r05_b4/bench_${b}_host_$m.json $P/b4/bench_${b}_host_$m.json; done; done
  cp $G/r05_b4/convergence.txt $P/convergence_final.txt
fi
if [ -f $G/r05_h/b4_step_serial.txt ]; then cp $G/r05_h/b4_step_serial.txt $P/b4/step_serial_timeline.txt; fi
ls -R $P | head -60
