#!/bin/bash
# Copy the round-6 evidence from gpurun_out/ (scratch) into profiles/r06/ (tracked).  Run from the repo root after the
# `gpurun ... scripts/collect_r06.sh {A,B}` calls.
set -eu
G=gpurun_out
P=profiles/r06
mkdir -p $P/final $P/cfg5 $P/b4
F=$G/r06_final
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
  if [ -f $G/r06_cfg5/b$B.json ]; then
    cp $G/r06_cfg5/b${B}_kernel_stats.csv $P/cfg5/b${B}_kernel_stats.csv
    cp $G/r06_cfg5/b$B.json $P/cfg5/b$B.json
  fi
done
B4=$G/r06_b4
if [ -d $B4 ]; then
  cp $B4/kernel_stats.csv $P/b4/kernel_stats.csv
  cp $B4/bench.json $P/b4/bench_cfg3_b4.json
  cp $B4/bench_under_rocprof.json $P/b4/bench_cfg3_b4_under_rocprof.json
  cp $B4/step_overlapped_timeline.txt $P/b4/step_overlapped_timeline.txt
  cp $B4/bench_one_rank_rccl.json $P/b4/bench_cfg3_b4_one_rank_rccl.json
  cp $B4/bench_b32_one_rank_rccl.json $P/b4/bench_cfg3_b32_one_rank_rccl.json
  cp $B4/bench_one_rank_rccl_reduce_scatter.json $P/b4/bench_cfg3_b4_one_rank_rccl_reduce_scatter.json
  cp $B4/ddp_overhead_b4.txt $P/b4/ddp_overhead_final.txt
  cp $B4/host_profile_b4.txt $P/b4/host_profile_final.txt
  for m in pinned pageable; do for b in b32 b4; do cp $B4/bench_${b}_host_$m.json $P/b4/bench_${b}_host_$m.json; done; done
  cp $B4/convergence.txt $P/convergence_final.txt
fi
ls -R $P | head -80
