#!/bin/bash
# Round-3 evidence, in three gpurun calls (each well under 20 minutes on the box):
#   gpurun --timeout 1500 -- 'bash scripts/collect_r03.sh A'   headline cfg3 (trace, PMC traffic, bench) + cfg5 inference
#   gpurun --timeout 1500 -- 'bash scripts/collect_r03.sh B'   cfg4 in bf16-mixed (trace, per-layer, PMC traffic) + other configs
#   gpurun --timeout 1500 -- 'bash scripts/collect_r03.sh C'   per-layer A/B 256-pixel vs wide kernel, clock / MFMA-busy counters
# Everything lands in gpurun_out/r03_*; scripts/stage_r03.sh copies what is judged into profiles/r03/.
set -u
PART=${1:-A}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
case "$PART" in
  A)
    bash scripts/collect_profiles.sh r03_final
    bash scripts/collect_inference_profile.sh r03_cfg5
    ;;
  B)
    BENCH_ARGS="--config cfg4" MIMO_PRECISION=bf16-mixed TRACE_CONV_ARGS="4 30 16 256 256 2" FINAL_STEPS=10 \
      bash scripts/collect_profiles.sh r03_cfg4_bf16mixed
    bash scripts/collect_other_configs.sh
    ;;
  C)
    bash scripts/layer_ab.sh r03_ab_split16 MIMO_CONV_WIDE=0 - MIMO_CONV_WIDE=2
    MIMO_LAYER_BENCH_PREC=3 bash scripts/layer_ab.sh r03_ab_bf16mixed MIMO_CONV_WIDE=0 - MIMO_CONV_WIDE=2
    MIMO_LAYER_BENCH_WGRAD=0 MIMO_LAYER_BENCH_ONLY=0,4,8,11,14,15 bash scripts/layer_pmc.sh r03_pmc MIMO_CONV_WIDE=0 MIMO_CONV_WIDE=2
    MIMO_LAYER_BENCH_WGRAD=0 MIMO_LAYER_BENCH_SHAPES='8,256,256,30,30;16,256,256,30,30;32,256,256,30,30;64,256,256,30,30;128,256,256,30,30' \
      bash scripts/layer_ab.sh r03_nscale MIMO_CONV_WIDE=0 MIMO_CONV_WIDE=2
    ;;
esac
