#!/bin/bash
# round-4 GPU call E: clock / matrix-pipe-busy / wait counters of the weight gradient, read pipeline as before vs pinned one tap ahead
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_e
mkdir -p $O
cd $R
export MIMO_LAYER_BENCH_ONLY=4,6,8,10 REPS=3
V=$R/build/variants
bash scripts/layer_pmc.sh r04_e "MIMO_HIP_LIB=$V/libmimo_wg_nopin.so" "-"
cat $O/pmc_0.txt $O/pmc_1.txt | grep -E "==|wgrad"
