#!/bin/bash
# round-4 GPU call C: weight-gradient staging ablations (how much would sharing halo rows between vertically adjacent tiles buy?)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_c
mkdir -p $O
cd $R
export REPS=3
V=$R/build/variants
bash scripts/layer_ab.sh r04_c/wgabl "-" "MIMO_HIP_LIB=$V/libmimo_wgabl_halfA.so" "MIMO_HIP_LIB=$V/libmimo_wgabl_noA.so" "MIMO_HIP_LIB=$V/libmimo_wgabl_noAnoD.so"
python3 scripts/layer_ab_table.py $O/wgabl full half-A-staging no-A-staging no-staging > $O/wgabl.txt 2>&1
grep -E "^wgrad|^#" $O/wgabl.txt
