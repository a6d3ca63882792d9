#!/bin/bash
# round-4 GPU call Y: the two heaviest thin layers (90->45, 45->30 at 256x256) on the other convolution family
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_y
mkdir -p $O
cd $R
export MIMO_LAYER_BENCH_WGRAD=0 MIMO_LAYER_BENCH_ONLY=0,14,15 REPS=3
bash scripts/layer_ab.sh r04_y/fam "-" "MIMO_CONV_WIDE=0" "MIMO_CONV_WIDE=2"
python3 scripts/layer_ab_table.py $O/fam by-rule 256-pixel-kernel wide-kernel > $O/fam.txt 2>&1
cat $O/fam.txt | cut -c1-200
