#!/bin/bash
# round-4 GPU call Z: cfg3 layers at 4 images per GPU (strong-scaling regime): are the dispatch rules right there?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_z
mkdir -p $O
cd $R
export MIMO_LAYER_BENCH_WGRAD=1 REPS=3
export MIMO_LAYER_BENCH_SHAPES="4,256,256,30,30;4,128,128,30,60;4,128,128,60,60;4,64,64,120,240;4,64,64,240,240;4,32,32,240,480;4,32,32,480,480;4,16,16,480,480;4,32,32,960,480;4,32,32,480,240;4,64,64,480,240;4,64,64,240,120;4,128,128,240,120;4,128,128,120,60;4,256,256,90,45;4,256,256,45,30"
bash scripts/layer_ab.sh r04_z/b4 "-" "MIMO_CONV_WIDE=0" "MIMO_CONV_WIDE=2" "MIMO_CONV_WS_MF2=0" "MIMO_CONV_ADAPTIVE_NF=0" "MIMO_WGRAD_SPLIT_MODE=0"
python3 scripts/layer_ab_table.py $O/b4 by-rule no-wide wide-forced no-mf2 no-adaptive-nf wgrad-fixed-splits > $O/b4.txt 2>&1
cat $O/b4.txt | cut -c1-175
