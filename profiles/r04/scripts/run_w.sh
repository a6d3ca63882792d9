#!/bin/bash
# round-4 GPU call W: role ablation of the THIN 256x256 / 128x128 layers (timing-only builds of both convolution families:
# -DMIMO_WIDE_ABLATE / -DMIMO_CONV_ABLATE = 56 producers alone, 7 consumers alone, 55 MFMAs alone, 32 no epilogue, 1 no input loads)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_w
mkdir -p $O
cd $R
export MIMO_LAYER_BENCH_WGRAD=0 MIMO_LAYER_BENCH_ONLY=0,2,14,15 REPS=3
V=$R/build/variants
bash scripts/layer_ab.sh r04_w/tabl "-" "MIMO_HIP_LIB=$V/libmimo_tabl_prod.so" "MIMO_HIP_LIB=$V/libmimo_tabl_cons.so" "MIMO_HIP_LIB=$V/libmimo_tabl_mfma.so" "MIMO_HIP_LIB=$V/libmimo_tabl_noepi.so" "MIMO_HIP_LIB=$V/libmimo_tabl_noload.so"
python3 scripts/layer_ab_table.py $O/tabl full producers-alone consumers-alone mfma-alone no-epilogue no-input-loads > $O/tabl.txt 2>&1
cat $O/tabl.txt
