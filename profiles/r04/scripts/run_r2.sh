cd "${GRAFT_REPO_ROOT:-$(pwd)}"
export MIMO_LAYER_BENCH_WGRAD=1 REPS=3 MIMO_LAYER_BENCH_SHAPES="32,256,256,2,30;32,256,256,3,21;4,256,256,2,30"
bash scripts/layer_ab.sh r04_r/layer "MIMO_CONV_THIN=1" "MIMO_CONV_THIN=0"
python3 scripts/layer_ab_table.py gpurun_out/r04_r/layer thin mfma > gpurun_out/r04_r/layer.txt 2>&1
cut -c1-200 gpurun_out/r04_r/layer.txt
