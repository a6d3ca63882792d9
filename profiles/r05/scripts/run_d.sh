#!/bin/bash
# round-5 GPU call D: where the deferred epilogue differs from the old one (op level), data-path tests, the two-MFMA weight-gradient timing probe
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_d
mkdir -p $O
cd "$R"
MIMO_AB_LIB=$R/build/variants/libmimo_nodefer.so python profiles/r05/scripts/op_bitcmp.py > $O/op_bitcmp.txt 2>&1
cat $O/op_bitcmp.txt
timeout 600 python -m pytest tests/test_data_gpu.py tests/test_network_gpu.py -q -m gpu -k "data or caller_masks" > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
bash scripts/ab_conv_layers.sh r05_d/np MIMO_WGRAD_NP 3 2
grep -h "wgrad" $O/np/conv_layers_3.txt > $O/np3.txt; grep -h "wgrad" $O/np/conv_layers_2.txt > $O/np2.txt
paste $O/np3.txt $O/np2.txt | awk '{print $1,$2,$3,$4,$5,$6,"|",$(NF/2+5),$(NF/2+6)}' | head -40
cd "$R"
for i in 1 2 3; do
  for v in 3 2; do
    MIMO_WGRAD_NP=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print('np=$v', l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in r['kernels'].items()}, l['config']['final_loss'])" >> $O/step_ab.txt
  done
done
cat $O/step_ab.txt
