#!/bin/bash
# round-5 GPU call E: the two-MFMA (fp16) weight gradient wired in: full GPU suite, op bit-compare of the deferred data-gradient
# epilogue, step A/B against the three-MFMA arithmetic, per-layer table, convergence probe
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_e
mkdir -p $O
cd "$R"
MIMO_PARITY_LOG=$O/parity_errors.txt timeout 2000 python -m pytest tests -q -m gpu > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
MIMO_AB_LIB=$R/build/variants/libmimo_nodefer.so python profiles/r05/scripts/op_bitcmp.py > $O/op_bitcmp.txt 2>&1
cat $O/op_bitcmp.txt
for i in 1 2 3; do
  for v in 3 2; do
    MIMO_WGRAD_NP=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print('np=$v', l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in r['kernels'].items()}, r['bandwidth_kernels']['ms_per_step'], l['config']['final_loss'])" >> $O/step_ab.txt
  done
done
cat $O/step_ab.txt
MIMO_WGRAD_NP=3 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --batch 4 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('b4 np3', l['value'], l['ms_per_step'])" | tee -a $O/step_ab.txt
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --batch 4 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('b4 np2', l['value'], l['ms_per_step'])" | tee -a $O/step_ab.txt
bash scripts/ab_conv_layers.sh r05_e/np MIMO_WGRAD_NP 3 2
grep -h "wgrad" $O/np/conv_layers_3.txt > $O/np3.txt; grep -h "wgrad" $O/np/conv_layers_2.txt > $O/np2.txt
paste $O/np3.txt $O/np2.txt | head -30
cd "$R"
python tests/tools/convergence_probe.py 300 > $O/convergence.txt 2>&1
tail -16 $O/convergence.txt
