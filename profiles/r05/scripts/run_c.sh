#!/bin/bash
# round-5 GPU call C: deferred epilogue of the 32-channel wide instances (conv_wide.hip DEFER): bit-identity against the
# build without it, parity subset, per-layer times, step A/B
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_c
mkdir -p $O
cd "$R"
V=$R/build/variants/libmimo_nodefer.so
for a in "4 256" "32 256" "3 100" "2 64"; do
  python profiles/r05/scripts/step_hash.py $a >> $O/hash_defer.txt 2>&1
  MIMO_HIP_LIB=$V python profiles/r05/scripts/step_hash.py $a >> $O/hash_nodefer.txt 2>&1
done
diff $O/hash_defer.txt $O/hash_nodefer.txt && echo "BIT-IDENTICAL" | tee $O/bitcmp.txt
cat $O/hash_defer.txt
python profiles/r05/scripts/data_path_probe.py > $O/data_path_probe.txt 2>&1; tail -3 $O/data_path_probe.txt
MIMO_PARITY_LOG=$O/parity_errors.txt timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_network_gpu.py tests/test_data_gpu.py tests/test_configs_gpu.py -q -m gpu -x > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
export MIMO_LAYER_BENCH_ONLY=0,1,2,13,14,15 MIMO_LAYER_BENCH_WGRAD=0 REPS=6
bash scripts/layer_ab.sh r05_c/layers "-" "MIMO_HIP_LIB=$V" "-" "MIMO_HIP_LIB=$V"
cat $O/layers/layers_*.txt
cd "$R"
for i in 1 2 3; do
  for v in defer nodefer; do
    if [ $v = nodefer ]; then export MIMO_HIP_LIB=$V; else unset MIMO_HIP_LIB; fi
    python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=l['roofline']; print('$v', l['value'], l['ms_per_step'], {k:v['ms_per_step'] for k,v in r['kernels'].items()}, r['tiers']['256x256']['kernels_ms'])" >> $O/step_ab.txt
  done
done
unset MIMO_HIP_LIB
cat $O/step_ab.txt
python bench.py --steps 50 --warmup 10 --no-cpu-baseline --batch 4 --host-batches pageable 2>/dev/null | tail -1 | python -c "import json,sys; l=json.loads(sys.stdin.read()); print('b4 pageable', l['value'], l['ms_per_step'], l['config']['inputs'])"
