#!/bin/bash
# round-4 GPU call B: the rest of the GPU suite, a batch-4 kernel trace, wide-kernel role ablations (Winograd feasibility)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_b
mkdir -p $O
cd $R
export MIMO_PARITY_LOG=$O/parity_errors.txt
python -m pytest tests/test_ddp_gpu.py -x -q --tb=short 2>&1 | tail -30 > $O/pytest_ddp.txt
python -m pytest tests -m gpu -q --tb=short --deselect tests/test_ddp_gpu.py 2>&1 | tail -40 > $O/pytest_rest.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/b4 -o t --output-format csv -- python3 $R/bench.py --batch 4 --steps 20 --warmup 5 --profile-steps 0 --no-cpu-baseline > $O/b4_bench.json 2> $O/b4.err
rocprofv3 --kernel-trace --stats -d $O/b4s -o t --output-format csv -- python3 $R/bench.py --batch 4 --steps 40 --warmup 5 --profile-steps 0 --no-cpu-baseline > $O/b4s_bench.json 2>> $O/b4.err
cd $R
export MIMO_LAYER_BENCH_WGRAD=0 MIMO_LAYER_BENCH_ONLY=4,6,8 REPS=3
bash scripts/layer_ab.sh r04_b/wabl "MIMO_CONV_WIDE=2" "MIMO_CONV_WIDE=2 MIMO_HIP_LIB=$R/build/variants/libmimo_wabl_prod.so" "MIMO_CONV_WIDE=2 MIMO_HIP_LIB=$R/build/variants/libmimo_wabl_cons.so" "MIMO_CONV_WIDE=2 MIMO_HIP_LIB=$R/build/variants/libmimo_wabl_mfma.so" "MIMO_CONV_WIDE=2 MIMO_HIP_LIB=$R/build/variants/libmimo_wabl_read.so"
python3 scripts/layer_ab_table.py $O/wabl full producers-alone consumers-alone mfma-alone reads-alone > $O/wabl.txt 2>&1
find $O -name "*kernel_trace.csv" -size +20M -delete
tail -5 $O/pytest_ddp.txt $O/pytest_rest.txt; cat $O/wabl.txt
