#!/bin/bash
# Round-5 evidence, in gpurun calls (each well under 20 minutes on the box):
#   gpurun --timeout 1800 -- 'bash scripts/collect_r05.sh A'   GPU suite with the parity log; headline cfg3: trace, PMC traffic, bench
#   gpurun --timeout 1800 -- 'bash scripts/collect_r05.sh B'   other configs, cfg5 inference profile, batch-4 kernel trace (strong-scaling regime)
# Everything lands in gpurun_out/r05_*; scripts/stage_r05.sh copies what is judged into profiles/r05/.
set -u
PART=${1:-A}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
case "$PART" in
  A)
    mkdir -p gpurun_out/r05_final
    MIMO_PARITY_LOG=$R/gpurun_out/r05_final/parity_errors.txt python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/r05_final/pytest.txt
    PMC_INSTALL_DIR=profiles/r05/final bash scripts/collect_profiles.sh r05_final
    ;;
  B)
    bash scripts/collect_other_configs.sh
    bash scripts/collect_inference_profile.sh r05_cfg5
    mkdir -p gpurun_out/r05_b4
    cd /tmp && export TMPDIR=/tmp
    MIMO_WGRAD_STREAM=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r05_b4/t -o b4 --output-format csv -- python3 $R/bench.py --batch 4 --steps 40 --warmup 5 --profile-steps 0 --no-cpu-baseline > $R/gpurun_out/r05_b4/bench_under_rocprof.json 2> $R/gpurun_out/r05_b4/err.txt
    cp $R/gpurun_out/r05_b4/t/b4_kernel_stats.csv $R/gpurun_out/r05_b4/kernel_stats.csv
    rm -rf $R/gpurun_out/r05_b4/t
    cd $R && python3 bench.py --batch 4 --steps 40 --warmup 10 --no-cpu-baseline > gpurun_out/r05_b4/bench.json 2>> gpurun_out/r05_b4/err.txt
    tail -c 400 gpurun_out/r05_b4/bench.json
    for m in pinned pageable; do
      python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --host-batches $m 2>/dev/null | tail -1 > gpurun_out/r05_b4/bench_b32_host_$m.json
      python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --batch 4 --host-batches $m 2>/dev/null | tail -1 > gpurun_out/r05_b4/bench_b4_host_$m.json
    done
    for i in 1 2 3; do for v in 3 2; do
      MIMO_WGRAD_NP=$v python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --batch 4 --profile-steps 0 2>/dev/null | tail -1 | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('b4 np=$v', l['value'], l['ms_per_step'])" >> gpurun_out/r05_b4/b4_np_ab.txt
    done; done
    python3 tests/tools/convergence_probe.py 300 > gpurun_out/r05_b4/convergence.txt 2>&1
    ;;
esac
