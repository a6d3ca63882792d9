#!/bin/bash
# Round-6 evidence, in gpurun calls (each well under 25 minutes on the box):
#   gpurun --timeout 2400 -- 'bash scripts/collect_r06.sh A'   GPU suite with the parity log; headline cfg3: trace, PMC traffic, bench
#   gpurun --timeout 2400 -- 'bash scripts/collect_r06.sh B'   other configs, cfg5 inference profile, the 4-images-per-GPU regime (plain +
#                                                              one-rank RCCL route, kernel trace, host loop), data-parallel overhead table
# Everything lands in gpurun_out/r06_*; scripts/stage_r06.sh copies what is judged into profiles/r06/.
set -u
PART=${1:-A}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
case "$PART" in
  A)
    mkdir -p gpurun_out/r06_final
    MIMO_PARITY_LOG=$R/gpurun_out/r06_final/parity_errors.txt python -m pytest tests -m gpu -q 2>&1 | tail -5 > gpurun_out/r06_final/pytest.txt
    cat gpurun_out/r06_final/pytest.txt
    PMC_INSTALL_DIR=profiles/r06/final bash scripts/collect_profiles.sh r06_final
    ;;
  B)
    bash scripts/collect_other_configs.sh
    bash scripts/collect_inference_profile.sh r06_cfg5
    O=$R/gpurun_out/r06_b4
    mkdir -p $O
    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats -d $O/t -o b4 --output-format csv -- python3 $R/bench.py --batch 4 --steps 40 --warmup 5 --profile-steps 0 --no-cpu-baseline --no-strict > $O/bench_under_rocprof.json 2> $O/err.txt
    cp $O/t/b4_kernel_stats.csv $O/kernel_stats.csv
    python3 $R/scripts/trace_overlap.py $O/t > $O/step_overlapped_timeline.txt 2>&1
    rm -rf $O/t
    cd $R
    python3 bench.py --batch 4 --steps 80 --warmup 15 --no-cpu-baseline > $O/bench.json 2>> $O/err.txt
    tail -c 300 $O/bench.json
    export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
    MIMO_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python3 bench.py --batch 4 --steps 80 --warmup 15 --no-cpu-baseline --no-strict 2>/dev/null | grep "^{" > $O/bench_one_rank_rccl.json
    MIMO_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-strict 2>/dev/null | grep "^{" > $O/bench_b32_one_rank_rccl.json
    MIMO_DDP_ALGO=reduce_scatter MIMO_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python3 bench.py --batch 4 --steps 40 --warmup 10 --no-cpu-baseline --no-strict --profile-steps 0 2>/dev/null | grep "^{" > $O/bench_one_rank_rccl_reduce_scatter.json
    python3 scripts/ddp_overhead.py 4 2>/dev/null | grep -v "version\|Hostname\|path" > $O/ddp_overhead_b4.txt
    python3 scripts/host_profile.py 4 200 0 > $O/host_profile_b4.txt 2>&1
    for m in pinned pageable; do
      python3 bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-strict --host-batches $m 2>/dev/null | tail -1 > $O/bench_b32_host_$m.json
      python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-strict --batch 4 --host-batches $m 2>/dev/null | tail -1 > $O/bench_b4_host_$m.json
    done
    python3 tests/tools/convergence_probe.py 300 > $O/convergence.txt 2>&1
    python3 - <<PY
import json
for n in ("bench", "bench_one_rank_rccl", "bench_b32_one_rank_rccl", "bench_one_rank_rccl_reduce_scatter"):
    try:
        d = json.load(open("$O/" + n + ".json")); print(n, d["ms_per_step"], "ms/step", d["value"], "images/s", d.get("value_strict"))
    except Exception as e:
        print(n, "FAILED", e)
PY
    ;;
esac
