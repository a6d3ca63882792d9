#!/bin/bash
# round 6, call A: training-step hipGraphs (MIMO_TRAIN_GRAPH) — new bit-identity tests, then eager / graph A/B of the bench
# line at 4 images per GPU and at batch 32 (alternating, one box), then the whole GPU suite on the new default
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_a
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_streams_gpu.py -m gpu -q -x 2>&1 | tail -15 > $O/pytest_streams.txt
cat $O/pytest_streams.txt
for rep in 1 2 3; do
  for g in 0 1; do
    MIMO_TRAIN_GRAPH=$g timeout 300 python bench.py --batch 4 --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 3 > $O/b4_graph${g}_$rep.json 2> $O/b4_graph${g}_$rep.err
    python - <<PY
import json
try:
    d = json.load(open("$O/b4_graph${g}_$rep.json"))
    print("b4 graph=$g rep=$rep ms/step", d["ms_per_step"], "host enqueue", d["config"]["host_enqueue_ms_per_step"])
except Exception as e:
    print("b4 graph=$g rep=$rep FAILED", e)
PY
  done
done
for rep in 1 2; do
  for g in 0 1; do
    MIMO_TRAIN_GRAPH=$g timeout 300 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --profile-steps 3 > $O/b32_graph${g}_$rep.json 2> $O/b32_graph${g}_$rep.err
    python - <<PY
import json
try:
    d = json.load(open("$O/b32_graph${g}_$rep.json"))
    print("b32 graph=$g rep=$rep ms/step", d["ms_per_step"], "host enqueue", d["config"]["host_enqueue_ms_per_step"])
except Exception as e:
    print("b32 graph=$g rep=$rep FAILED", e)
PY
  done
done
# the DDP route on one rank over RCCL (staged backward, six collectives): eager / graph
for g in 0 1; do
  MIMO_TRAIN_GRAPH=$g MIMO_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 300 python bench.py --batch 4 --steps 60 --warmup 15 --no-cpu-baseline --profile-steps 3 > $O/b4_dist_graph$g.json 2> $O/b4_dist_graph$g.err
  python - <<PY
import json
try:
    d = json.load(open("$O/b4_dist_graph$g.json"))
    print("b4 one-rank RCCL graph=$g ms/step", d["ms_per_step"], "host enqueue", d["config"]["host_enqueue_ms_per_step"])
except Exception as e:
    print("b4 dist graph=$g FAILED", e)
PY
done
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $O/pytest_gpu_tail.txt
cat $O/pytest_gpu_tail.txt
