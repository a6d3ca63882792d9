#!/bin/bash
# round 6, call N: data-parallel route at 4 images per GPU on the current tree (one-rank RCCL group), and a re-scan of the side
# stream's CU share at this batch (MIMO_WGRAD_CUS) now that the hand-offs are cheaper and the deep layers are K-split
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_n
mkdir -p $O
cd $R
timeout 600 python scripts/ddp_overhead.py 4 2>/dev/null | grep -v "version\|Hostname\|path" | tee $O/ddp_overhead_b4.txt
run() {  # name, batch, steps, env...
  local name=$1 batch=$2 steps=$3; shift 3
  env "$@" timeout 300 python bench.py --batch $batch --steps $steps --warmup 10 --no-cpu-baseline --no-strict --profile-steps 0 2>/dev/null | tail -1 > $O/$name.json
  python - <<PY
import json
try:
    d = json.load(open("$O/$name.json")); print("$name", d["ms_per_step"], "ms/step", d["value"], "images/s")
except Exception as e:
    print("$name FAILED", e)
PY
}
for rep in 1 2; do
  for cus in 96 128 160 192; do
    run b4_cus${cus}_$rep 4 80 MIMO_WGRAD_CUS=$cus
  done
done
MIMO_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 timeout 300 python bench.py --batch 4 --steps 80 --warmup 15 --no-cpu-baseline --no-strict --profile-steps 3 2>/dev/null | grep "^{" > $O/bench_b4_one_rank_rccl.json
python - <<PY
import json
d = json.load(open("$O/bench_b4_one_rank_rccl.json")); c = d["config"]
print("one-rank RCCL bench line: ", d["ms_per_step"], "ms/step; collectives", c["collectives_per_step"], c["collective_mbytes"], "host enqueue", c["host_enqueue_ms_per_step"])
PY
