#!/bin/bash
# host-fed loop: event hand-offs resolved on the GPU queues (stream waits) or on the worker thread (host waits)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_p
mkdir -p $O
cd $R
for i in 1 2; do for b in 4 32; do
  for m in pinned pageable; do for h in gpu host; do
    MIMO_PREFETCH_HANDOFF=$h python3 bench.py --batch $b --steps $((b==4?60:25)) --warmup 8 --profile-steps 0 --no-cpu-baseline --host-batches $m 2>/dev/null | tail -1 | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('b$b $m handoff=$h', l['value'], l['ms_per_step'], l['config']['inputs'][-90:])" >> $O/ab.txt
  done; done
  python3 bench.py --batch $b --steps $((b==4?60:25)) --warmup 8 --profile-steps 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; l=json.loads(sys.stdin.read()); print('b$b resident', l['value'], l['ms_per_step'])" >> $O/ab.txt
done; done
cat $O/ab.txt
