#!/bin/bash
# round 6, call S: kernel traces of the one-rank RCCL route — bench.py (5.8 ms per step) against the same step driven by
# profiles/r06/scripts/ddp_bisect.py (4.4 ms): what is different on the GPU?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_s
mkdir -p $O
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
cd /tmp && export TMPDIR=/tmp
MIMO_BENCH_FORCE_DIST=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 rocprofv3 --kernel-trace -d $O/tb -o t --output-format csv -- python3 $R/bench.py --batch 4 --steps 12 --warmup 6 --profile-steps 0 --no-cpu-baseline --no-strict > $O/bench.json 2> $O/bench.err
python3 $R/scripts/trace_overlap.py $O/tb > $O/overlap_bench_ddp.txt 2>&1
rm -rf $O/tb
BISECT_PROPS=0 BISECT_ONLY=1 rocprofv3 --kernel-trace -d $O/td -o t --output-format csv -- python3 $R/profiles/r06/scripts/ddp_bisect.py > $O/bisect.txt 2> $O/bisect.err
python3 $R/scripts/trace_overlap.py $O/td > $O/overlap_bisect_ddp.txt 2>&1
rm -rf $O/td
tail -5 $O/overlap_bench_ddp.txt; tail -5 $O/overlap_bisect_ddp.txt
grep -c "q" $O/overlap_bench_ddp.txt
