#!/bin/bash
# round 6, last call: the GPU suite, smoke() and the default bench line on the final tree
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_final_check
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu_tail.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
timeout 900 python bench.py 2>/dev/null | tail -1 | tee $O/bench_default.json
