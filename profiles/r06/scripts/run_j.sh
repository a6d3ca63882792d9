#!/bin/bash
# round 6, call J: the whole GPU suite with the parity log (after the loss-buffer override fix)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_j
mkdir -p $O
cd $R
MIMO_PARITY_LOG=$O/parity_errors.txt timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $O/pytest_gpu_tail.txt
