#!/bin/bash
# GPU suite on the final tree, then the same suite with the round-4 fusions off (separate-kernel paths) and with the
# three-MFMA weight gradient
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05_s
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x 2>&1 | tail -4 > $O/pytest_default.txt
MIMO_FUSE_BN_IN=0 MIMO_FUSE_BWD_SRC=0 MIMO_CONV_THIN=0 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_fusions_off.txt
MIMO_WGRAD_NP=3 python -m pytest tests -m gpu -q 2>&1 | tail -8 > $O/pytest_np3.txt
cat $O/pytest_default.txt $O/pytest_fusions_off.txt $O/pytest_np3.txt
