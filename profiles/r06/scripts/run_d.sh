#!/bin/bash
# round 6, call D: host profile after the first host-side fixes; bench at 4 images per GPU (plain + one-rank RCCL), graph 0/1
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_d
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_streams_gpu.py tests/test_network_gpu.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.txt
timeout 300 python scripts/host_profile.py 4 200 0 > $O/host_profile_b4.txt 2>&1
timeout 300 python scripts/host_profile.py 4 200 1 > $O/host_profile_b4_ddp.txt 2>&1
head -32 $O/host_profile_b4.txt
grep "loss.backward\|host loop" $O/host_profile_b4_ddp.txt
for g in 0 1; do
  echo "== MIMO_TRAIN_GRAPH=$g, batch 4" | tee -a $O/ddp_overhead_b4.txt
  MIMO_TRAIN_GRAPH=$g timeout 600 python scripts/ddp_overhead.py 4 2>/dev/null | grep -v "version\|Hostname\|path" | tee -a $O/ddp_overhead_b4.txt
done
