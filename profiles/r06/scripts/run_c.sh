#!/bin/bash
# round 6, call C: host profile of the steady-state step at 4 images per GPU (plain and data-parallel route)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c
mkdir -p $O
cd $R
timeout 300 python scripts/host_profile.py 4 200 0 > $O/host_profile_b4.txt 2>&1
timeout 300 python scripts/host_profile.py 4 200 1 > $O/host_profile_b4_ddp.txt 2>&1
head -60 $O/host_profile_b4.txt
